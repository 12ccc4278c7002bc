// alga_amd/csrc/pkb_kernels.hip -- gfx950 kernels of the approximate supplement (error_rate > 0.01).
//
// Replaces, for the caller at src/main.cpp:300-347 of the reference (paths relative to its root):
//   Read::getLIKmers                                   src/DataStructures/Read.cpp:145-226          -> k_li_kmers
//   GraphCreatorKmerBased bucket sort + grouping       src/GraphCreators/GraphCreatorKmerBased.cpp   -> radix sort by k-mer hash
//   GraphCreatorPairwiseKmerBranch::createAlignmentsForKmers  .../GraphCreatorPairwiseKmerBranch.cpp:16-97 -> k_pkb_groups
//   AlignmentControllerHybrid / LowErrorRate::canAlign src/AlignmentControllers/*.cpp               -> can_align()
//   Graph::addDirectedEdge / retainOnlySmallestOffset  src/DataStructures/Graph.cpp:53-71,348-387   -> sort + unique by (src, dst)
//
// Semantics: groups of one round are independent here (every group sees the graph as it was when the round started
// plus its own additions); the reference walks the groups one after the other (and races between threads when
// --threads > 1).  The CPU checker under tests/ implements both and quantifies the difference.  Equal k-mers
// (same hash, position and read length) are ordered by read id where the reference leaves the order to std::sort.
#include <hip/hip_runtime.h>
#include <algorithm>
#include "prefsuf_common.h"
#include "prefsuf_kernels.h"
#include "pkb_kernels.h"

namespace alga {

__device__ __forceinline__ int pkb_lane() { return (int) (threadIdx.x & 63u); }
__device__ __forceinline__ uint32_t pkb_funnel(uint32_t lo, uint32_t hi, int r) { return __funnelshift_r(lo, hi, r); }
constexpr int PKB_INF = 1000000001;                      // Params::INF (include/Params.h:40)

// ------------------------------------------------------------------------------------------
// canAlign: AlignmentControllerHybrid::canAlign (Hybrid.cpp:46-83) -> AlignmentControllerLowErrorRate::canAlign
// (LowErrorRate.cpp:15-49) under the reference's defaults (USE_LCS_LOW_ERROR_FILTER = USE_ACLER_INSTEAD_OF_ACLCS = 1).
//   X = (r1 >> 2*off) ^ r2 ; mismatching BITS (not nucleotides) are counted over the overlap; the first 2*se+1 bits
//   and the last 2*se bits of the overlap must be equal; accept iff 100 * ((2*ov - diffbits) >> 1) >= min_identity * ov.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ bool can_align(const NodesDev &nd, int r1, int r2, int off, const PkbCfg &c) {
    const int l1 = nd.len[r1], l2 = nd.len[r2];
    if (100 * off > c.max_offset_pct * l1) return false;                     // Hybrid :50-52
    if (off < 0) return false;                                               // MIN_OFFSET_FOR_ALIGNMENT = 0
    const int ov = (l1 < l2 + off ? l1 : l2 + off) - off;                    // Read::calculateReadOverlap
    if (ov < c.min_overlap_area) return false;
    if (l2 + off - l1 < 0) return false;                                     // Read::getRightOffset
    const uint32_t *a = nd.words + (size_t) r1 * nd.stride, *b = nd.words + (size_t) r2 * nd.stride;
    const int n1 = blocks_of(l1), n2 = blocks_of(l2);
    const int m = n1 < n2 ? n1 : n2;
    const int bit = 2 * off, q = bit >> 5, r = bit & 31;
    const int nbits = 2 * ov;
    const int t0 = 2 * (ov - c.same_ends);                                   // first bit of the tail window
    int total = 0, head = 0, tail = 0;
    const int nblk = (nbits + 31) >> 5;
    for (int k = 0; k < nblk; k++) {
        const uint32_t lo = (q + k) < n1 ? a[q + k] : 0u;
        const uint32_t hi = (q + k + 1) < n1 ? a[q + k + 1] : 0u;
        uint32_t x = pkb_funnel(lo, hi, r);
        if (k < m) x ^= b[k];
        const int rem = nbits - 32 * k;                                      // valid bits in this block
        const uint32_t vmask = rem >= 32 ? 0xFFFFFFFFu : ((1u << rem) - 1u);
        x &= vmask;
        total += __popc(x);
        if (k == 0) head = __popc(x & ((2u << (2 * c.same_ends)) - 1u));     // bits 0 .. 2*se inclusive (LowErrorRate :43)
        const int lo_t = t0 - 32 * k;                                        // tail window: bits >= t0
        if (lo_t < 32) tail += __popc(lo_t <= 0 ? x : (x & ~((1u << lo_t) - 1u)));
    }
    if (head != 0 || tail != 0) return false;
    const int seq = (nbits - total) >> 1;
    return 100 * seq >= c.min_identity_pct * ov;
}

__global__ void __launch_bounds__(256) k_can_align_batch(NodesDev nd, PkbCfg c, const int32_t *__restrict__ triples, uint64_t n,
                                                          uint8_t *__restrict__ out) {
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
        const int r1 = triples[3 * i], r2 = triples[3 * i + 1], off = triples[3 * i + 2];
        bool ok = false;
        if (r1 >= 0 && r1 < nd.n && r2 >= 0 && r2 < nd.n && nd.len[r1] > 0 && nd.len[r2] > 0) ok = can_align(nd, r1, r2, off, c);
        out[i] = ok ? 1 : 0;
    }
}

// ------------------------------------------------------------------------------------------
// LI k-mers (Read::getLIKmers): per interval of start positions the k-mer that is smallest as a base-4 number under
// the alphabet permutation prio[]; hash = value mod 10^18+3.  Start positions only move forward, so the intervals are
// finished one after the other and no per-interval array is needed.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t mod_hash_u128(unsigned __int128 v) {   // v mod (10^18 + 3), four bits at a time
    const uint64_t M = 1000000000000000003ull;
    uint64_t r = 0;
#pragma unroll 1
    for (int sh = 124; sh >= 0; sh -= 4) {
        r = r * 16ull + (uint64_t) ((v >> sh) & 15u);                        // r < M < 2^60: no overflow
        r %= M;
    }
    return r;
}

// v mod (10^18 + 3) for v < 2^70 (k-mers of up to 35 nucleotides, the reference's LI_KMER_LENGTH): one 64-bit remainder of the low
// word plus a table look-up for the 6 high bits, T[h] = (h * 2^64) mod M  (the nibble loop above costs 32 remainders per k-mer and was
// three quarters of the k-mer kernel)
__device__ __forceinline__ uint64_t mod_hash_u70(unsigned __int128 v, const uint64_t *T) {
    const uint64_t M = 1000000000000000003ull;
    uint64_t r = (uint64_t) v % M + T[(uint32_t) (v >> 64) & 63u];               // both terms < M < 2^60
    return r >= M ? r - M : r;
}

__device__ __forceinline__ void mod_table_fill(uint64_t *T) {                    // 64 threads of the workgroup, then __syncthreads()
    if (threadIdx.x < 64) T[threadIdx.x] = mod_hash_u128((unsigned __int128) threadIdx.x << 64);
}

// returns the number of k-mers written to hash_out / ind_out (at most `intervals`).  T: table of mod_hash_u70 (LDS) or null
__device__ __forceinline__ int li_kmers(const uint32_t *row, int len, int k, int intervals, const int *prio, uint64_t *hash_out, int32_t *ind_out,
                                        const uint64_t *T = nullptr) {
    typedef unsigned __int128 u128;
    const bool fast = T != nullptr && k <= 35;
    auto mh = [&](u128 v) { return fast ? mod_hash_u70(v, T) : mod_hash_u128(v); };
    if (k > len || intervals <= 0) return 0;
    auto digit = [&](int pos) { return (u128) (uint32_t) prio[(row[pos >> 4] >> ((pos & 15) << 1)) & 3u]; };
    u128 h = 0;
    for (int q = 0; q < k; q++) h = (h << 2) + digit(q);
    const u128 low_mask = (((u128) 1) << (2 * (k - 1))) - 1;                  // factor - 1, factor = 4^(k-1)
    const int il = (len - k + 1 + intervals - 1) / intervals;                 // ceil((size - length + 1) / intervals)
    u128 best = h; int best_p = 0, cur = 0, cnt = 0;
    for (int p = 1; p + k <= len; p++) {
        h = ((h & low_mask) << 2) + digit(p + k - 1);                         // hash -= factor * first; hash <<= 2; hash += next
        const int iv = p / il;
        if (iv != cur) { hash_out[cnt] = mh(best); ind_out[cnt] = best_p; cnt++; cur = iv; best = h; best_p = p; }
        else if (h < best) { best = h; best_p = p; }
    }
    hash_out[cnt] = mh(best); ind_out[cnt] = best_p; cnt++;
    return cnt;
}

// fixed-slot form for tests / function-level parity: slots [node * intervals + j]
__global__ void __launch_bounds__(256) k_li_kmers_slots(NodesDev nd, PkbCfg c, int4 prio4, uint64_t *__restrict__ hash, int32_t *__restrict__ ind,
                                                         int32_t *__restrict__ count) {
    __shared__ uint64_t T[64];
    mod_table_fill(T);
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nd.n) return;
    const int prio[4] = {prio4.x, prio4.y, prio4.z, prio4.w};
    uint64_t h[PKB_MAX_INTERVALS]; int32_t p[PKB_MAX_INTERVALS];
    int cnt = 0;
    if (nd.len[i] >= c.li_k) cnt = li_kmers(nd.words + (size_t) i * nd.stride, nd.len[i], c.li_k, c.li_intervals, prio, h, p, T);
    count[i] = cnt;
    for (int j = 0; j < c.li_intervals; j++) { hash[(size_t) i * c.li_intervals + j] = j < cnt ? h[j] : 0ull; ind[(size_t) i * c.li_intervals + j] = j < cnt ? p[j] : 0; }
}

// ------------------------------------------------------------------------------------------
// The graph of the supplement: sorted unique 64-bit keys  src << 36 | dst << 9 | offset  (one key per (src, dst): the smallest
// offset) + row pointers.  Graph::addDirectedEdge / retainOnlySmallestOffset (src/DataStructures/Graph.cpp:53-71,348-387) become
// "sort the additions, merge, keep the first key of every (src, dst) run".
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long pkb_edge_key(int src, int dst, int off) {
    return ((unsigned long long) (uint32_t) src << 36) | ((unsigned long long) (uint32_t) dst << 9) | (uint32_t) (off & 511);
}
__device__ __forceinline__ int pkb_key_src(unsigned long long k) { return (int) (k >> 36); }
__device__ __forceinline__ int pkb_key_dst(unsigned long long k) { return (int) ((k >> 9) & 0x7FFFFFFull); }
__device__ __forceinline__ int pkb_key_off(unsigned long long k) { return (int) (k & 511ull); }

// edge list -> keys; counts[0] += edges whose offset does not fit, counts[1] += positions where (src, dst) does not increase
__global__ void __launch_bounds__(256) k_pkb_edge_keys(const alga_edge_dev *__restrict__ e, uint64_t n, unsigned long long *__restrict__ keys,
                                                        unsigned long long *__restrict__ counts) {
    unsigned bad = 0, unsorted = 0;
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
        const alga_edge_dev x = e[i];
        bad += (uint32_t) x.offset > 511u;                                      // never on reads the supplement is meant for (<= 500 nt)
        const unsigned long long k = pkb_edge_key(x.src, x.dst, x.offset);
        keys[i] = k;
        if (i > 0) { const alga_edge_dev y = e[i - 1]; unsorted += (pkb_edge_key(y.src, y.dst, 0) >> 9) >= (k >> 9); }
    }
    if (bad) atomicAdd(&counts[0], (unsigned long long) bad);
    if (unsorted) atomicAdd(&counts[1], (unsigned long long) unsorted);
}

// row pointers of a sorted key list: rowptr[s] = first key with src >= s, rowptr[n] = E
__global__ void __launch_bounds__(256) k_pkb_rowptr(const unsigned long long *__restrict__ keys, uint64_t E, int32_t n, uint32_t *__restrict__ rowptr) {
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i <= E; i += (uint64_t) gridDim.x * blockDim.x) {
        const int s_prev = i == 0 ? -1 : pkb_key_src(keys[i - 1]);
        const int s_cur = i == E ? n : pkb_key_src(keys[i]);
        for (int s = s_prev + 1; s <= s_cur; s++) rowptr[s] = (uint32_t) i;
    }
}

__global__ void __launch_bounds__(256) k_pkb_keys_to_edges(const unsigned long long *__restrict__ keys, uint64_t E, alga_edge_dev *__restrict__ out) {
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < E; i += (uint64_t) gridDim.x * blockDim.x) {
        const unsigned long long k = keys[i];
        alga_edge_dev e; e.src = pkb_key_src(k); e.dst = pkb_key_dst(k); e.offset = pkb_key_off(k);
        out[i] = e;
    }
}

// ------------------------------------------------------------------------------------------
// masks of the supplement (src/main.cpp:308-322): alignTo = no in-edge but out-edges, alignFrom = in-edges but no out-edge
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_pkb_indeg(const unsigned long long *__restrict__ keys, uint64_t m, uint32_t *__restrict__ indeg) {
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (uint64_t) gridDim.x * blockDim.x)
        atomicAdd(&indeg[pkb_key_dst(keys[i])], 1u);
}

__global__ void __launch_bounds__(256) k_pkb_masks(int32_t n, const uint32_t *__restrict__ rowptr, const uint32_t *__restrict__ indeg,
                                                    uint8_t *__restrict__ mask /* bit0 from, bit1 to */) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t outd = rowptr[i + 1] - rowptr[i], ind = indeg[i];
    mask[i] = (uint8_t) (((ind > 0 && outd == 0) ? 1 : 0) | ((ind == 0 && outd > 0) ? 2 : 0));
}

// nodes that take part in the supplement (the masks never change between the rounds): dense id list, so that the k-mer kernel
// runs with full waves (the tips are ~1 node in 5).  Flags -> scan -> scatter: a block-aggregated append would still issue one
// same-address atomic per workgroup (78 k of them at 20 M nodes: 0.9 ms).
__global__ void __launch_bounds__(256) k_pkb_tip_flags(NodesDev nd, PkbCfg c, const uint8_t *__restrict__ mask, uint32_t *__restrict__ flag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nd.n) flag[i] = (mask[i] != 0 && nd.len[i] >= c.li_k && nd.len[i] >= c.kmer_length_bucket) ? 1u : 0u;   // Read::getKmers: length > size() -> none
}

// tips[] and, per tip, the number of LI k-mers it yields -- a function of its length alone (the intervals that hold a start
// position), the same in every round: the k-mer kernel writes at fixed offsets, no append counter.  max_len: longest tip.
__global__ void __launch_bounds__(256) k_pkb_tip_list(NodesDev nd, PkbCfg c, const uint32_t *__restrict__ flag, const uint32_t *__restrict__ pos,
                                                       uint32_t *__restrict__ tips, uint32_t *__restrict__ kcount, unsigned long long *__restrict__ max_len) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int len = 0;
    if (i < nd.n && flag[i]) {
        len = nd.len[i];
        const int il = (len - c.li_k + 1 + c.li_intervals - 1) / c.li_intervals;
        tips[pos[i]] = (uint32_t) i;
        kcount[pos[i]] = (uint32_t) ((len - c.li_k) / il + 1);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const int t = __shfl_xor(len, o); len = t > len ? t : len; }
    if ((threadIdx.x & 63u) == 0 && len > 0) atomicMax(max_len, (unsigned long long) len);
}

// A k-mer entry: key = hash, val = (4095 - indInRead) << 40 | read length << 28 | node id.  Ascending val == the order of the
// reference inside a group (indInRead descending, read length ascending; Kmer::operator<, Kmer.cpp:58-64) with ties by node id.
constexpr unsigned long long PKB_ID_MASK = (1ull << 28) - 1;
__device__ __forceinline__ int pkb_val_id(unsigned long long v) { return (int) (v & PKB_ID_MASK); }
__device__ __forceinline__ int pkb_val_len(unsigned long long v) { return (int) ((v >> 28) & 0xFFFull); }
__device__ __forceinline__ int pkb_val_ind(unsigned long long v) { return 4095 - (int) (v >> 40); }

// k-mers of every node that takes part (GraphCreatorKmerBased::getKmersForBucketJob :202-259) at the tip's fixed offset
__global__ void __launch_bounds__(256) k_pkb_kmers(NodesDev nd, PkbCfg c, int4 prio4, const uint32_t *__restrict__ tips, const uint32_t *__restrict__ koff,
                                                    uint32_t n_tips, unsigned long long *__restrict__ keys, unsigned long long *__restrict__ vals) {
    __shared__ uint64_t T[64];
    mod_table_fill(T);
    __syncthreads();
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tips) return;
    const int prio[4] = {prio4.x, prio4.y, prio4.z, prio4.w};
    uint64_t h[PKB_MAX_INTERVALS]; int32_t p[PKB_MAX_INTERVALS];
    const uint32_t i = tips[t];
    const int len = nd.len[i];
    const int cnt = li_kmers(nd.words + (size_t) i * nd.stride, len, c.li_k, c.li_intervals, prio, h, p, T);
    const uint32_t base = koff[t];
    for (int j = 0; j < cnt; j++) {
        keys[base + j] = h[j];
        vals[base + j] = ((unsigned long long) (4095 - p[j]) << 40) | ((unsigned long long) (uint32_t) len << 28) | i;
    }
}

// The k-mer entries are radix-sorted on the LOW `bits` bits of the hash only (half the passes of a full 60-bit sort).  A run of equal
// low bits nearly always is one group; where two hashes share their low bits (n^2 / 2^(bits+1) pairs) the run is ordered by the full
// hash here, so that equal hashes are contiguous for everything downstream.
__global__ void __launch_bounds__(256) k_pkb_fix_runs(unsigned long long *__restrict__ keys, unsigned long long *__restrict__ vals, uint64_t n, int bits) {
    const unsigned long long lm = bits >= 64 ? ~0ull : ((1ull << bits) - 1ull);
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
        const unsigned long long k0 = keys[i];
        if (i > 0 && (keys[i - 1] & lm) == (k0 & lm)) continue;                  // not the first entry of its run
        uint64_t e = i + 1;
        bool mixed = false;
        while (e < n) { const unsigned long long k = keys[e]; if ((k & lm) != (k0 & lm)) break; mixed |= k != k0; e++; }
        if (!mixed) continue;
        for (uint64_t a = i + 1; a < e; a++) {                                  // insertion sort of the run by the full hash
            const unsigned long long k = keys[a], v = vals[a];
            uint64_t b = a;
            while (b > i && keys[b - 1] > k) { keys[b] = keys[b - 1]; vals[b] = vals[b - 1]; b--; }
            keys[b] = k; vals[b] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------
// k_pkb_groups: one thread per group of equal hash.
//   createAlignmentsForKmers (PairwiseKmerBranch.cpp:16-97): entries ordered by (indInRead desc, read length asc, id asc);
//   i from the last-but-one down to the first is the "from" k-mer, j > i the "to" k-mers; offset = ind_i - ind_j.
//   branchMarkers rows are 64-bit masks kept in marks[] (one word per entry); groups larger than 64 use rows of
//   ceil(D/64) words carved from big_marks.
//   The groups are handed out in order of their size (heads sorted by D): the lanes of a wave then run loops of the same length --
//   in entry order nearly every wave held one group of 8+ entries and 60 of two, and waited for it.
//   New edges are written as keys to add_keys at [2 * group_start ...) (capacity 2 * D per group, their count to n_add[t]); the
//   rare overflow goes through an atomic cursor behind the dense part.
// ------------------------------------------------------------------------------------------
struct PkbGraph { const uint32_t *rowptr; const unsigned long long *keys; };   // snapshot of the round's start

__device__ __forceinline__ int snapshot_offset(const PkbGraph &g, int a, int b) {
    uint32_t lo = g.rowptr[a], hi = g.rowptr[a + 1];
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        const unsigned long long k = g.keys[mid];
        const int d = pkb_key_dst(k);
        if (d == b) return pkb_key_off(k);
        if (d < b) lo = mid + 1; else hi = mid;
    }
    return PKB_INF;
}

__global__ void __launch_bounds__(256) k_pkb_group_sizes(const unsigned long long *__restrict__ keys, uint64_t n,
                                                          unsigned long long *__restrict__ big_words /* total words for groups > 64 */,
                                                          unsigned long long *__restrict__ stats /* [0] groups >= 2, [1] max D */,
                                                          uint32_t *__restrict__ head_flag /* 1 = entry heads a group of >= 2 */,
                                                          uint32_t *__restrict__ gsize /* at such an entry: min(D, 255) */) {
    // per-thread tallies, one atomic per workgroup at the end (a contended atomic per group cost 0.8 ms per round)
    unsigned long long n2 = 0, mx = 0, big = 0;
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
        if (i > 0 && keys[i] == keys[i - 1]) { head_flag[i] = 0u; continue; }
        uint64_t e = i + 1;
        while (e < n && keys[e] == keys[i]) e++;
        const uint64_t D = e - i;
        head_flag[i] = D >= 2 ? 1u : 0u;
        gsize[i] = (uint32_t) (D < 255 ? D : 255);
        n2 += D >= 2;
        mx = D > mx ? D : mx;
        if (D > 64) big += D * ((D + 63) / 64);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        n2 += __shfl_xor(n2, o);
        big += __shfl_xor(big, o);
        const unsigned long long t = __shfl_xor(mx, o);
        mx = t > mx ? t : mx;
    }
    __shared__ unsigned long long s_n2[4], s_mx[4], s_big[4];
    const int wv = (int) (threadIdx.x >> 6);
    if ((threadIdx.x & 63u) == 0) { s_n2[wv] = n2; s_mx[wv] = mx; s_big[wv] = big; }
    __syncthreads();
    if (threadIdx.x == 0) {                      // same-address atomics retire at ~88 per microsecond chip-wide: one set per workgroup
        for (int k = 1; k < 4; k++) { n2 += s_n2[k]; big += s_big[k]; mx = s_mx[k] > mx ? s_mx[k] : mx; }
        if (n2) atomicAdd(&stats[0], n2);
        if (mx) atomicMax(&stats[1], mx);
        if (big) atomicAdd(big_words, big);
    }
}

// dense list of the entries that head a group of >= 2 (flags from k_pkb_group_sizes, positions from their scan) + their sizes as sort keys
__global__ void __launch_bounds__(256) k_pkb_head_list(const uint32_t *__restrict__ head_flag, const uint32_t *__restrict__ pos, const uint32_t *__restrict__ gsize,
                                                        uint64_t n, uint32_t *__restrict__ heads, uint32_t *__restrict__ hsize) {
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x)
        if (head_flag[i]) { heads[pos[i]] = (uint32_t) i; hsize[pos[i]] = gsize[i]; }
}

// one group; returns the number of canAlign calls, *n_added = additions kept in the dense slots
__device__ __forceinline__ unsigned long long pkb_group(const NodesDev &nd, const PkbCfg &c, const PkbGraph &g, const unsigned long long *__restrict__ keys,
                                                       unsigned long long *__restrict__ vals, uint64_t n, uint64_t gs, int D, unsigned long long *__restrict__ marks,
                                                       unsigned long long *__restrict__ big_marks, unsigned long long *__restrict__ big_cursor,
                                                       unsigned long long *__restrict__ add_keys, uint64_t add_dense, uint64_t add_cap,
                                                       unsigned long long *__restrict__ add_overflow, uint32_t *n_added) {
    if (D >= 255) {                                                          // the size key saturates: count
        uint64_t ge = gs + 1;
        while (ge < n && keys[ge] == keys[gs]) ge++;
        D = (int) (ge - gs);
    }
    unsigned long long *v = vals + gs;
    for (int i = 1; i < D; i++) {                                            // order the group: ascending val
        const unsigned long long x = v[i];
        int j = i;
        while (j > 0 && v[j - 1] > x) { v[j] = v[j - 1]; j--; }
        v[j] = x;
    }
    // does any read occur twice in the group?  (then additions made inside the group must be visible to later pairs)
    bool dup = false;
    for (int i = 0; i < D && !dup; i++) for (int j = i + 1; j < D; j++) if (pkb_val_id(v[i]) == pkb_val_id(v[j])) { dup = true; break; }
    const int RW = (D + 63) >> 6;                                            // words per branch-marker row
    unsigned long long *rows;
    if (D <= 64) rows = marks + gs;
    else rows = big_marks + atomicAdd(big_cursor, (unsigned long long) ((uint64_t) D * RW));
    for (int i = 0; i < D * RW; i++) rows[i] = 0ull;
    unsigned long long *mine = add_keys + 2 * gs;                            // dense slots of this group: 2 * D
    int n_add = 0;
    unsigned long long calls = 0;
    for (int i = D - 2; i >= 0; i--) {
        const unsigned long long vi = v[i];
        const int id1 = pkb_val_id(vi), ind1 = pkb_val_ind(vi), len1 = pkb_val_len(vi);
        unsigned long long *row_i = rows + (size_t) i * RW;
        for (int j = i + 1; j < D; j++) {
            const unsigned long long vj = v[j];
            const int id2 = pkb_val_id(vj);
            if (id1 == id2) continue;
            const int off = ind1 - pkb_val_ind(vj);
            if (off < 0) continue;
            if (100 * off > c.max_offset_pct * len1) break;                  // :55
            const int len2 = pkb_val_len(vj);
            const int ov = (len1 < len2 + off ? len1 : len2 + off) - off;
            if (ov < c.min_overlap_area) continue;
            if (len2 + off - len1 < 0) continue;
            if ((row_i[j >> 6] >> (j & 63)) & 1ull) continue;                // already reachable inside the group (:62)
            int cur = snapshot_offset(g, id1, id2);                          // neighbors[id2]
            if (dup) {                                                       // additions this group already made (dense slots only)
                const int lim = n_add < 2 * D ? n_add : 2 * D;
                for (int t = 0; t < lim; t++)
                    if (pkb_key_src(mine[t]) == id1 && pkb_key_dst(mine[t]) == id2 && pkb_key_off(mine[t]) < cur) cur = pkb_key_off(mine[t]);
            }
            if (cur > off) {
                calls++;
                if (can_align(nd, id1, id2, off, c)) {                       // :66
                    const unsigned long long ne = pkb_edge_key(id1, id2, off);
                    if (n_add < 2 * D) mine[n_add] = ne;
                    else {
                        const unsigned long long k = atomicAdd(add_overflow, 1ull);
                        if (add_dense + k < add_cap) add_keys[add_dense + k] = ne;
                    }
                    n_add++;
                    cur = off;
                }
            }
            if (cur != PKB_INF) {                                            // :73-77
                row_i[j >> 6] |= 1ull << (j & 63);
                const unsigned long long *row_j = rows + (size_t) j * RW;
                for (int t = 0; t < RW; t++) row_i[t] |= row_j[t];
            }
        }
    }
    *n_added = (uint32_t) (n_add < 2 * D ? n_add : 2 * D);
    return calls;
}

__global__ void __launch_bounds__(64) k_pkb_groups(NodesDev nd, PkbCfg c, PkbGraph g, const unsigned long long *__restrict__ keys,
                                                    const uint32_t *__restrict__ heads, const uint32_t *__restrict__ hsize, uint32_t n_heads,
                                                    unsigned long long *__restrict__ vals, uint64_t n, unsigned long long *__restrict__ marks,
                                                    unsigned long long *__restrict__ big_marks, unsigned long long *__restrict__ big_cursor,
                                                    unsigned long long *__restrict__ add_keys, uint64_t add_dense, uint64_t add_cap,
                                                    unsigned long long *__restrict__ add_overflow, unsigned long long *__restrict__ counters,
                                                    uint32_t *__restrict__ n_add) {
    unsigned long long calls = 0;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n_heads) {
        uint32_t na = 0;
        calls = pkb_group(nd, c, g, keys, vals, n, (uint64_t) heads[t], (int) hsize[t], marks, big_marks, big_cursor, add_keys, add_dense, add_cap, add_overflow, &na);
        n_add[t] = na;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) calls += __shfl_xor(calls, o);          // one counter update per wave
    if ((threadIdx.x & 63u) == 0 && calls) atomicAdd(&counters[0], calls);
}

// the additions of a round as one dense key list: group t's n_add[t] dense slots at pos[t], the overflow entries behind them
__global__ void __launch_bounds__(256) k_pkb_gather_adds(const uint32_t *__restrict__ heads, const uint32_t *__restrict__ n_add, const uint32_t *__restrict__ pos,
                                                          uint32_t n_heads, const unsigned long long *__restrict__ add_keys, uint64_t add_dense, uint64_t n_dense_total,
                                                          uint64_t n_ovf, unsigned long long *__restrict__ out) {
    const uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_heads) {
        const uint32_t k = n_add[i];
        const unsigned long long *mine = add_keys + 2 * (uint64_t) heads[i];
        for (uint32_t t = 0; t < k; t++) out[pos[i] + t] = mine[t];
    }
    if (i < n_ovf) out[n_dense_total + i] = add_keys[add_dense + i];
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
static inline unsigned pkb_grid(uint64_t n, int block, unsigned cap) { return (unsigned) std::max<uint64_t>(1, std::min<uint64_t>((n + block - 1) / block, cap)); }

void launch_can_align_batch(const NodesDev &nd, const PkbCfg &c, const int32_t *triples, uint64_t n, uint8_t *out, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_can_align_batch, dim3(pkb_grid(n, 256, 16384)), dim3(256), 0, s, nd, c, triples, n, out);
}

void launch_li_kmers_slots(const NodesDev &nd, const PkbCfg &c, const int32_t prio[4], uint64_t *hash, int32_t *ind, int32_t *count, hipStream_t s) {
    if (nd.n <= 0) return;
    hipLaunchKernelGGL(k_li_kmers_slots, dim3((nd.n + 255) / 256), dim3(256), 0, s, nd, c, make_int4(prio[0], prio[1], prio[2], prio[3]), hash, ind, count);
}

void launch_pkb_edge_keys(const alga_edge_dev *e, uint64_t n, unsigned long long *keys, unsigned long long *counts, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_pkb_edge_keys, dim3(pkb_grid(n, 256, 8192)), dim3(256), 0, s, e, n, keys, counts);
}

void launch_pkb_rowptr(const unsigned long long *keys, uint64_t E, int32_t n, uint32_t *rowptr, hipStream_t s) {
    hipLaunchKernelGGL(k_pkb_rowptr, dim3(pkb_grid(E + 1, 256, 16384)), dim3(256), 0, s, keys, E, n, rowptr);
}

void launch_pkb_keys_to_edges(const unsigned long long *keys, uint64_t E, alga_edge_dev *out, hipStream_t s) {
    if (E == 0) return;
    hipLaunchKernelGGL(k_pkb_keys_to_edges, dim3(pkb_grid(E, 256, 16384)), dim3(256), 0, s, keys, E, out);
}

void launch_pkb_masks(int32_t n, const uint32_t *rowptr, const unsigned long long *keys, uint64_t m, uint32_t *indeg, uint8_t *mask, hipStream_t s) {
    if (n <= 0) return;
    (void) hipMemsetAsync(indeg, 0, sizeof(uint32_t) * (size_t) n, s);
    if (m) hipLaunchKernelGGL(k_pkb_indeg, dim3(pkb_grid(m, 256, 8192)), dim3(256), 0, s, keys, m, indeg);
    hipLaunchKernelGGL(k_pkb_masks, dim3((n + 255) / 256), dim3(256), 0, s, n, rowptr, indeg, mask);
}

void launch_pkb_tip_flags(const NodesDev &nd, const PkbCfg &c, const uint8_t *mask, uint32_t *flag, hipStream_t s) {
    if (nd.n <= 0) return;
    hipLaunchKernelGGL(k_pkb_tip_flags, dim3((nd.n + 255) / 256), dim3(256), 0, s, nd, c, mask, flag);
}

void launch_pkb_tip_list(const NodesDev &nd, const PkbCfg &c, const uint32_t *flag, const uint32_t *pos, uint32_t *tips, uint32_t *kcount,
                         unsigned long long *max_len, hipStream_t s) {
    if (nd.n <= 0) return;
    hipLaunchKernelGGL(k_pkb_tip_list, dim3((nd.n + 255) / 256), dim3(256), 0, s, nd, c, flag, pos, tips, kcount, max_len);
}

void launch_pkb_kmers(const NodesDev &nd, const PkbCfg &c, const int32_t prio[4], const uint32_t *tips, const uint32_t *koff, uint32_t n_tips,
                      unsigned long long *keys, unsigned long long *vals, hipStream_t s) {
    if (n_tips == 0) return;
    hipLaunchKernelGGL(k_pkb_kmers, dim3((n_tips + 255) / 256), dim3(256), 0, s, nd, c, make_int4(prio[0], prio[1], prio[2], prio[3]), tips, koff, n_tips, keys, vals);
}

void launch_pkb_fix_runs(unsigned long long *keys, unsigned long long *vals, uint64_t n, int bits, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_pkb_fix_runs, dim3(pkb_grid(n, 256, 16384)), dim3(256), 0, s, keys, vals, n, bits);
}

void launch_pkb_group_sizes(const unsigned long long *keys, uint64_t n, unsigned long long *big_words, unsigned long long *stats, uint32_t *head_flag,
                            uint32_t *gsize, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_pkb_group_sizes, dim3(pkb_grid(n, 256 * 8, 1024)), dim3(256), 0, s, keys, n, big_words, stats, head_flag, gsize);
}

void launch_pkb_head_list(const uint32_t *head_flag, const uint32_t *pos, const uint32_t *gsize, uint64_t n, uint32_t *heads, uint32_t *hsize, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_pkb_head_list, dim3(pkb_grid(n, 256, 8192)), dim3(256), 0, s, head_flag, pos, gsize, n, heads, hsize);
}

void launch_pkb_groups(const NodesDev &nd, const PkbCfg &c, const uint32_t *rowptr, const unsigned long long *gkeys, const unsigned long long *keys,
                       const uint32_t *heads, const uint32_t *hsize, uint32_t n_heads, unsigned long long *vals, uint64_t n, unsigned long long *marks,
                       unsigned long long *big_marks, unsigned long long *big_cursor, unsigned long long *add_keys, uint64_t add_dense, uint64_t add_cap,
                       unsigned long long *add_overflow, unsigned long long *counters, uint32_t *n_add, hipStream_t s) {
    if (n == 0 || n_heads == 0) return;
    PkbGraph g{rowptr, gkeys};
    hipLaunchKernelGGL(k_pkb_groups, dim3((n_heads + 63) / 64), dim3(64), 0, s, nd, c, g, keys, heads, hsize, n_heads, vals, n, marks, big_marks, big_cursor,
                       add_keys, add_dense, add_cap, add_overflow, counters, n_add);
}

void launch_pkb_gather_adds(const uint32_t *heads, const uint32_t *n_add, const uint32_t *pos, uint32_t n_heads, const unsigned long long *add_keys,
                            uint64_t add_dense, uint64_t n_dense_total, uint64_t n_ovf, unsigned long long *out, hipStream_t s) {
    const uint64_t m = std::max<uint64_t>(n_heads, n_ovf);
    if (m == 0) return;
    hipLaunchKernelGGL(k_pkb_gather_adds, dim3((unsigned) ((m + 255) / 256)), dim3(256), 0, s, heads, n_add, pos, n_heads, add_keys, add_dense, n_dense_total, n_ovf, out);
}

} // namespace alga
