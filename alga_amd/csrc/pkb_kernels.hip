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
    const uint32_t pp = (uint32_t) prio[0] | ((uint32_t) prio[1] << 2) | ((uint32_t) prio[2] << 4) | ((uint32_t) prio[3] << 6);
    auto digit = [&](int pos) { return (u128) ((pp >> (((row[pos >> 4] >> ((pos & 15) << 1)) & 3u) << 1)) & 3u); };
    u128 h = 0;
    for (int q = 0; q < k; q++) h = (h << 2) + digit(q);
    const u128 low_mask = (((u128) 1) << (2 * (k - 1))) - 1;                  // factor - 1, factor = 4^(k-1)
    const int il = (len - k + 1 + intervals - 1) / intervals;                 // ceil((size - length + 1) / intervals)
    u128 best = h; int best_p = 0, cnt = 0;
    int next = il;                                                            // first start position of the next interval (p / il changes there)
    for (int p = 1; p + k <= len; p++) {
        h = ((h & low_mask) << 2) + digit(p + k - 1);                         // hash -= factor * first; hash <<= 2; hash += next
        if (p == next) { hash_out[cnt] = mh(best); ind_out[cnt] = best_p; cnt++; next += il; best = h; best_p = p; }
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

// The additions of a round in key order without a 61-bit sort (round 5; the library's took twelve kernels, 0.6 ms for 3.6 M keys): the engine's
// (u32, u32) radix sort orders (src, position) on the node-id bits alone, the keys follow their positions, and every run of one src -- a handful
// of keys -- is put in order by the one thread that finds its start.
__global__ void __launch_bounds__(256) k_pkb_src_keys(const unsigned long long *__restrict__ keys, uint64_t n, int shift, uint32_t *__restrict__ k32) {
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x)
        k32[i] = (uint32_t) pkb_key_src(keys[i]) << shift;
}

// the thread at the first position of a src run (read off the SORTED 32-bit keys) fetches the run's keys by their positions, orders them in
// registers (runs of up to eight keys: all but a handful) and writes them where they belong
__global__ void __launch_bounds__(256) k_pkb_gather_sorted_runs(const unsigned long long *__restrict__ keys, const uint32_t *__restrict__ k32s, const uint32_t *__restrict__ idx,
                                                                 uint64_t n, unsigned long long *__restrict__ out) {
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
        const uint32_t ks = k32s[i];
        if (i > 0 && k32s[i - 1] == ks) continue;                                // not the first key of its run
        uint64_t e = i + 1;
        while (e < n && k32s[e] == ks) e++;
        const int L = (int) min(e - i, (uint64_t) 9);
        if (L <= 8) {
            unsigned long long k[8];
#pragma unroll
            for (int q = 0; q < 8; q++) k[q] = q < L ? keys[idx[i + q]] : ~0ull;
#pragma unroll
            for (int a = 0; a < 8; a++)                                          // odd-even transposition: eight keys, sorted after eight rounds
#pragma unroll
                for (int b = (a & 1); b + 1 < 8; b += 2) { const unsigned long long lo = min(k[b], k[b + 1]), hi = max(k[b], k[b + 1]); k[b] = lo; k[b + 1] = hi; }
#pragma unroll
            for (int q = 0; q < 8; q++) if (q < L) out[i + q] = k[q];
        } else {
            for (uint64_t a = i; a < e; a++) {                                    // a long run: insertion sort while gathering (nobody else touches out[i .. e))
                const unsigned long long kx = keys[idx[a]];
                uint64_t b = a;
                while (b > i && out[b - 1] > kx) { out[b] = out[b - 1]; b--; }
                out[b] = kx;
            }
        }
    }
}

// "first key of every (src, dst) run" of a sorted key list (retainOnlySmallestOffset after a merge) in flags -> scan -> scatter, the scatter
// writing the row pointers of the result as it goes (round 5: the library's unique took 0.19 ms per round at 9 M keys, the row-pointer pass 0.06)
__global__ void __launch_bounds__(256) k_pkb_unique_flags(const unsigned long long *__restrict__ in, uint64_t n, uint32_t *__restrict__ flag) {
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x)
        flag[i] = (i == 0 || (in[i] >> 9) != (in[i - 1] >> 9)) ? 1u : 0u;
}

// pos: exclusive scan of flag (pos[n] = number of keys kept).  rowptr[s] = first kept key with src >= s, rowptr[n_nodes] = the number kept
__global__ void __launch_bounds__(256) k_pkb_unique_scatter(const unsigned long long *__restrict__ in, uint64_t n, const uint32_t *__restrict__ flag,
                                                             const uint32_t *__restrict__ pos, int32_t n_nodes, unsigned long long *__restrict__ out,
                                                             uint32_t *__restrict__ rowptr) {
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += (uint64_t) gridDim.x * blockDim.x) {
        if (i < n && !flag[i]) continue;                                          // (a dropped key has the src of the key before it: no row starts here)
        const int s_prev = i == 0 ? -1 : pkb_key_src(in[i - 1]);
        const int s_cur = i == n ? n_nodes : pkb_key_src(in[i]);
        const uint32_t at = pos[i];
        if (i < n) out[at] = in[i];
        for (int s = s_prev + 1; s <= s_cur; s++) rowptr[s] = at;
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
                                                       uint32_t *__restrict__ tips, uint32_t *__restrict__ kcount, unsigned long long *__restrict__ max_len,
                                                       uint32_t *__restrict__ tipidx /* node -> its place in tips[], ~0: takes no part */) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int len = 0;
    if (i < nd.n) tipidx[i] = flag[i] ? pos[i] : 0xFFFFFFFFu;
    if (i < nd.n && flag[i]) {
        len = nd.len[i];
        const int il = (len - c.li_k + 1 + c.li_intervals - 1) / c.li_intervals;
        tips[pos[i]] = (uint32_t) i;
        kcount[pos[i]] = (uint32_t) ((len - c.li_k) / il + 1);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const int t = __shfl_xor(len, o); len = t > len ? t : len; }
    // same-address atomics retire at ~88 per microsecond chip-wide (one per wave here was 3.5 ms at 20 M nodes) and reads of one hot
    // address are not much better: one guarded update per workgroup
    __shared__ int s_len[4];
    if ((threadIdx.x & 63u) == 0) s_len[threadIdx.x >> 6] = len;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; k++) len = s_len[k] > len ? s_len[k] : len;
        if (len > 0 && (unsigned long long) len > __hip_atomic_load(max_len, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(max_len, (unsigned long long) len);
    }
}

// A k-mer entry: key = hash, val = (4095 - indInRead) << 40 | read length << 28 | tip (the node's position in the dense tip list = its
// record, PkbTipRec below; the list ascends with the node id).  Ascending val == the order of the reference inside a group (indInRead
// descending, read length ascending; Kmer::operator<, Kmer.cpp:58-64) with ties by node id.
constexpr unsigned long long PKB_ID_MASK = (1ull << 28) - 1;
__device__ __forceinline__ int pkb_val_id(unsigned long long v) { return (int) (v & PKB_ID_MASK); }
__device__ __forceinline__ int pkb_val_len(unsigned long long v) { return (int) ((v >> 28) & 0xFFFull); }
__device__ __forceinline__ int pkb_val_ind(unsigned long long v) { return 4095 - (int) (v >> 40); }

// k-mers of every node that takes part (GraphCreatorKmerBased::getKmersForBucketJob :202-259) at the tip's fixed offset.
// STAGED (rows of up to 16 words): the workgroup's rows go through LDS -- one thread walking its own row in global memory costs
// a 64-line access per load instruction of the wave (116 of them per row: 0.7 of the kernel's 1.25 ms at 3.6 M tips).
// The sort key of an entry is hash * PKB_KEY_MIX mod 2^64 (odd multiplier: a bijection, equal keys == equal hashes), rotated so
// that the product's top `sort_bits` bits are the key's low bits: the entries are radix-sorted on those only.  The hash itself will
// not do: it is the k-mer's base-4 value when that is
// below 10^18 + 3 -- true of most minimizers -- so its low 32 bits are the last 16 nucleotides, shared by every k-mer that differs
// from another by a sequencing error further left (1.4 M such neighbours among 21 M entries).
constexpr unsigned long long PKB_KEY_MIX = 0x9E3779B97F4A7C15ull;
constexpr int PKB_ROW_WORDS = 16;
// a node that takes part, in one 128-byte line (see "Tip records" below)
constexpr int PKB_REC_KEYS = 7;
struct __attribute__((aligned(128))) PkbTipRec {
    uint32_t words[PKB_ROW_WORDS];                       // the node's row (zero behind its last word)
    uint32_t id;                                         // node id
    uint32_t nkeys;                                      // out-edges in the round's snapshot; the first PKB_REC_KEYS of them:
    unsigned long long keys[PKB_REC_KEYS];
};
static_assert(sizeof(PkbTipRec) == 128, "one record == one 128-byte line");

// li_kmers for k <= 48 on a staged row, the rolling value in three 32-bit words (round 5: the 128-bit form cost 37 vector instructions per start
// position -- a quarter of them moving the upper 58 bits of zeros -- and made the k-mer kernel VALU-bound at 0.5 ms per round).  Same walk, same
// ties (strictly smaller replaces), same interval borders; returns the k-mers written.
__device__ __forceinline__ int li_kmers96(const uint32_t *row, int len, int k, int intervals, uint32_t pp /* prio[s] at bits 2 s */, int wave_max_len,
                                          uint64_t *hash_out, int32_t *ind_out, const uint64_t *T) {
    typedef unsigned __int128 u128;
    if (k > len || intervals <= 0) return 0;
    const int nb = 2 * k;                                                      // bits of a k-mer value
    const uint32_t m0 = nb >= 32 ? 0xFFFFFFFFu : ((1u << nb) - 1u);
    const uint32_t m1 = nb >= 64 ? 0xFFFFFFFFu : (nb > 32 ? ((1u << (nb - 32)) - 1u) : 0u);
    const uint32_t m2 = nb > 64 ? ((1u << (nb - 64)) - 1u) : 0u;
    uint32_t h0 = 0u, h1 = 0u, h2 = 0u;
    uint32_t cur = 0u;
    auto step = [&](int pos) {                                                 // append the digit at `pos` (pos is the same in every lane)
        if ((pos & 15) == 0) cur = row[pos >> 4];
        const uint32_t d = __builtin_amdgcn_ubfe(cur, (uint32_t) ((pos & 15) << 1), 2u);
        const uint32_t dm = __builtin_amdgcn_ubfe(pp, d << 1, 2u);
        h2 = __funnelshift_l(h1, h2, 2) & m2;
        h1 = __funnelshift_l(h0, h1, 2) & m1;
        h0 = ((h0 << 2) | dm) & m0;
    };
    for (int q = 0; q < k; q++) step(q);
    const int il = (len - k + 1 + intervals - 1) / intervals;
    uint32_t b0 = h0, b1 = h1, b2 = h2;
    int best_p = 0, cnt = 0, next = il;
    auto emit = [&]() {
        const u128 v = ((u128) b2 << 64) | ((u128) b1 << 32) | (u128) b0;
        hash_out[cnt] = (T != nullptr && k <= 35) ? mod_hash_u70(v, T) : mod_hash_u128(v);
        ind_out[cnt] = best_p; cnt++;
    };
    for (int p = 1; p + k <= wave_max_len; p++) {
        step(p + k - 1);
        if (p + k <= len) {
            const unsigned long long hl = ((unsigned long long) h1 << 32) | h0, bl = ((unsigned long long) b1 << 32) | b0;
            if (p == next) { emit(); next += il; b0 = h0; b1 = h1; b2 = h2; best_p = p; }
            else if (h2 < b2 || (h2 == b2 && hl < bl)) { b0 = h0; b1 = h1; b2 = h2; best_p = p; }
        }
    }
    emit();
    return cnt;
}

template <bool STAGED>
__global__ void __launch_bounds__(256) k_pkb_kmers(NodesDev nd, PkbCfg c, int4 prio4, const uint32_t *__restrict__ tips, const uint32_t *__restrict__ koff,
                                                    uint32_t n_tips, int sort_bits /* 1 .. 63 */, unsigned long long *__restrict__ keys,
                                                    unsigned long long *__restrict__ vals, const PkbTipRec *__restrict__ rec, int wide /* 1: the 128-bit walk */) {
    __shared__ uint64_t T[64];
    __shared__ uint32_t srow[STAGED ? 256 : 1][PKB_ROW_WORDS + 1];
    mod_table_fill(T);
    const uint32_t base_t = blockIdx.x * blockDim.x;
    if (STAGED) {
        const int pc = (int) (threadIdx.x & 3u);                              // (dense: the block's 256 records are 32 KB in a row; 16-byte pieces)
        for (int r = (int) (threadIdx.x >> 2); r < 256; r += 64) {
            const uint32_t t = base_t + (uint32_t) r;
            const uint4 x = t < n_tips ? reinterpret_cast<const uint4 *>(rec[t].words)[pc] : make_uint4(0u, 0u, 0u, 0u);
            srow[r][4 * pc] = x.x; srow[r][4 * pc + 1] = x.y; srow[r][4 * pc + 2] = x.z; srow[r][4 * pc + 3] = x.w;
        }
    }
    __syncthreads();
    const uint32_t t = base_t + threadIdx.x;
    const int prio[4] = {prio4.x, prio4.y, prio4.z, prio4.w};
    uint64_t h[PKB_MAX_INTERVALS]; int32_t p[PKB_MAX_INTERVALS];
    const uint32_t i = t < n_tips ? tips[t] : 0u;
    const int len = t < n_tips ? nd.len[i] : 0;
    int cnt;
    if (STAGED && !wide && c.li_k <= 48) {
        int wl = len;                                                          // the longest read of the wave: every lane walks that far
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) wl = max(wl, __shfl_xor(wl, o));
        cnt = li_kmers96(srow[threadIdx.x], len, c.li_k, c.li_intervals, (uint32_t) prio[0] | ((uint32_t) prio[1] << 2) | ((uint32_t) prio[2] << 4) | ((uint32_t) prio[3] << 6),
                         wl, h, p, T);
        if (t >= n_tips) return;
    } else {
        if (t >= n_tips) return;
        const uint32_t *row = STAGED ? srow[threadIdx.x] : nd.words + (size_t) i * nd.stride;
        cnt = li_kmers(row, len, c.li_k, c.li_intervals, prio, h, p, T);
    }
    const uint32_t base = koff[t];
    for (int j = 0; j < cnt; j++) {
        const unsigned long long m = h[j] * PKB_KEY_MIX;
        keys[base + j] = (m << sort_bits) | (m >> (64 - sort_bits));            // the product's top bits, where the sort looks
        vals[base + j] = ((unsigned long long) (4095 - p[j]) << 40) | ((unsigned long long) (uint32_t) len << 28) | t;      // t: the tip's record (ascends with the node id)
    }
}

// The k-mers of ALL rounds in one walk (round 5).  The rounds differ in the alphabet priority alone (GraphCreatorLI.cpp:26 rotates it by one per
// round: round r reads prio[(s + r) & 3]), the tips and their rows are the same: one staging of the rows, one loop over the start positions with
// R rolling values side by side -- the digit is extracted once, the interval borders are the same for every round, the loop's scalar bookkeeping
// (a third of the per-round kernel's issue slots) is paid once.  k-mers go straight to memory when an interval closes (R x 16 buffered hashes would
// not fit the registers).  keys / vals: R arrays `round_stride` entries apart.  Rows staged (stride <= PKB_ROW_WORDS), k <= 48.
template <int R>
__global__ void __launch_bounds__(256) k_pkb_kmers_all(NodesDev nd, PkbCfg c, uint32_t pp0 /* prio[s] at bits 2 s, round 0 */, int rounds, const uint32_t *__restrict__ tips,
                                                        const uint32_t *__restrict__ koff, uint32_t n_tips, int sort_bits, unsigned long long *__restrict__ keys,
                                                        unsigned long long *__restrict__ vals, size_t round_stride, const PkbTipRec *__restrict__ rec) {
    typedef unsigned __int128 u128;
    __shared__ uint64_t T[64];
    __shared__ uint32_t srow[256][PKB_ROW_WORDS + 1];
    mod_table_fill(T);
    const uint32_t base_t = blockIdx.x * blockDim.x;
    {
        const int pc = (int) (threadIdx.x & 3u);                              // (the rows in 16-byte pieces: a quarter of the load instructions)
        for (int r = (int) (threadIdx.x >> 2); r < 256; r += 64) {
            const uint32_t t = base_t + (uint32_t) r;
            const uint4 x = t < n_tips ? reinterpret_cast<const uint4 *>(rec[t].words)[pc] : make_uint4(0u, 0u, 0u, 0u);
            srow[r][4 * pc] = x.x; srow[r][4 * pc + 1] = x.y; srow[r][4 * pc + 2] = x.z; srow[r][4 * pc + 3] = x.w;
        }
    }
    __syncthreads();
    const uint32_t t = base_t + threadIdx.x;
    const uint32_t node = t < n_tips ? tips[t] : 0u;
    const int len = t < n_tips ? nd.len[node] : 0;
    const int k = c.li_k, intervals = c.li_intervals;
    const bool valid = t < n_tips && k <= len && intervals > 0;
    int wl = len;                                                              // the longest read of the wave: every lane walks that far
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) wl = max(wl, __shfl_xor(wl, o));
    const int nb = 2 * k;
    const uint32_t m0 = nb >= 32 ? 0xFFFFFFFFu : ((1u << nb) - 1u);
    const uint32_t m1 = nb >= 64 ? 0xFFFFFFFFu : (nb > 32 ? ((1u << (nb - 32)) - 1u) : 0u);
    const uint32_t m2 = nb > 64 ? ((1u << (nb - 64)) - 1u) : 0u;
    uint32_t pp[R], h0[R], h1[R], h2[R], b0[R], b1[R], b2[R];
    int bp[R];
#pragma unroll
    for (int r = 0; r < R; r++) { pp[r] = ((pp0 >> (2 * r)) | (pp0 << (8 - 2 * r))) & 0xFFu; h0[r] = h1[r] = h2[r] = 0u; bp[r] = 0; }
    const uint32_t *row = srow[threadIdx.x];
    uint32_t cur = 0u;
    auto step = [&](int pos) {                                                 // append the digit at `pos` to every round's value (pos: the same in every lane)
        if ((pos & 15) == 0) cur = row[pos >> 4];
        const uint32_t d2 = __builtin_amdgcn_ubfe(cur, (uint32_t) ((pos & 15) << 1), 2u) << 1;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const uint32_t dm = __builtin_amdgcn_ubfe(pp[r], d2, 2u);
            h2[r] = __funnelshift_l(h1[r], h2[r], 2) & m2;
            h1[r] = __funnelshift_l(h0[r], h1[r], 2) & m1;
            h0[r] = ((h0[r] << 2) | dm) & m0;
        }
    };
    for (int q = 0; q < k; q++) step(q);
#pragma unroll
    for (int r = 0; r < R; r++) { b0[r] = h0[r]; b1[r] = h1[r]; b2[r] = h2[r]; }
    const int il = valid ? (len - k + 1 + intervals - 1) / intervals : 1;
    int cnt = 0, next = il;
    const uint32_t kbase = valid ? koff[t] : 0u;
    auto emit = [&]() {                                                        // the interval's k-mer of every round -> memory
#pragma unroll
        for (int r = 0; r < R; r++) {
            if (r >= rounds) continue;
            const u128 v = ((u128) b2[r] << 64) | ((u128) b1[r] << 32) | (u128) b0[r];
            const unsigned long long hsh = k <= 35 ? mod_hash_u70(v, T) : mod_hash_u128(v);
            const unsigned long long m = hsh * PKB_KEY_MIX;
            const size_t at = (size_t) r * round_stride + kbase + (uint32_t) cnt;
            keys[at] = (m << sort_bits) | (m >> (64 - sort_bits));
            vals[at] = ((unsigned long long) (4095 - bp[r]) << 40) | ((unsigned long long) (uint32_t) len << 28) | t;
        }
        cnt++;
    };
    for (int p = 1; p + k <= wl; p++) {
        step(p + k - 1);
        const bool inr = valid && p + k <= len;
        const bool border = inr && p == next;
        if (__ballot(border) != 0ull) {                                        // uniform: an interval closes in some lane
            if (border) {
                emit();
                next += il;
#pragma unroll
                for (int r = 0; r < R; r++) { b0[r] = h0[r]; b1[r] = h1[r]; b2[r] = h2[r]; bp[r] = p; }
            }
        }
#pragma unroll
        for (int r = 0; r < R; r++) {
            const unsigned long long hl = ((unsigned long long) h1[r] << 32) | h0[r], bl = ((unsigned long long) b1[r] << 32) | b0[r];
            const bool lt = inr & !border & ((h2[r] < b2[r]) | ((h2[r] == b2[r]) & (hl < bl)));
            b0[r] = lt ? h0[r] : b0[r]; b1[r] = lt ? h1[r] : b1[r]; b2[r] = lt ? h2[r] : b2[r]; bp[r] = lt ? p : bp[r];
        }
    }
    if (valid) emit();
}

// The k-mer entries are radix-sorted on the low `bits` bits of the key only (half the passes of a full sort).  A run of equal
// low bits nearly always is one group; where two keys share them (n^2 / 2^(bits+1) pairs) the run is ordered by the full
// key, so that equal hashes are contiguous for everything downstream.  k_pkb_fix_flag lists the places (one streaming pass, no
// loops), k_pkb_fix_apply repairs the listed runs; a list that overflows is answered with the loop form k_pkb_fix_runs.
__device__ __forceinline__ void pkb_sort_run(unsigned long long *keys, unsigned long long *vals, uint64_t r, uint64_t e) {
    for (uint64_t a = r + 1; a < e; a++) {                                      // insertion sort of the run by the full hash
        const unsigned long long k = keys[a], v = vals[a];
        uint64_t b = a;
        while (b > r && keys[b - 1] > k) { keys[b] = keys[b - 1]; vals[b] = vals[b - 1]; b--; }
        keys[b] = k; vals[b] = v;
    }
}

constexpr int PKB_FLAG_IPT = 16;                                              // entries per thread of k_pkb_fix_flag
__global__ void __launch_bounds__(256) k_pkb_fix_flag(const unsigned long long *__restrict__ keys, uint64_t n, int bits, uint32_t *__restrict__ list,
                                                       uint32_t list_cap, unsigned long long *__restrict__ counter) {
    // the places of a block of 4096 entries are collected in LDS and appended with ONE atomic (sorted on 30 bits, 21 M entries have 200 k such
    // places: an atomic each on the one counter would retire at ~90 per microsecond)
    __shared__ uint32_t s_list[256 * PKB_FLAG_IPT];
    __shared__ uint32_t s_n, s_base;
    const unsigned long long lm = bits >= 64 ? ~0ull : ((1ull << bits) - 1ull);
    if (threadIdx.x == 0) s_n = 0u;
    __syncthreads();
    const uint64_t base = (uint64_t) blockIdx.x * (256 * PKB_FLAG_IPT);
    unsigned long long k[PKB_FLAG_IPT], kp[PKB_FLAG_IPT];                  // every load on its way before the first compare
#pragma unroll
    for (int j = 0; j < PKB_FLAG_IPT; j++) {
        const uint64_t i = base + (uint64_t) j * 256 + threadIdx.x;
        const bool ok = i != 0 && i < n;
        k[j] = ok ? keys[i] : 0ull; kp[j] = ok ? keys[i - 1] : 0ull;
    }
#pragma unroll
    for (int j = 0; j < PKB_FLAG_IPT; j++)
        if (k[j] != kp[j] && (k[j] & lm) == (kp[j] & lm)) s_list[atomicAdd(&s_n, 1u)] = (uint32_t) (base + (uint64_t) j * 256 + threadIdx.x);
    __syncthreads();
    const uint32_t cnt = s_n;
    if (cnt == 0u) return;
    if (threadIdx.x == 0) s_base = (uint32_t) min(atomicAdd(counter, (unsigned long long) cnt), (unsigned long long) list_cap);
    __syncthreads();
    for (uint32_t q = threadIdx.x; q < cnt; q += 256) if (s_base + q < list_cap) list[s_base + q] = s_list[q];
}

// which listed place repairs its run: the first one of the run (decided while nobody writes: bit 31 of the list entry)
__global__ void __launch_bounds__(64) k_pkb_fix_owner(const unsigned long long *__restrict__ keys, int bits, uint32_t *__restrict__ list, uint32_t list_cap,
                                                       const unsigned long long *__restrict__ counter) {
    const unsigned long long lm = bits >= 64 ? ~0ull : ((1ull << bits) - 1ull);
    const unsigned long long cnt = *counter;
    if (cnt > list_cap) return;                                                 // the host runs k_pkb_fix_runs instead
    const uint64_t t = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= cnt) return;
    const uint64_t i = list[t];
    const unsigned long long low = keys[i] & lm;
    uint64_t r = i;
    while (r > 0 && (keys[r - 1] & lm) == low) r--;                             // start of the run
    bool first = true;
    for (uint64_t j = r + 1; j < i; j++) if (keys[j] != keys[j - 1]) { first = false; break; }
    if (first) list[t] = (uint32_t) i | 0x80000000u;
}

__global__ void __launch_bounds__(64) k_pkb_fix_apply(unsigned long long *__restrict__ keys, unsigned long long *__restrict__ vals, uint64_t n, int bits,
                                                       const uint32_t *__restrict__ list, uint32_t list_cap, const unsigned long long *__restrict__ counter) {
    const unsigned long long lm = bits >= 64 ? ~0ull : ((1ull << bits) - 1ull);
    const unsigned long long cnt = *counter;
    if (cnt > list_cap) return;
    const uint64_t t = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= cnt || !(list[t] & 0x80000000u)) return;
    const uint64_t i = list[t] & 0x7FFFFFFFu;
    const unsigned long long low = keys[i] & lm;                                // runs are disjoint: nobody else touches this one
    uint64_t r = i, e = i + 1;
    while (r > 0 && (keys[r - 1] & lm) == low) r--;
    while (e < n && (keys[e] & lm) == low) e++;
    pkb_sort_run(keys, vals, r, e);
}

__global__ void __launch_bounds__(256) k_pkb_fix_runs(unsigned long long *__restrict__ keys, unsigned long long *__restrict__ vals, uint64_t n, int bits) {
    const unsigned long long lm = bits >= 64 ? ~0ull : ((1ull << bits) - 1ull);
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
        const unsigned long long k0 = keys[i];
        if (i > 0 && (keys[i - 1] & lm) == (k0 & lm)) continue;                  // not the first entry of its run
        uint64_t e = i + 1;
        bool mixed = false;
        while (e < n) { const unsigned long long k = keys[e]; if ((k & lm) != (k0 & lm)) break; mixed |= k != k0; e++; }
        if (mixed) pkb_sort_run(keys, vals, i, e);
    }
}

// ------------------------------------------------------------------------------------------
// The pairwise join inside a group of equal hash.
//   createAlignmentsForKmers (PairwiseKmerBranch.cpp:16-97): entries ordered by (indInRead desc, read length asc, id asc);
//   i from the last-but-one down to the first is the "from" k-mer, j > i the "to" k-mers; offset = ind_i - ind_j; a pair is skipped
//   when j is already reachable from i inside the group (branchMarkers: one bit row per entry), tested with canAlign when the
//   graph has no edge of at most that offset yet.
// Groups are handed out in DESCENDING size (heads sorted by 255 - min(D, 255)): the long groups start first and the lanes of a wave
// replay groups of the same size.  Three kernels share the list:
//   k_pkb_groups_serial  D > 64, or a read twice in the group, or rows too long to stage: one thread, everything in global memory
//   k_pkb_groups_wave    8 <= D <= 64: one wave per group, the group's rows staged in LDS once; all pairs at once (lane = pair:
//                        offset look-up + canAlign speculatively), then the reference's i / j loops replayed on the ballot masks
//   k_pkb_groups_small   2 <= D <= 7: one thread per group, entries and marker rows in LDS / a register, rows staged per pair
// New edges are written as keys to add_keys at [2 * group_start ...) (capacity 2 * D per group, their count to n_add[t]); the
// rare overflow goes through an atomic cursor behind the dense part.
// ------------------------------------------------------------------------------------------
struct PkbGraph { const uint32_t *rowptr; const unsigned long long *keys; };   // snapshot of the round's start
constexpr int PKB_SMALL_MAX = 7;
constexpr int PKB_WAVE_MAX = 64;

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

// ------------------------------------------------------------------------------------------
// Tip records (round 5).  The group kernels used to fetch THREE isolated 128-byte lines per k-mer entry -- the node's row (64 bytes of one), its two
// row pointers (8 bytes of another) and its snapshot keys (a third) -- and ran at the rate HBM serves isolated lines (5.7 GB per round at 10 M
// reads).  Now every node that takes part has ONE 128-byte record: its row, its id and its first snapshot keys; a k-mer entry names the record
// (the tip's position in the dense tip list, which ascends with the node id: the order inside a group is unchanged) instead of the node.
// Rows longer than PKB_ROW_WORDS words are not staged at all (the serial kernel reads them through the node id).
// ------------------------------------------------------------------------------------------
// rows, once per supplement: four lanes per tip, sixteen bytes each (rows whose stride is not a multiple of four words: word by word)
__global__ void __launch_bounds__(256) k_pkb_tiprec_rows(NodesDev nd, const uint32_t *__restrict__ tips, uint32_t n_tips, PkbTipRec *__restrict__ rec) {
    const uint32_t t = (blockIdx.x * blockDim.x + threadIdx.x) >> 2, w = (threadIdx.x & 3u) * 4u;
    if (t >= n_tips) return;
    const uint32_t id = tips[t];
    uint4 x = make_uint4(0u, 0u, 0u, 0u);
    if (nd.stride <= PKB_ROW_WORDS) {
        const uint32_t *src = nd.words + (size_t) id * nd.stride;
        if ((nd.stride & 3) == 0) { if ((int) w < nd.stride) x = *reinterpret_cast<const uint4 *>(src + w); }
        else {
            if ((int) w < nd.stride) x.x = src[w];
            if ((int) w + 1 < nd.stride) x.y = src[w + 1];
            if ((int) w + 2 < nd.stride) x.z = src[w + 2];
            if ((int) w + 3 < nd.stride) x.w = src[w + 3];
        }
    }
    *reinterpret_cast<uint4 *>(rec[t].words + w) = x;                          // (the id goes in with the snapshot half: k_pkb_tiprec_snap)
}

// the other half of the record -- id, snapshot row size, its first keys -- once per round: four lanes per tip, sixteen bytes each, so that the
// half-line is written WHOLE (a record whose 4-byte count alone was rewritten cost a read-modify-write in memory: 0.16 ms for 3.6 M tips)
__global__ void __launch_bounds__(256) k_pkb_tiprec_snap(const uint32_t *__restrict__ tips, uint32_t n_tips, const uint32_t *__restrict__ rowptr,
                                                          const unsigned long long *__restrict__ gkeys, PkbTipRec *__restrict__ rec) {
    const uint32_t t = (blockIdx.x * blockDim.x + threadIdx.x) >> 2, q = threadIdx.x & 3u;
    if (t >= n_tips) return;
    const uint32_t id = tips[t], r0 = rowptr[id], r1 = rowptr[id + 1];
    // lane q holds the 16 bytes [16 q, 16 q + 16) of the half: q = 0: id, nkeys, key 0; q > 0: keys 2 q - 1 and 2 q
    const uint32_t ka = q == 0u ? 0u : 2u * q - 1u, kb = 2u * q;
    const unsigned long long a = (q != 0u && r0 + ka < r1) ? gkeys[r0 + ka] : 0ull, b = r0 + kb < r1 ? gkeys[r0 + kb] : 0ull;
    uint4 x;
    if (q == 0u) x = make_uint4(id, r1 - r0, (uint32_t) b, (uint32_t) (b >> 32));
    else x = make_uint4((uint32_t) a, (uint32_t) (a >> 32), (uint32_t) b, (uint32_t) (b >> 32));
    *reinterpret_cast<uint4 *>(reinterpret_cast<char *>(rec + t) + 64 + 16 * q) = x;
}

// ... and between the rounds only where a row changed: the sources of the round's additions (sorted keys: the first key of a source's run acts)
__global__ void __launch_bounds__(256) k_pkb_tiprec_snap_srcs(const unsigned long long *__restrict__ adds, uint64_t n_adds, const uint32_t *__restrict__ tipidx,
                                                               const uint32_t *__restrict__ rowptr, const unsigned long long *__restrict__ gkeys, PkbTipRec *__restrict__ rec) {
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n_adds; i += (uint64_t) gridDim.x * blockDim.x) {
        const int src = pkb_key_src(adds[i]);
        if (i > 0 && pkb_key_src(adds[i - 1]) == src) continue;
        const uint32_t t = tipidx[src];
        if (t == 0xFFFFFFFFu) continue;                                        // (cannot happen: only tips are sources of additions)
        const uint32_t r0 = rowptr[src], r1 = rowptr[src + 1];
        unsigned long long k[PKB_REC_KEYS];
#pragma unroll
        for (int q = 0; q < PKB_REC_KEYS; q++) k[q] = r0 + (uint32_t) q < r1 ? gkeys[r0 + q] : 0ull;
        uint4 *h = reinterpret_cast<uint4 *>(reinterpret_cast<char *>(rec + t) + 64);       // the half-line written whole, as k_pkb_tiprec_snap does
        h[0] = make_uint4((uint32_t) src, r1 - r0, (uint32_t) k[0], (uint32_t) (k[0] >> 32));
        h[1] = make_uint4((uint32_t) k[1], (uint32_t) (k[1] >> 32), (uint32_t) k[2], (uint32_t) (k[2] >> 32));
        h[2] = make_uint4((uint32_t) k[3], (uint32_t) (k[3] >> 32), (uint32_t) k[4], (uint32_t) (k[4] >> 32));
        h[3] = make_uint4((uint32_t) k[5], (uint32_t) (k[5] >> 32), (uint32_t) k[6], (uint32_t) (k[6] >> 32));
    }
}

// neighbors[b] of node a in the round's snapshot through a's record (ra, rb: the records' positions)
__device__ __forceinline__ int rec_offset(const PkbGraph &g, const PkbTipRec *__restrict__ rec, int ra, int id_b) {
    const uint32_t nk = rec[ra].nkeys;
    if (nk > (uint32_t) PKB_REC_KEYS) return snapshot_offset(g, (int) rec[ra].id, id_b);
    int off = PKB_INF;
    for (uint32_t q = 0; q < nk; q++) { const unsigned long long k = rec[ra].keys[q]; if (pkb_key_dst(k) == id_b) off = pkb_key_off(k); }
    return off;
}

// a record's row into LDS: PKB_ROW_WORDS words + a zero word behind them
__device__ __forceinline__ void pkb_stage_rec_row(const PkbTipRec *__restrict__ r, uint32_t *dst) {
    const uint4 *src = reinterpret_cast<const uint4 *>(r->words);
#pragma unroll
    for (int q = 0; q < PKB_ROW_WORDS / 4; q++) { const uint4 x = src[q]; dst[4 * q] = x.x; dst[4 * q + 1] = x.y; dst[4 * q + 2] = x.z; dst[4 * q + 3] = x.w; }
    dst[PKB_ROW_WORDS] = 0u;
}

// canAlign (see can_align above) on staged rows; l1, l2 <= 16 * PKB_ROW_WORDS.  Every block the loop reads lies inside both reads
// (32 k < 2 ov <= 2 min(l1 - off, l2)), words past a read's end are zero or masked.
__device__ __forceinline__ bool can_align_rows_loop(const uint32_t *a, const uint32_t *b, int ov, int off, const PkbCfg &c) {
    const int bit = 2 * off, q = bit >> 5, r = bit & 31;
    const int nbits = 2 * ov;
    const int t0 = 2 * (ov - c.same_ends);
    int total = 0, head = 0, tail = 0;
    const int nblk = (nbits + 31) >> 5;
    for (int k = 0; k < nblk; k++) {
        uint32_t x = pkb_funnel(a[q + k], a[q + k + 1], r) ^ b[k];
        const int rem = nbits - 32 * k;
        const uint32_t vmask = rem >= 32 ? 0xFFFFFFFFu : ((1u << rem) - 1u);
        x &= vmask;
        total += __popc(x);
        if (k == 0) head = __popc(x & ((2u << (2 * c.same_ends)) - 1u));
        const int lo_t = t0 - 32 * k;
        if (lo_t < 32) tail += __popc(lo_t <= 0 ? x : (x & ~((1u << lo_t) - 1u)));
    }
    if (head != 0 || tail != 0) return false;
    const int seq = (nbits - total) >> 1;
    return 100 * seq >= c.min_identity_pct * ov;
}

// 32 bits of a staged row from bit `pos` on (the word behind the row is zero)
__device__ __forceinline__ uint32_t pkb_row_bits(const uint32_t *row, int pos) { return pkb_funnel(row[pos >> 5], row[(pos >> 5) + 1], pos & 31); }

// The same decision with the two end windows looked at FIRST and on their own (bits 0 .. 2 se and the last 2 se bits of the overlap: two funnel
// shifts each), then one popcount per word of the overlap with the row word carried over: half the vector instructions of the loop above, which
// stays for overlaps too short for the windows to be apart (never under the reference's parameters: min_overlap_area is half a read).
__device__ __forceinline__ bool can_align_rows(const uint32_t *a, const uint32_t *b, int l1, int l2, int off, const PkbCfg &c) {
    if (100 * off > c.max_offset_pct * l1) return false;
    if (off < 0) return false;
    const int ov = (l1 < l2 + off ? l1 : l2 + off) - off;
    if (ov < c.min_overlap_area) return false;
    if (l2 + off - l1 < 0) return false;
    const int se = c.same_ends;
    if (ov < 32 || se > 7 || se < 1) return can_align_rows_loop(a, b, ov, off, c);
    const int bit = 2 * off, nbits = 2 * ov, t0 = nbits - 2 * se;
    if (((pkb_row_bits(a, bit) ^ b[0]) & ((2u << (2 * se)) - 1u)) != 0u) return false;                     // head: bits 0 .. 2 se inclusive (LowErrorRate :43)
    if (((pkb_row_bits(a, bit + t0) ^ pkb_row_bits(b, t0)) & ((1u << (2 * se)) - 1u)) != 0u) return false;   // tail: bits t0 .. nbits - 1
    const int q = bit >> 5, r = bit & 31, nw = nbits >> 5, rem = nbits & 31;
    int total = 0;
    uint32_t lo = a[q];
    for (int k = 0; k < nw; k++) {
        const uint32_t hi = a[q + k + 1];
        total += __popc(pkb_funnel(lo, hi, r) ^ b[k]);
        lo = hi;
    }
    if (rem) total += __popc((pkb_funnel(lo, a[q + nw + 1], r) ^ b[nw]) & ((1u << rem) - 1u));
    const int seq = (nbits - total) >> 1;
    return 100 * seq >= c.min_identity_pct * ov;
}

// thread per entry: which entries head a group of >= 2 and how long it is.  The number of such groups is the total of the flag scan,
// the largest size is the first key of the sorted head list; the only atomics are the rare ones (groups of more than 64: *max_d
// is exact for those)
__global__ void __launch_bounds__(256) k_pkb_group_sizes(const unsigned long long *__restrict__ keys, uint64_t n,
                                                          unsigned long long *__restrict__ big_words /* total words for groups > 64 */,
                                                          unsigned long long *__restrict__ max_d,
                                                          uint32_t *__restrict__ head_flag /* 1 = entry heads a group of >= 2 */,
                                                          uint32_t *__restrict__ gsize /* at such an entry: min(D, 255) */,
                                                          uint32_t rank, uint32_t n_ranks /* N ranks: a group belongs to rank mix(k-mer key) mod N */) {
    const uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long k = keys[i];
    uint32_t flag = 0;
    const bool mine = n_ranks <= 1u || (uint32_t) (((k ^ (k >> 29)) * 0x9E3779B97F4A7C15ull) >> 40) % n_ranks == rank;
    if (mine && (i == 0 || keys[i - 1] != k) && i + 1 < n && keys[i + 1] == k) {
        uint64_t e = i + 2;
        while (e < n && keys[e] == k) e++;
        const uint64_t D = e - i;
        flag = 1u;
        gsize[i] = (uint32_t) (D < 255 ? D : 255);
        if (D > 64) { atomicAdd(big_words, (unsigned long long) (D * ((D + 63) / 64))); atomicMax(max_d, (unsigned long long) D); }   // rare
    }
    head_flag[i] = flag;
}

// dense list of the entries that head a group of >= 2 (flags from k_pkb_group_sizes, positions from their scan) + sort keys that
// put the longest groups first
__global__ void __launch_bounds__(256) k_pkb_head_list(const uint32_t *__restrict__ head_flag, const uint32_t *__restrict__ pos, const uint32_t *__restrict__ gsize,
                                                        uint64_t n, uint32_t *__restrict__ heads, uint32_t *__restrict__ hkey) {
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x)
        if (head_flag[i]) { heads[pos[i]] = (uint32_t) i; hkey[pos[i]] = 255u - gsize[i]; }
}

// first list position of every size key in the sorted head list: bound[k] = first t with hkey[t] >= k (k = 0 .. 256; bound[256] = n_heads).
// Groups of exactly D members are [bound[255 - D], bound[256 - D]): the group kernels take their ranges from here, the host the histogram.
__global__ void __launch_bounds__(256) k_pkb_class_bounds(const uint32_t *__restrict__ hkey, uint32_t n_heads, uint32_t *__restrict__ bound) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t > n_heads) return;
    const int k0 = t == 0 ? 0 : (int) hkey[t - 1] + 1;
    const int k1 = t == n_heads ? 256 : (int) hkey[t];
    for (int k = k0; k <= k1; k++) bound[k] = t;
}

// k_pkb_group_sizes + scan + k_pkb_head_list in ONE pass over the sorted keys (round 5): a block of 256 threads looks at 4096 entries, counts the
// heads it finds, takes its place in the list with one atomic and writes (entry, 255 - min(D, 255)).  The list's order is whatever order the blocks
// arrive in; it is sorted by size next, the additions are sorted by key at the merge: nothing downstream depends on it.
constexpr int PKB_HEADS_IPT = 16;
__global__ void __launch_bounds__(256) k_pkb_heads(const unsigned long long *__restrict__ keys, uint64_t n, unsigned long long *__restrict__ big_words,
                                                    unsigned long long *__restrict__ max_d, unsigned long long *__restrict__ n_heads /* zeroed by the caller */,
                                                    uint32_t *__restrict__ heads, uint32_t *__restrict__ hkey, uint32_t rank, uint32_t n_ranks) {
    constexpr int TILE = 256 * PKB_HEADS_IPT;
    __shared__ unsigned long long sk[TILE + 2];                              // the block's keys, the one before and the one behind
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t s_base;
    const uint64_t base = (uint64_t) blockIdx.x * TILE;
    {
        unsigned long long x[PKB_HEADS_IPT];
#pragma unroll
        for (int j = 0; j < PKB_HEADS_IPT; j++) { const uint64_t i = base + (uint64_t) j * 256 + threadIdx.x; x[j] = i < n ? keys[i] : 0ull; }
#pragma unroll
        for (int j = 0; j < PKB_HEADS_IPT; j++) sk[1 + j * 256 + threadIdx.x] = x[j];
        if (threadIdx.x == 0) sk[0] = base > 0 ? keys[base - 1] : 0ull;
        if (threadIdx.x == 1) sk[TILE + 1] = base + TILE < n ? keys[base + TILE] : 0ull;
    }
    __syncthreads();
    uint32_t flags = 0u, cnt = 0u;
    uint32_t sz[PKB_HEADS_IPT / 4] = {0u, 0u, 0u, 0u};                          // min(D, 255), a byte per entry of this thread
#pragma unroll
    for (int j = 0; j < PKB_HEADS_IPT; j++) {
        const uint32_t li = (uint32_t) j * 256 + threadIdx.x;                  // position in the tile
        const uint64_t i = base + li;
        if (i >= n) continue;
        const unsigned long long k = sk[1 + li];
        const bool mine = n_ranks <= 1u || (uint32_t) (((k ^ (k >> 29)) * 0x9E3779B97F4A7C15ull) >> 40) % n_ranks == rank;
        if (mine && (i == 0 || sk[li] != k) && i + 1 < n && sk[2 + li] == k) {
            uint64_t e = i + 2;
            while (e < n && (e - base <= (uint64_t) TILE ? sk[1 + (e - base)] : keys[e]) == k) e++;
            const uint64_t D = e - i;
            flags |= 1u << j; cnt++;
            sz[j >> 2] |= (uint32_t) (D < 255 ? D : 255) << (8 * (j & 3));
            if (D > 64) { atomicAdd(big_words, (unsigned long long) (D * ((D + 63) / 64))); atomicMax(max_d, (unsigned long long) D); }   // rare
        }
    }
    // exclusive scan of cnt over the block
    uint32_t incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(incl, o, 64); if ((int) (threadIdx.x & 63u) >= o) incl += y; }
    if ((threadIdx.x & 63u) == 63u) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t before = 0u, total = 0u;
#pragma unroll
    for (int q = 0; q < 4; q++) { const uint32_t x = wsum[q]; before += (uint32_t) q < (threadIdx.x >> 6) ? x : 0u; total += x; }
    if (total == 0u) return;
    if (threadIdx.x == 0) s_base = (uint32_t) atomicAdd(n_heads, (unsigned long long) total);
    __syncthreads();
    uint32_t at = s_base + before + incl - cnt;
#pragma unroll
    for (int j = 0; j < PKB_HEADS_IPT; j++)
        if ((flags >> j) & 1u) {
            heads[at] = (uint32_t) (base + (uint64_t) j * 256 + threadIdx.x);
            hkey[at] = 255u - ((sz[j >> 2] >> (8 * (j & 3))) & 0xFFu);
            at++;
        }
}

struct PkbAdd {                                                             // where a round's additions go
    unsigned long long *keys; uint64_t dense, cap; unsigned long long *overflow;
};

__device__ __forceinline__ void pkb_add_edge(const PkbAdd &ad, unsigned long long *mine, int slot, int cap2d, unsigned long long ne) {
    if (slot < cap2d) mine[slot] = ne;
    else {
        const unsigned long long k = atomicAdd(ad.overflow, 1ull);
        if (ad.dense + k < ad.cap) ad.keys[ad.dense + k] = ne;
    }
}

// one group, one thread, global memory only; returns the number of canAlign calls, *n_added = additions kept in the dense slots
__device__ __forceinline__ unsigned long long pkb_group_serial(const NodesDev &nd, const PkbCfg &c, const PkbGraph &g, const unsigned long long *__restrict__ keys,
                                                              unsigned long long *__restrict__ vals, uint64_t n, uint64_t gs, int D, unsigned long long *__restrict__ marks,
                                                              unsigned long long *__restrict__ big_marks, unsigned long long *__restrict__ big_cursor,
                                                              const PkbAdd &ad, uint32_t *n_added, const uint32_t *__restrict__ tips) {
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
    unsigned long long *mine = ad.keys + 2 * gs;                             // dense slots of this group: 2 * D
    int n_add = 0;
    unsigned long long calls = 0;
    for (int i = D - 2; i >= 0; i--) {
        const unsigned long long vi = v[i];
        const int id1 = (int) tips[pkb_val_id(vi)], ind1 = pkb_val_ind(vi), len1 = pkb_val_len(vi);      // (an entry names the tip: its node id is tips[tip])
        unsigned long long *row_i = rows + (size_t) i * RW;
        for (int j = i + 1; j < D; j++) {
            const unsigned long long vj = v[j];
            if (pkb_val_id(vi) == pkb_val_id(vj)) continue;
            const int id2 = (int) tips[pkb_val_id(vj)];
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
                    pkb_add_edge(ad, mine, n_add, 2 * D, pkb_edge_key(id1, id2, off));
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

// The groups the other two kernels leave: the front of the list (D > 64), any group whose flag[t] they set (a read twice in the
// group), or all of them (`all`: rows too long to stage)
__global__ void __launch_bounds__(64) k_pkb_groups_serial(NodesDev nd, PkbCfg c, PkbGraph g, const unsigned long long *__restrict__ keys,
                                                           const uint32_t *__restrict__ heads, const uint32_t *__restrict__ hkey, uint32_t n_heads, int all,
                                                           const uint32_t *__restrict__ left /* per group: 1 = left for this kernel */,
                                                           unsigned long long *__restrict__ vals, uint64_t n, unsigned long long *__restrict__ marks,
                                                           unsigned long long *__restrict__ big_marks, unsigned long long *__restrict__ big_cursor, PkbAdd ad,
                                                           unsigned long long *__restrict__ counters, uint32_t *__restrict__ n_add, const uint32_t *__restrict__ tips) {
    unsigned long long calls = 0;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n_heads) {
        const int D = 255 - (int) hkey[t];
        if (all || D > PKB_WAVE_MAX || left[t]) {
            uint32_t na = 0;
            calls = pkb_group_serial(nd, c, g, keys, vals, n, (uint64_t) heads[t], D, marks, big_marks, big_cursor, ad, &na, tips);
            n_add[t] = na;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) calls += __shfl_xor(calls, o);          // one counter update per wave
    if ((threadIdx.x & 63u) == 0 && calls) atomicAdd(&counters[0], calls);
}

// 8 <= D <= 64: one wave per group.  The list is in descending size: a wave stops at the first group that is too small.
__global__ void __launch_bounds__(64) k_pkb_groups_wave(NodesDev nd, PkbCfg c, PkbGraph g, const uint32_t *__restrict__ heads, const uint32_t *__restrict__ hkey,
                                                         const uint32_t *__restrict__ bound, int d_lo /* sizes d_lo .. PKB_WAVE_MAX */,
                                                         const unsigned long long *__restrict__ vals, PkbAdd ad,
                                                         unsigned long long *__restrict__ counters, uint32_t *__restrict__ n_add, uint32_t *__restrict__ left,
                                                         const PkbTipRec *__restrict__ rec) {
    __shared__ uint32_t srow[64][PKB_ROW_WORDS + 1];
    __shared__ unsigned long long sv[64];
    __shared__ uint32_t sid[64], snk[64];                                    // entry i: node id, size of its snapshot row ...
    __shared__ unsigned long long sk[64][PKB_REC_KEYS];                      // ... and its first keys (all of them, usually)
    const int lane = (int) threadIdx.x;
    unsigned long long calls = 0;
    const uint32_t t_lo = bound[255 - PKB_WAVE_MAX], t_hi = bound[256 - d_lo];
    for (uint32_t t = t_lo + blockIdx.x; t < t_hi; t += gridDim.x) {
        const int D = 255 - (int) hkey[t];
        const uint64_t gs = heads[t];
        __syncthreads();                                                     // the previous group's LDS is no longer read
        // order the group: ascending val (vals are distinct: a read yields one k-mer per start position)
        const unsigned long long mv = lane < D ? vals[gs + lane] : ~0ull;
        sv[lane] = mv;
        __syncthreads();
        int rank = 0;
        for (int k = 0; k < D; k++) rank += sv[k] < mv;
        __syncthreads();
        if (lane < D) sv[rank] = mv;
        __syncthreads();
        const unsigned long long vj = lane < D ? sv[lane] : 0ull;
        const int idj = pkb_val_id(vj), indj = pkb_val_ind(vj);
        bool twice = false;
        for (int k = 0; k < D; k++) twice |= (lane < D && k != lane && pkb_val_id(sv[k]) == idj);
        if (__ballot(twice) != 0ull) {                                       // a read twice in the group: the serial kernel replays it
            if (lane == 0) { left[t] = 1u; n_add[t] = 0u; }
            continue;
        }
        int nidj = 0;                                                        // this lane's entry: its node id
        if (lane < D) {
            const PkbTipRec *r = rec + idj;                                  // the whole entry in one line: row, id, snapshot keys
            pkb_stage_rec_row(r, srow[lane]);
            nidj = (int) r->id;
            sid[lane] = r->id; snk[lane] = r->nkeys;
#pragma unroll
            for (int q = 0; q < PKB_REC_KEYS; q++) sk[lane][q] = r->keys[q];
        }
        __syncthreads();
        unsigned long long *mine = ad.keys + 2 * gs;
        // All pairs (i, j > i) at once, 64 per pass, lane = pair: eligibility, offset look-up and canAlign -- speculatively, the replay
        // below decides which of them the reference would have reached.  Pair order: rows ascending, j ascending; lane r collects
        // the bits of row r (bit j = pair (r, j)).
        unsigned long long Em = 0ull, Sm = 0ull, Cm = 0ull, Am = 0ull;
        const int P = D * (D - 1) / 2;
        const int my_start = lane * (D - 1) - lane * (lane - 1) / 2, my_cnt = D - 1 - lane;
        for (int base = 0; base < P; base += 64) {
            const int p = base + lane;
            int i = 0, rem = p < P ? p : 0;
            while (rem >= D - 1 - i) { rem -= D - 1 - i; i++; }
            const int j = i + 1 + rem;
            const unsigned long long vi = sv[i], vj2 = sv[j];
            const int ind1 = pkb_val_ind(vi), len1 = pkb_val_len(vi);
            const int id2 = (int) sid[j], len2 = pkb_val_len(vj2);
            const int off = ind1 - pkb_val_ind(vj2);
            bool elig = p < P && off >= 0 && !(100 * off > c.max_offset_pct * len1);     // the `break` of :55 is monotone in j: part of the mask
            if (elig) {
                const int ov = (len1 < len2 + off ? len1 : len2 + off) - off;
                elig = ov >= c.min_overlap_area && len2 + off - len1 >= 0;
            }
            int snap = PKB_INF;
            if (elig) {
                const uint32_t nk = snk[i];
                if (nk <= (uint32_t) PKB_REC_KEYS) {                             // the row is in LDS
                    for (uint32_t q = 0; q < nk; q++) { const unsigned long long k = sk[i][q]; if (pkb_key_dst(k) == id2) snap = pkb_key_off(k); }
                } else snap = snapshot_offset(g, (int) sid[i], id2);
            }
            const bool call = elig && snap > off;
            const bool ok = call && can_align_rows(srow[i], srow[j], len1, len2, off, c);
            const unsigned long long be = __ballot(elig), bs = __ballot(snap != PKB_INF), bc = __ballot(call), ba = __ballot(ok);
            const int lo = my_start > base ? my_start : base, hi = (my_start + my_cnt) < (base + 64) ? (my_start + my_cnt) : (base + 64);
            if (lane < D - 1 && lo < hi) {
                const int w = hi - lo;
                const unsigned long long m = w >= 64 ? ~0ull : ((1ull << w) - 1ull);
                const int from = lo - base, to = lane + 1 + (lo - my_start);
                Em |= ((be >> from) & m) << to; Sm |= ((bs >> from) & m) << to; Cm |= ((bc >> from) & m) << to; Am |= ((ba >> from) & m) << to;
            }
        }
        // the i / j loops of the reference on the masks (scalar: the same in every lane); lane r keeps the marker row of entry r
        auto lane64 = [&](unsigned long long x, int l) -> unsigned long long {
            return ((unsigned long long) (uint32_t) __builtin_amdgcn_readlane((int) (x >> 32), l) << 32) | (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) x, l);
        };
        unsigned long long myrow = 0ull;
        int n_added = 0;
        for (int i = D - 2; i >= 0; i--) {
            const unsigned long long e_i = lane64(Em, i), s_i = lane64(Sm, i), c_i = lane64(Cm, i), a_i = lane64(Am, i);
            unsigned long long row = 0ull, addm = 0ull;
            for (unsigned long long m = e_i; m; m &= m - 1ull) {
                const int j = __builtin_ctzll(m);
                if ((row >> j) & 1ull) continue;                             // already reachable inside the group (:62)
                bool reach = true;                                           // an edge of at most this offset exists
                if ((c_i >> j) & 1ull) {
                    calls++;
                    if ((a_i >> j) & 1ull) addm |= 1ull << j;
                    else reach = (s_i >> j) & 1ull;
                }
                if (reach) row |= (1ull << j) | lane64(myrow, j);
            }
            if (lane == i) myrow = row;
            if ((addm >> lane) & 1ull) {
                const unsigned long long vi = sv[i];
                pkb_add_edge(ad, mine, n_added + __popcll(addm & ((1ull << lane) - 1ull)), 2 * D, pkb_edge_key((int) sid[i], nidj, pkb_val_ind(vi) - indj));
            }
            n_added += __popcll(addm);
        }
        if (lane == 0) { n_add[t] = (uint32_t) (n_added < 2 * D ? n_added : 2 * D); left[t] = 0u; }
    }
    if (lane == 0 && calls) atomicAdd(&counters[0], calls);                  // every lane counted the same calls
}


// 8 <= D <= 16: FOUR groups per wave, sixteen lanes each (round 5).  At 10 M reads with 2 % errors all but 400 of a round's 297 k groups above seven
// members have at most fifteen: a wave per group left three quarters of its lanes idle through a chain of four dependent memory round trips
// (list entry -> k-mer entries -> rows and snapshot rows -> snapshot keys), 13 us per group and wave.  Same plan as k_pkb_groups_wave -- every pair
// speculatively, then the reference's i / j loops on bit masks -- in 16-bit masks: a pass takes row i and row D - 2 - i of each group together
// (D - 1 - i and i + 1 pairs: D lanes), the replay runs once per wave with every quarter on its own masks (cross-lane reads stay inside a quarter).
// REPLAY = false (the default since round 5, second step): the kernel stops after the masks -- the replay of four groups side by side cost more
// vector instructions than everything before it (a wave executes every step of the longest of its four i / j walks) -- and leaves, per entry, its
// row of masks in marks[] and the group's entries in sorted order in vals[]; k_pkb_quarter_replay walks them with a THREAD per group.
constexpr int PKB_QUARTER_MAX = 16;
template <bool REPLAY>
__global__ void __launch_bounds__(64) k_pkb_groups_quarter(NodesDev nd, PkbCfg c, PkbGraph g, const uint32_t *__restrict__ heads, const uint32_t *__restrict__ hkey,
                                                            const uint32_t *__restrict__ bound, unsigned long long *__restrict__ vals, unsigned long long *__restrict__ marks,
                                                            PkbAdd ad, unsigned long long *__restrict__ counters, uint32_t *__restrict__ n_add, uint32_t *__restrict__ left,
                                                            const PkbTipRec *__restrict__ rec) {
    __shared__ uint32_t srow[64][PKB_ROW_WORDS + 1];
    __shared__ unsigned long long sv[64];
    __shared__ uint32_t sid[64], snk[64];
    __shared__ unsigned long long sk[64][PKB_REC_KEYS];
    const int lane = (int) threadIdx.x, sub = lane & 15, qb = lane & 48;     // qb: the quarter's first lane == its bit position in a ballot
    const uint32_t t_lo = bound[255 - PKB_QUARTER_MAX], t_hi = bound[256 - (PKB_SMALL_MAX + 1)];
    unsigned long long calls = 0;                                            // the quarter's calls, the same in its sixteen lanes
    for (uint32_t t4 = t_lo + blockIdx.x * 4u; t4 < t_hi; t4 += gridDim.x * 4u) {
        const uint32_t t = t4 + (uint32_t) (lane >> 4);
        const bool valid = t < t_hi;
        const int D = valid ? 255 - (int) hkey[t] : 0;
        const uint64_t gs = valid ? heads[t] : 0;
        __syncthreads();                                                     // the previous groups' LDS is no longer read
        const unsigned long long mv = sub < D ? vals[gs + sub] : ~0ull;
        sv[lane] = mv;
        __syncthreads();
        int rank = 0;
#pragma unroll
        for (int k = 0; k < PKB_QUARTER_MAX; k++) rank += sv[qb + k] < mv;   // (vals are distinct; the lanes past D hold ~0 and rank behind them)
        __syncthreads();
        if (sub < D) sv[qb + rank] = mv;
        __syncthreads();
        const unsigned long long vj = sub < D ? sv[lane] : 0ull;
        const int idj = pkb_val_id(vj), indj = pkb_val_ind(vj);
        bool twice = false;
        for (int k = 0; k < PKB_QUARTER_MAX; k++) twice |= (sub < D && k < D && k != sub && pkb_val_id(sv[qb + k]) == idj);
        const bool dup = ((__ballot(twice) >> qb) & 0xFFFFull) != 0ull;
        if (dup && sub == 0) { left[t] = 1u; n_add[t] = 0u; }               // a read twice in the group: the serial kernel replays it
        const int Dq = dup ? 0 : D;
        const int Dmax = max(max(__builtin_amdgcn_readlane(Dq, 0), __builtin_amdgcn_readlane(Dq, 16)), max(__builtin_amdgcn_readlane(Dq, 32), __builtin_amdgcn_readlane(Dq, 48)));
        int nidj = 0;                                                        // this lane's entry: its node id
        if (sub < Dq) {
            const PkbTipRec *r = rec + idj;                                  // the whole entry in one line: row, id, snapshot keys
            pkb_stage_rec_row(r, srow[lane]);
            nidj = (int) r->id;
            sid[lane] = r->id; snk[lane] = r->nkeys;
#pragma unroll
            for (int q = 0; q < PKB_REC_KEYS; q++) sk[lane][q] = r->keys[q];
        }
        __syncthreads();
        unsigned long long *mine = ad.keys + 2 * gs;
        // pass i: rows i (pairs with j = i + 1 + sub, sub < D - 1 - i) and D - 2 - i (j = sub >= D - 1 - i) of every quarter.  Masks: bit j = pair (row, j)
        uint32_t ES = 0u, CA = 0u;                                           // lane r: eligible | snapshot-known << 16, call | aligned << 16 of row r
        for (int i = 0; 2 * i <= Dmax - 2; i++) {
            const int na = Dq - 1 - i, rb = Dq - 2 - i;
            const bool on = 2 * i <= Dq - 2;
            const bool isA = sub < na;
            const int r = isA ? i : rb, j = isA ? i + 1 + sub : sub;
            const bool pair = on && sub < Dq && (isA || rb > i);
            const unsigned long long vi = sv[qb + (pair ? r : 0)], vj2 = sv[qb + (pair ? j : 0)];
            const int ind1 = pkb_val_ind(vi), len1 = pkb_val_len(vi);
            const int id2 = (int) sid[qb + (pair ? j : 0)], len2 = pkb_val_len(vj2);
            const int off = ind1 - pkb_val_ind(vj2);
            bool elig = pair && off >= 0 && !(100 * off > c.max_offset_pct * len1);
            if (elig) {
                const int ov = (len1 < len2 + off ? len1 : len2 + off) - off;
                elig = ov >= c.min_overlap_area && len2 + off - len1 >= 0;
            }
            int snap = PKB_INF;
            if (elig) {
                const uint32_t nk = snk[qb + r];
                if (nk <= (uint32_t) PKB_REC_KEYS) {
                    for (uint32_t q = 0; q < nk; q++) { const unsigned long long k = sk[qb + r][q]; if (pkb_key_dst(k) == id2) snap = pkb_key_off(k); }
                } else snap = snapshot_offset(g, (int) sid[qb + r], id2);
            }
            const bool call = elig && snap > off;
            const bool ok = call && can_align_rows(srow[qb + r], srow[qb + j], len1, len2, off, c);
            const uint32_t be = (uint32_t) (__ballot(elig) >> qb) & 0xFFFFu, bs = (uint32_t) (__ballot(snap != PKB_INF) >> qb) & 0xFFFFu;
            const uint32_t bc = (uint32_t) (__ballot(call) >> qb) & 0xFFFFu, ba = (uint32_t) (__ballot(ok) >> qb) & 0xFFFFu;
            const uint32_t ma = on ? ((1u << na) - 1u) : 0u;                   // lanes of row i
            if (on && sub == i) { ES = ((be & ma) << (i + 1)) | (((bs & ma) << (i + 1)) << 16); CA = ((bc & ma) << (i + 1)) | (((ba & ma) << (i + 1)) << 16); }
            if (on && rb > i && sub == rb) { ES = (be & ~ma) | ((bs & ~ma) << 16); CA = (bc & ~ma) | ((ba & ~ma) << 16); }
        }
        if (!REPLAY) {
            // (the sorted entries go back with the NODE id in place of the tip: the replay writes edges and reads nothing else)
            if (sub < Dq) { marks[gs + sub] = (unsigned long long) ES | ((unsigned long long) CA << 32); vals[gs + sub] = (vj & ~PKB_ID_MASK) | (unsigned long long) (uint32_t) nidj; }
            continue;
        }
        // the i / j loops of the reference on the masks, every quarter on its own; lane r keeps the marker row of entry r
        uint32_t myrow = 0u;
        int n_added = 0;
        for (int i = Dmax - 2; i >= 0; i--) {
            const bool on = i <= Dq - 2;
            const uint32_t es = (uint32_t) __shfl((int) ES, qb + i), ca = (uint32_t) __shfl((int) CA, qb + i);
            uint32_t m = on ? (es & 0xFFFFu) : 0u;
            const uint32_t s_i = es >> 16, c_i = ca & 0xFFFFu, a_i = ca >> 16;
            uint32_t row = 0u, addm = 0u;
            while (__ballot(m != 0u) != 0ull) {                              // (wave-uniform: every lane stays in, the cross-lane read below needs its source active)
                const bool act = m != 0u;
                const int j = act ? __builtin_ctz(m) : 0;
                const uint32_t rowj = (uint32_t) __shfl((int) myrow, qb + j);
                if (act) {
                    m &= m - 1u;
                    if (!((row >> j) & 1u)) {                                // else: already reachable inside the group (:62)
                        bool reach = true;                                   // an edge of at most this offset exists
                        if ((c_i >> j) & 1u) {
                            calls++;
                            if ((a_i >> j) & 1u) addm |= 1u << j;
                            else reach = (s_i >> j) & 1u;
                        }
                        if (reach) row |= (1u << j) | rowj;
                    }
                }
            }
            if (on && sub == i) myrow = row;
            if (on && ((addm >> sub) & 1u)) {
                const unsigned long long vi = sv[qb + i];
                pkb_add_edge(ad, mine, n_added + __popc(addm & ((1u << sub) - 1u)), 2 * Dq, pkb_edge_key((int) sid[qb + i], nidj, pkb_val_ind(vi) - indj));
            }
            n_added += __popc(addm);
        }
        if (sub == 0 && Dq > 0) { n_add[t] = (uint32_t) (n_added < 2 * Dq ? n_added : 2 * Dq); left[t] = 0u; }
    }
    calls = (sub == 0) ? calls : 0ull;
#pragma unroll
    for (int o = 32; o >= 16; o >>= 1) calls += __shfl_xor(calls, o);
    if (lane == 0 && calls) atomicAdd(&counters[0], calls);
}

// the i / j loops of the reference (createAlignmentsForKmers :40-97) on the masks k_pkb_groups_quarter<false> left: one THREAD per group, the marker
// rows of the group in a column of LDS.  marks[gs + i]: eligible | snapshot-known << 16 | call << 32 | aligned << 48 of row i (bit j = pair (i, j)).
__global__ void __launch_bounds__(256) k_pkb_quarter_replay(const uint32_t *__restrict__ heads, const uint32_t *__restrict__ hkey, const uint32_t *__restrict__ bound,
                                                             const unsigned long long *__restrict__ vals, const unsigned long long *__restrict__ marks, PkbAdd ad,
                                                             unsigned long long *__restrict__ counters, uint32_t *__restrict__ n_add, uint32_t *__restrict__ left) {
    __shared__ unsigned short srows[PKB_QUARTER_MAX][256];
    const uint32_t t_lo = bound[255 - PKB_QUARTER_MAX], t_hi = bound[256 - (PKB_SMALL_MAX + 1)];
    unsigned long long calls = 0;
    for (uint32_t t = t_lo + blockIdx.x * blockDim.x + threadIdx.x; t < t_hi; t += gridDim.x * blockDim.x) {
        if (left[t]) continue;                                               // a read twice in the group: k_pkb_groups_serial
        const int D = 255 - (int) hkey[t];
        const uint64_t gs = heads[t];
        unsigned long long *mine = ad.keys + 2 * gs;
        int n_added = 0;
        unsigned long long mk[PKB_QUARTER_MAX - 1];                          // all rows of masks on their way at once (the walk below is a chain)
#pragma unroll
        for (int i = 0; i < PKB_QUARTER_MAX - 1; i++) mk[i] = i <= D - 2 ? marks[gs + i] : 0ull;
        srows[D - 1][threadIdx.x] = 0;
#pragma unroll
        for (int i = PKB_QUARTER_MAX - 2; i >= 0; i--) {
            if (i > D - 2) continue;
            const uint32_t s_i = (uint32_t) (mk[i] >> 16) & 0xFFFFu, c_i = (uint32_t) (mk[i] >> 32) & 0xFFFFu, a_i = (uint32_t) (mk[i] >> 48);
            uint32_t row = 0u, addm = 0u;
            for (uint32_t m = (uint32_t) mk[i] & 0xFFFFu; m; m &= m - 1u) {
                const int j = __builtin_ctz(m);
                if ((row >> j) & 1u) continue;                               // already reachable inside the group (:62)
                bool reach = true;                                           // an edge of at most this offset exists
                if ((c_i >> j) & 1u) {
                    calls++;
                    if ((a_i >> j) & 1u) addm |= 1u << j;
                    else reach = (s_i >> j) & 1u;
                }
                if (reach) row |= (1u << j) | srows[j][threadIdx.x];
            }
            srows[i][threadIdx.x] = (unsigned short) row;
            if (addm) {
                const unsigned long long vi = vals[gs + i];
                for (uint32_t m = addm; m; m &= m - 1u) {
                    const unsigned long long vj = vals[gs + __builtin_ctz(m)];
                    pkb_add_edge(ad, mine, n_added, 2 * D, pkb_edge_key(pkb_val_id(vi), pkb_val_id(vj), pkb_val_ind(vi) - pkb_val_ind(vj)));
                    n_added++;
                }
            }
        }
        n_add[t] = (uint32_t) (n_added < 2 * D ? n_added : 2 * D);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) calls += __shfl_xor(calls, o);
    if ((threadIdx.x & 63u) == 0 && calls) atomicAdd(&counters[0], calls);
}

// 2 <= D <= 7: one THREAD per group; the entries in LDS, the marker rows in one register (8 bits per row), rows staged per pair.
// Persistent grid (a thread walks the list with the grid's stride): the call counter is updated once per wave of the GRID, not of
// the list -- with one launch wave per 64 groups the 55 k same-address atomics were a third of the kernel's time (0.89 -> 0.6 ms per
// round at 10 M reads).  Measured and rejected (round 3): eight LANES per group -- the pairs (i, j > i) of a row side by side,
// speculatively, then the replay on 8 x 8 bit masks as k_pkb_groups_wave does it, a group's chain of dependent accesses two long
// whatever its size -- 1.8 ms against 0.6: most groups have two or three members, seven of eight lanes idle, and the kernel turns
// VALU-bound; no split point between the two shapes (thread per group up to D = 1 .. 7, lanes above) beat the thread per group alone.
constexpr int PKB_SMALL_WG = 256;
__global__ void __launch_bounds__(PKB_SMALL_WG) k_pkb_groups_small(NodesDev nd, PkbCfg c, PkbGraph g, const uint32_t *__restrict__ heads, const uint32_t *__restrict__ hkey,
                                                          const uint32_t *__restrict__ bound, const unsigned long long *__restrict__ vals, PkbAdd ad,
                                                          unsigned long long *__restrict__ counters, uint32_t *__restrict__ n_add, uint32_t *__restrict__ left,
                                                          const PkbTipRec *__restrict__ rec) {
    __shared__ uint32_t srow[PKB_SMALL_WG][2 * PKB_ROW_WORDS + 3];           // a | 0 | b | 0 (+1: odd stride, conflict-free)
    __shared__ unsigned long long sv[PKB_SMALL_MAX][PKB_SMALL_WG];
    const int lane = (int) threadIdx.x;                                      // slot in the workgroup's LDS arrays
    unsigned long long calls = 0;
    const uint32_t t_lo = bound[255 - PKB_SMALL_MAX], t_hi = bound[254];
    for (uint32_t t = t_lo + blockIdx.x * blockDim.x + threadIdx.x; t < t_hi; t += gridDim.x * blockDim.x) {
    const int D = 255 - (int) hkey[t];
    if (D >= 2 && D <= PKB_SMALL_MAX) {
        const uint64_t gs = heads[t];
#pragma unroll
        for (int k = 0; k < PKB_SMALL_MAX; k++) if (k < D) sv[k][lane] = vals[gs + k];
        for (int i = 1; i < D; i++) {                                        // order the group: ascending val
            const unsigned long long x = sv[i][lane];
            int j = i;
            while (j > 0 && sv[j - 1][lane] > x) { sv[j][lane] = sv[j - 1][lane]; j--; }
            sv[j][lane] = x;
        }
        bool dup = false;
        for (int i = 0; i < D; i++) for (int j = i + 1; j < D; j++) dup |= pkb_val_id(sv[i][lane]) == pkb_val_id(sv[j][lane]);
        if (dup) { left[t] = 1u; n_add[t] = 0u; }
        else {
            uint32_t *ra = srow[lane], *rb = srow[lane] + PKB_ROW_WORDS + 1;
            unsigned long long *mine = ad.keys + 2 * gs;
            unsigned long long rows = 0ull;                                  // row i = bits 8 i .. 8 i + 7
            int n_added = 0;
            for (int i = D - 2; i >= 0; i--) {
                const unsigned long long vi = sv[i][lane];
                const int t1 = pkb_val_id(vi), ind1 = pkb_val_ind(vi), len1 = pkb_val_len(vi);       // t1, t2: the entries' records (row, node id, snapshot keys in one line)
                bool staged = false;
                uint32_t row_i = 0u;
                for (int j = i + 1; j < D; j++) {
                    const unsigned long long vj = sv[j][lane];
                    const int t2 = pkb_val_id(vj);
                    const int off = ind1 - pkb_val_ind(vj);
                    if (off < 0) continue;
                    if (100 * off > c.max_offset_pct * len1) break;          // :55
                    const int len2 = pkb_val_len(vj);
                    const int ov = (len1 < len2 + off ? len1 : len2 + off) - off;
                    if (ov < c.min_overlap_area) continue;
                    if (len2 + off - len1 < 0) continue;
                    if ((row_i >> j) & 1u) continue;                         // already reachable inside the group (:62)
                    const int id2 = (int) rec[t2].id;
                    int cur = rec_offset(g, rec, t1, id2);                   // neighbors[id2]
                    if (cur > off) {
                        calls++;
                        if (!staged) { pkb_stage_rec_row(rec + t1, ra); staged = true; }
                        pkb_stage_rec_row(rec + t2, rb);
                        if (can_align_rows(ra, rb, len1, len2, off, c)) {    // :66
                            pkb_add_edge(ad, mine, n_added, 2 * D, pkb_edge_key((int) rec[t1].id, id2, off));
                            n_added++;
                            cur = off;
                        }
                    }
                    if (cur != PKB_INF) row_i |= (1u << j) | (uint32_t) ((rows >> (8 * j)) & 0xFFull);      // :73-77
                }
                rows |= (unsigned long long) row_i << (8 * i);
            }
            n_add[t] = (uint32_t) (n_added < 2 * D ? n_added : 2 * D);
            left[t] = 0u;
        }
    }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) calls += __shfl_xor(calls, o);          // one counter update per wave
    if ((lane & 63) == 0 && calls) atomicAdd(&counters[0], calls);
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

void launch_pkb_src_keys(const unsigned long long *keys, uint64_t n, int shift, uint32_t *k32, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_pkb_src_keys, dim3(pkb_grid(n, 256, 8192)), dim3(256), 0, s, keys, n, shift, k32);
}
void launch_pkb_gather_sorted_runs(const unsigned long long *keys, const uint32_t *k32_sorted, const uint32_t *idx, uint64_t n, unsigned long long *out, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_pkb_gather_sorted_runs, dim3(pkb_grid(n, 256, 16384)), dim3(256), 0, s, keys, k32_sorted, idx, n, out);
}

void launch_pkb_unique_flags(const unsigned long long *in, uint64_t n, uint32_t *flag, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_pkb_unique_flags, dim3(pkb_grid(n, 256, 16384)), dim3(256), 0, s, in, n, flag);
}

void launch_pkb_unique_scatter(const unsigned long long *in, uint64_t n, const uint32_t *flag, const uint32_t *pos, int32_t n_nodes, unsigned long long *out,
                               uint32_t *rowptr, hipStream_t s) {
    hipLaunchKernelGGL(k_pkb_unique_scatter, dim3(pkb_grid(n + 1, 256, 16384)), dim3(256), 0, s, in, n, flag, pos, n_nodes, out, rowptr);
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
                         unsigned long long *max_len, uint32_t *tipidx, hipStream_t s) {
    if (nd.n <= 0) return;
    hipLaunchKernelGGL(k_pkb_tip_list, dim3((nd.n + 255) / 256), dim3(256), 0, s, nd, c, flag, pos, tips, kcount, max_len, tipidx);
}

void launch_pkb_kmers(const NodesDev &nd, const PkbCfg &c, const int32_t prio[4], const uint32_t *tips, const uint32_t *koff, uint32_t n_tips, int sort_bits,
                      unsigned long long *keys, unsigned long long *vals, const void *tiprec, int wide, hipStream_t s) {
    if (n_tips == 0) return;
    const int4 pr = make_int4(prio[0], prio[1], prio[2], prio[3]);
    const PkbTipRec *rec = (const PkbTipRec *) tiprec;
    if (nd.stride <= PKB_ROW_WORDS) hipLaunchKernelGGL(k_pkb_kmers<true>, dim3((n_tips + 255) / 256), dim3(256), 0, s, nd, c, pr, tips, koff, n_tips, sort_bits, keys, vals, rec, wide);
    else hipLaunchKernelGGL(k_pkb_kmers<false>, dim3((n_tips + 255) / 256), dim3(256), 0, s, nd, c, pr, tips, koff, n_tips, sort_bits, keys, vals, rec, wide);
}

// every round's k-mers at once (round 0's priority in prio[]; the arrays of round r start r * round_stride entries in).  false: the shape is
// not this kernel's (rows not staged, k > 48, more than four rounds) -- the caller takes launch_pkb_kmers per round
// rounds [first, first + count) of the sequence whose round 0 reads `prio`; keys / vals: the arrays of round 0 (round r lies r * round_stride entries on)
bool launch_pkb_kmers_all(const NodesDev &nd, const PkbCfg &c, const int32_t prio[4], int first, int count, const uint32_t *tips, const uint32_t *koff, uint32_t n_tips,
                          int sort_bits, unsigned long long *keys, unsigned long long *vals, size_t round_stride, const void *tiprec, hipStream_t s) {
    if (nd.stride > PKB_ROW_WORDS || c.li_k > 48 || first < 0 || count < 1 || first + count > 4) return false;
    if (n_tips == 0) return true;
    uint32_t pp0 = (uint32_t) prio[0] | ((uint32_t) prio[1] << 2) | ((uint32_t) prio[2] << 4) | ((uint32_t) prio[3] << 6);
    pp0 = ((pp0 >> (2 * first)) | (pp0 << (8 - 2 * first))) & 0xFFu;              // the priority of round `first`
    keys += (size_t) first * round_stride; vals += (size_t) first * round_stride;
    const dim3 grid((n_tips + 255) / 256), block(256);
    const PkbTipRec *rec = (const PkbTipRec *) tiprec;
    if (count == 1) hipLaunchKernelGGL((k_pkb_kmers_all<1>), grid, block, 0, s, nd, c, pp0, count, tips, koff, n_tips, sort_bits, keys, vals, round_stride, rec);
    else if (count <= 3) hipLaunchKernelGGL((k_pkb_kmers_all<3>), grid, block, 0, s, nd, c, pp0, count, tips, koff, n_tips, sort_bits, keys, vals, round_stride, rec);
    else hipLaunchKernelGGL((k_pkb_kmers_all<4>), grid, block, 0, s, nd, c, pp0, count, tips, koff, n_tips, sort_bits, keys, vals, round_stride, rec);
    return true;
}

size_t pkb_tiprec_bytes(uint32_t n_tips) { return ((size_t) n_tips + 1) * sizeof(PkbTipRec); }

void launch_pkb_tiprec_rows(const NodesDev &nd, const uint32_t *tips, uint32_t n_tips, void *tiprec, hipStream_t s) {
    if (n_tips == 0) return;
    hipLaunchKernelGGL(k_pkb_tiprec_rows, dim3((unsigned) (((uint64_t) n_tips * 4 + 255) / 256)), dim3(256), 0, s, nd, tips, n_tips, (PkbTipRec *) tiprec);
}

void launch_pkb_tiprec_snap_srcs(const unsigned long long *adds, uint64_t n_adds, const uint32_t *tipidx, const uint32_t *rowptr, const unsigned long long *gkeys, void *tiprec,
                                 hipStream_t s) {
    if (n_adds == 0) return;
    hipLaunchKernelGGL(k_pkb_tiprec_snap_srcs, dim3(pkb_grid(n_adds, 256, 8192)), dim3(256), 0, s, adds, n_adds, tipidx, rowptr, gkeys, (PkbTipRec *) tiprec);
}

void launch_pkb_tiprec_snap(const uint32_t *tips, uint32_t n_tips, const uint32_t *rowptr, const unsigned long long *gkeys, void *tiprec, hipStream_t s) {
    if (n_tips == 0) return;
    hipLaunchKernelGGL(k_pkb_tiprec_snap, dim3((unsigned) (((uint64_t) n_tips * 4 + 255) / 256)), dim3(256), 0, s, tips, n_tips, rowptr, gkeys, (PkbTipRec *) tiprec);
}

// counter: zeroed by the caller; after the call *counter > list_cap means "run launch_pkb_fix_runs_loop"
void launch_pkb_fix_runs(unsigned long long *keys, unsigned long long *vals, uint64_t n, int bits, uint32_t *list, uint32_t list_cap, unsigned long long *counter,
                         hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_pkb_fix_flag, dim3((unsigned) ((n + 256 * PKB_FLAG_IPT - 1) / (256 * PKB_FLAG_IPT))), dim3(256), 0, s, (const unsigned long long *) keys, n, bits,
                       list, list_cap, counter);
    hipLaunchKernelGGL(k_pkb_fix_owner, dim3((list_cap + 63) / 64), dim3(64), 0, s, (const unsigned long long *) keys, bits, list, list_cap, (const unsigned long long *) counter);
    hipLaunchKernelGGL(k_pkb_fix_apply, dim3((list_cap + 63) / 64), dim3(64), 0, s, keys, vals, n, bits, (const uint32_t *) list, list_cap, (const unsigned long long *) counter);
}

void launch_pkb_fix_runs_loop(unsigned long long *keys, unsigned long long *vals, uint64_t n, int bits, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_pkb_fix_runs, dim3(pkb_grid(n, 256, 16384)), dim3(256), 0, s, keys, vals, n, bits);
}

void launch_pkb_group_sizes(const unsigned long long *keys, uint64_t n, unsigned long long *big_words, unsigned long long *max_d, uint32_t *head_flag,
                            uint32_t *gsize, int rank, int n_ranks, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_pkb_group_sizes, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, s, keys, n, big_words, max_d, head_flag, gsize, (uint32_t) rank, (uint32_t) n_ranks);
}

void launch_pkb_heads(const unsigned long long *keys, uint64_t n, unsigned long long *big_words, unsigned long long *max_d, unsigned long long *n_heads,
                      uint32_t *heads, uint32_t *hkey, int rank, int n_ranks, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_pkb_heads, dim3((unsigned) ((n + 256 * PKB_HEADS_IPT - 1) / (256 * PKB_HEADS_IPT))), dim3(256), 0, s, keys, n, big_words, max_d, n_heads,
                       heads, hkey, (uint32_t) rank, (uint32_t) n_ranks);
}

void launch_pkb_class_bounds(const uint32_t *hkey, uint32_t n_heads, uint32_t *bound, hipStream_t s) {
    hipLaunchKernelGGL(k_pkb_class_bounds, dim3(n_heads / 256 + 1), dim3(256), 0, s, hkey, n_heads, bound);
}

void launch_pkb_head_list(const uint32_t *head_flag, const uint32_t *pos, const uint32_t *gsize, uint64_t n, uint32_t *heads, uint32_t *hkey, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_pkb_head_list, dim3(pkb_grid(n, 256, 8192)), dim3(256), 0, s, head_flag, pos, gsize, n, heads, hkey);
}

// heads / hkey: the group list in descending size (hkey = 255 - min(D, 255), ascending)
void launch_pkb_groups(const NodesDev &nd, const PkbCfg &c, const uint32_t *rowptr, const unsigned long long *gkeys, const unsigned long long *keys,
                       const uint32_t *heads, const uint32_t *hkey, const uint32_t *bound, uint32_t n_heads, unsigned long long *vals, uint64_t n, unsigned long long *marks,
                       unsigned long long *big_marks, unsigned long long *big_cursor, unsigned long long *add_keys, uint64_t add_dense, uint64_t add_cap,
                       unsigned long long *add_overflow, unsigned long long *counters, uint32_t *n_add, uint32_t *left, int n_cu, const uint32_t *tips,
                       const void *tiprec, int legacy, hipStream_t s) {
    if (n == 0 || n_heads == 0) return;
    const PkbTipRec *rec = (const PkbTipRec *) tiprec;
    PkbGraph g{rowptr, gkeys};
    PkbAdd ad{add_keys, add_dense, add_cap, add_overflow};
    const unsigned blocks = (n_heads + 63) / 64;
    const bool staged = nd.stride <= PKB_ROW_WORDS;
    const unsigned cus = (unsigned) std::max(1, n_cu);
    if (staged) {
        (void) hipMemsetAsync(left, 0, (size_t) n_heads * sizeof(uint32_t), s);
        const bool quarter = !(legacy & 1);
        // the ranges of the three shapes come from `bound` on the device: the grids are sized for the chip, not for the list
        hipLaunchKernelGGL(k_pkb_groups_wave, dim3(std::min<unsigned>(n_heads, cus * (quarter ? 8u : 24u))), dim3(64), 0, s, nd, c, g, heads, hkey, bound,
                           quarter ? PKB_QUARTER_MAX + 1 : PKB_SMALL_MAX + 1, (const unsigned long long *) vals, ad, counters, n_add, left, rec);
        if (quarter && (legacy & 8))
            hipLaunchKernelGGL(k_pkb_groups_quarter<true>, dim3(std::min<unsigned>((n_heads + 3) / 4, cus * 17u)), dim3(64), 0, s, nd, c, g, heads, hkey, bound, vals, marks, ad,
                               counters, n_add, left, rec);
        else if (quarter) {
            hipLaunchKernelGGL(k_pkb_groups_quarter<false>, dim3(std::min<unsigned>((n_heads + 3) / 4, cus * 17u)), dim3(64), 0, s, nd, c, g, heads, hkey, bound, vals, marks, ad,
                               counters, n_add, left, rec);
            hipLaunchKernelGGL(k_pkb_quarter_replay, dim3(std::min<unsigned>((n_heads + 255) / 256, cus * 8u)), dim3(256), 0, s, heads, hkey, bound,
                               (const unsigned long long *) vals, (const unsigned long long *) marks, ad, counters, n_add, left);
        }
        hipLaunchKernelGGL(k_pkb_groups_small, dim3(std::min<unsigned>((n_heads + PKB_SMALL_WG - 1) / PKB_SMALL_WG, cus * 3u)), dim3(PKB_SMALL_WG), 0, s, nd, c, g,
                           heads, hkey, bound, (const unsigned long long *) vals, ad, counters, n_add, left, rec);
    }
    hipLaunchKernelGGL(k_pkb_groups_serial, dim3(blocks), dim3(64), 0, s, nd, c, g, keys, heads, hkey, n_heads, staged ? 0 : 1, (const uint32_t *) left, vals, n, marks,
                       big_marks, big_cursor, ad, counters, n_add, tips);
}

void launch_pkb_gather_adds(const uint32_t *heads, const uint32_t *n_add, const uint32_t *pos, uint32_t n_heads, const unsigned long long *add_keys,
                            uint64_t add_dense, uint64_t n_dense_total, uint64_t n_ovf, unsigned long long *out, hipStream_t s) {
    const uint64_t m = std::max<uint64_t>(n_heads, n_ovf);
    if (m == 0) return;
    hipLaunchKernelGGL(k_pkb_gather_adds, dim3((unsigned) ((m + 255) / 256)), dim3(256), 0, s, heads, n_add, pos, n_heads, add_keys, add_dense, n_dense_total, n_ovf, out);
}

} // namespace alga
