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

// returns the number of k-mers written to hash_out / ind_out (at most `intervals`)
__device__ __forceinline__ int li_kmers(const uint32_t *row, int len, int k, int intervals, const int *prio, uint64_t *hash_out, int32_t *ind_out) {
    typedef unsigned __int128 u128;
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
        if (iv != cur) { hash_out[cnt] = mod_hash_u128(best); ind_out[cnt] = best_p; cnt++; cur = iv; best = h; best_p = p; }
        else if (h < best) { best = h; best_p = p; }
    }
    hash_out[cnt] = mod_hash_u128(best); ind_out[cnt] = best_p; cnt++;
    return cnt;
}

// fixed-slot form for tests / function-level parity: slots [node * intervals + j]
__global__ void __launch_bounds__(256) k_li_kmers_slots(NodesDev nd, PkbCfg c, int4 prio4, uint64_t *__restrict__ hash, int32_t *__restrict__ ind,
                                                         int32_t *__restrict__ count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nd.n) return;
    const int prio[4] = {prio4.x, prio4.y, prio4.z, prio4.w};
    uint64_t h[PKB_MAX_INTERVALS]; int32_t p[PKB_MAX_INTERVALS];
    int cnt = 0;
    if (nd.len[i] >= c.li_k) cnt = li_kmers(nd.words + (size_t) i * nd.stride, nd.len[i], c.li_k, c.li_intervals, prio, h, p);
    count[i] = cnt;
    for (int j = 0; j < c.li_intervals; j++) { hash[(size_t) i * c.li_intervals + j] = j < cnt ? h[j] : 0ull; ind[(size_t) i * c.li_intervals + j] = j < cnt ? p[j] : 0; }
}

// ------------------------------------------------------------------------------------------
// masks of the supplement (src/main.cpp:308-322): alignTo = no in-edge but out-edges, alignFrom = in-edges but no out-edge
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_pkb_indeg(const alga_edge_dev *__restrict__ e, uint64_t m, uint32_t *__restrict__ indeg) {
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (uint64_t) gridDim.x * blockDim.x)
        atomicAdd(&indeg[e[i].dst], 1u);
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

__global__ void __launch_bounds__(256) k_pkb_tip_list(int32_t n, const uint32_t *__restrict__ flag, const uint32_t *__restrict__ pos, uint32_t *__restrict__ tips) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && flag[i]) tips[pos[i]] = (uint32_t) i;
}

// k-mers of every node that takes part (GraphCreatorKmerBased::getKmersForBucketJob :202-259), appended to a dense list
//   key = hash, val = node id | (indInRead << 32)
__global__ void __launch_bounds__(256) k_pkb_kmers(NodesDev nd, PkbCfg c, int4 prio4, const uint32_t *__restrict__ tips, uint32_t n_tips,
                                                    unsigned long long *__restrict__ keys, unsigned long long *__restrict__ vals,
                                                    unsigned long long *__restrict__ counter) {
    __shared__ uint32_t s_cnt, s_base_lo, s_base_hi;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const int prio[4] = {prio4.x, prio4.y, prio4.z, prio4.w};
    uint64_t h[PKB_MAX_INTERVALS]; int32_t p[PKB_MAX_INTERVALS];
    int cnt = 0;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    uint32_t i = 0;
    if (t < n_tips) {
        i = tips[t];
        cnt = li_kmers(nd.words + (size_t) i * nd.stride, nd.len[i], c.li_k, c.li_intervals, prio, h, p);
    }
    uint32_t my = 0;
    if (cnt) my = atomicAdd(&s_cnt, (uint32_t) cnt);
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long b = s_cnt ? atomicAdd(counter, (unsigned long long) s_cnt) : 0ull;
        s_base_lo = (uint32_t) b; s_base_hi = (uint32_t) (b >> 32);
    }
    __syncthreads();
    const unsigned long long base = ((unsigned long long) s_base_hi << 32) | s_base_lo;
    for (int j = 0; j < cnt; j++) {
        keys[base + my + j] = h[j];
        vals[base + my + j] = (unsigned long long) i | ((unsigned long long) (uint32_t) p[j] << 32);
    }
}

// ------------------------------------------------------------------------------------------
// k_pkb_groups: one thread per group of equal hash (the thread of the group's first entry).
//   createAlignmentsForKmers (PairwiseKmerBranch.cpp:16-97): entries ordered by (indInRead desc, read length asc, id asc);
//   i from the last-but-one down to the first is the "from" k-mer, j > i the "to" k-mers; offset = ind_i - ind_j.
//   branchMarkers rows are 64-bit masks kept in marks[] (one word per entry); groups larger than 64 use rows of
//   ceil(D/64) words carved from big_marks (offsets from k_pkb_group_sizes + scan on the host side).
//   New edges are written to add_edges at [2 * group_start ...) (capacity 2 * D per group); the rare overflow goes
//   through an atomic cursor behind the dense part.
// ------------------------------------------------------------------------------------------
struct PkbGraph { const uint32_t *rowptr; const alga_edge_dev *edges; };     // snapshot, rows sorted by dst

__device__ __forceinline__ int snapshot_offset(const PkbGraph &g, int a, int b) {
    uint32_t lo = g.rowptr[a], hi = g.rowptr[a + 1];
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        const int d = g.edges[mid].dst;
        if (d == b) return g.edges[mid].offset;
        if (d < b) lo = mid + 1; else hi = mid;
    }
    return PKB_INF;
}

__device__ __forceinline__ uint64_t pkb_order_key(const NodesDev &nd, unsigned long long v) {
    // ascending key == (indInRead descending, read length ascending, node id ascending)   (Kmer::operator<, Kmer.cpp:58-64)
    const uint32_t id = (uint32_t) v, ind = (uint32_t) (v >> 32);
    return ((uint64_t) (0xFFFFu - ind) << 48) | ((uint64_t) (uint32_t) nd.len[id] << 32) | id;
}

__global__ void __launch_bounds__(256) k_pkb_group_sizes(const unsigned long long *__restrict__ keys, uint64_t n,
                                                          unsigned long long *__restrict__ big_words /* total words for groups > 64 */,
                                                          unsigned long long *__restrict__ stats /* [0] groups >= 2, [1] max D */,
                                                          uint32_t *__restrict__ head_flag /* 1 = entry heads a group of >= 2 */) {
    // per-thread tallies, one atomic per wave at the end (a contended atomic per group cost 0.8 ms per round)
    unsigned long long n2 = 0, mx = 0, big = 0;
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
        if (i > 0 && keys[i] == keys[i - 1]) { head_flag[i] = 0u; continue; }
        uint64_t e = i + 1;
        while (e < n && keys[e] == keys[i]) e++;
        const uint64_t D = e - i;
        head_flag[i] = D >= 2 ? 1u : 0u;
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

// one group (the thread owns the entry that heads it); returns the number of canAlign calls
__device__ __forceinline__ unsigned long long pkb_group(const NodesDev &nd, const PkbCfg &c, const PkbGraph &g, const unsigned long long *__restrict__ keys,
                                                       unsigned long long *__restrict__ vals, uint64_t n, uint64_t gs, unsigned long long *__restrict__ marks,
                                                       unsigned long long *__restrict__ big_marks, unsigned long long *__restrict__ big_cursor,
                                                       alga_edge_dev *__restrict__ add_edges, uint64_t add_dense, uint64_t add_cap,
                                                       unsigned long long *__restrict__ add_overflow) {
    if (gs > 0 && keys[gs] == keys[gs - 1]) return 0;                        // not the first entry of its group
    uint64_t ge = gs + 1;
    while (ge < n && keys[ge] == keys[gs]) ge++;
    const int D = (int) (ge - gs);
    if (D < 2) return 0;
    unsigned long long *v = vals + gs;
    for (int i = 1; i < D; i++) {                                            // order the group
        const unsigned long long x = v[i];
        const uint64_t kx = pkb_order_key(nd, x);
        int j = i;
        while (j > 0 && pkb_order_key(nd, v[j - 1]) > kx) { v[j] = v[j - 1]; j--; }
        v[j] = x;
    }
    // does any read occur twice in the group?  (then additions made inside the group must be visible to later pairs)
    bool dup = false;
    for (int i = 0; i < D && !dup; i++) for (int j = i + 1; j < D; j++) if ((uint32_t) v[i] == (uint32_t) v[j]) { dup = true; break; }
    const int RW = (D + 63) >> 6;                                            // words per branch-marker row
    unsigned long long *rows;
    if (D <= 64) rows = marks + gs;
    else rows = big_marks + atomicAdd(big_cursor, (unsigned long long) ((uint64_t) D * RW));
    for (int i = 0; i < D * RW; i++) rows[i] = 0ull;
    alga_edge_dev *mine = add_edges + 2 * gs;                                // dense slots of this group: 2 * D
    int n_add = 0;
    unsigned long long calls = 0;
    for (int i = D - 2; i >= 0; i--) {
        const int id1 = (int) (uint32_t) v[i], ind1 = (int) (uint32_t) (v[i] >> 32);
        const int len1 = nd.len[id1];
        unsigned long long *row_i = rows + (size_t) i * RW;
        for (int j = i + 1; j < D; j++) {
            const int id2 = (int) (uint32_t) v[j];
            if (id1 == id2) continue;
            const int off = ind1 - (int) (uint32_t) (v[j] >> 32);
            if (off < 0) continue;
            if (100 * off > c.max_offset_pct * len1) break;                  // :55
            const int len2 = nd.len[id2];
            const int ov = (len1 < len2 + off ? len1 : len2 + off) - off;
            if (ov < c.min_overlap_area) continue;
            if (len2 + off - len1 < 0) continue;
            if ((row_i[j >> 6] >> (j & 63)) & 1ull) continue;                // already reachable inside the group (:62)
            int cur = snapshot_offset(g, id1, id2);                          // neighbors[id2]
            if (dup) {                                                       // additions this group already made (dense slots only)
                const int lim = n_add < 2 * D ? n_add : 2 * D;
                for (int t = 0; t < lim; t++)
                    if (mine[t].src == id1 && mine[t].dst == id2 && mine[t].offset < cur) cur = mine[t].offset;
            }
            if (cur > off) {
                calls++;
                if (can_align(nd, id1, id2, off, c)) {                       // :66
                    alga_edge_dev ne; ne.src = id1; ne.dst = id2; ne.offset = off;
                    if (n_add < 2 * D) mine[n_add] = ne;
                    else {
                        const unsigned long long k = atomicAdd(add_overflow, 1ull);
                        if (add_dense + k < add_cap) add_edges[add_dense + k] = ne;
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
    // unused dense slots are marked invalid
    for (int t = n_add; t < 2 * D; t++) mine[t].src = -1;
    return calls;
}

// dense list of the entries that head a group of >= 2 (flags from k_pkb_group_sizes, positions from their scan): the group kernel
// then runs with one group per lane instead of one lane in six
__global__ void __launch_bounds__(256) k_pkb_head_list(const uint32_t *__restrict__ head_flag, const uint32_t *__restrict__ pos, uint64_t n,
                                                        uint32_t *__restrict__ heads) {
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x)
        if (head_flag[i]) heads[pos[i]] = (uint32_t) i;
}

__global__ void __launch_bounds__(64) k_pkb_groups(NodesDev nd, PkbCfg c, PkbGraph g, const unsigned long long *__restrict__ keys,
                                                    const uint32_t *__restrict__ heads, uint32_t n_heads,
                                                    unsigned long long *__restrict__ vals, uint64_t n, unsigned long long *__restrict__ marks,
                                                    unsigned long long *__restrict__ big_marks, unsigned long long *__restrict__ big_cursor,
                                                    alga_edge_dev *__restrict__ add_edges, uint64_t add_dense, uint64_t add_cap,
                                                    unsigned long long *__restrict__ add_overflow, unsigned long long *__restrict__ counters) {
    unsigned long long calls = 0;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n_heads) calls = pkb_group(nd, c, g, keys, vals, n, (uint64_t) heads[t], marks, big_marks, big_cursor, add_edges, add_dense, add_cap, add_overflow);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) calls += __shfl_xor(calls, o);          // one counter update per wave
    if ((threadIdx.x & 63u) == 0 && calls) atomicAdd(&counters[0], calls);
}

__global__ void __launch_bounds__(256) k_pkb_mark_unused(const unsigned long long *__restrict__ keys, uint64_t n, alga_edge_dev *__restrict__ add_edges) {
    // slots of entries that head no group of size >= 2 were never touched by k_pkb_groups: mark them invalid
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
        const bool head = i == 0 || keys[i] != keys[i - 1];
        const bool multi = head && (i + 1 < n) && keys[i + 1] == keys[i];
        if (!multi) {
            // find whether this entry's slots belong to a group's dense region: region of group starting at gs covers [2gs, 2gs+2D)
            // -> slots 2i, 2i+1 belong to the group that contains entry i; they are owned (and fully written) by that group.
            if (head) { add_edges[2 * i].src = -1; add_edges[2 * i + 1].src = -1; }     // singleton group: nobody wrote them
        }
    }
}

// edges (old graph + additions) -> sort key (src << 36 | dst << 9 | offset).
// The addition slots of a round are mostly unused (src < 0): only the used entries are worth sorting.
__global__ void __launch_bounds__(256) k_pkb_valid_flags(const alga_edge_dev *__restrict__ e, uint64_t n, uint32_t *__restrict__ flag) {
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) flag[i] = e[i].src >= 0 ? 1u : 0u;
}

__global__ void __launch_bounds__(256) k_pkb_edge_keys_dense(const alga_edge_dev *__restrict__ e, const uint32_t *__restrict__ flag,
                                                              const uint32_t *__restrict__ pos, uint64_t n, unsigned long long *__restrict__ keys,
                                                              unsigned long long *__restrict__ bad /* edges whose offset does not fit the key */) {
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
        if (!flag[i]) continue;
        const alga_edge_dev x = e[i];
        if ((uint32_t) x.offset > 511u) atomicAdd(bad, 1ull);                // never on reads the supplement is meant for (<= 500 nt)
        keys[pos[i]] = ((unsigned long long) (uint32_t) x.src << 36) | ((unsigned long long) (uint32_t) x.dst << 9) | (uint32_t) (x.offset & 511);
    }
}

// after the sort: keep the first key of every (src, dst) run == the smallest offset (Graph::retainOnlySmallestOffset)
__global__ void __launch_bounds__(256) k_pkb_unique_flags(const unsigned long long *__restrict__ keys, uint64_t n, uint32_t *__restrict__ flag) {
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
        const unsigned long long k = keys[i];
        flag[i] = (k != ~0ull && (i == 0 || (keys[i - 1] >> 9) != (k >> 9))) ? 1u : 0u;
    }
}

__global__ void __launch_bounds__(256) k_pkb_compact(const unsigned long long *__restrict__ keys, const uint32_t *__restrict__ flag,
                                                      const uint32_t *__restrict__ pos, uint64_t n, alga_edge_dev *__restrict__ out,
                                                      uint32_t *__restrict__ outdeg) {
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
        if (!flag[i]) continue;
        const unsigned long long k = keys[i];
        alga_edge_dev e; e.src = (int32_t) (k >> 36); e.dst = (int32_t) ((k >> 9) & 0x7FFFFFFull); e.offset = (int32_t) (k & 511ull);
        out[pos[i]] = e;
        atomicAdd(&outdeg[e.src], 1u);
    }
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

void launch_pkb_masks(int32_t n, const uint32_t *rowptr, const alga_edge_dev *edges, uint64_t m, uint32_t *indeg, uint8_t *mask, hipStream_t s) {
    if (n <= 0) return;
    (void) hipMemsetAsync(indeg, 0, sizeof(uint32_t) * (size_t) n, s);
    if (m) hipLaunchKernelGGL(k_pkb_indeg, dim3(pkb_grid(m, 256, 8192)), dim3(256), 0, s, edges, m, indeg);
    hipLaunchKernelGGL(k_pkb_masks, dim3((n + 255) / 256), dim3(256), 0, s, n, rowptr, indeg, mask);
}

void launch_pkb_tip_flags(const NodesDev &nd, const PkbCfg &c, const uint8_t *mask, uint32_t *flag, hipStream_t s) {
    if (nd.n <= 0) return;
    hipLaunchKernelGGL(k_pkb_tip_flags, dim3((nd.n + 255) / 256), dim3(256), 0, s, nd, c, mask, flag);
}

void launch_pkb_tip_list(int32_t n, const uint32_t *flag, const uint32_t *pos, uint32_t *tips, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_pkb_tip_list, dim3((n + 255) / 256), dim3(256), 0, s, n, flag, pos, tips);
}

void launch_pkb_kmers(const NodesDev &nd, const PkbCfg &c, const int32_t prio[4], const uint32_t *tips, uint32_t n_tips, unsigned long long *keys,
                      unsigned long long *vals, unsigned long long *counter, hipStream_t s) {
    if (n_tips == 0) return;
    hipLaunchKernelGGL(k_pkb_kmers, dim3((n_tips + 255) / 256), dim3(256), 0, s, nd, c, make_int4(prio[0], prio[1], prio[2], prio[3]), tips, n_tips, keys, vals,
                       counter);
}

void launch_pkb_group_sizes(const unsigned long long *keys, uint64_t n, unsigned long long *big_words, unsigned long long *stats, uint32_t *head_flag,
                            hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_pkb_group_sizes, dim3(pkb_grid(n, 256 * 8, 1024)), dim3(256), 0, s, keys, n, big_words, stats, head_flag);
}

void launch_pkb_head_list(const uint32_t *head_flag, const uint32_t *pos, uint64_t n, uint32_t *heads, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_pkb_head_list, dim3(pkb_grid(n, 256, 8192)), dim3(256), 0, s, head_flag, pos, n, heads);
}

void launch_pkb_groups(const NodesDev &nd, const PkbCfg &c, const uint32_t *rowptr, const alga_edge_dev *edges, const unsigned long long *keys,
                       const uint32_t *heads, uint32_t n_heads, unsigned long long *vals, uint64_t n, unsigned long long *marks, unsigned long long *big_marks,
                       unsigned long long *big_cursor, alga_edge_dev *add_edges, uint64_t add_dense, uint64_t add_cap,
                       unsigned long long *add_overflow, unsigned long long *counters, hipStream_t s) {
    if (n == 0) return;
    PkbGraph g{rowptr, edges};
    hipLaunchKernelGGL(k_pkb_mark_unused, dim3(pkb_grid(n, 256, 16384)), dim3(256), 0, s, keys, n, add_edges);
    if (n_heads)
        hipLaunchKernelGGL(k_pkb_groups, dim3((n_heads + 63) / 64), dim3(64), 0, s, nd, c, g, keys, heads, n_heads, vals, n, marks, big_marks, big_cursor,
                           add_edges, add_dense, add_cap, add_overflow, counters);
}

void launch_pkb_valid_flags(const alga_edge_dev *e, uint64_t n, uint32_t *flag, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_pkb_valid_flags, dim3(pkb_grid(n, 256, 16384)), dim3(256), 0, s, e, n, flag);
}

void launch_pkb_edge_keys_dense(const alga_edge_dev *e, const uint32_t *flag, const uint32_t *pos, uint64_t n, unsigned long long *keys,
                                unsigned long long *bad, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_pkb_edge_keys_dense, dim3(pkb_grid(n, 256, 16384)), dim3(256), 0, s, e, flag, pos, n, keys, bad);
}

void launch_pkb_unique_flags(const unsigned long long *keys, uint64_t n, uint32_t *flag, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_pkb_unique_flags, dim3(pkb_grid(n, 256, 16384)), dim3(256), 0, s, keys, n, flag);
}

void launch_pkb_compact(const unsigned long long *keys, const uint32_t *flag, const uint32_t *pos, uint64_t n, alga_edge_dev *out,
                        uint32_t *outdeg, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_pkb_compact, dim3(pkb_grid(n, 256, 16384)), dim3(256), 0, s, keys, flag, pos, n, out, outdeg);
}

} // namespace alga
