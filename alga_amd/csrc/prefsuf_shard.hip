// alga_amd/csrc/prefsuf_shard.hip -- the PrefSuf overlap graph with the INDEX SHARDED BY SEED BUCKET over the N GPUs of a node (gfx950).
//
// The reference's parallelism is a thread pool over read ids inside one address space (src/GraphCreators/GraphCreatorPrefSuf.cpp:150-161,
// 299-306: every thread reads every bucket).  Across GPUs the bucket table itself is what has to be divided: building the entry array
// of ALL targets on every GPU (round 3's replicated form) costs 6.4 ms per rank at the north-star size whatever N is.  Here rank g
// owns the targets whose minimizer bucket lies in its slice of the bucket space (n_buckets / N consecutive buckets):
//
//   keys       (existing) k_node_runs on the rank's own node range: the node's sort key as a target, its runs as a source
//   exchange 1 (driver)   the per-node key array is all-gathered (4 B / node)
//   k_shard_select        the (key, id) pairs of MY bucket range out of all keys; sort + k_tgt_gather + k_tgt_dir on those alone
//                         (prefsuf_cluster.hip: launch_cluster_store_slice): 1 / N of the entry array and of its directory
//   k_shard_export        the rank's own sources' runs as DESCRIPTORS {cluster key, source id, q | p0 | p1} (12 B), partitioned by the
//                         rank that owns the run's bucket; sources whose runs k_node_runs could not list are re-done by brute force
//   exchange 2 (driver)   all-to-all of the descriptors (the rows themselves are replicated: nothing but 12 B per run travels)
//   sort by bucket, then
//   k_shard_join          a wave per BUCKET: every (source run, target) pair of the bucket is verified bit for bit, and the transitive
//                         reduction is decided THERE, per target, from the target's complete candidate list -- every via B of an
//                         overlap A -> C is itself a source of C, so it reached C's bucket too (tests/bucket_side_rule.py is the
//                         executable statement, checked against the oracle's literal replay).  Survivors with a big overlap are
//                         final; a surviving SMALL overlap (L < RSOEMO) still has to pass its source's cap of three
//                         (GraphCreatorPrefSuf.cpp:397-401), the one decision that needs the source's OTHER overlaps:
//   exchange 3, 4 (driver) the ids of the sources with such pending edges are all-gathered (a few thousand), every rank lists the
//                         top-3 small (L, C) keys its descriptors of those sources saw, the lists are all-gathered, and
//   k_shard_resolve       drops the pending edges that are not among their source's three largest small keys
//   exchange 5 (driver)   final edges to the rank that owns the SOURCE id range (all-to-all, 12 B / edge)
//   k_shard_place_*       adjacency lists of the rank's sources: count, scan, fill, per-row order
//   gather     (existing) the per-range lists concatenate to the single-GPU byte order on rank 0.
//
// Nothing here approximates: verification is the exact 2-bit compare, the reduction is the reference's rule.  Inputs the form does
// not take (the clustered probe's own limits, a bucket with more than SJ_DMAX descriptors) make every rank fall back together.
#include <hip/hip_runtime.h>
#include <algorithm>
#include "prefsuf_common.h"
#include "prefsuf_kernels.h"
#include "prefsuf_device.h"
#include "prefsuf_cluster_device.h"
#include "prefsuf_shard.h"

namespace alga {

constexpr int SJ_QW = 17;            // staged words per source row: 13 row words + the compare's reach past them (offset <= 63 nt = 3 words), odd = conflict-free
constexpr int SJ_DMAX = 4096;        // descriptors of one bucket the join takes (64 V masks per wave); beyond: the build declines
constexpr int SJ_WAVES = 4;

__device__ __forceinline__ uint32_t shard_owner(uint32_t key, int shift, uint32_t bpr, uint32_t n_ranks) {
    const uint32_t o = (key >> shift) / bpr;
    return o < n_ranks ? o : n_ranks - 1u;
}

// min over the 64 lanes, uniform (lanes without a value pass 0xFFFFFFFF)
__device__ __forceinline__ uint32_t wave_min_u32_dpp(uint32_t v) {
    uint32_t t;
    t = (uint32_t) __builtin_amdgcn_update_dpp(-1, (int) v, 0x111, 0xF, 0xF, false); v = t < v ? t : v;
    t = (uint32_t) __builtin_amdgcn_update_dpp(-1, (int) v, 0x112, 0xF, 0xF, false); v = t < v ? t : v;
    t = (uint32_t) __builtin_amdgcn_update_dpp(-1, (int) v, 0x114, 0xF, 0xF, false); v = t < v ? t : v;
    t = (uint32_t) __builtin_amdgcn_update_dpp(-1, (int) v, 0x118, 0xF, 0xF, false); v = t < v ? t : v;
    t = (uint32_t) __builtin_amdgcn_update_dpp(-1, (int) v, 0x142, 0xA, 0xF, false); v = t < v ? t : v;
    t = (uint32_t) __builtin_amdgcn_update_dpp(-1, (int) v, 0x143, 0xC, 0xF, false); v = t < v ? t : v;
    return (uint32_t) __builtin_amdgcn_readlane((int) v, 63);
}

// max over the 64 lanes, uniform
__device__ __forceinline__ uint32_t wave_max_u32_dpp_or0(uint32_t v) {
    uint32_t t;
    t = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x111, 0xF, 0xF, false); v = t > v ? t : v;
    t = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x112, 0xF, 0xF, false); v = t > v ? t : v;
    t = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x114, 0xF, 0xF, false); v = t > v ? t : v;
    t = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x118, 0xF, 0xF, false); v = t > v ? t : v;
    t = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x142, 0xA, 0xF, false); v = t > v ? t : v;
    t = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x143, 0xC, 0xF, false); v = t > v ? t : v;
    return (uint32_t) __builtin_amdgcn_readlane((int) v, 63);
}

// ------------------------------------------------------------------------------------------
// k_shard_select : (key, id) of the targets whose bucket lies in [b_lo, b_hi)
// ------------------------------------------------------------------------------------------
// One global atomic per BLOCK of 8192 keys (a returning atomic on one address sustains ~88 ops / us chip-wide): every thread keeps
// the matches of its 32 keys as a bit mask, the block scans the counts in LDS, reserves its output range once.
constexpr int SEL_IT = 32, SEL_BLOCK = 256;
__global__ void __launch_bounds__(SEL_BLOCK) k_shard_select(const uint32_t *__restrict__ keys, uint32_t n, int shift, uint32_t b_lo, uint32_t b_hi,
                                                             uint32_t *__restrict__ okeys, uint32_t *__restrict__ ovals, unsigned long long *__restrict__ cursor) {
    __shared__ uint32_t s_cnt[SEL_BLOCK];
    __shared__ unsigned long long s_base;
    const uint64_t base = (uint64_t) blockIdx.x * (SEL_BLOCK * SEL_IT);
    const int t = (int) threadIdx.x;
    uint32_t mask = 0, mine = 0;
    // thread t takes the keys base + k * SEL_BLOCK + t: coalesced
#pragma unroll 4
    for (int k = 0; k < SEL_IT; k++) {
        const uint64_t i = base + (uint64_t) k * SEL_BLOCK + t;
        if (i < n) {
            const uint32_t x = keys[i];
            const uint32_t b = x >> shift;
            if (x != 0xFFFFFFFFu && b >= b_lo && b < b_hi) { mask |= 1u << k; mine++; }
        }
    }
    s_cnt[t] = mine;
    __syncthreads();
    for (int d = 1; d < SEL_BLOCK; d <<= 1) {              // inclusive scan
        const uint32_t v = t >= d ? s_cnt[t - d] : 0u;
        __syncthreads();
        s_cnt[t] += v;
        __syncthreads();
    }
    if (t == SEL_BLOCK - 1) s_base = s_cnt[t] ? atomicAdd(cursor, (unsigned long long) s_cnt[t]) : 0ull;
    __syncthreads();
    unsigned long long at = s_base + (s_cnt[t] - mine);
    while (mask) {
        const int k = __builtin_ctz(mask);
        mask &= mask - 1u;
        const uint64_t i = base + (uint64_t) k * SEL_BLOCK + t;
        okeys[at] = keys[i]; ovals[at] = (uint32_t) i;
        at++;
    }
}

// ------------------------------------------------------------------------------------------
// k_shard_export : runs of the rank's own sources -> descriptors, partitioned by the owner of the run's bucket
// ------------------------------------------------------------------------------------------
// COUNT = true : counts[o] += descriptors for owner o; sources k_node_runs flagged (a window without a class-0 k-mer, more runs or
//                records than it stores) are appended to flagged_list (their descriptors are made by k_shard_export_flagged).
// COUNT = false: descriptors written to out[3 * (seg_off[o] + position)], positions from cursor[o] (one global atomic per block and owner).
constexpr int EXP_BLOCK = 256;
template <bool COUNT>
__global__ void __launch_bounds__(EXP_BLOCK) k_shard_export(const uint2 *__restrict__ runs, int32_t node_begin, int32_t node_end, int shift, uint32_t bpr, uint32_t n_ranks,
                                                             unsigned long long *__restrict__ counts, int32_t *__restrict__ flagged_list, unsigned long long *__restrict__ flagged_count,
                                                             uint32_t flagged_cap, const unsigned long long *__restrict__ seg_off, unsigned long long *__restrict__ cursor,
                                                             uint32_t *__restrict__ out) {
    __shared__ uint32_t s_cnt[64];
    __shared__ unsigned long long s_base[64];
    const int t = (int) threadIdx.x;
    if (t < 64) s_cnt[t] = 0u;
    __syncthreads();
    const int64_t i = (int64_t) node_begin + (int64_t) blockIdx.x * EXP_BLOCK + t;
    uint2 r[CL_RMAX];
    uint32_t slot[CL_RMAX];
    int nr = 0;
    if (i < node_end) {
        const uint2 r0 = runs[(size_t) i * CL_RMAX];
        nr = (int) (r0.y >> 24);
        if (nr == CL_RUNS_FLAGGED) {
            if (COUNT) { const unsigned long long k = atomicAdd(flagged_count, 1ull); if (k < flagged_cap) flagged_list[k] = (int32_t) i; }
            nr = 0;
        }
        nr = nr > CL_RMAX ? CL_RMAX : nr;
#pragma unroll
        for (int k = 0; k < CL_RMAX; k++) {
            if (k < nr) {
                r[k] = k == 0 ? r0 : runs[(size_t) i * CL_RMAX + k];
                slot[k] = atomicAdd(&s_cnt[shard_owner(r[k].x, shift, bpr, n_ranks)], 1u);
            }
        }
    }
    __syncthreads();
    if (t < (int) n_ranks && s_cnt[t]) {
        if (COUNT) atomicAdd(&counts[t], (unsigned long long) s_cnt[t]);
        else s_base[t] = seg_off[t] + atomicAdd(&cursor[t], (unsigned long long) s_cnt[t]);
    }
    if (COUNT) return;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < CL_RMAX; k++) {
        if (k < nr) {
            const unsigned long long at = s_base[shard_owner(r[k].x, shift, bpr, n_ranks)] + slot[k];
            out[3 * at] = r[k].x; out[3 * at + 1] = (uint32_t) i; out[3 * at + 2] = r[k].y & 0xFFFFFFu;
        }
    }
}

// The flagged sources (1 in ~10^3): window minimizers by brute force (lane p scans the w k-mers of window p), runs by ballot -- what
// k_probe_clustered's slow path does for such a source -- one wave per source; a descriptor per run, appended behind the owner's
// regular ones (cursor[o]; the owner's segment has room for 64 per flagged source).
__global__ void __launch_bounds__(256) k_shard_export_flagged(NodesDev nd, PrefSufCfg cfg, ClusterCfg cc, const int32_t *__restrict__ flagged_list, uint32_t n_flagged,
                                                               uint32_t bpr, uint32_t n_ranks, const unsigned long long *__restrict__ seg_off,
                                                               unsigned long long *__restrict__ cursor, uint32_t *__restrict__ out) {
    __shared__ uint32_t sB[4][STAGE_WORDS];
    const int wave = (int) (threadIdx.x >> 6), lane = lane_id();
    const uint32_t k = blockIdx.x * 4u + (uint32_t) wave;
    if (k >= n_flagged) return;
    const int b = flagged_list[k];
    const int lenB = nd.len[b];
    const int nwB = blocks_of(lenB);
    if (lane < STAGE_WORDS) sB[wave][lane] = (lane < nwB && lane < nd.stride) ? nd.words[(size_t) b * nd.stride + lane] : 0u;
    wave_lds_fence();
    const uint32_t *sb = sB[wave];
    const int nwin = lenB - cfg.Lmin + 1;
    uint32_t wm = 0xFFFFFFFFu;
    for (int j = 0; j < cc.w; j++) {                       // uniform
        uint32_t h, pk;
        kmer_key(sb, (lane < nwin ? lane : 0) + j, true, cc, h, pk);
        wm = pk < wm ? pk : wm;
    }
    const bool wv = lane < nwin;
    const uint32_t prev = bperm(wm, (lane + 63) & 63);
    const bool start = wv && (lane == 0 || wm != prev);
    const uint64_t runmask = __ballot(start);
    const uint64_t higher = lane >= 63 ? 0ull : runmask & ~((2ull << lane) - 1ull);
    const int p1 = higher ? __builtin_ctzll(higher) : nwin;
    if (start) {
        uint32_t h, pk;
        const int q = (int) (wm & 255u);
        kmer_key(sb, q < 128 ? q : 0, true, cc, h, pk);
        const uint32_t key = cluster_key(h, cc.idx_shift - CL_MBITS);
        const uint32_t o = shard_owner(key, cc.idx_shift, bpr, n_ranks);
        const unsigned long long at = seg_off[o] + atomicAdd(&cursor[o], 1ull);
        out[3 * at] = key; out[3 * at + 1] = (uint32_t) b; out[3 * at + 2] = (uint32_t) q | ((uint32_t) lane << 8) | ((uint32_t) p1 << 16);
    }
}

// received descriptors {key, src, y} -> sort key + payload
// (the key RELATIVE to the first key of the rank's bucket range: its bits above the range's width are zero and need not be sorted)
__global__ void __launch_bounds__(256) k_shard_desc_split(const uint32_t *__restrict__ in, uint64_t n, uint32_t key_base, uint32_t *__restrict__ dkey, unsigned long long *__restrict__ dval) {
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
        dkey[i] = in[3 * i] - key_base;
        dval[i] = (unsigned long long) in[3 * i + 1] | ((unsigned long long) in[3 * i + 2] << 32);
    }
}

// ------------------------------------------------------------------------------------------
// k_shard_groups : the descriptor groups (one per bucket) of the bucket-sorted descriptor array
// ------------------------------------------------------------------------------------------
// flag[i] = 1 where descriptor i opens a group; the exclusive scan of the flags numbers the groups; gstart[g] = first descriptor of
// group g, gstart[n_groups] = n_desc.  (A list, because the join packs WHOLE groups onto a wave's lanes.)
__global__ void __launch_bounds__(256) k_shard_group_flags(const uint32_t *__restrict__ dkey, uint64_t n, int shift, uint32_t *__restrict__ flag) {
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x)
        flag[i] = (i == 0 || (dkey[i] >> shift) != (dkey[i - 1] >> shift)) ? 1u : 0u;
}
__global__ void __launch_bounds__(256) k_shard_group_starts(const uint32_t *__restrict__ flag, const uint32_t *__restrict__ pos, uint64_t n, uint32_t *__restrict__ gstart) {
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += (uint64_t) gridDim.x * blockDim.x) {
        if (i == n) gstart[pos[n]] = (uint32_t) n;          // pos[n] = number of groups (launch_exclusive_scan writes the total there)
        else if (flag[i]) gstart[pos[i]] = (uint32_t) i;
    }
}

// ------------------------------------------------------------------------------------------
// k_shard_join
// ------------------------------------------------------------------------------------------
// Descriptors sorted by bucket, groups listed (gstart).  A wave takes chunks of SJ_GCH consecutive groups and packs WHOLE groups
// onto its 64 lanes for as long as they fit (a 150-bp bucket at 30x holds ~20 descriptors: three groups a pass) -- lanes =
// descriptors (source runs), every lane with its own bucket's entry range:
//   pass 1  entry c of the lane's own bucket, c = 0 .. the longest bucket of the pass (lanes of one group read the same entry: one
//           fetch): the lane verifies its source against the entry at the one offset their minimizers allow (p = q - m_C, inside the
//           run's windows), exact 2-bit compare from the source's row staged in LDS -> bit c of the lane's candidate mask; the top-3
//           small (L, C) keys of the run go to small_top[] (the cap's raw material);
//   pass 2  entry c again (= one TARGET per group): the vias B of a group in the order of their offsets, smallest first (a
//           segmented minimum through one LDS word per group); B removes candidate A when B -> C is big, B sits at offset
//           delta = p_A - p_B >= 0 of A, reaches at least to A's end but not past C's, and A[delta, p_A) == B[0, p_B) -- beyond that
//           both already equal C (the reference's compare, GraphCreatorPrefSuf.cpp:434-451: p_B <= 63 nucleotides, four words); a second
//           descriptor of the SAME source at a smaller offset supersedes.  On error-free data the first B removes every other
//           candidate and the loop ends after one step.
//   Survivors leave as records {target, offset | L | small, source} through the chunked record list of the probes
//   (prefsuf_device.h flush_records); `small` marks the ones that still have to pass their source's cap.
// A group with more than 64 descriptors or entries takes the same steps alone, chunk by chunk, with the candidate masks of one
// target in LDS and the vias' rows from global memory (repeats, 400x coverage): slow and rare.
constexpr int SJ_GCH = 32;           // groups per chunk of a wave's share
constexpr int SJ_WB = 128;           // per-wave LDS record buffer
constexpr int SJ_ES = 96;            // entries of one pass whose id / meta stay in LDS for pass 2
template <int EQ, int KF>
__global__ void __launch_bounds__(SJ_WAVES * 64)
k_shard_join(NodesDev nd, PrefSufCfg cfg, ClusterCfg cc, int ulen, const uint4 *__restrict__ store, const uint4 *__restrict__ dir, uint32_t bucket_base,
             const uint32_t *__restrict__ dkey, const unsigned long long *__restrict__ dval, uint64_t n_desc, const uint32_t *__restrict__ gstart, uint32_t n_groups,
             ProbeOut o, unsigned long long *__restrict__ small_top, unsigned long long *__restrict__ declined, int dmax /* <= SJ_DMAX */) {
    constexpr int WC = 4 * EQ - 3;
    __shared__ uint32_t sA[SJ_WAVES][64][SJ_QW];
    __shared__ uint32_t sSrc[SJ_WAVES][64], sLen[SJ_WAVES][64], sMin[SJ_WAVES][64];
    __shared__ uint32_t sEI[SJ_WAVES][SJ_ES], sEM[SJ_WAVES][SJ_ES];     // id / meta of the pass's entries, group by group
    __shared__ uint32_t sTab[SJ_WAVES][SJ_ES];                          // ... and the first candidate (offset << 6 | lane) of each
    __shared__ unsigned long long sVM[SJ_WAVES][SJ_DMAX / 64];
    __shared__ uint32_t sRecC[SJ_WAVES][SJ_WB];
    __shared__ unsigned long long sRecV[SJ_WAVES][SJ_WB];
    __shared__ uint32_t sCnt[SJ_WAVES][2];
    const int wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    const int lane = lane_id();
    WaveLds w{nullptr, nullptr, nullptr, nullptr, sRecC[wave], sRecV[wave], &sCnt[wave][0]};
    if (lane == 0) *w.recN = 0;
    wave_lds_fence();
    uint64_t chunk_base = 0;
    int chunk_fill = REC_CHUNK;
    uint64_t st_rec = 0;
    uint32_t st_pass = 0, st_serial = 0;                   // uniform: passes of this wave, passes decided target by target
    const int fs = cc.idx_shift - CL_MBITS, shift = cc.idx_shift;
    const int Lbig = cfg.rsoemo > cfg.Lmin ? cfg.rsoemo : cfg.Lmin;
    const int kfull = KF ? KF : (2 * cfg.Lmin) >> 5;       // row words every overlap covers entirely
    uint32_t *sb = sA[wave][lane];

    // per-lane state of the descriptors in the lanes
    bool act = false;
    uint32_t key = 0, src = 0;
    int q = 0, p0 = 0, p1 = 0, lenA = 0;
    auto load_descs = [&](uint64_t first, int count) {     // descriptors first .. first + count - 1 -> lanes 0 .. count - 1, rows staged
        act = lane < count;
        const uint64_t idx = act ? first + (uint64_t) lane : first;
        const unsigned long long v = dval[idx];
        key = dkey[idx] + (bucket_base << shift);           // (stored relative to the rank's first key)
        src = (uint32_t) v;
        const uint32_t y = (uint32_t) (v >> 32);
        q = (int) (y & 255u); p0 = (int) ((y >> 8) & 255u); p1 = (int) ((y >> 16) & 255u);
        lenA = act ? (ulen > 0 ? ulen : nd.len[src]) : 0;
        const int nw = blocks_of(lenA);
        const uint32_t *row = nd.words + (size_t) src * nd.stride;
        wave_lds_fence();
        if ((nd.stride & 3) == 0) {                        // the engine's own layout: 16-byte pieces
            const uint4 *r4 = reinterpret_cast<const uint4 *>(row);
#pragma unroll
            for (int k4 = 0; k4 < EQ; k4++) {
                const uint4 x = 4 * k4 < nd.stride ? r4[k4] : make_uint4(0u, 0u, 0u, 0u);
                if (4 * k4 + 0 < SJ_QW) sb[4 * k4 + 0] = 4 * k4 + 0 < nw ? x.x : 0u;
                if (4 * k4 + 1 < SJ_QW) sb[4 * k4 + 1] = 4 * k4 + 1 < nw ? x.y : 0u;
                if (4 * k4 + 2 < SJ_QW) sb[4 * k4 + 2] = 4 * k4 + 2 < nw ? x.z : 0u;
                if (4 * k4 + 3 < SJ_QW) sb[4 * k4 + 3] = 4 * k4 + 3 < nw ? x.w : 0u;
            }
#pragma unroll
            for (int k = 4 * EQ; k < SJ_QW; k++) sb[k] = 0u;
        } else {
#pragma unroll
            for (int k = 0; k < SJ_QW; k++) sb[k] = (k < nw && k < nd.stride) ? row[k] : 0u;
        }
        sSrc[wave][lane] = src; sLen[wave][lane] = (uint32_t) lenA;
        wave_lds_fence();
    };
    // does this lane's run overlap the entry?  p = the one offset their minimizers allow
    auto verify = [&](bool has, const uint32_t (&ew)[4 * EQ], int &p) -> bool {
        const uint32_t id = ew[4 * EQ - 3], ekey = ew[4 * EQ - 2], meta = ew[4 * EQ - 1];
        const int lenC = (int) ((meta >> 8) & 0xFFFu);
        p = q - (int) (meta & 255u);
        const bool ok = has & same_cluster(ekey, key, fs) & (p >= p0) & (p < p1) & (id != src) & (lenC >= lenA - p);
        p = ok ? p : 0;
        const int nb = 2 * (lenA - p), qw = (2 * p) >> 5, sh = (2 * p) & 31;
        uint32_t diff = 0;
#pragma unroll
        for (int k = 0; k < WC; k++) {
            const uint32_t x = funnel(sb[qw + k], sb[qw + k + 1], sh) ^ ew[k];
            if (k < kfull) diff |= x;                      // (compile time with KF)
            else diff |= x & low_bits32(nb - 32 * k);
        }
        return ok && diff == 0;
    };
    // does via B (id, length, offset pB into C, its first four row words) remove this lane's candidate (offset p into C, rho_c past A's end)?
    auto via = [&](uint32_t srcB, int lenB, int pB, uint32_t b0, uint32_t b1, uint32_t b2, uint32_t b3, int p, int rho_c) -> bool {
        const int delta = p - pB;
        const bool same = srcB == src;
        const int rho_b = lenB - (lenA - delta);
        const bool vok = (!same) & (pB > 0) & (delta >= 0) & (lenB - pB >= Lbig) & (rho_b >= 0) & (rho_b <= rho_c) & ((rho_b > 0) | (srcB > src));
        const int dd = delta < 0 ? 0 : delta;
        const int nb = 2 * pB, qw = (2 * dd) >> 5, sh = (2 * dd) & 31;       // A[delta, delta + pB) against B[0, pB): pB <= 63
        const uint32_t diff = ((funnel(sb[qw], sb[qw + 1], sh) ^ b0) & low_bits32(nb)) | ((funnel(sb[qw + 1], sb[qw + 2], sh) ^ b1) & low_bits32(nb - 32)) |
                              ((funnel(sb[qw + 2], sb[qw + 3], sh) ^ b2) & low_bits32(nb - 64)) | ((funnel(sb[qw + 3], sb[qw + 4], sh) ^ b3) & low_bits32(nb - 96));
        return (vok && diff == 0) || (same && pB < p);
    };
    auto push = [&](uint32_t C, int p) {
        const int L = lenA - p;
        const unsigned long long val = ((unsigned long long) ol_pack(p, L, L < cfg.rsoemo) << 32) | src;
        const uint32_t i = atomicAdd(w.recN, 1u);
        if (i < (uint32_t) SJ_WB) { w.recC[i] = C; w.recV[i] = val; }
        else store_record(o, atomicAdd(&o.counters[CNT_RECORDS], 1ull), C, val);
        st_rec++;
    };
    auto maybe_flush = [&]() {
        wave_lds_fence();
        const int nb2 = (int) __builtin_amdgcn_readfirstlane((int) *w.recN);
        if (nb2 >= SJ_WB / 2) flush_records<REC_CHUNK, SJ_WB>(o, w, chunk_base, chunk_fill);
    };

    // ---------------- one big bucket alone: chunks of 64 descriptors, candidate masks of one target in LDS, via rows from global memory ----------------
    auto big_group = [&](uint64_t gs, int D) {
        if (D > dmax) { if (lane == 0) atomicOr(declined, 1ull); return; }
        const uint32_t b = (uint32_t) __builtin_amdgcn_readfirstlane((int) (dkey[gs] >> shift));       // relative to the rank's first bucket
        const uint4 rec = dir[b];
        const uint32_t e0 = rec.x, ecnt = rec.y;
        const int nch = (D + 63) >> 6;
        for (int cc2 = 0; cc2 < nch; cc2++) {
            const int cnt = D - 64 * cc2 < 64 ? D - 64 * cc2 : 64;
            if (lane < cnt) { const uint64_t i = gs + 64u * (uint64_t) cc2 + lane; small_top[3 * i] = 0ull; small_top[3 * i + 1] = 0ull; small_top[3 * i + 2] = 0ull; }
        }
        for (uint32_t c = 0; c < ecnt; c++) {              // uniform
            uint32_t ew[4 * EQ];
            {
                const uint4 *ent = store + (size_t) (e0 + c) * EQ;
#pragma unroll
                for (int k4 = 0; k4 < EQ; k4++) { const uint4 v = ent[k4]; ew[4 * k4] = v.x; ew[4 * k4 + 1] = v.y; ew[4 * k4 + 2] = v.z; ew[4 * k4 + 3] = v.w; }
            }
            const uint32_t meta = ew[4 * EQ - 1], idC = ew[4 * EQ - 3];
            const int lenC = (int) ((meta >> 8) & 0xFFFu);
            uint64_t any = 0ull;
            for (int cc2 = 0; cc2 < nch; cc2++) {
                const int cnt = D - 64 * cc2 < 64 ? D - 64 * cc2 : 64;
                load_descs(gs + 64u * (uint64_t) cc2, cnt);
                int p;
                const bool V = verify(act, ew, p);
                const uint64_t m = __ballot(V);
                if (lane == 0) sVM[wave][cc2] = m;
                any |= m;
                const int L = lenA - p;
                if (V && L < cfg.rsoemo) {
                    const uint64_t i = gs + 64u * (uint64_t) cc2 + lane;
                    uint64_t a = small_top[3 * i], b2 = small_top[3 * i + 1], c2 = small_top[3 * i + 2];
                    top3_insert(a, b2, c2, ((uint64_t) (uint32_t) L << 32) | idC);
                    small_top[3 * i] = a; small_top[3 * i + 1] = b2; small_top[3 * i + 2] = c2;
                }
            }
            wave_lds_fence();
            if (any == 0ull) continue;
            for (int cc2 = 0; cc2 < nch; cc2++) {          // candidate chunk
                const uint64_t cm = sVM[wave][cc2];
                if (cm == 0ull) continue;
                const int cnt = D - 64 * cc2 < 64 ? D - 64 * cc2 : 64;
                if (nch > 1) load_descs(gs + 64u * (uint64_t) cc2, cnt);
                const bool V = ((cm >> lane) & 1ull) != 0ull;
                const int p = q - (int) (meta & 255u);
                const int rho_c = lenC - (lenA - p);
                bool removed = false;
                for (int bb = 0; bb < nch; bb++) {
                    for (uint64_t bm = sVM[wave][bb]; bm != 0ull; bm &= bm - 1ull) {     // uniform: every via, in index order
                        const int bl = __builtin_ctzll(bm);
                        const bool test = V && !removed && !(bb == cc2 && bl == lane);
                        if (__ballot(test) == 0ull) continue;
                        const uint64_t bi = gs + 64u * (uint64_t) bb + (uint64_t) bl;
                        const unsigned long long vB = dval[bi];
                        const uint32_t srcB = (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) vB);
                        const int pB = (int) ((uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) (vB >> 32)) & 255u) - (int) (meta & 255u);
                        const int lenB = ulen > 0 ? ulen : __builtin_amdgcn_readfirstlane(nd.len[srcB]);
                        const int nwB = blocks_of(lenB);
                        const uint32_t *rb = nd.words + (size_t) srcB * nd.stride;
                        uint32_t rB[4];
#pragma unroll
                        for (int k = 0; k < 4; k++) rB[k] = (k < nwB && k < nd.stride) ? rb[k] : 0u;
                        removed = removed || (test && p >= pB && via(srcB, lenB, pB, rB[0], rB[1], rB[2], rB[3], p, rho_c));
                    }
                }
                if (V && !removed) push(idC, p);
                maybe_flush();
            }
        }
    };

    const uint32_t total_waves = gridDim.x * SJ_WAVES;
    for (uint32_t g0 = ((uint32_t) blockIdx.x * SJ_WAVES + (uint32_t) wave) * SJ_GCH; g0 < n_groups; g0 += total_waves * SJ_GCH) {
        const uint32_t gend = g0 + SJ_GCH < n_groups ? g0 + SJ_GCH : n_groups;
        uint32_t ga = g0;
        while (ga < gend) {                                // uniform: one pass per iteration
            // whole groups from ga on for as long as they fit the 64 lanes
            const uint32_t s0 = gstart[ga];
            const uint32_t gi = ga + 1u + (uint32_t) lane;
            const uint32_t ge = gstart[gi <= gend ? gi : gend];
            const uint64_t fits = __ballot(gi <= gend && ge - s0 <= 64u);
            const int ng = fits == ~0ull ? 64 : __builtin_ctzll(~fits);
            if (ng == 0) {                                 // the group at ga alone exceeds a pass
                big_group((uint64_t) s0, (int) min(gstart[ga + 1u] - s0, (uint32_t) SJ_DMAX + 1u));
                ga++;
                continue;
            }
            const int cnt = (int) ((uint32_t) __builtin_amdgcn_readlane((int) ge, ng - 1) - s0);
            ga += (uint32_t) ng;
            load_descs((uint64_t) s0, cnt);
            // ---- the lanes' groups: bucket, entry range, first lane ----
            const uint32_t b = act ? key >> shift : 0xFFFFFFFFu;
            const uint32_t bprev = bperm(b, (lane + 63) & 63);
            const uint64_t starts = __ballot(act && (lane == 0 || b != bprev));
            const int gfirst = 63 - __builtin_clzll((long long) ((starts & (lane >= 63 ? ~0ull : ((2ull << lane) - 1ull))) | 1ull));
            const uint4 rec = dir[act ? b - bucket_base : 0u];       // (key is absolute again: load_descs)
            uint32_t e0 = rec.x, ecnt = act ? rec.y : 0u;
            // a bucket with more than 64 entries (several loci in one bucket at extreme coverage): that group alone, afterwards
            const uint64_t wide = __ballot(act && ecnt > 64u);
            if (wide != 0ull) ecnt = ecnt > 64u ? 0u : ecnt;
            const uint32_t max_e = wave_max_u32_dpp_or0(ecnt);
            // slots of the pass's entries in LDS (id, meta: what pass 2 needs of an entry): group by group, SJ_ES in all (beyond: re-read)
            int ebase;
            {
                uint32_t inc = (act && lane == gfirst) ? ecnt : 0u, t;      // inclusive scan over the lanes
                t = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) inc, 0x111, 0xF, 0xF, false); inc += t;
                t = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) inc, 0x112, 0xF, 0xF, false); inc += t;
                t = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) inc, 0x114, 0xF, 0xF, false); inc += t;
                t = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) inc, 0x118, 0xF, 0xF, false); inc += t;
                t = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) inc, 0x142, 0xA, 0xF, false); inc += t;
                t = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) inc, 0x143, 0xC, 0xF, false); inc += t;
                ebase = (int) (bperm(inc, gfirst) - ecnt);                    // entries of the groups before this lane's
            }
            // ---- pass 1: the entry loads of step c + 1 are in flight while step c is verified ----
            uint64_t vmask = 0, k0 = 0, k1 = 0, k2 = 0;
            uint32_t en[4 * EQ];
            auto load_entry = [&](uint32_t c, uint32_t (&ew)[4 * EQ]) {
                const uint4 *ent = store + (size_t) (c < ecnt ? e0 + c : e0) * EQ;
#pragma unroll
                for (int k4 = 0; k4 < EQ; k4++) { const uint4 v = ent[k4]; ew[4 * k4] = v.x; ew[4 * k4 + 1] = v.y; ew[4 * k4 + 2] = v.z; ew[4 * k4 + 3] = v.w; }
            };
            load_entry(0u, en);
            for (uint32_t c = 0; c < max_e; c++) {         // uniform
                uint32_t ew[4 * EQ];
#pragma unroll
                for (int k = 0; k < 4 * EQ; k++) ew[k] = en[k];
                load_entry(c + 1u, en);                    // (past the bucket's end: its first entry again, never used)
                const bool has = c < ecnt;
                int p;
                const bool V = verify(has, ew, p);
                vmask |= V ? 1ull << c : 0ull;
                const int L = lenA - p;
                if (V && L < cfg.rsoemo) top3_insert(k0, k1, k2, ((uint64_t) (uint32_t) L << 32) | ew[4 * EQ - 3]);
                if (has && lane == gfirst && ebase + (int) c < SJ_ES) { sEI[wave][ebase + (int) c] = ew[4 * EQ - 3]; sEM[wave][ebase + (int) c] = ew[4 * EQ - 1]; }
            }
            if (act && ((wide >> gfirst) & 1ull) == 0ull) { small_top[3 * ((uint64_t) s0 + lane)] = k0; small_top[3 * ((uint64_t) s0 + lane) + 1] = k1; small_top[3 * ((uint64_t) s0 + lane) + 2] = k2; }
            wave_lds_fence();
            // ---- pass 2, all targets at once: every candidate against the FIRST via of its target (the candidate with the smallest offset: a
            //      segmented minimum through one LDS word per target).  On error-free data that via removes every other candidate and itself
            //      stands: one step per overlap a lane holds (~4) instead of two per entry of the longest bucket (~11).  Anything else -- a
            //      candidate the first via does not remove (another one still might), two candidates at one offset, entries beyond the LDS
            //      slots -- and the pass is decided target by target below instead; nothing has been pushed by then. ----
            bool serial = __ballot(act && vmask != 0ull && ebase + 64 - (int) __builtin_clzll(vmask | 1ull) > SJ_ES) != 0ull;
            if (!serial) {
                const int total_e = (int) bperm((uint32_t) (ebase + (int) ecnt), 63 - (int) __builtin_clzll((long long) (starts | 1ull)));   // slots used: base + count of the last group
                wave_lds_fence();
                for (int k = lane; k < total_e && k < SJ_ES; k += 64) sTab[wave][k] = 0xFFFFFFFFu;
                wave_lds_fence();
                for (uint64_t m = vmask; m != 0ull; m &= m - 1ull) {           // per lane: its own overlaps
                    const int es = ebase + (int) __builtin_ctzll(m);
                    atomicMin(&sTab[wave][es], ((uint32_t) (q - (int) (sEM[wave][es] & 255u)) << 6) | (uint32_t) lane);
                }
                wave_lds_fence();
                bool undecided = false;
                uint64_t mine = 0ull;                                           // the targets whose first candidate this lane is
                for (uint64_t m = vmask; m != 0ull; m &= m - 1ull) {
                    const int c = (int) __builtin_ctzll(m), es = ebase + c;
                    const uint32_t meta = sEM[wave][es], t = sTab[wave][es];
                    const int p = q - (int) (meta & 255u), Bl = (int) (t & 63u), pB = (int) (t >> 6);
                    if (Bl == lane) { mine |= 1ull << c; continue; }
                    const int rho_c = (int) ((meta >> 8) & 0xFFFu) - (lenA - p);
                    const uint32_t *rB = sA[wave][Bl];
                    const bool removed = via(sSrc[wave][Bl], (int) sLen[wave][Bl], pB, rB[0], rB[1], rB[2], rB[3], p, rho_c);
                    // Not removed by the first via.  Reads of ONE length: when even the first via's overlap with the target is small, every later
                    // candidate's is smaller still -- there is no via at all and every candidate of the target stands (a coverage gap: the
                    // pending small survivors).  Otherwise another via may still remove it: undecided.
                    const bool none_big = ulen > 0 && ulen - pB < Lbig;
                    if (!removed && none_big && p != pB) mine |= 1ull << c;
                    undecided = undecided || (!removed && !none_big) || p == pB;
                }
                serial = __ballot(undecided) != 0ull;
                st_pass++; st_serial += serial ? 1u : 0u;
                if (!serial)
                    for (uint64_t m = mine; m != 0ull; m &= m - 1ull) {
                        const int es = ebase + (int) __builtin_ctzll(m);
                        push(sEI[wave][es], q - (int) (sEM[wave][es] & 255u));
                    }
            }
            // ---- pass 2, target by target: the vias of each group by ascending offset, until nobody is left whom a later one could remove ----
            for (uint32_t c = 0; serial && c < max_e; c++) {         // uniform
                const bool V = ((vmask >> c) & 1ull) != 0ull;
                if (__ballot(V) == 0ull) continue;
                uint32_t idC, meta;
                const int es = ebase + (int) c;
                if (__ballot(V && es >= SJ_ES) == 0ull) { const int e2 = (es >= 0 && es < SJ_ES) ? es : 0; idC = sEI[wave][e2]; meta = sEM[wave][e2]; }
                else { const uint4 tail = store[(size_t) (c < ecnt ? e0 + c : e0) * EQ + (EQ - 1)]; idC = tail.y; meta = tail.w; }
                const int lenC = (int) ((meta >> 8) & 0xFFFu);
                const int p = q - (int) (meta & 255u);
                const int rho_c = lenC - (lenA - p);
                bool removed = false, used = false;
                for (;;) {                                 // uniform
                    wave_lds_fence();
                    if (lane == gfirst) sMin[wave][lane] = 0xFFFFFFFFu;
                    wave_lds_fence();
                    if (V && !used) atomicMin(&sMin[wave][gfirst], ((uint32_t) p << 6) | (uint32_t) lane);
                    wave_lds_fence();
                    const uint32_t m = sMin[wave][gfirst];
                    const bool hasB = act && m != 0xFFFFFFFFu;
                    const int Bl = (int) (m & 63u), pB = (int) (m >> 6);
                    used = used || (hasB && lane == Bl);
                    const bool test = hasB && V && !removed && lane != Bl && p >= pB;
                    if (__ballot(test) == 0ull) break;     // in no group is anybody left whom this or a later via (they all sit further right) could remove
                    const uint32_t srcB = sSrc[wave][Bl];
                    const int lenB = (int) sLen[wave][Bl];
                    const uint32_t *rB = sA[wave][Bl];
                    removed = removed || (test && via(srcB, lenB, pB, rB[0], rB[1], rB[2], rB[3], p, rho_c));
                }
                if (V && !removed) push(idC, p);
            }
            maybe_flush();
            for (uint64_t wg = wide & starts; wg != 0ull; wg &= wg - 1ull) {      // uniform: the wide buckets of this pass, one by one
                const int fl = __builtin_ctzll(wg);
                const uint64_t after = fl >= 63 ? 0ull : starts & ~((2ull << fl) - 1ull);
                const int nl = after ? __builtin_ctzll(after) : cnt;
                big_group((uint64_t) s0 + (uint64_t) fl, nl - fl);
            }
        }
    }
    flush_records<REC_CHUNK, SJ_WB>(o, w, chunk_base, chunk_fill);
    close_chunk<REC_CHUNK>(o, chunk_base, chunk_fill);
    st_rec = wave_sum_u64(st_rec);
    if (lane == 0 && st_rec) atomicAdd(&o.counters[CNT_VALID_RECORDS], (unsigned long long) st_rec);
    if (lane == 0 && st_pass) { atomicAdd(&declined[1], (unsigned long long) st_pass); atomicAdd(&declined[2], (unsigned long long) st_serial); }
}

// ------------------------------------------------------------------------------------------
// the per-source cap on the pending (small) survivors
// ------------------------------------------------------------------------------------------
// Appends to a list from a tiled kernel with ONE global atomic per block (a returning atomic on one address sustains ~88 ops / us
// chip-wide): the threads take their slots from an LDS counter, thread 0 reserves the block's range.  Call convergently.
struct BlockAppend {
    uint32_t *s_cnt; unsigned long long *s_base;
    __device__ void begin() { if (threadIdx.x == 0) *s_cnt = 0u; __syncthreads(); }
    __device__ uint32_t take() { return atomicAdd(s_cnt, 1u); }
    __device__ void reserve(unsigned long long *count) { __syncthreads(); if (threadIdx.x == 0) *s_base = *s_cnt ? atomicAdd(count, (unsigned long long) *s_cnt) : 0ull; __syncthreads(); }
    __device__ unsigned long long at(uint32_t slot) const { return *s_base + slot; }
};

// sources of the pending records -> list (few: a small overlap survives only where no big one reaches its target)
constexpr int PEND_IT = 8;
__global__ void __launch_bounds__(256) k_shard_pending_src(const uint32_t *__restrict__ rec_dst, const unsigned long long *__restrict__ rec_val, uint64_t n_rec,
                                                            uint32_t *__restrict__ list, uint32_t cap, unsigned long long *__restrict__ count) {
    __shared__ uint32_t s_cnt; __shared__ unsigned long long s_base;
    BlockAppend ap{&s_cnt, &s_base};
    ap.begin();
    const uint64_t base = (uint64_t) blockIdx.x * (256 * PEND_IT);
    uint32_t slot[PEND_IT], src[PEND_IT];
#pragma unroll
    for (int k = 0; k < PEND_IT; k++) {
        const uint64_t i = base + (uint64_t) k * 256 + threadIdx.x;
        slot[k] = 0xFFFFFFFFu;
        if (i < n_rec && rec_dst[i] != REC_INVALID) {
            const unsigned long long v = rec_val[i];
            if (ol_small((uint32_t) (v >> 32))) { src[k] = (uint32_t) v; slot[k] = ap.take(); }
        }
    }
    ap.reserve(count);
#pragma unroll
    for (int k = 0; k < PEND_IT; k++)
        if (slot[k] != 0xFFFFFFFFu && ap.at(slot[k]) < cap) list[ap.at(slot[k])] = src[k];
}

__global__ void __launch_bounds__(256) k_shard_bitmap_set(const uint32_t *__restrict__ ids, uint64_t n, uint32_t n_nodes, uint32_t *__restrict__ bitmap) {
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
        const uint32_t id = ids[i];
        if (id < n_nodes) atomicOr(&bitmap[id >> 5], 1u << (id & 31u));
    }
}

// the small keys my descriptors saw for the listed sources: {src, L, C} triples
__global__ void __launch_bounds__(256) k_shard_small_emit(const unsigned long long *__restrict__ dval, const unsigned long long *__restrict__ small_top, uint64_t n_desc,
                                                           const uint32_t *__restrict__ bitmap, uint32_t *__restrict__ out, uint32_t cap, unsigned long long *__restrict__ count) {
    __shared__ uint32_t s_cnt; __shared__ unsigned long long s_base;
    BlockAppend ap{&s_cnt, &s_base};
    ap.begin();
    const uint64_t base = (uint64_t) blockIdx.x * (256 * PEND_IT);
    uint32_t slot[PEND_IT], nk[PEND_IT];
#pragma unroll
    for (int k = 0; k < PEND_IT; k++) {
        const uint64_t i = base + (uint64_t) k * 256 + threadIdx.x;
        nk[k] = 0u; slot[k] = 0u;
        if (i < n_desc) {
            const uint32_t src = (uint32_t) dval[i];
            if ((bitmap[src >> 5] >> (src & 31u)) & 1u) {
                uint32_t c = 0;
                while (c < 3u && small_top[3 * i + c] != 0ull) c++;
                nk[k] = c;
                if (c) slot[k] = atomicAdd(&s_cnt, c);
            }
        }
    }
    ap.reserve(count);
#pragma unroll
    for (int k = 0; k < PEND_IT; k++) {
        const uint64_t i = base + (uint64_t) k * 256 + threadIdx.x;
        for (uint32_t c = 0; c < nk[k]; c++) {
            const unsigned long long at = ap.at(slot[k] + c);
            if (at < cap) { const unsigned long long key = small_top[3 * i + c]; out[3 * at] = (uint32_t) dval[i]; out[3 * at + 1] = (uint32_t) (key >> 32); out[3 * at + 2] = (uint32_t) key; }
        }
    }
}

__global__ void __launch_bounds__(256) k_shard_small_split(const uint32_t *__restrict__ in, uint64_t n, uint32_t *__restrict__ ssrc, unsigned long long *__restrict__ skey) {
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
        ssrc[i] = in[3 * i];
        skey[i] = ((unsigned long long) in[3 * i + 1] << 32) | in[3 * i + 2];
    }
}

// a pending record (A -> C, L) stands iff fewer than three small keys of A are larger than (L, C): ssrc sorted, all keys of a source adjacent
__global__ void __launch_bounds__(256) k_shard_resolve(uint32_t *__restrict__ rec_dst, const unsigned long long *__restrict__ rec_val, uint64_t n_rec,
                                                        const uint32_t *__restrict__ ssrc, const unsigned long long *__restrict__ skey, uint64_t n_small,
                                                        unsigned long long *__restrict__ dropped) {
    uint64_t n_drop = 0;
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n_rec; i += (uint64_t) gridDim.x * blockDim.x) {
        const uint32_t C = rec_dst[i];
        if (C == REC_INVALID) continue;
        const unsigned long long v = rec_val[i];
        const uint32_t ol = (uint32_t) (v >> 32);
        if (!ol_small(ol)) continue;
        const uint32_t A = (uint32_t) v;
        const unsigned long long mine = ((unsigned long long) (uint32_t) ol_len(ol) << 32) | C;
        uint64_t lo = 0, hi = n_small;                      // first index with ssrc >= A
        while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (ssrc[mid] < A) lo = mid + 1; else hi = mid; }
        int larger = 0;
        for (uint64_t k = lo; k < n_small && ssrc[k] == A; k++) larger += skey[k] > mine ? 1 : 0;
        if (larger >= 3) { rec_dst[i] = REC_INVALID; n_drop++; }
    }
    // one atomic per wave that dropped anything (pending edges are few)
    const uint64_t tot = wave_sum_u64(n_drop);
    if (lane_id() == 0 && tot) atomicAdd(dropped, (unsigned long long) tot);
}

// ------------------------------------------------------------------------------------------
// final edges -> the rank that owns the source id, then adjacency lists there
// ------------------------------------------------------------------------------------------
// records -> alga_edge triples partitioned by owner = src / chunk (COUNT: counts[o]; else out[seg_off[o] + position])
template <bool COUNT>
__global__ void __launch_bounds__(256) k_shard_edges_out(const uint32_t *__restrict__ rec_dst, const unsigned long long *__restrict__ rec_val, uint64_t n_rec, uint32_t chunk,
                                                          uint32_t n_ranks, unsigned long long *__restrict__ counts, const unsigned long long *__restrict__ seg_off,
                                                          unsigned long long *__restrict__ cursor, alga_edge_dev *__restrict__ out) {
    __shared__ uint32_t s_cnt[64];
    __shared__ unsigned long long s_base[64];
    const int t = (int) threadIdx.x;
    constexpr int IT = 8;
    const uint64_t base = (uint64_t) blockIdx.x * (256 * IT);
    if (t < 64) s_cnt[t] = 0u;
    __syncthreads();
    uint32_t slot[IT], own[IT];
#pragma unroll
    for (int k = 0; k < IT; k++) {
        const uint64_t i = base + (uint64_t) k * 256 + t;
        own[k] = 0xFFFFFFFFu;
        if (i < n_rec && rec_dst[i] != REC_INVALID) {
            uint32_t o = (uint32_t) rec_val[i] / chunk;
            o = o < n_ranks ? o : n_ranks - 1u;
            own[k] = o;
            slot[k] = atomicAdd(&s_cnt[o], 1u);
        }
    }
    __syncthreads();
    if (t < (int) n_ranks && s_cnt[t]) {
        if (COUNT) atomicAdd(&counts[t], (unsigned long long) s_cnt[t]);
        else s_base[t] = seg_off[t] + atomicAdd(&cursor[t], (unsigned long long) s_cnt[t]);
    }
    if (COUNT) return;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < IT; k++) {
        if (own[k] != 0xFFFFFFFFu) {
            const uint64_t i = base + (uint64_t) k * 256 + t;
            const unsigned long long v = rec_val[i];
            alga_edge_dev e;
            e.src = (int32_t) (uint32_t) v; e.dst = (int32_t) rec_dst[i]; e.offset = ol_off((uint32_t) (v >> 32));
            out[s_base[own[k]] + slot[k]] = e;
        }
    }
}

__global__ void __launch_bounds__(256) k_shard_place_count(const alga_edge_dev *__restrict__ in, uint64_t n, int32_t src_base, int32_t n_src, uint32_t *__restrict__ deg,
                                                            unsigned long long *__restrict__ bad) {
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
        const int32_t a = in[i].src - src_base;
        if (a < 0 || a >= n_src) { atomicOr(bad, 1ull); continue; }
        atomicAdd(&deg[a], 1u);
    }
}

__global__ void __launch_bounds__(256) k_shard_place_fill(const alga_edge_dev *__restrict__ in, uint64_t n, int32_t src_base, int32_t n_src, const uint32_t *__restrict__ rowptr,
                                                           uint32_t *__restrict__ cursor, alga_edge_dev *__restrict__ out) {
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
        const alga_edge_dev e = in[i];
        const int32_t a = e.src - src_base;
        if (a < 0 || a >= n_src) continue;
        out[rowptr[a] + atomicAdd(&cursor[a], 1u)] = e;
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
static unsigned grid_stride_blocks(uint64_t n, int per_block, unsigned cap = 16384u) {
    return (unsigned) std::max<uint64_t>(1, std::min<uint64_t>((n + (uint64_t) per_block - 1) / (uint64_t) per_block, cap));
}

void launch_shard_select(const uint32_t *keys, uint32_t n, int shift, uint32_t b_lo, uint32_t b_hi, uint32_t *okeys, uint32_t *ovals, unsigned long long *cursor, hipStream_t s) {
    if (n == 0) return;
    const unsigned g = (unsigned) (((uint64_t) n + SEL_BLOCK * SEL_IT - 1) / (SEL_BLOCK * SEL_IT));
    hipLaunchKernelGGL(k_shard_select, dim3(g), dim3(SEL_BLOCK), 0, s, keys, n, shift, b_lo, b_hi, okeys, ovals, cursor);
}

void launch_shard_export(bool count, const void *runs, int32_t node_begin, int32_t node_end, int shift, uint32_t bpr, uint32_t n_ranks, unsigned long long *counts,
                         int32_t *flagged_list, unsigned long long *flagged_count, uint32_t flagged_cap, const unsigned long long *seg_off, unsigned long long *cursor,
                         uint32_t *out, hipStream_t s) {
    if (node_end <= node_begin) return;
    const unsigned g = (unsigned) (((uint64_t) (node_end - node_begin) + EXP_BLOCK - 1) / EXP_BLOCK);
    if (count) hipLaunchKernelGGL(k_shard_export<true>, dim3(g), dim3(EXP_BLOCK), 0, s, (const uint2 *) runs, node_begin, node_end, shift, bpr, n_ranks, counts, flagged_list,
                                  flagged_count, flagged_cap, seg_off, cursor, out);
    else hipLaunchKernelGGL(k_shard_export<false>, dim3(g), dim3(EXP_BLOCK), 0, s, (const uint2 *) runs, node_begin, node_end, shift, bpr, n_ranks, counts, flagged_list,
                            flagged_count, flagged_cap, seg_off, cursor, out);
}

void launch_shard_export_flagged(const NodesDev &nd, const PrefSufCfg &cfg, const ClusterCfg &cc, const int32_t *flagged_list, uint32_t n_flagged, uint32_t bpr, uint32_t n_ranks,
                                 const unsigned long long *seg_off, unsigned long long *cursor, uint32_t *out, hipStream_t s) {
    if (n_flagged == 0) return;
    hipLaunchKernelGGL(k_shard_export_flagged, dim3((n_flagged + 3) / 4), dim3(256), 0, s, nd, cfg, cc, flagged_list, n_flagged, bpr, n_ranks, seg_off, cursor, out);
}

void launch_shard_desc_split(const uint32_t *in, uint64_t n, uint32_t key_base, uint32_t *dkey, unsigned long long *dval, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_shard_desc_split, dim3(grid_stride_blocks(n, 256)), dim3(256), 0, s, in, n, key_base, dkey, dval);
}

uint64_t shard_join_record_slack(int n_cu) { return (uint64_t) std::max(1, n_cu) * 8 * SJ_WAVES * (uint64_t) REC_CHUNK; }

// flag / pos / gstart: n_desc + 2 uint32 each; scan_scratch: scan_scratch_bytes(n_desc + 1).  *n_groups_dev = pos[n_desc] afterwards.
void launch_shard_groups(const uint32_t *dkey, uint64_t n_desc, int shift, uint32_t *flag, uint32_t *pos, uint32_t *gstart, uint64_t *scan_scratch, hipStream_t s) {
    if (n_desc == 0) return;
    hipLaunchKernelGGL(k_shard_group_flags, dim3(grid_stride_blocks(n_desc, 256)), dim3(256), 0, s, dkey, n_desc, shift, flag);
    launch_exclusive_scan(flag, n_desc, pos, scan_scratch, s);
    hipLaunchKernelGGL(k_shard_group_starts, dim3(grid_stride_blocks(n_desc + 1, 256)), dim3(256), 0, s, (const uint32_t *) flag, (const uint32_t *) pos, n_desc, gstart);
}

void launch_shard_join(const NodesDev &nd, const PrefSufCfg &cfg, const ClusterCfg &cc, int eq, int uniform_len, const void *store, const void *dir, uint32_t bucket_base,
                       const uint32_t *dkey, const unsigned long long *dval, uint64_t n_desc, const uint32_t *gstart, uint32_t n_groups, uint32_t *rec_dst,
                       unsigned long long *rec_val, uint64_t rec_cap, unsigned long long *counters, unsigned long long *small_top, unsigned long long *declined, int dmax, int n_cu,
                       hipStream_t s) {
    if (n_desc == 0 || n_groups == 0) return;
    const uint64_t chunks = ((uint64_t) n_groups + SJ_GCH - 1) / SJ_GCH;
    dim3 grid((unsigned) std::max<uint64_t>(1, std::min<uint64_t>((chunks + SJ_WAVES - 1) / SJ_WAVES, (uint64_t) std::max(1, n_cu) * 8))), block(SJ_WAVES * 64);
    ProbeOut o{rec_dst, rec_val, rec_cap, counters};
    const int kf = (2 * cfg.Lmin) >> 5;
#define SJ_LAUNCH(E, K) hipLaunchKernelGGL((k_shard_join<E, K>), grid, block, 0, s, nd, cfg, cc, uniform_len, (const uint4 *) store, (const uint4 *) dir, bucket_base, dkey, dval, n_desc, gstart, n_groups, o, small_top, declined, std::max(1, std::min(dmax, SJ_DMAX)))
    if (eq == 3 && kf == 5)      SJ_LAUNCH(3, 5);
    else if (eq == 3 && kf == 3) SJ_LAUNCH(3, 3);
    else if (eq == 2 && kf == 3) SJ_LAUNCH(2, 3);
    else if (eq == 2)            SJ_LAUNCH(2, 0);
    else if (eq == 3)            SJ_LAUNCH(3, 0);
    else                         SJ_LAUNCH(4, 0);
#undef SJ_LAUNCH
}

void launch_shard_pending_src(const uint32_t *rec_dst, const unsigned long long *rec_val, uint64_t n_rec, uint32_t *list, uint32_t cap, unsigned long long *count, hipStream_t s) {
    if (n_rec == 0) return;
    hipLaunchKernelGGL(k_shard_pending_src, dim3((unsigned) ((n_rec + 256 * PEND_IT - 1) / (256 * PEND_IT))), dim3(256), 0, s, rec_dst, rec_val, n_rec, list, cap, count);
}

void launch_shard_bitmap_set(const uint32_t *ids, uint64_t n, uint32_t n_nodes, uint32_t *bitmap, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_shard_bitmap_set, dim3(grid_stride_blocks(n, 256)), dim3(256), 0, s, ids, n, n_nodes, bitmap);
}

void launch_shard_small_emit(const unsigned long long *dval, const unsigned long long *small_top, uint64_t n_desc, const uint32_t *bitmap, uint32_t *out, uint32_t cap,
                             unsigned long long *count, hipStream_t s) {
    if (n_desc == 0) return;
    hipLaunchKernelGGL(k_shard_small_emit, dim3((unsigned) ((n_desc + 256 * PEND_IT - 1) / (256 * PEND_IT))), dim3(256), 0, s, dval, small_top, n_desc, bitmap, out, cap, count);
}

void launch_shard_small_split(const uint32_t *in, uint64_t n, uint32_t *ssrc, unsigned long long *skey, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_shard_small_split, dim3(grid_stride_blocks(n, 256)), dim3(256), 0, s, in, n, ssrc, skey);
}

void launch_shard_resolve(uint32_t *rec_dst, const unsigned long long *rec_val, uint64_t n_rec, const uint32_t *ssrc, const unsigned long long *skey, uint64_t n_small,
                          unsigned long long *dropped, hipStream_t s) {
    if (n_rec == 0) return;
    hipLaunchKernelGGL(k_shard_resolve, dim3(grid_stride_blocks(n_rec, 256)), dim3(256), 0, s, rec_dst, rec_val, n_rec, ssrc, skey, n_small, dropped);
}

void launch_shard_edges_out(bool count, const uint32_t *rec_dst, const unsigned long long *rec_val, uint64_t n_rec, uint32_t chunk, uint32_t n_ranks,
                            unsigned long long *counts, const unsigned long long *seg_off, unsigned long long *cursor, alga_edge_dev *out, hipStream_t s) {
    if (n_rec == 0) return;
    const unsigned g = (unsigned) ((n_rec + 2047) / 2048);
    if (count) hipLaunchKernelGGL(k_shard_edges_out<true>, dim3(g), dim3(256), 0, s, rec_dst, rec_val, n_rec, chunk, n_ranks, counts, seg_off, cursor, out);
    else hipLaunchKernelGGL(k_shard_edges_out<false>, dim3(g), dim3(256), 0, s, rec_dst, rec_val, n_rec, chunk, n_ranks, counts, seg_off, cursor, out);
}

void launch_shard_place_count(const alga_edge_dev *in, uint64_t n, int32_t src_base, int32_t n_src, uint32_t *deg, unsigned long long *bad, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_shard_place_count, dim3(grid_stride_blocks(n, 256)), dim3(256), 0, s, in, n, src_base, n_src, deg, bad);
}

void launch_shard_place_fill(const alga_edge_dev *in, uint64_t n, int32_t src_base, int32_t n_src, const uint32_t *rowptr, uint32_t *cursor, alga_edge_dev *out, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_shard_place_fill, dim3(grid_stride_blocks(n, 256)), dim3(256), 0, s, in, n, src_base, n_src, rowptr, cursor, out);
}

} // namespace alga
