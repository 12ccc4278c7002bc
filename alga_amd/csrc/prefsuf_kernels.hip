// alga_amd/csrc/prefsuf_kernels.hip -- gfx950 kernels of the PrefSuf overlap engine.
//
// Replaces the per-overlap-length hash join + edge policy of
// src/GraphCreators/GraphCreatorPrefSuf.cpp:238-488 (reference paths relative to its root).
// Integer / byte work only: no MFMA.  Wavefront = 64 lanes throughout.
//
//   k_node_stats        max read length, live-node count
//   k_seed_build        fingerprint of the first min(min_overlap, 64) nucleotides of every target ->
//                       bucketised seed table (64-byte buckets of 8 entries: one cache line answers one
//                       probe) + L2-resident prefilter bitmap
//   k_probe_sources     persistent wavefronts, one source at a time: the source's tail is staged in
//                       LDS, lane p fingerprints suffix window p; the windows that pass the prefilter are
//                       compacted, four lanes read one bucket; tag hits become candidates in LDS, four
//                       lanes verify one candidate with a wide row load and an exact 2-bit compare.
//                       LOCAL = true : verified overlaps become items, the transitive reduction runs in
//                                      the wave (prefsuf_device.h local_reduce), final edges leave
//                       LOCAL = false: per-source small-overlap top-3 (wave max-reduction), all capped
//                                      overlaps leave as records for the per-target pipeline below
//   k_local_emit_*      source-side form: adjacency lists from out-degrees, one-edge slots, record list
//   k_make_keys         record -> sort key (target id local to the owned range, invalid last)
//   (radix sort by key: sort_records.hip)
//   k_rowptr_from_sorted   row pointers of the per-target segments
//   k_reduce_targets    per target: replay of the reference's insertion order (small-overlap dedupe,
//                       transitive reduction with 2-bit compares); fast path entirely in LDS
//   k_scan_*            exclusive scan (out-degree -> row pointers)
//   k_scatter_by_source / k_sort_rows   final adjacency lists sorted by (dst, offset)
#include <hip/hip_runtime.h>
#include <algorithm>
#include "prefsuf_common.h"
#include "prefsuf_kernels.h"
#include "prefsuf_device.h"

namespace alga {

// ------------------------------------------------------------------------------------------
// k_node_stats : few workgroups, grid-stride (one contended atomic per wave would dominate)
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_node_stats(NodesDev nd, unsigned long long *counters, int *max_len) {
    __shared__ int s_max[4], s_min[4];
    __shared__ unsigned long long s_live[4];
    int m = 0, mn = 0x7FFFFFFF;
    unsigned long long live = 0, asym = 0;
    auto one = [&](int l, int64_t i) {
        m = l > m ? l : m;
        mn = (l > 0 && l < mn) ? l : mn;
        live += l > 0;
        if (nd.to && l > 0 && !nd.to[i] && (nd.from == nullptr || nd.from[i])) asym++;
    };
    const int64_t gid = (int64_t) blockIdx.x * blockDim.x + threadIdx.x, gsz = (int64_t) gridDim.x * blockDim.x;
    int64_t done = 0;                                      // lengths taken four at a time (16-byte loads) where the array allows it
    if ((reinterpret_cast<uintptr_t>(nd.len) & 15u) == 0) {
        const int64_t nq = (int64_t) nd.n >> 2;
        const int4 *len4 = reinterpret_cast<const int4 *>(nd.len);
        for (int64_t q = gid; q < nq; q += gsz) { const int4 l = len4[q]; one(l.x, 4 * q); one(l.y, 4 * q + 1); one(l.z, 4 * q + 2); one(l.w, 4 * q + 3); }
        done = 4 * nq;
    }
    for (int64_t i = done + gid; i < nd.n; i += gsz) one(nd.len[i], i);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { int t = __shfl_xor(m, o); m = t > m ? t : m; t = __shfl_xor(mn, o); mn = t < mn ? t : mn; }
    live = wave_sum_u64(live);
    asym = wave_sum_u64(asym);
    const int wave = (int) (threadIdx.x >> 6);
    if (lane_id() == 0) { s_max[wave] = m; s_min[wave] = mn; s_live[wave] = live; if (asym) atomicAdd(&counters[CNT_MASK_ASYM], asym); }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) { m = s_max[w] > m ? s_max[w] : m; mn = s_min[w] < mn ? s_min[w] : mn; live += s_live[w]; }
        if (m > 0) { atomicMax(max_len, m); atomicMax(max_len + 1, 0x7FFFFFFF - mn); }
        if (live) atomicAdd(&counters[CNT_LIVE_NODES], live);
    }
}

// ------------------------------------------------------------------------------------------
// seed table: buckets of 8 x u64, entry = (tag23 | len9) << 32 | node id, empty = all ones
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t seed_bucket(uint64_t h, uint32_t n_buckets) { return __umulhi((uint32_t) h, n_buckets); }
__device__ __forceinline__ uint32_t seed_taglen(uint64_t h, int len) {
    return ((uint32_t) (h >> 41) << 9) | (uint32_t) (len > 511 ? 511 : len);
}

// one thread per target node
//   replaces updatePrefixHash + putKmersIntoBucketsJob (GraphCreatorPrefSuf.cpp:213-223,323-332)
//   for the single length min_overlap; longer overlaps are found by extension, not by re-hashing.
// Prefilter: one bit per fingerprint in a bitmap small enough to live in every XCD's L2 (<= 4 MB).  ~72 % of the
// suffix windows of BASELINE configs[1] match no target; their bucket read (a random 64-byte line from the
// Infinity Cache / HBM, the dominant cost of the probe) is skipped when the bit is clear.
__device__ __forceinline__ uint32_t seed_filter_index(uint64_t h, uint32_t filter_mask) { return (uint32_t) (h >> 32) & filter_mask; }

__global__ void __launch_bounds__(256) k_seed_build(NodesDev nd, PrefSufCfg cfg, unsigned long long *table, uint32_t n_buckets,
                                                     uint32_t *filter, uint32_t filter_mask) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nd.n) return;
    int len = nd.len[i];
    if (len <= 0 || len < cfg.Lmin) return;
    if (nd.to && !nd.to[i]) return;
    const uint32_t *row = nd.words + (size_t) i * nd.stride;
    uint64_t h = fp_init();
    for (int k = 0; k < cfg.seed_words; k++) {
        uint32_t w = row[k];
        if (k == cfg.seed_words - 1) w &= cfg.seed_last_mask;
        h = fp_step(h, w);
    }
    h = fp_final(h);
    uint32_t b = seed_bucket(h, n_buckets);
    if (filter) { const uint32_t fi = seed_filter_index(h, filter_mask); atomicOr(&filter[fi >> 5], 1u << (fi & 31)); }
    const unsigned long long entry = ((unsigned long long) seed_taglen(h, len) << 32) | (uint32_t) i;
    for (;;) {
        unsigned long long *bp = table + (size_t) b * SEED_BUCKET;
        for (int j = 0; j < SEED_BUCKET; j++) {
            if (bp[j] != SEED_EMPTY) continue;                 // slots only ever go empty -> full
            if (atomicCAS(&bp[j], (unsigned long long) SEED_EMPTY, entry) == SEED_EMPTY) return;
        }
        b = (b + 1 == n_buckets) ? 0u : b + 1;                 // bucket full: spill to the next one
    }
}

#ifndef PROBE_OCC
#define PROBE_OCC 4                   // minimum waves per SIMD requested from the register allocator
#endif
// LOCAL != 0: the wave also performs the transitive reduction for its source (prefsuf_device.h: local_reduce; LOCAL = width of
// its offset masks / overhangs: 1 for max_len - Lmin <= 63, 2 for <= 127) and what it emits are the final edges; LOCAL == 0: the
// records are all capped raw overlaps, reduced per target later.
// BIG (LOCAL only): second pass over the sources the first pass listed because they have more raw overlaps than a wave's LDS holds
// (repeats): same probe, the items live in a global slice per wave, every source takes the all-pairs evaluation.
template <bool STATS, int NQ, int LOCAL, bool BIG = false>
__global__ void __launch_bounds__(PROBE_WAVES * 64, PROBE_OCC)
k_probe_sources(NodesDev nd, PrefSufCfg cfg, const unsigned long long *__restrict__ table, uint32_t n_buckets,
                const uint32_t *__restrict__ filter, uint32_t filter_mask, int32_t src_begin, int32_t src_end, ProbeOut o) {
    __shared__ uint32_t sB[PROBE_WAVES][STAGE_WORDS];
    __shared__ uint32_t sCandC[PROBE_WAVES][CANDMAX];
    __shared__ uint32_t sCandW[PROBE_WAVES][CANDMAX];
    constexpr int WB = LOCAL ? WBUF_LOCAL : WBUF;
    __shared__ uint32_t sRecC[PROBE_WAVES][WB];
    __shared__ unsigned long long sRecV[PROBE_WAVES][WB];
    __shared__ uint32_t sCnt[PROBE_WAVES][3];
    __shared__ uint32_t sItemC[PROBE_WAVES][LOCAL && !BIG ? ITEMMAX : 1];
    __shared__ uint32_t sItemM[PROBE_WAVES][LOCAL && !BIG ? ITEMMAX : 1];
    __shared__ uint4 sItemO[PROBE_WAVES][LOCAL && !BIG ? ITEMMAX * LOCAL : 1];
    __shared__ uint8_t sItemT[PROBE_WAVES][LOCAL > 1 ? 64 * LOCAL : 64];
    __shared__ uint32_t sActB[PROBE_WAVES][64];
    __shared__ uint32_t sActT[PROBE_WAVES][64];
    constexpr int CHUNK = LOCAL ? REC_CHUNK_LOCAL : REC_CHUNK;
    const int wave = (int) (threadIdx.x >> 6);
    const int lane = lane_id();
    WaveLds w{sB[wave], sCandC[wave], sCandW[wave], &sCnt[wave][0], sRecC[wave], sRecV[wave], &sCnt[wave][1], sActB[wave], sActT[wave]};
    const size_t wave_gid = (size_t) blockIdx.x * PROBE_WAVES + (size_t) wave;
    ItemLds it{BIG ? o.bigC + wave_gid * o.big_cap : sItemC[wave], BIG ? o.bigM + wave_gid * o.big_cap : sItemM[wave],
               BIG ? o.bigO + wave_gid * o.big_cap * (LOCAL ? LOCAL : 1) : sItemO[wave], sItemT[wave], &sCnt[wave][2]};
    const int item_cap = BIG ? (int) o.big_cap : ITEMMAX;
    if (lane == 0) { *w.candN = 0; *w.recN = 0; *it.N = 0; }
    uint64_t chunk_base = 0;
    int chunk_fill = CHUNK;                                // "no chunk yet"
    uint64_t st_raw = 0, st_slots = 0, st_win = 0, st_rec = 0, st_cmp = 0, st_generic = 0;
    int f_left = filter != nullptr ? 16 : -1;              // uniform: batches still sampled for the pass rate; < 0 = filter not used
    int f_bal = 0;                                         // 2 * passed - tested over the sampled batches
    const int64_t total_waves = (int64_t) gridDim.x * PROBE_WAVES;
    const uint32_t *sb = w.sb;

    // software pipeline over sources: length, mask and row word of the NEXT source are requested before the
    // current one is probed, so their latency hides behind the bucket and candidate-row round trips
    const int pre_words = nd.stride < STAGE_WORDS ? nd.stride : STAGE_WORDS;
    // BIG: the loop runs over the positions of the source list instead of over source ids
    const int64_t it_end = BIG ? (int64_t) o.big_count : (int64_t) src_end;
    int64_t Bl = (BIG ? 0 : (int64_t) src_begin) + (int64_t) blockIdx.x * PROBE_WAVES + wave;
    int n_id = 0, n_len = 0; uint32_t n_word = 0; uint8_t n_from = 1;
    if (Bl < it_end) {
        n_id = BIG ? o.big_list[Bl] : (int) Bl;
        n_len = nd.len[n_id];
        n_word = lane < pre_words ? nd.words[(size_t) n_id * nd.stride + lane] : 0u;
        if (nd.from) n_from = nd.from[n_id];
    }
    while (Bl < it_end) {
        const int B = n_id;
        const int lenB = n_len;
        const uint32_t word0 = n_word;
        const bool from_ok = n_from != 0;
        Bl += total_waves;
        if (Bl < it_end) {
            n_id = BIG ? o.big_list[Bl] : (int) Bl;
            n_len = nd.len[n_id];
            n_word = lane < pre_words ? nd.words[(size_t) n_id * nd.stride + lane] : 0u;
            if (nd.from) n_from = nd.from[n_id];
        }
        if (!(lenB >= cfg.Lmin && lenB > 0 && from_ok)) continue;                       // wave-uniform
        // stage the last Lspan nucleotides of B (all an overlap of length <= Lcap can touch)
        const int Lspan = lenB < cfg.Lcap ? lenB : cfg.Lcap;
        const int w0 = (2 * (lenB - Lspan)) >> 5;         // first staged word of the row (0 unless the read is > 501 nt)
        const int nwB = blocks_of(lenB) - w0;              // staged words (<= 33)
        {
            wave_lds_fence();
            uint32_t x = word0;
            if (w0 != 0) x = (lane < nwB) ? nd.words[(size_t) B * nd.stride + w0 + lane] : 0u;   // long read: re-read the tail
            if (lane < STAGE_WORDS) w.sb[lane] = lane < nwB ? x : 0u;
            wave_lds_fence();
        }
        const int nwin = Lspan - cfg.Lmin + 1;             // overlap lengths Lmin..Lspan
        uint64_t k0 = 0, k1 = 0, k2 = 0;                   // per-lane top-3 small overlaps, key=(L<<32)|C

        // A verified overlap (B -> C, length L): small ones compete for the per-source top 3, big ones are records.
        auto classify = [&](int C, int L, int lenC) -> int {                  // LOCAL: the item slot (may be >= ITEMMAX: not stored)
            if (STATS) st_raw++;
            if (LOCAL) {
                uint32_t m = (uint32_t) (lenB - L) | ((uint32_t) lenC << 9);
                if (nd.from == nullptr || nd.from[C]) m |= ITEM_FROM;
                const uint32_t i = atomicAdd(it.N, 1u);                // LDS atomic
                if (i < (uint32_t) item_cap) { it.C[i] = (uint32_t) C; it.M[i] = m; }
                return (int) i;
            }
            if (L < cfg.rsoemo) {
                top3_insert(k0, k1, k2, ((uint64_t) (uint32_t) L << 32) | (uint32_t) C);       // GraphCreatorPrefSuf.cpp:397-401
            } else {
                st_rec++;
                const unsigned long long val = ((unsigned long long) ol_pack(lenB - L, L, false) << 32) | (uint32_t) B;
                const uint32_t i = atomicAdd(w.recN, 1u);              // LDS atomic
                if (i < (uint32_t) WBUF) { w.recC[i] = (uint32_t) C; w.recV[i] = val; }
                else store_record(o, atomicAdd(&o.counters[CNT_RECORDS], 1ull), (uint32_t) C, val);   // buffer full: direct, slow
            }
            return -1;
        };

        int n_items = 0;                                   // LOCAL: verified overlaps of this source so far (uniform)
        for (int base = 0; base < nwin; base += 64) {
            // ---- phase 1a: one window per lane -> fingerprint of its first seed_nt nucleotides -> (bucket, tag) ----
            const int widx = base + lane;
            const bool wvalid = widx < nwin;
            uint32_t my_b = 0, my_tag = 0xFFFFFFFFu;       // tag 0xFFFFFFFF (> 23 bits) never matches an entry
            int my_q = 0, my_r = 0;
            if (wvalid) {
                const int L = Lspan - widx;
                const int bit = 2 * (lenB - L) - 32 * w0;
                my_q = bit >> 5; my_r = bit & 31;
                uint64_t h = fp_init();
                uint32_t x[SEED_MAX_WORDS + 1];            // all reads in flight before the first use (the tail has slack words)
#pragma unroll
                for (int k = 0; k <= SEED_MAX_WORDS; k++) x[k] = sb[my_q + k];
#pragma unroll
                for (int k = 0; k < SEED_MAX_WORDS; k++) {
                    if (k < cfg.seed_words) {              // uniform
                        uint32_t v = funnel(x[k], x[k + 1], my_r);
                        if (k == cfg.seed_words - 1) v &= cfg.seed_last_mask;
                        h = fp_step(h, v);
                    }
                }
                h = fp_final(h);
                my_tag = (uint32_t) (h >> 41);
                my_b = seed_bucket(h, n_buckets);
                if (STATS) st_win++;
                if (f_left >= 0) {
                    const uint32_t fi = seed_filter_index(h, filter_mask);
                    if (!((filter[fi >> 5] >> (fi & 31)) & 1u)) my_tag = 0xFFFFFFFFu;      // no target has this fingerprint
                }
            }
            // ---- phase 1b: the windows that passed the filter are compacted; FOUR lanes read one 64-byte bucket with ONE
            // request (16 buckets per instruction).  A lane-private 4 x 16 B read of a random line costs four requests in
            // the vector memory path.  Candidates are appended with ballot + prefix count (no LDS atomics).
            const int sub = lane & 3;
            const uint64_t amask = __ballot(my_tag != 0xFFFFFFFFu);
            const int nact = __popcll(amask);
            if (f_left > 0) {                              // pass rate over the wave's first batches: above 50 % the lookups are wasted
                f_bal += 2 * nact - ((nwin - base) < 64 ? (nwin - base) : 64);
                if (--f_left == 0 && f_bal > 0) f_left = -1;
            }
            if (my_tag != 0xFFFFFFFFu) {
                const int rank = (int) __builtin_amdgcn_mbcnt_hi((uint32_t) (amask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) amask, 0u));
                w.actB[rank] = my_b;
                w.actT[rank] = my_tag | ((uint32_t) lane << 23);
            }
            wave_lds_fence();
            int ncand = 0;                                 // uniform
            auto append = [&](bool hit, uint32_t id, uint32_t tl, int wl) {      // convergent
                const uint64_t m = __ballot(hit);
                if (hit) {
                    const int ci = ncand + (int) __builtin_amdgcn_mbcnt_hi((uint32_t) (m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) m, 0u));
                    if (ci < CANDMAX) { w.candC[ci] = id; w.candW[ci] = (uint32_t) (base + wl) | ((tl & 511u) << 16); }
                }
                ncand += __popcll(m);
            };
            for (int r0 = 0; r0 < nact; r0 += 16) {
                const int qi = r0 + (lane >> 2);
                const bool qv = qi < nact;
                uint32_t bw = 0, tw = 0xFFFFFFFFu;
                int wl = 0;
                uint4 e = make_uint4(0xFFFFFFFFu, 0u, 0xFFFFFFFFu, 0u);
                if (qv) {
                    bw = w.actB[qi];
                    const uint32_t t = w.actT[qi];
                    tw = t & 0x7FFFFFu; wl = (int) (t >> 23);
                    e = reinterpret_cast<const uint4 *>(table + (size_t) bw * SEED_BUCKET)[sub];
                    if (STATS) st_slots += 2;
                }
                const int Lw = Lspan - (base + wl);
                // candidate: same tag, long enough for a prefix of length L (:215), not B itself (:386)
                append(e.x != 0xFFFFFFFFu && (e.y >> 9) == tw && (int) (e.y & 511u) >= Lw && (int) e.x != B, e.x, e.y, wl);
                append(e.z != 0xFFFFFFFFu && (e.w >> 9) == tw && (int) (e.w & 511u) >= Lw && (int) e.z != B, e.z, e.w, wl);
                // ---- phase 1c (rare): a full bucket may have spilled entries into the following ones; the wave walks them
                uint64_t fm = __ballot(qv && sub == 3 && e.z != 0xFFFFFFFFu);
                while (fm) {                               // uniform
                    const int fl = __builtin_ctzll(fm);
                    fm &= fm - 1;
                    uint32_t b = (uint32_t) __builtin_amdgcn_readlane((int) bw, fl);
                    const uint32_t stag = (uint32_t) __builtin_amdgcn_readlane((int) tw, fl);
                    const int swl = __builtin_amdgcn_readlane(wl, fl);
                    const int sL = Lspan - (base + swl);
                    for (;;) {
                        b = (b + 1 == n_buckets) ? 0u : b + 1;
                        uint4 f = make_uint4(0xFFFFFFFFu, 0u, 0xFFFFFFFFu, 0u);
                        if (lane < 4) f = reinterpret_cast<const uint4 *>(table + (size_t) b * SEED_BUCKET)[lane];
                        if (STATS && lane < 4) st_slots += 2;
                        append(f.x != 0xFFFFFFFFu && (f.y >> 9) == stag && (int) (f.y & 511u) >= sL && (int) f.x != B, f.x, f.y, swl);
                        append(f.z != 0xFFFFFFFFu && (f.w >> 9) == stag && (int) (f.w & 511u) >= sL && (int) f.z != B, f.z, f.w, swl);
                        if ((uint32_t) __builtin_amdgcn_readlane((int) f.z, 3) == 0xFFFFFFFFu) break;     // bucket not full: nothing spilled past it
                    }
                }
            }
            // ---- phase 2: candidates -> exact 2-bit compare of C[0, L) with B[off, off+L) ----------------------------
            wave_lds_fence();
            if (ncand > CANDMAX || NQ == 0) {
                // NQ == 0: rows are not 16-byte aligned / longer than the wide path takes.  ncand > CANDMAX: more tag hits
                // than the buffer holds (heavy repeats): the buffered ones are dropped and the whole 64-window batch is
                // verified the slow way, window by window, straight from the table.  Both use LDS-atomic bookkeeping.
                if (LOCAL) { if (lane == 0) *it.N = (uint32_t) n_items; wave_lds_fence(); }
                if (ncand > CANDMAX) {
                    if (wvalid && my_tag != 0xFFFFFFFFu) {
                        const int L = Lspan - widx;
                        uint32_t b = my_b;
                        for (;;) {
                            const unsigned long long *bp = table + (size_t) b * SEED_BUCKET;
                            bool full = true;
                            for (int j = 0; j < SEED_BUCKET; j++) {
                                const unsigned long long en = bp[j];
                                const uint32_t id = (uint32_t) en, tl = (uint32_t) (en >> 32);
                                if (id == 0xFFFFFFFFu) { full = false; break; }
                                if ((tl >> 9) == my_tag && (int) (tl & 511u) >= L && (int) id != B &&
                                    verify_overlap<0>(nd, sb, (int) id, my_q, my_r, L)) {
                                    const int slot = classify((int) id, L, (int) (tl & 511u));
                                    if (LOCAL) item_overhang_global<(LOCAL ? LOCAL : 1)>(nd, it, item_cap, slot, (int) id, L, (int) (tl & 511u));
                                }
                            }
                            if (!full) break;
                            b = (b + 1 == n_buckets) ? 0u : b + 1;
                        }
                    }
                } else {
                    for (int c0 = 0; c0 < ncand; c0 += 64) {
                        const int ci = c0 + lane;
                        if (ci < ncand) {
                            const int C = (int) w.candC[ci];
                            const uint32_t cw = w.candW[ci];
                            const int L = Lspan - (int) (cw & 0xFFFFu);
                            const int bit = 2 * (lenB - L) - 32 * w0;
                            if (verify_overlap<0>(nd, sb, C, bit >> 5, bit & 31, L)) {
                                const int slot = classify(C, L, (int) (cw >> 16));
                                if (LOCAL) item_overhang_global<(LOCAL ? LOCAL : 1)>(nd, it, item_cap, slot, C, L, (int) (cw >> 16));
                            }
                        }
                    }
                }
                if (LOCAL) { wave_lds_fence(); n_items = (int) __builtin_amdgcn_readfirstlane((int) *it.N); }
            } else if (NQ == 3 || NQ == 2) {
                // rows of 48 (32) used bytes: THREE (TWO) lanes per candidate, five (eight) candidates per 16-lane row, 20 (32) per
                // group: 76 % of the sources of configs[1] fit one group (41 % with quads of which one lane loads nothing)
                constexpr int GPR = NQ == 3 ? 5 : 8;              // candidates per 16-lane row
                const int pos = lane & 15;
                const int grp = NQ == 3 ? (pos * 11) >> 5 : pos >> 1;                          // pos / NQ for pos < 16
                const int sub3 = pos - NQ * grp;
                const bool lane_ok = NQ == 2 || pos < 15;         // lane 15 of a row idles with three lanes per candidate
                for (int c0 = 0; c0 < ncand; c0 += 8 * GPR) {     // two groups per trip: both row loads in flight
                    int Cc[2], Lc[2], Nc[2];
                    uint4 cc[2];
                    bool act[2];
#pragma unroll
                    for (int g = 0; g < 2; g++) {
                        const int ci = c0 + 4 * GPR * g + GPR * (lane >> 4) + grp;
                        act[g] = lane_ok && ci < ncand;
                        Cc[g] = 0; Lc[g] = Lspan; Nc[g] = 0;
                        cc[g] = make_uint4(0u, 0u, 0u, 0u);
                        if (act[g]) {
                            Cc[g] = (int) w.candC[ci];
                            const uint32_t cw = w.candW[ci];
                            Lc[g] = Lspan - (int) (cw & 0xFFFFu);
                            Nc[g] = (int) (cw >> 16);
                            cc[g] = reinterpret_cast<const uint4 *>(nd.words + (size_t) Cc[g] * nd.stride)[sub3];
                        }
                    }
#pragma unroll
                    for (int g = 0; g < 2; g++) {
                        uint32_t diff = 0;
                        if (act[g]) {
                            const int L = Lc[g];
                            const int bit = 2 * (lenB - L) - 32 * w0;
                            const int q = bit >> 5, r = bit & 31;
                            const int nwL = (2 * L + 31) >> 5;
                            const uint32_t lastmask = (2 * L & 31) ? ((1u << (2 * L & 31)) - 1u) : 0xFFFFFFFFu;
                            const uint32_t cw[4] = {cc[g].x, cc[g].y, cc[g].z, cc[g].w};
                            uint32_t y[5];
#pragma unroll
                            for (int j = 0; j < 5; j++) y[j] = sb[q + 4 * sub3 + j];
#pragma unroll
                            for (int j = 0; j < 4; j++) {
                                const int k = 4 * sub3 + j;
                                const uint32_t m = k < nwL - 1 ? 0xFFFFFFFFu : (k == nwL - 1 ? lastmask : 0u);
                                diff |= (funnel(y[j], y[j + 1], r) ^ cw[j]) & m;
                            }
                        }
                        const uint32_t d1 = (uint32_t) row_from_above<1>((int) diff);
                        const uint32_t d2 = NQ == 3 ? (uint32_t) row_from_above<2>((int) diff) : 0u;
                        const bool pass = act[g] && sub3 == 0 && (diff | d1 | d2) == 0;       // the group's first lane decides
                        if (LOCAL) {
                            const uint64_t pm = __ballot(pass);
                            int slot = -1;
                            if (pass) {
                                if (STATS) st_raw++;
                                slot = n_items + (int) __builtin_amdgcn_mbcnt_hi((uint32_t) (pm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) pm, 0u));
                                if (slot < item_cap) {
                                    uint32_t m = (uint32_t) (lenB - Lc[g]) | ((uint32_t) Nc[g] << 9);
                                    if (nd.from == nullptr || nd.from[Cc[g]]) m |= ITEM_FROM;
                                    it.C[slot] = (uint32_t) Cc[g]; it.M[slot] = m;
                                }
                            }
                            n_items += __popcll(pm);
                            const int s1 = row_from_below<1>(slot);                                 // the first lane's slot for the others
                            const int s2 = NQ == 3 ? row_from_below<2>(slot) : -1;
                            slot = sub3 == 0 ? slot : (sub3 == 1 ? s1 : s2);
                            const uint32_t nq = (uint32_t) row_from_above<1>((int) cc[g].x);
                            const uint32_t nxt = sub3 == NQ - 1 ? 0u : nq;
                            if (lane_ok && slot >= 0 && slot < item_cap) {
                                const int ws = (2 * Lc[g]) >> 5, r = (2 * Lc[g]) & 31;
                                uint32_t *ow = reinterpret_cast<uint32_t *>(&it.O[slot * (LOCAL ? LOCAL : 1)]);
                                const uint32_t cw[5] = {cc[g].x, cc[g].y, cc[g].z, cc[g].w, nxt};
#pragma unroll
                                for (int j = 0; j < 4; j++) {
                                    const int k = 4 * sub3 + j - ws;
                                    if (k >= 0 && k < 4 * LOCAL) ow[k] = funnel(cw[j], cw[j + 1], r);
                                }
                            }
                        } else if (pass) classify(Cc[g], Lc[g], Nc[g]);
                    }
                }
            } else {
                // four lanes per candidate: lane `sub` loads 16 bytes of C's row (one request per row), compares its
                // four words, the group ORs its differences
                for (int c0 = 0; c0 < ncand; c0 += 32) {          // two groups of 16 candidates per trip: both loads in flight
                    int Cc[2], Lc[2], Nc[2];
                    uint4 cc[2];
                    bool act[2];
#pragma unroll
                    for (int g = 0; g < 2; g++) {
                        const int ci = c0 + 16 * g + (lane >> 2);
                        act[g] = ci < ncand;
                        Cc[g] = 0; Lc[g] = Lspan; Nc[g] = 0;
                        cc[g] = make_uint4(0u, 0u, 0u, 0u);
                        if (act[g]) {
                            Cc[g] = (int) w.candC[ci];
                            const uint32_t cw = w.candW[ci];
                            Lc[g] = Lspan - (int) (cw & 0xFFFFu);
                            Nc[g] = (int) (cw >> 16);
                            if (sub < NQ) cc[g] = reinterpret_cast<const uint4 *>(nd.words + (size_t) Cc[g] * nd.stride)[sub];
                        }
                    }
#pragma unroll
                    for (int g = 0; g < 2; g++) {
                        uint32_t diff = 0;
                        if (act[g] && sub < NQ) {
                            const int L = Lc[g];
                            const int bit = 2 * (lenB - L) - 32 * w0;
                            const int q = bit >> 5, r = bit & 31;
                            const int nwL = (2 * L + 31) >> 5;
                            const uint32_t lastmask = (2 * L & 31) ? ((1u << (2 * L & 31)) - 1u) : 0xFFFFFFFFu;
                            const uint32_t cw[4] = {cc[g].x, cc[g].y, cc[g].z, cc[g].w};
                            uint32_t y[5];
#pragma unroll
                            for (int j = 0; j < 5; j++) y[j] = sb[q + 4 * sub + j];
#pragma unroll
                            for (int j = 0; j < 4; j++) {
                                const int k = 4 * sub + j;
                                const uint32_t m = k < nwL - 1 ? 0xFFFFFFFFu : (k == nwL - 1 ? lastmask : 0u);
                                diff |= (funnel(y[j], y[j + 1], r) ^ cw[j]) & m;
                            }
                        }
                        diff = quad_or(diff);
                        const bool pass = act[g] && sub == 0 && diff == 0;
                        if (LOCAL) {
                            // item slots by ballot + prefix count; the quad that holds C's row fills the item's overhang:
                            // word k of it = bits [2L + 32k, 2L + 32k + 32) of the row
                            const uint64_t pm = __ballot(pass);
                            int slot = -1;
                            if (pass) {
                                if (STATS) st_raw++;
                                slot = n_items + (int) __builtin_amdgcn_mbcnt_hi((uint32_t) (pm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) pm, 0u));
                                if (slot < item_cap) {
                                    uint32_t m = (uint32_t) (lenB - Lc[g]) | ((uint32_t) Nc[g] << 9);
                                    if (nd.from == nullptr || nd.from[Cc[g]]) m |= ITEM_FROM;
                                    it.C[slot] = (uint32_t) Cc[g]; it.M[slot] = m;
                                }
                            }
                            n_items += __popcll(pm);
                            slot = quad_bcast0(slot);
                            const uint32_t nq = quad_next(cc[g].x);         // every lane of the quad must execute the DPP read
                            const uint32_t nxt = sub == 3 ? 0u : nq;
                            if (slot >= 0 && slot < item_cap) {
                                const int ws = (2 * Lc[g]) >> 5, r = (2 * Lc[g]) & 31;
                                uint32_t *ow = reinterpret_cast<uint32_t *>(&it.O[slot * (LOCAL ? LOCAL : 1)]);
                                const uint32_t cw[5] = {cc[g].x, cc[g].y, cc[g].z, cc[g].w, nxt};
#pragma unroll
                                for (int j = 0; j < 4; j++) {
                                    const int k = 4 * sub + j - ws;
                                    if (k >= 0 && k < 4 * LOCAL) ow[k] = funnel(cw[j], cw[j + 1], r);
                                }
                            }
                        } else if (pass) classify(Cc[g], Lc[g], Nc[g]);
                    }
                }
            }
        }
        if (LOCAL) {
            wave_lds_fence();
            if (n_items > item_cap) {
                // first pass: the source goes on the list of the second pass; second pass (or a full list): the engine falls
                // back to the per-target pipeline
                if (lane == 0) {
                    const unsigned long long k = atomicAdd(&o.counters[CNT_LOCAL_OVERFLOW], 1ull);
                    if (!BIG && k < (unsigned long long) o.big_list_cap) o.big_list[k] = B;
                    atomicMax(&o.counters[CNT_LOCAL_MAXITEMS], (unsigned long long) n_items);
                    if (STATS) { st_raw -= (uint64_t) n_items; st_win -= (uint64_t) nwin; }     // the second pass counts this source
                }
            } else if (n_items > 0) {
                if (BIG) __threadfence();                  // the items were written to global memory by other lanes of this wave
                local_reduce<STATS, WB, (LOCAL ? LOCAL : 1)>(nd, cfg, it, w, o, B, lenB, n_items, st_rec, st_cmp, st_generic);
            }
            const int nb = (int) __builtin_amdgcn_readfirstlane((int) *w.recN);
            if (nb >= WFLUSH_LOCAL) flush_records<CHUNK, WB>(o, w, chunk_base, chunk_fill);
            continue;
        }
        // per-source small-overlap cap: the reference keeps the LAST `SOES`=3 pushes in (L asc, C asc)
        // order (GraphCreatorPrefSuf.cpp:400-401) == the 3 largest (L, C) keys.
        uint64_t win0, win1, win2;
        wave_top3(k0, k1, k2, win0, win1, win2);
        const int nwon = (win0 != 0) + (win1 != 0) + (win2 != 0);
        wave_lds_fence();
        int nbuf = (int) __builtin_amdgcn_readfirstlane((int) *w.recN);
        if (nbuf > WBUF) nbuf = WBUF;
        if (nbuf + nwon > WBUF) { flush_records<CHUNK>(o, w, chunk_base, chunk_fill); nbuf = 0; }
        if (lane < nwon) {
            const uint64_t m = lane == 0 ? win0 : (lane == 1 ? win1 : win2);
            const int L = (int) (m >> 32);
            w.recC[nbuf + lane] = (uint32_t) m;
            w.recV[nbuf + lane] = ((unsigned long long) ol_pack(lenB - L, L, true) << 32) | (uint32_t) B;
            st_rec++;
        }
        wave_lds_fence();
        if (lane == 0) *w.recN = (uint32_t) (nbuf + nwon);
        wave_lds_fence();
        if (nbuf + nwon >= WFLUSH) flush_records<CHUNK>(o, w, chunk_base, chunk_fill);
    }
    flush_records<CHUNK, WB>(o, w, chunk_base, chunk_fill);
    close_chunk<CHUNK>(o, chunk_base, chunk_fill);
    st_rec = wave_sum_u64(st_rec);
    if (lane == 0 && st_rec) atomicAdd(&o.counters[CNT_VALID_RECORDS], (unsigned long long) st_rec);
    if (STATS) {
        st_raw = wave_sum_u64(st_raw); st_slots = wave_sum_u64(st_slots); st_win = wave_sum_u64(st_win);
        st_cmp = wave_sum_u64(st_cmp); st_generic = wave_sum_u64(st_generic);
        if (lane == 0) {
            atomicAdd(&o.counters[CNT_RAW], (unsigned long long) st_raw);
            atomicAdd(&o.counters[CNT_SLOTS], (unsigned long long) st_slots);
            atomicAdd(&o.counters[CNT_WINDOWS], (unsigned long long) st_win);
            if (LOCAL) {
                atomicAdd(&o.counters[CNT_TR_COMPARES], (unsigned long long) st_cmp);
                atomicAdd(&o.counters[CNT_LOCAL_GENERIC], (unsigned long long) st_generic);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// k_make_keys : sort key of a record = target id relative to the owned range; everything else
// (chunk padding, foreign targets) gets the all-ones key and sorts behind the valid records
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_make_keys(const uint32_t *__restrict__ rec_dst, uint64_t n_rec, int32_t dst_begin, int32_t dst_end,
                                                    uint32_t *__restrict__ keys, unsigned long long *__restrict__ n_valid) {
    __shared__ unsigned long long s_cnt[4];
    unsigned long long cnt = 0;
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n_rec; i += (uint64_t) gridDim.x * blockDim.x) {
        const uint32_t c = rec_dst[i];
        const bool ok = c != REC_INVALID && (int) c >= dst_begin && (int) c < dst_end;
        keys[i] = ok ? c - (uint32_t) dst_begin : 0xFFFFFFFFu;
        cnt += ok;
    }
    cnt = wave_sum_u64(cnt);
    if (lane_id() == 0) s_cnt[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) { cnt = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3]; if (cnt) atomicAdd(n_valid, cnt); }
}

// rowptr[t] = first index i in the sorted keys with key[i] >= t, for t in [0, n_owned]
__global__ void __launch_bounds__(256) k_rowptr_from_sorted(const uint32_t *__restrict__ keys, const unsigned long long *__restrict__ n_valid_ptr,
                                                             int32_t n_owned, uint32_t *__restrict__ rowptr) {
    const uint64_t nv = (uint64_t) *n_valid_ptr;
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i <= nv; i += (uint64_t) gridDim.x * blockDim.x) {
        const int64_t lo = i == 0 ? 0 : (int64_t) keys[i - 1] + 1;
        const int64_t hi = i == nv ? (int64_t) n_owned : (int64_t) keys[i];
        for (int64_t t = lo; t <= hi; t++) rowptr[t] = (uint32_t) i;
    }
}

// ------------------------------------------------------------------------------------------
// exclusive scan of uint32 (n up to 2^31): reduce tiles, scan tile sums, scan tiles
// ------------------------------------------------------------------------------------------
constexpr int SCAN_BLOCK = 256;
constexpr int SCAN_ITEMS = 16;                             // per thread: four 16-byte loads / stores of consecutive values
constexpr int SCAN_TILE = SCAN_BLOCK * SCAN_ITEMS;

__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t *total, uint32_t *lds /*>= 8 words*/) {
    const int lane = lane_id(), wave = (int) (threadIdx.x >> 6);
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { uint32_t t = (uint32_t) __shfl_up((int) inc, o); if (lane >= o) inc += t; }
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    uint32_t wave_off = 0, tot = 0;
    const int nw = (int) (blockDim.x >> 6);
    for (int w = 0; w < nw; w++) { uint32_t s = lds[w]; if (w < wave) wave_off += s; tot += s; }
    __syncthreads();
    *total = tot;
    return wave_off + inc - v;
}

__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_tile_sums(const uint32_t *__restrict__ in, uint64_t n, uint64_t *__restrict__ tile_sums) {
    __shared__ uint32_t lds[8];
    const uint64_t base = (uint64_t) blockIdx.x * SCAN_TILE;
    uint32_t s = 0;
    if (base + SCAN_TILE <= n && (reinterpret_cast<uintptr_t>(in) & 15u) == 0) {
        const uint4 *in4 = reinterpret_cast<const uint4 *>(in + base);
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS / 4; k++) { const uint4 v = in4[k * SCAN_BLOCK + threadIdx.x]; s += v.x + v.y + v.z + v.w; }
    } else {
        for (int k = 0; k < SCAN_ITEMS; k++) {
            uint64_t i = base + (uint64_t) k * SCAN_BLOCK + threadIdx.x;
            if (i < n) s += in[i];
        }
    }
    uint32_t tot;
    block_exclusive_scan(s, &tot, lds);
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = tot;
}

// single workgroup: exclusive scan of the tile sums (64-bit), total written to tile_sums[n_tiles]
__global__ void __launch_bounds__(1024) k_scan_spine(uint64_t *tile_sums, uint32_t n_tiles) {
    __shared__ uint64_t lds[1024];
    __shared__ uint64_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n_tiles; base += 1024) {
        uint32_t i = base + threadIdx.x;
        uint64_t v = i < n_tiles ? tile_sums[i] : 0;
        lds[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {                 // Hillis-Steele inclusive
            uint64_t t = threadIdx.x >= (unsigned) o ? lds[threadIdx.x - o] : 0;
            __syncthreads();
            lds[threadIdx.x] += t;
            __syncthreads();
        }
        uint64_t inc = lds[threadIdx.x];
        uint64_t c = carry;
        if (i < n_tiles) tile_sums[i] = c + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = c + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) tile_sums[n_tiles] = carry;
}

// out[i] = exclusive prefix (uint32; the host checks the 64-bit total fits); out[n] = total.  Full tiles: a wave owns 1024
// consecutive values and moves them as four fully coalesced 1 KB loads / stores (lane l holds the values 256 k + 4 l .. + 3 of the
// wave's region, k = 0 .. 3); four wave scans of the lanes' sums, the waves' totals through LDS.  (One value per load instruction and
// SCAN_ITEMS consecutive values per thread left a wave's 64 loads 32 bytes apart: 0.30 ms for 90.6 M values, against 0.09 ms for the
// reduction pass over the same input.)
__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_tiles(const uint32_t *__restrict__ in, uint64_t n,
                                                            const uint64_t *__restrict__ tile_sums, uint32_t *__restrict__ out) {
    __shared__ uint32_t lds[8];
    const bool full = (uint64_t) (blockIdx.x + 1) * SCAN_TILE <= n && (reinterpret_cast<uintptr_t>(in) & 15u) == 0 && (reinterpret_cast<uintptr_t>(out) & 15u) == 0;
    if (full) {
        constexpr int G = SCAN_ITEMS / 4;                  // 16-byte groups per lane
        const int lane = lane_id(), wave = (int) (threadIdx.x >> 6);
        const uint64_t wbase = (uint64_t) blockIdx.x * SCAN_TILE + (uint64_t) wave * (64 * SCAN_ITEMS);
        const uint4 *in4 = reinterpret_cast<const uint4 *>(in + wbase);
        uint4 q[G];
        uint32_t ex[G];
#pragma unroll
        for (int k = 0; k < G; k++) q[k] = in4[k * 64 + lane];
        uint32_t run = 0;                                  // values of the wave before group k
#pragma unroll
        for (int k = 0; k < G; k++) {
            const uint32_t sk = q[k].x + q[k].y + q[k].z + q[k].w;
            uint32_t inc = sk;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const uint32_t t = (uint32_t) __shfl_up((int) inc, o); if (lane >= o) inc += t; }
            ex[k] = run + inc - sk;
            run += (uint32_t) __shfl((int) inc, 63);
        }
        if (lane == 0) lds[wave] = run;
        __syncthreads();
        uint32_t woff = (uint32_t) tile_sums[blockIdx.x];
        for (int w = 0; w < wave; w++) woff += lds[w];
        uint4 *out4 = reinterpret_cast<uint4 *>(out + wbase);
#pragma unroll
        for (int k = 0; k < G; k++) {
            uint32_t e = woff + ex[k];
            uint4 r;
            r.x = e; e += q[k].x; r.y = e; e += q[k].y; r.z = e; e += q[k].z; r.w = e;
            out4[k * 64 + lane] = r;
        }
    } else {
        const uint64_t base = (uint64_t) blockIdx.x * SCAN_TILE + (uint64_t) threadIdx.x * SCAN_ITEMS;
        uint32_t v[SCAN_ITEMS];
        uint32_t s = 0;
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; k++) { uint64_t i = base + k; v[k] = i < n ? in[i] : 0u; s += v[k]; }
        uint32_t tot;
        uint32_t ex = block_exclusive_scan(s, &tot, lds) + (uint32_t) tile_sums[blockIdx.x];
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; k++) { uint64_t i = base + k; if (i < n) out[i] = ex; ex += v[k]; }
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) out[n] = (uint32_t) tile_sums[gridDim.x];
}

// ------------------------------------------------------------------------------------------
// k_reduce_targets : per-target replay of the reference's insertion order on the sorted segments
//   processing order == the reference's with --threads=1: small overlaps first (they exist before
//   the reversal at L == rsoemo, GraphCreatorPrefSuf.cpp:288-296), then big overlaps by
//   (L ascending, source id ascending) (:94-100, :369).
//   Survivors are written back, compacted, to the front of the target's segment.
//
//   Fast path (whole workgroup): the records of the workgroup's targets and the first 64 nt of each
//   record's source read are staged in LDS with coalesced / independent loads; the sequential replay
//   then touches LDS only.  A transitive check compares A[d, d+off) with B[0, off) where
//   d + off = off_A: both operands lie inside the first off_A / off_B nucleotides of the SOURCE reads.
//   Slow path: one thread per target on global memory (segments too long for LDS, offsets > 64 nt).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t seg_key(unsigned long long val) {
    const uint32_t ol = (uint32_t) (val >> 32);
    return ((uint64_t) (ol_small(ol) ? 0u : 1u) << 63) | ((uint64_t) (uint32_t) ol_len(ol) << 32) | (uint32_t) val;
}

// A[d .. d+nnt) == B[0 .. nnt) on 2-bit packed rows; replaces the Bitset temp/shift/mismatchBounded
// sequence of GraphCreatorPrefSuf.cpp:434-451 (Bitset.cpp:116-163,879-909).  `wordsA` = readable words of A.
__device__ __forceinline__ bool eq_shifted(const uint32_t *rowA, const uint32_t *rowB, int d, int nnt, int wordsA) {
    const int bit = 2 * d, q = bit >> 5, r = bit & 31;
    const int nbits = 2 * nnt;
    const int nw = (nbits + 31) >> 5;
    uint32_t diff = 0;
    for (int k = 0; k < nw; k++) {
        const uint32_t lo = rowA[q + k];
        const uint32_t hi = (r != 0 && q + k + 1 < wordsA) ? rowA[q + k + 1] : 0u;
        uint32_t x = funnel(lo, hi, r) ^ rowB[k];
        if (k == nw - 1 && (nbits & 31)) x &= (1u << (nbits & 31)) - 1u;
        diff |= x;
    }
    return diff == 0;
}

constexpr int RED_BLOCK = 256;
constexpr int RED_CAP = 2048;        // records staged per workgroup (52 KB of LDS: three workgroups per CU)
constexpr int RED_HEAD_NT = 64;      // nucleotides of each source read staged (4 words)

// heads[i] = first 16 bytes (64 nt) of the source read of sorted record i.  A separate, fully parallel gather:
// inside the reduction the same loads would sit in a dependent chain at one workgroup per CU.
__global__ void __launch_bounds__(256) k_gather_heads(NodesDev nd, const unsigned long long *__restrict__ seg_val,
                                                       const unsigned long long *__restrict__ n_valid_ptr, uint4 *__restrict__ heads) {
    const uint64_t nv = (uint64_t) *n_valid_ptr;
    const bool vec = (nd.stride & 3) == 0 && ((uintptr_t) nd.words & 15u) == 0;
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (uint64_t) gridDim.x * blockDim.x) {
        const uint32_t *row = nd.words + (size_t) (uint32_t) seg_val[i] * nd.stride;
        uint4 h;
        if (vec) h = *reinterpret_cast<const uint4 *>(row);
        else { h.x = row[0]; h.y = nd.stride > 1 ? row[1] : 0u; h.z = nd.stride > 2 ? row[2] : 0u; h.w = nd.stride > 3 ? row[3] : 0u; }
        heads[i] = h;
    }
}

template <bool STATS>
__global__ void __launch_bounds__(RED_BLOCK)
k_reduce_targets(NodesDev nd, PrefSufCfg cfg, int32_t dst_begin, int32_t n_owned, int32_t targets_per_block,
                 const uint32_t *__restrict__ rowptr, unsigned long long *__restrict__ seg_val, const uint4 *__restrict__ heads,
                 uint32_t *__restrict__ out_cnt, uint32_t *__restrict__ outdeg, unsigned long long *__restrict__ counters) {
    __shared__ unsigned long long sVal[RED_CAP];
    __shared__ uint32_t sHead[RED_CAP * 4];
    __shared__ uint16_t sIdx[RED_CAP];
    __shared__ int sSlow;
    const int t0 = blockIdx.x * targets_per_block;
    const int t1 = min(t0 + targets_per_block, n_owned);
    const uint32_t r0 = rowptr[t0], r1 = rowptr[t1];
    const uint32_t cnt = r1 - r0;
    if (threadIdx.x == 0) sSlow = (cnt > (uint32_t) RED_CAP) ? 1 : 0;
    __syncthreads();
    uint64_t st_listed = 0, st_cmp = 0, st_rem = 0;
    uint32_t nlive = 0, k = 0;
    const int t = t0 + (int) threadIdx.x;
    const bool has_target = (int) threadIdx.x < targets_per_block && t < t1;
    bool fast = sSlow == 0;
    if (fast) {
        // stage the records and the first 64 nt of every record's source read (both streams are contiguous)
        int slow = 0;
        for (uint32_t i = threadIdx.x; i < cnt; i += RED_BLOCK) {
            const unsigned long long v = seg_val[r0 + i];
            const uint4 h = heads[r0 + i];
            sVal[i] = v;
            sIdx[i] = (uint16_t) i;
            if (ol_off((uint32_t) (v >> 32)) > RED_HEAD_NT) slow = 1;
            sHead[4 * i + 0] = h.x; sHead[4 * i + 1] = h.y; sHead[4 * i + 2] = h.z; sHead[4 * i + 3] = h.w;
        }
        if (slow) sSlow = 1;
        __syncthreads();
        fast = sSlow == 0;
    }
    if (has_target) {
        const uint32_t gbeg = rowptr[t];
        k = rowptr[t + 1] - gbeg;
        if (fast) {
            uint16_t *ix = sIdx + (gbeg - r0);
            for (uint32_t i = 1; i < k; i++) {                            // order the index list by processing order
                const uint16_t x = ix[i];
                const uint64_t kx = seg_key(sVal[x]);
                uint32_t j = i;
                while (j > 0 && seg_key(sVal[ix[j - 1]]) > kx) { ix[j] = ix[j - 1]; j--; }
                ix[j] = x;
            }
            // the live list is the front of the index list: it never grows past the processed prefix
            for (uint32_t i = 0; i < k; i++) {
                const uint16_t xi = ix[i];
                const unsigned long long vb = sVal[xi];
                const uint32_t B = (uint32_t) vb, ol = (uint32_t) (vb >> 32);
                const int off = ol_off(ol), L = ol_len(ol);
                if (ol_small(ol)) {
                    // Graph::retainOnlySmallestOffset after the reversal (Graph.cpp:348-387): one entry per source
                    bool found = false;
                    for (uint32_t u = 0; u < nlive; u++) {
                        const unsigned long long va = sVal[ix[u]];
                        if ((uint32_t) va == B) { if (off < ol_off((uint32_t) (va >> 32))) ix[u] = xi; found = true; break; }
                    }
                    if (!found) { ix[nlive] = xi; nlive++; }
                } else {
                    if (STATS && off > 0) st_listed += nlive;
                    uint32_t w = 0;
                    for (uint32_t u = 0; u < nlive; u++) {
                        const uint16_t ya = ix[u];
                        const unsigned long long va = sVal[ya];
                        const uint32_t olA = (uint32_t) (va >> 32);
                        bool remove = ((uint32_t) va == B);                       // toRemove[suffId], :461-462
                        if (!remove && off > 0) {                                 // :406
                            const int d = ol_off(olA) - off;                      // offsetDiff, :417
                            if (d >= 0) {                                         // :420
                                if (STATS) st_cmp++;
                                // Read::getRightOffset(rA, rB, d) = |B| + d - |A| = L_B - L_A  (:429)
                                if (L - ol_len(olA) >= 0 && eq_shifted(&sHead[4 * ya], &sHead[4 * xi], d, off, 4)) {
                                    remove = true;
                                    if (STATS) st_rem++;
                                }
                            }
                        }
                        if (!remove) { ix[w] = ya; w++; }
                    }
                    ix[w] = xi;                                                   // pushDirectedEdge(prefId, suffId, offset), :477
                    nlive = w + 1;
                }
            }
            for (uint32_t u = 0; u < nlive; u++) seg_val[gbeg + u] = sVal[ix[u]];
        } else {
            unsigned long long *sv = seg_val + gbeg;
            for (uint32_t i = 1; i < k; i++) {
                const unsigned long long x = sv[i];
                const uint64_t kx = seg_key(x);
                uint32_t j = i;
                while (j > 0 && seg_key(sv[j - 1]) > kx) { sv[j] = sv[j - 1]; j--; }
                sv[j] = x;
            }
            for (uint32_t i = 0; i < k; i++) {
                const unsigned long long vb = sv[i];
                const uint32_t B = (uint32_t) vb, ol = (uint32_t) (vb >> 32);
                const int off = ol_off(ol), L = ol_len(ol);
                if (ol_small(ol)) {
                    bool found = false;
                    for (uint32_t u = 0; u < nlive; u++) {
                        const unsigned long long va = sv[u];
                        if ((uint32_t) va == B) { if (off < ol_off((uint32_t) (va >> 32))) sv[u] = vb; found = true; break; }
                    }
                    if (!found) { sv[nlive] = vb; nlive++; }
                } else {
                    if (STATS && off > 0) st_listed += nlive;
                    const uint32_t *rowB = nd.words + (size_t) B * nd.stride;
                    uint32_t w = 0;
                    for (uint32_t u = 0; u < nlive; u++) {
                        const unsigned long long va = sv[u];
                        const uint32_t A = (uint32_t) va, olA = (uint32_t) (va >> 32);
                        bool remove = (A == B);
                        if (!remove && off > 0) {
                            const int d = ol_off(olA) - off;
                            if (d >= 0) {
                                if (STATS) st_cmp++;
                                if (L - ol_len(olA) >= 0 && eq_shifted(nd.words + (size_t) A * nd.stride, rowB, d, off, nd.stride)) {
                                    remove = true;
                                    if (STATS) st_rem++;
                                }
                            }
                        }
                        if (!remove) { if (w != u) sv[w] = va; w++; }
                    }
                    sv[w] = vb;
                    nlive = w + 1;
                }
            }
        }
        out_cnt[t] = nlive;
        if (cfg.reversed) { if (nlive) outdeg[dst_begin + t] = nlive; }       // never-reversed quirk: rows are the targets
        else for (uint32_t u = 0; u < nlive; u++) atomicAdd(&outdeg[(uint32_t) seg_val[gbeg + u]], 1u);
    }
    if (STATS) {
        st_listed = wave_sum_u64(st_listed); st_cmp = wave_sum_u64(st_cmp); st_rem = wave_sum_u64(st_rem);
        uint64_t mk = wave_max_u64((uint64_t) k);
        if (lane_id() == 0) {
            if (st_listed) atomicAdd(&counters[CNT_TR_LISTED], (unsigned long long) st_listed);
            if (st_cmp) atomicAdd(&counters[CNT_TR_COMPARES], (unsigned long long) st_cmp);
            if (st_rem) atomicAdd(&counters[CNT_TR_REMOVED], (unsigned long long) st_rem);
            atomicMax(&counters[CNT_MAX_IN], (unsigned long long) mk);
        }
    }
}

// ------------------------------------------------------------------------------------------
// final adjacency: scatter survivors to their source row, sort each row by (dst, offset)
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_scatter_by_source(PrefSufCfg cfg, int32_t dst_begin, int32_t n_owned, const uint32_t *__restrict__ rowptr,
                    const unsigned long long *__restrict__ seg_val, const uint32_t *__restrict__ out_cnt,
                    const uint32_t *__restrict__ out_rowptr, uint32_t *__restrict__ out_cursor, alga_edge_dev *__restrict__ edges) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_owned) return;
    const uint32_t beg = rowptr[t], cnt = out_cnt[t];
    const int C = dst_begin + t;
    for (uint32_t u = 0; u < cnt; u++) {
        const unsigned long long v = seg_val[beg + u];
        const int A = (int) (uint32_t) v;
        const int off = ol_off((uint32_t) (v >> 32));
        const int row = cfg.reversed ? C : A;
        const int col = cfg.reversed ? A : C;
        const uint32_t pos = out_rowptr[row] + (atomicSub(&out_cursor[row], 1u) - 1u);
        edges[pos].src = row; edges[pos].dst = col; edges[pos].offset = off;
    }
}

__global__ void __launch_bounds__(256)
k_sort_rows(int32_t n, const uint32_t *__restrict__ out_rowptr, alga_edge_dev *__restrict__ edges) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= n) return;
    const uint32_t beg = out_rowptr[a], k = out_rowptr[a + 1] - beg;
    alga_edge_dev *e = edges + beg;
    for (uint32_t i = 1; i < k; i++) {                    // Graph::retainOnlySmallestOffsetJob's sort, Graph.cpp:367
        const alga_edge_dev x = e[i];
        uint32_t j = i;
        while (j > 0 && (e[j - 1].dst > x.dst || (e[j - 1].dst == x.dst && e[j - 1].offset > x.offset))) { e[j] = e[j - 1]; j--; }
        e[j] = x;
    }
}

// ------------------------------------------------------------------------------------------
// ordering of a gathered edge list (multi-GPU): key = (src << 32) | dst, value = offset
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_edges_to_keys(const alga_edge_dev *__restrict__ e, uint64_t n, unsigned long long *__restrict__ keys,
                                                        uint32_t *__restrict__ vals) {
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
        keys[i] = ((unsigned long long) (uint32_t) e[i].src << 32) | (uint32_t) e[i].dst;
        vals[i] = (uint32_t) e[i].offset;
    }
}

__global__ void __launch_bounds__(256) k_keys_to_edges(const unsigned long long *__restrict__ keys, const uint32_t *__restrict__ vals, uint64_t n,
                                                        alga_edge_dev *__restrict__ e) {
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
        e[i].src = (int32_t) (keys[i] >> 32); e[i].dst = (int32_t) (uint32_t) keys[i]; e[i].offset = (int32_t) vals[i];
    }
}

// Source-side form, final adjacency lists of the sources [src_base, src_base + n_src): the row pointers are the scan of
// the out-degrees the probe wrote; a one-edge source (the fast path) left its edge in first[], every other source's edges
// are in the record list and take their slot with a cursor (rows with more than one edge are ordered by k_sort_rows).
// `second` (may be null): a source with out-degree 2 whose first slot is set has its other edge there (the pair kernel of the
// clustered probe finishes two-edge sources in slots too); k_sort_rows orders the two.
// slot_stride != 0 (round 5): a source with out-degree 3 .. LOCAL_SLOTS_MAX whose first slot is set has its further edges in second[(k - 2) * slot_stride + i]
// (k_probe_stream finishes sources with up to that many standing items).
__global__ void __launch_bounds__(256) k_local_emit_first(int32_t src_base, int32_t n_src, const uint32_t *__restrict__ deg,
                                                           const unsigned long long *__restrict__ first, const unsigned long long *__restrict__ second,
                                                           const uint32_t *__restrict__ rowptr, alga_edge_dev *__restrict__ edges, uint32_t slot_stride) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_src) return;
    const uint32_t dg = deg[i];
    if (dg == 0) return;
    const unsigned long long f = first[i];
    if (f == LOCAL_FIRST_NONE) return;
    alga_edge_dev e;
    e.src = src_base + i; e.dst = (int32_t) (uint32_t) (f >> 32); e.offset = (int32_t) (uint32_t) f;
    const uint32_t at = rowptr[i];
    edges[at] = e;
    if (dg >= 3u && dg <= (uint32_t) LOCAL_SLOTS_MAX && second != nullptr && slot_stride != 0u) {      // up to LOCAL_SLOTS_MAX slots: left in (dst, offset) order as well
        unsigned long long k[LOCAL_SLOTS_MAX];
        k[0] = f;
#pragma unroll
        for (int q = 1; q < LOCAL_SLOTS_MAX; q++) k[q] = (uint32_t) q < dg ? second[(size_t) (q - 1) * slot_stride + (size_t) i] : ~0ull;   // (dst in the high half: order by (dst, offset) == by value)
#pragma unroll
        for (int a = 0; a < LOCAL_SLOTS_MAX; a++)          // odd-even transposition
#pragma unroll
            for (int b = (a & 1); b + 1 < LOCAL_SLOTS_MAX; b += 2) { const unsigned long long lo = min(k[b], k[b + 1]), hi = max(k[b], k[b + 1]); k[b] = lo; k[b + 1] = hi; }
#pragma unroll
        for (int q = 0; q < LOCAL_SLOTS_MAX; q++)
            if ((uint32_t) q < dg) { alga_edge_dev x; x.src = src_base + i; x.dst = (int32_t) (uint32_t) (k[q] >> 32); x.offset = (int32_t) (uint32_t) k[q]; edges[at + q] = x; }
        return;
    }
    if (dg == 2u && second != nullptr) {                   // both slots: left in (dst, offset) order, so that such a row needs no k_sort_rows
        const unsigned long long g = second[i];
        alga_edge_dev e2 = e;
        e2.dst = (int32_t) (uint32_t) (g >> 32); e2.offset = (int32_t) (uint32_t) g;
        if (e2.dst < e.dst || (e2.dst == e.dst && e2.offset < e.offset)) { edges[at] = e2; e2 = e; }
        edges[at + 1] = e2;
    }
}

// cursor[id - src_base] = 0 for the sources of a list (its length read from the device): the only cursors k_local_emit_records will touch
__global__ void __launch_bounds__(256) k_zero_cursors_list(const int32_t *__restrict__ list, const unsigned long long *__restrict__ count, uint32_t cap, int32_t src_base,
                                                            uint32_t *__restrict__ cursor) {
    unsigned long long m = *count;
    if (m > (unsigned long long) cap) m = cap;
    for (unsigned long long q = (unsigned long long) blockIdx.x * blockDim.x + threadIdx.x; q < m; q += (unsigned long long) gridDim.x * blockDim.x) cursor[list[q] - src_base] = 0u;
}

// the rows of a LIST of sources (ids; the list's length is read from the device): the deferred sources of the clustered probe, the
// only ones whose rows are filled from records -- in arbitrary order -- when k_probe_stream ran first
__global__ void __launch_bounds__(256) k_sort_rows_list(const int32_t *__restrict__ list, const unsigned long long *__restrict__ count, uint32_t cap, int32_t src_base,
                                                         const uint32_t *__restrict__ out_rowptr, alga_edge_dev *__restrict__ edges) {
    unsigned long long m = *count;
    if (m > (unsigned long long) cap) m = cap;
    for (unsigned long long q = (unsigned long long) blockIdx.x * blockDim.x + threadIdx.x; q < m; q += (unsigned long long) gridDim.x * blockDim.x) {
        const int a = list[q] - src_base;
        const uint32_t beg = out_rowptr[a], k = out_rowptr[a + 1] - beg;
        alga_edge_dev *e = edges + beg;
        for (uint32_t i = 1; i < k; i++) {
            const alga_edge_dev x = e[i];
            uint32_t j = i;
            while (j > 0 && (e[j - 1].dst > x.dst || (e[j - 1].dst == x.dst && e[j - 1].offset > x.offset))) { e[j] = e[j - 1]; j--; }
            e[j] = x;
        }
    }
}

__global__ void __launch_bounds__(256) k_local_emit_records(int32_t src_base, const uint32_t *__restrict__ rec_dst,
                                                             const unsigned long long *__restrict__ rec_val, uint64_t n_rec,
                                                             const uint32_t *__restrict__ rowptr, uint32_t *__restrict__ cursor,
                                                             alga_edge_dev *__restrict__ edges) {
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n_rec; i += (uint64_t) gridDim.x * blockDim.x) {
        const uint32_t c = rec_dst[i];
        if (c == REC_INVALID) continue;
        const unsigned long long v = rec_val[i];
        const int32_t src = (int32_t) (uint32_t) v;
        alga_edge_dev e;
        e.src = src; e.dst = (int32_t) c; e.offset = ol_off((uint32_t) (v >> 32));
        edges[rowptr[src - src_base] + atomicAdd(&cursor[src - src_base], 1u)] = e;
    }
}

// rows at the caller's stride -> rows at the engine's HBM stride (zero padded); one thread per output word
__global__ void __launch_bounds__(256) k_restride(const uint32_t *__restrict__ in, int stride_in, uint32_t *__restrict__ out, int stride_out, uint64_t n) {
    const uint64_t total = n * (uint64_t) stride_out;
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t) gridDim.x * blockDim.x) {
        const uint64_t r = i / (uint64_t) stride_out;
        const int c = (int) (i - r * (uint64_t) stride_out);
        out[i] = c < stride_in ? in[r * (uint64_t) stride_in + c] : 0u;
    }
}

// The node set of ALGA comes in TWIN PAIRS: node 2k is the reverse complement of node 2k + 1 (src/IO/InputReader.cpp:78-80,363-377; the
// duplicate removal deletes twins together, src/main.cpp:150-232).  A caller that says so (alga_prefsuf_params.twin_rows) sends the rows of the
// odd nodes only -- half of the PCIe upload -- and the even rows are made here: row 2k + 1 = in[k] (re-strided), row 2k = its reverse
// complement over len[2k] nucleotides (0: removed node, an all-zero row).  One thread per output word: nucleotide-reversed, complemented
// input words taken from the end, shifted down by the padding of the last word.
__device__ __forceinline__ uint32_t nuc_revcomp32(uint32_t x) {      // 16 nucleotides reversed and complemented (A0 C1 G2 T3: 3 - c = ~c)
    x = __brev(x);
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
    return ~x;
}
__global__ void __launch_bounds__(256) k_expand_twins(const uint32_t *__restrict__ in, int stride_in, const int32_t *__restrict__ len, uint32_t *__restrict__ out, int stride_out,
                                                       uint64_t n_pairs) {
    const uint64_t total = n_pairs * (uint64_t) stride_out;
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t) gridDim.x * blockDim.x) {
        const uint64_t k = i / (uint64_t) stride_out;
        const int c = (int) (i - k * (uint64_t) stride_out);
        const uint32_t *row = in + k * (uint64_t) stride_in;
        out[(2 * k + 1) * (uint64_t) stride_out + c] = c < stride_in ? row[c] : 0u;
        const int n = len[2 * k];
        uint32_t v = 0u;
        if (n > 0) {
            const int wn = (2 * n + 31) >> 5, pad = 32 * wn - 2 * n;              // words of the read; unused bits of its last word
            if (c < wn) {
                const uint32_t f0 = (wn - 1 - c) < stride_in ? nuc_revcomp32(row[wn - 1 - c]) : 0u;
                const uint32_t f1 = (c + 1 < wn && (wn - 2 - c) < stride_in) ? nuc_revcomp32(row[wn - 2 - c]) : 0u;
                v = pad ? ((f0 >> pad) | (f1 << (32 - pad))) : f0;
            }
        }
        out[(2 * k) * (uint64_t) stride_out + c] = v;
    }
}

// lengths as they crossed PCIe (one or two bytes per node where every read is short enough: 0.09 GB instead of 0.36 GB at 90 M nodes) -> int32
template <typename T>
__global__ void __launch_bounds__(256) k_widen_len(const T *__restrict__ in, int32_t *__restrict__ out, uint64_t n) {
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) out[i] = (int32_t) in[i];
}

// The finished edge list (grouped by src, lists sorted) in the COMPACT form of alga_prefsuf_build_host_compact / alga_download_edges_compact: a
// byte per node (its out-degree; zeroed before this kernel) + per edge the neighbour (4 bytes) and the offset (1 byte) -- 5.1 bytes per edge
// instead of 12 on the way down.  The first edge of a list counts the list (99 % of the lists hold one edge).  *bad is set where a degree or
// an offset does not fit a byte (the caller then takes the triples).
__global__ void __launch_bounds__(256) k_compact_edges(const alga_edge_dev *__restrict__ edges, int32_t n, uint64_t n_edges,
                                                       uint8_t *__restrict__ deg, uint32_t *__restrict__ dst, uint8_t *__restrict__ off, unsigned long long *__restrict__ bad) {
    bool b = false;
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n_edges; i += (uint64_t) gridDim.x * blockDim.x) {
        const alga_edge_dev e = edges[i];
        dst[i] = (uint32_t) e.dst; off[i] = (uint8_t) e.offset;
        b = b || (uint32_t) e.offset > 255u || e.src < 0 || e.src >= n;
        if ((i == 0 || edges[i - 1].src != e.src) && e.src >= 0 && e.src < n) {
            uint32_t d = 1;
            while (i + d < n_edges && d < 256u && edges[i + d].src == e.src) d++;
            deg[e.src] = (uint8_t) d;
            b = b || d > 255u;
        }
    }
    if (b) atomicOr(bad, 1ull);
}

// ------------------------------------------------------------------------------------------
// launch wrappers (host)
// ------------------------------------------------------------------------------------------
static inline unsigned grid_for(uint64_t n, int block) { return (unsigned) ((n + (uint64_t) block - 1) / (uint64_t) block); }

void launch_widen_len(const void *in, int elem_bytes, int32_t *out, uint64_t n, hipStream_t s) {
    if (n == 0) return;
    const dim3 g((unsigned) std::min<uint64_t>((n + 255) / 256, 1u << 16)), b(256);
    if (elem_bytes == 1) hipLaunchKernelGGL((k_widen_len<uint8_t>), g, b, 0, s, (const uint8_t *) in, out, n);
    else hipLaunchKernelGGL((k_widen_len<uint16_t>), g, b, 0, s, (const uint16_t *) in, out, n);
}

void launch_compact_edges(const alga_edge_dev *edges, int32_t n, uint64_t n_edges, uint8_t *deg, uint32_t *dst, uint8_t *off, unsigned long long *bad, hipStream_t s) {
    if (n > 0) (void) hipMemsetAsync(deg, 0, (size_t) n, s);
    if (n_edges == 0) return;
    hipLaunchKernelGGL(k_compact_edges, dim3((unsigned) std::min<uint64_t>((n_edges + 255) / 256, 1u << 16)), dim3(256), 0, s, edges, n, n_edges, deg, dst, off, bad);
}

void launch_expand_twins(const uint32_t *in, int stride_in, const int32_t *len, uint32_t *out, int stride_out, uint64_t n_pairs, hipStream_t s) {
    if (n_pairs == 0) return;
    hipLaunchKernelGGL(k_expand_twins, dim3((unsigned) std::min<uint64_t>((n_pairs * (uint64_t) stride_out + 255) / 256, 1u << 16)), dim3(256), 0, s, in, stride_in, len, out,
                       stride_out, n_pairs);
}

void launch_restride(const uint32_t *in, int stride_in, uint32_t *out, int stride_out, uint64_t n, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_restride, dim3((unsigned) std::min<uint64_t>((n * (uint64_t) stride_out + 255) / 256, 1u << 16)), dim3(256), 0, s, in, stride_in, out, stride_out, n);
}

void launch_node_stats(const NodesDev &nd, unsigned long long *counters, int *max_len, hipStream_t s) {
    if (nd.n <= 0) return;
    unsigned g = std::min<unsigned>(grid_for((uint64_t) nd.n, 256), 1024u);
    hipLaunchKernelGGL(k_node_stats, dim3(g), dim3(256), 0, s, nd, counters, max_len);
}

uint32_t seed_buckets_for(uint64_t live, int fill_x10) {   // average entries per 8-entry bucket = fill_x10 / 10
    if (fill_x10 < 5) fill_x10 = 5;
    if (fill_x10 > 60) fill_x10 = 60;
    uint64_t nb = (live * 10 + (uint64_t) fill_x10 - 1) / (uint64_t) fill_x10;
    if (nb < 64) nb = 64;
    return (uint32_t) std::min<uint64_t>(nb, 0x7FFFFFFFull);
}

// Prefilter size in bits (a power of two), 0 = off.  Measured on MI355X:
//  * up to 4 M nodes a bitmap of 8 bits/node fits every XCD's 4 MB L2 next to the streaming traffic: -12 % probe time at
//    1.7 M nodes;
//  * 6 M .. 14 M nodes: an L2-sized bitmap rejects too little, a bigger one misses L2 as often as the bucket reads it saves,
//    and its build (one more random atomic per node) costs what the probe gains: off;
//  * from 16 M nodes table + rows outgrow the 256 MB Infinity Cache and every bucket read goes to HBM, while a bitmap of up to
//    128 MB still lives in that cache: -16 % at 90 M nodes (30x coverage).  At high coverage most windows are real hits and
//    the lookups are wasted: the probing waves measure their pass rate and stop consulting the filter above 50 %.
uint32_t seed_filter_bits_for(uint64_t live) {
    const uint64_t want = live * 8;
    uint64_t bits = 1ull << 16;
    while (bits < want) bits <<= 1;
    if (bits <= (1ull << 25)) return (uint32_t) bits;
    if (live < (1ull << 24)) return 0;
    return (uint32_t) std::min<uint64_t>(bits, 1ull << 30);
}

void launch_seed_build(const NodesDev &nd, const PrefSufCfg &cfg, unsigned long long *table, uint32_t n_buckets, uint32_t *filter,
                       uint32_t filter_bits, hipStream_t s) {
    if (nd.n <= 0) return;
    hipLaunchKernelGGL(k_seed_build, dim3(grid_for((uint64_t) nd.n, 256)), dim3(256), 0, s, nd, cfg, table, n_buckets,
                       filter_bits ? filter : nullptr, filter_bits ? filter_bits - 1 : 0u);
}

static uint64_t probe_blocks(int n_cu, uint64_t n_src) {
    // persistent grid: 8 workgroups of 4 waves per CU fill the 32 wave slots of a CU
    return std::max<uint64_t>(1, std::min<uint64_t>((n_src + PROBE_WAVES - 1) / PROBE_WAVES, (uint64_t) std::max(1, n_cu) * 8));
}

int local_item_capacity() { return ITEMMAX; }
int local_big_limit() { return ITEM_BIG_MAX; }
size_t probe_big_bytes(int n_cu, uint32_t count, int local, uint32_t item_cap) {
    return probe_blocks(n_cu, (uint64_t) count) * PROBE_WAVES * (size_t) item_cap * (8 + 16 * (size_t) local) + 64;
}

uint64_t probe_record_slack(int n_cu, uint64_t n_src, bool local) {  // worst-case invalid padding of one launch
    return probe_blocks(n_cu, n_src) * PROBE_WAVES * (uint64_t) (local ? REC_CHUNK_LOCAL : REC_CHUNK);
}

struct ProbeTable { const unsigned long long *table; uint32_t n_buckets; const uint32_t *filter; uint32_t filter_mask; };

template <int NQ, int LOCAL>
static void launch_probe_nql(const NodesDev &nd, const PrefSufCfg &cfg, const ProbeTable &t,
                             int32_t src_begin, int32_t src_end, const ProbeOut &o, dim3 grid, dim3 block, hipStream_t s) {
    if constexpr (LOCAL != 0) {
        if (o.big_count) {                                 // second pass over the listed sources (statistics always on: it is rare)
            hipLaunchKernelGGL((k_probe_sources<true, NQ, LOCAL, true>), grid, block, 0, s, nd, cfg, t.table, t.n_buckets, t.filter, t.filter_mask, src_begin, src_end, o);
            return;
        }
    }
    if (cfg.stats) hipLaunchKernelGGL((k_probe_sources<true, NQ, LOCAL>), grid, block, 0, s, nd, cfg, t.table, t.n_buckets, t.filter, t.filter_mask, src_begin, src_end, o);
    else           hipLaunchKernelGGL((k_probe_sources<false, NQ, LOCAL>), grid, block, 0, s, nd, cfg, t.table, t.n_buckets, t.filter, t.filter_mask, src_begin, src_end, o);
}
template <int NQ>
static void launch_probe_nq(const NodesDev &nd, const PrefSufCfg &cfg, const ProbeTable &t, int local,
                            int32_t src_begin, int32_t src_end, const ProbeOut &o, dim3 grid, dim3 block, hipStream_t s) {
    if (local == 1)      launch_probe_nql<NQ, 1>(nd, cfg, t, src_begin, src_end, o, grid, block, s);
    else if (local == 2) launch_probe_nql<NQ, 2>(nd, cfg, t, src_begin, src_end, o, grid, block, s);
    else                 launch_probe_nql<NQ, 0>(nd, cfg, t, src_begin, src_end, o, grid, block, s);
}

void launch_probe(const NodesDev &nd, const PrefSufCfg &cfg, const unsigned long long *table, uint32_t n_buckets,
                  const uint32_t *filter, uint32_t filter_bits,
                  int32_t src_begin, int32_t src_end, uint32_t *rec_dst, unsigned long long *rec_val, uint64_t rec_cap,
                  unsigned long long *counters, int n_cu, int local, uint32_t *deg, unsigned long long *first, const ProbeBig *big, hipStream_t s) {
    const int64_t ns = (int64_t) src_end - src_begin;
    if (ns <= 0) return;
    const bool second = local && big && big->count > 0;    // second pass: the grid covers the listed sources
    dim3 grid((unsigned) probe_blocks(n_cu, second ? (uint64_t) big->count : (uint64_t) ns)), block(PROBE_WAVES * 64);
    ProbeOut o{rec_dst, rec_val, rec_cap, counters, deg, first, src_begin};
    if (local && big) {
        o.big_list = big->list; o.big_list_cap = big->list_cap;
        if (second) {                                      // items: [O: uint4 x local | C: u32 | M: u32] per item, `item_cap` items per wave
            const size_t items = (size_t) grid.x * PROBE_WAVES * (size_t) big->item_cap;
            o.big_count = big->count;
            o.bigO = (uint4 *) big->items;
            o.bigC = (uint32_t *) (o.bigO + items * (size_t) local);
            o.bigM = o.bigC + items;
            o.big_cap = big->item_cap;
        }
    }
    ProbeTable t{table, n_buckets, filter_bits ? filter : nullptr, filter_bits ? filter_bits - 1 : 0u};
    // widest prefix ever compared: Lcap nucleotides; wide path needs 16-byte aligned rows that hold it
    const int need_q = (((2 * cfg.Lcap + 31) >> 5) + 3) >> 2;
    const bool aligned = (nd.stride & 3) == 0 && ((uintptr_t) nd.words & 15u) == 0;
    const int row_q = nd.stride >> 2;
    if (aligned && need_q <= 2 && row_q >= 2)      launch_probe_nq<2>(nd, cfg, t, local, src_begin, src_end, o, grid, block, s);
    else if (aligned && need_q <= 3 && row_q >= 3) launch_probe_nq<3>(nd, cfg, t, local, src_begin, src_end, o, grid, block, s);
    else if (aligned && need_q <= 4 && row_q >= 4) launch_probe_nq<4>(nd, cfg, t, local, src_begin, src_end, o, grid, block, s);
    else                                           launch_probe_nq<0>(nd, cfg, t, local, src_begin, src_end, o, grid, block, s);
}

void launch_local_emit(int32_t src_base, int32_t n_src, const uint32_t *deg, const unsigned long long *first, const unsigned long long *second,
                       const uint32_t *rec_dst, const unsigned long long *rec_val, uint64_t n_rec, const uint32_t *rowptr, uint32_t *cursor,
                       alga_edge_dev *edges, const int32_t *record_sources, const unsigned long long *record_sources_count, uint32_t record_sources_cap, hipStream_t s,
                       uint32_t slot_stride) {
    if (n_src <= 0) return;
    // the cursors of the record rows start at zero: all of them, or those of the listed sources alone
    if (record_sources) hipLaunchKernelGGL(k_zero_cursors_list, dim3(std::min<unsigned>(grid_for((uint64_t) record_sources_cap, 256), 2048u)), dim3(256), 0, s, record_sources, record_sources_count,
                                           record_sources_cap, src_base, cursor);
    else (void) hipMemsetAsync(cursor, 0, (size_t) (n_src + 1) * sizeof(uint32_t), s);
    hipLaunchKernelGGL(k_local_emit_first, dim3(grid_for((uint64_t) n_src, 256)), dim3(256), 0, s, src_base, n_src, deg, first, second, rowptr, edges, slot_stride);
    if (n_rec) {
        unsigned g = std::min<unsigned>(grid_for(n_rec, 256), 4096u);
        hipLaunchKernelGGL(k_local_emit_records, dim3(std::max(1u, g)), dim3(256), 0, s, src_base, rec_dst, rec_val, n_rec, rowptr, cursor, edges);
    }
    // rows of two slot edges leave k_local_emit_first ordered; only rows filled from records need the sort: all rows, or -- when the caller
    // knows which sources those are (the clustered probe's defer list) -- that list alone (0.34 ms of reading 90.6 M row pointers for 43 k rows)
    if (record_sources) {
        if (n_rec) hipLaunchKernelGGL(k_sort_rows_list, dim3(std::min<unsigned>(grid_for((uint64_t) record_sources_cap, 256), 2048u)), dim3(256), 0, s, record_sources, record_sources_count,
                                      record_sources_cap, src_base, rowptr, edges);
    } else hipLaunchKernelGGL(k_sort_rows, dim3(grid_for((uint64_t) n_src, 256)), dim3(256), 0, s, n_src, rowptr, edges);
}

void launch_make_keys(const uint32_t *rec_dst, uint64_t n_rec, int32_t dst_begin, int32_t dst_end, uint32_t *keys,
                      unsigned long long *n_valid, hipStream_t s) {
    if (n_rec == 0) return;
    unsigned g = std::min<unsigned>(grid_for(n_rec, 256 * 4), 4096u);
    hipLaunchKernelGGL(k_make_keys, dim3(std::max(1u, g)), dim3(256), 0, s, rec_dst, n_rec, dst_begin, dst_end, keys, n_valid);
}

void launch_rowptr_from_sorted(const uint32_t *keys, const unsigned long long *n_valid_ptr, uint64_t n_rec_max, int32_t n_owned,
                               uint32_t *rowptr, hipStream_t s) {
    unsigned g = std::min<unsigned>(grid_for(n_rec_max + 1, 256), 8192u);
    hipLaunchKernelGGL(k_rowptr_from_sorted, dim3(std::max(1u, g)), dim3(256), 0, s, keys, n_valid_ptr, n_owned, rowptr);
}

size_t scan_scratch_bytes(uint64_t n) {
    uint64_t tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    return (size_t) (tiles + 2) * sizeof(uint64_t);
}

// out must hold n+1 entries; out[n] = total (low 32 bits); the 64-bit total is scratch[tiles]
void launch_exclusive_scan(const uint32_t *in, uint64_t n, uint32_t *out, uint64_t *scratch, hipStream_t s) {
    if (n == 0) { (void) hipMemsetAsync(out, 0, sizeof(uint32_t), s); (void) hipMemsetAsync(scratch, 0, 2 * sizeof(uint64_t), s); return; }
    const uint32_t tiles = (uint32_t) ((n + SCAN_TILE - 1) / SCAN_TILE);
    hipLaunchKernelGGL(k_scan_tile_sums, dim3(tiles), dim3(SCAN_BLOCK), 0, s, in, n, scratch);
    hipLaunchKernelGGL(k_scan_spine, dim3(1), dim3(1024), 0, s, scratch, tiles);
    hipLaunchKernelGGL(k_scan_tiles, dim3(tiles), dim3(SCAN_BLOCK), 0, s, in, n, scratch, out);
}

uint64_t scan_total_index(uint64_t n) { return (n + SCAN_TILE - 1) / SCAN_TILE; }

// The few words the host waits for at a sync point, from wherever they lie on the device, in ONE tiny kernel straight into the pinned host block
// (a device-to-host copy per word costs a blit kernel and a PCIe round trip each, one after the other on the stream).
__global__ void __launch_bounds__(256) k_mail(MailArgs a, uint32_t *__restrict__ dst) {
    for (int i = 0; i < a.n; i++)
        for (uint32_t w = threadIdx.x; w < a.seg[i].words; w += 256) dst[a.seg[i].dst + w] = a.seg[i].src[w];
}

void launch_mail(const MailArgs &a, uint32_t *host_block_dev, hipStream_t s) {
    if (a.n > 0) hipLaunchKernelGGL(k_mail, dim3(1), dim3(256), 0, s, a, host_block_dev);
}

int reduce_targets_per_block(uint64_t n_records, uint64_t n_targets) {
    // aim at ~60 % of the LDS record capacity so that ordinary fluctuations stay on the fast path
    const double avg = n_targets ? (double) n_records / (double) n_targets : 0.0;
    int tpb = avg > 0 ? (int) (0.6 * RED_CAP / avg) : RED_BLOCK;
    return std::max(16, std::min(RED_BLOCK, tpb));
}

void launch_gather_heads(const NodesDev &nd, const unsigned long long *seg_val, const unsigned long long *n_valid_ptr, uint64_t n_rec_max,
                         void *heads, hipStream_t s) {
    if (n_rec_max == 0) return;
    unsigned g = std::min<unsigned>(grid_for(n_rec_max, 256), 16384u);
    hipLaunchKernelGGL(k_gather_heads, dim3(std::max(1u, g)), dim3(256), 0, s, nd, seg_val, n_valid_ptr, (uint4 *) heads);
}

void launch_reduce_targets(const NodesDev &nd, const PrefSufCfg &cfg, int32_t dst_begin, int32_t n_owned, int32_t targets_per_block,
                           const uint32_t *rowptr, unsigned long long *seg_val, const void *heads, uint32_t *out_cnt, uint32_t *outdeg,
                           unsigned long long *counters, hipStream_t s) {
    if (n_owned <= 0) return;
    dim3 grid(grid_for((uint64_t) n_owned, targets_per_block)), block(RED_BLOCK);
    if (cfg.stats)
        hipLaunchKernelGGL(k_reduce_targets<true>, grid, block, 0, s, nd, cfg, dst_begin, n_owned, targets_per_block, rowptr, seg_val,
                           (const uint4 *) heads, out_cnt, outdeg, counters);
    else
        hipLaunchKernelGGL(k_reduce_targets<false>, grid, block, 0, s, nd, cfg, dst_begin, n_owned, targets_per_block, rowptr, seg_val,
                           (const uint4 *) heads, out_cnt, outdeg, counters);
}

void launch_scatter_by_source(const PrefSufCfg &cfg, int32_t dst_begin, int32_t n_owned, const uint32_t *rowptr,
                              const unsigned long long *seg_val, const uint32_t *out_cnt, const uint32_t *out_rowptr, uint32_t *out_cursor,
                              alga_edge_dev *edges, hipStream_t s) {
    if (n_owned <= 0) return;
    hipLaunchKernelGGL(k_scatter_by_source, dim3(grid_for((uint64_t) n_owned, 256)), dim3(256), 0, s, cfg, dst_begin, n_owned, rowptr, seg_val,
                       out_cnt, out_rowptr, out_cursor, edges);
}

void launch_edges_to_keys(const alga_edge_dev *e, uint64_t n, unsigned long long *keys, uint32_t *vals, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_edges_to_keys, dim3(std::min<unsigned>(grid_for(n, 256), 8192u)), dim3(256), 0, s, e, n, keys, vals);
}

void launch_keys_to_edges(const unsigned long long *keys, const uint32_t *vals, uint64_t n, alga_edge_dev *e, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_keys_to_edges, dim3(std::min<unsigned>(grid_for(n, 256), 8192u)), dim3(256), 0, s, keys, vals, n, e);
}

void launch_sort_rows(int32_t n, const uint32_t *out_rowptr, alga_edge_dev *edges, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_sort_rows, dim3(grid_for((uint64_t) n, 256)), dim3(256), 0, s, n, out_rowptr, edges);
}

} // namespace alga
