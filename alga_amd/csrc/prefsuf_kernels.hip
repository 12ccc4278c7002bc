// alga_amd/csrc/prefsuf_kernels.hip -- gfx950 kernels of the PrefSuf overlap engine.
//
// Replaces the per-overlap-length hash join + edge policy of
// src/GraphCreators/GraphCreatorPrefSuf.cpp:238-488 (reference paths relative to its root).
// Integer / byte work only: no MFMA.  Wavefront = 64 lanes throughout.
//
//   k_node_stats        max read length, live-node count
//   k_seed_build        fingerprint of the min_overlap-long prefix of every target -> seed table
//   k_probe_sources     one wavefront per source: the source's tail is staged in LDS, lane p
//                       fingerprints suffix window p, probes the table, verifies candidates with
//                       an exact 2-bit compare, keeps the per-source small-overlap top-3
//                       (wave max-reductions) and appends records (wave-aggregated atomics)
//   k_count_targets     in-degree histogram for records that came from another rank
//   k_scan_*            exclusive scan (row pointers)
//   k_scatter_by_target records -> per-target segments
//   k_reduce_targets    per target: replay of the reference's insertion order
//                       (small-overlap dedupe, transitive reduction with 2-bit compares)
//   k_scatter_by_source / k_sort_rows   final adjacency lists sorted by (dst, offset)
#include <hip/hip_runtime.h>
#include "prefsuf_common.h"
#include "prefsuf_kernels.h"
#include <algorithm>

namespace alga {

// ------------------------------------------------------------------------------------------
// wave helpers (64 lanes)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int) (threadIdx.x & 63u); }

__device__ __forceinline__ uint64_t shfl_u64(uint64_t v, int src_lane) {
    uint32_t lo = (uint32_t) v, hi = (uint32_t) (v >> 32);
    lo = (uint32_t) __shfl((int) lo, src_lane);
    hi = (uint32_t) __shfl((int) hi, src_lane);
    return ((uint64_t) hi << 32) | lo;
}

__device__ __forceinline__ uint64_t wave_max_u64(uint64_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        uint32_t lo = (uint32_t) __shfl_xor((int) (uint32_t) v, o);
        uint32_t hi = (uint32_t) __shfl_xor((int) (uint32_t) (v >> 32), o);
        uint64_t t = ((uint64_t) hi << 32) | lo;
        v = t > v ? t : v;
    }
    return v;
}

__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        uint32_t lo = (uint32_t) __shfl_xor((int) (uint32_t) v, o);
        uint32_t hi = (uint32_t) __shfl_xor((int) (uint32_t) (v >> 32), o);
        v += ((uint64_t) hi << 32) | lo;
    }
    return v;
}

// Slot in an append-only list for every CURRENTLY ACTIVE lane: one atomic per wave-instruction.
__device__ __forceinline__ uint64_t wave_append(unsigned long long *counter) {
    const uint64_t active = __ballot(1);
    const int leader = __ffsll((long long) active) - 1;
    const int lane = lane_id();
    uint64_t base = 0;
    if (lane == leader) base = atomicAdd(counter, (unsigned long long) __popcll(active));
    base = shfl_u64(base, leader);
    return base + (uint64_t) __popcll(active & ((1ull << lane) - 1ull));
}

// 32 bits of a staged bit string starting at word q, bit r (v_alignbit_b32)
__device__ __forceinline__ uint32_t funnel(uint32_t lo, uint32_t hi, int r) {
    return __funnelshift_r(lo, hi, r);
}

// ------------------------------------------------------------------------------------------
// k_node_stats
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_node_stats(NodesDev nd, unsigned long long *counters, int *max_len) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int l = 0;
    if (i < nd.n) l = nd.len[i];
    int m = l;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { int t = __shfl_xor(m, o); m = t > m ? t : m; }
    uint64_t live = __popcll(__ballot(l > 0));
    if (lane_id() == 0) {
        if (m > 0) atomicMax(max_len, m);
        if (live) atomicAdd(&counters[CNT_LIVE_NODES], (unsigned long long) live);
    }
}

// ------------------------------------------------------------------------------------------
// k_seed_build : one thread per target node
//   replaces updatePrefixHash + putKmersIntoBucketsJob (GraphCreatorPrefSuf.cpp:213-223,323-332)
//   for the single length min_overlap; longer overlaps are found by extension, not by re-hashing.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_seed_build(NodesDev nd, PrefSufCfg cfg, unsigned long long *table, uint32_t mask) {
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
    uint32_t slot = (uint32_t) h & mask;
    const unsigned long long entry = (h & 0xFFFFFFFF00000000ull) | (uint32_t) i;
    for (;;) {
        unsigned long long old = atomicCAS(&table[slot], (unsigned long long) SEED_EMPTY, entry);
        if (old == SEED_EMPTY) break;
        slot = (slot + 1) & mask;
    }
}

// ------------------------------------------------------------------------------------------
// k_probe_sources : persistent wavefronts, one source node B at a time per wavefront
//
//   * B's tail (<= 501 nt) is staged in the wave's LDS row; lane p owns suffix window p
//   * records are collected in a per-wave LDS buffer and flushed with coalesced 64-lane stores
//     into chunks of the global record list; a chunk is reserved with ONE global atomic
//     (a returning atomic on one address sustains only ~88 ops/us chip-wide, so per-record or
//     per-source reservations would cap the kernel at tens of milliseconds)
//   * unused chunk tails are filled with REC_INVALID and skipped downstream
// ------------------------------------------------------------------------------------------
constexpr int PROBE_WAVES = 4;        // waves per workgroup
constexpr int STAGE_WORDS = 36;       // (2*501+31)/32 + alignment slack + 1 zero word
constexpr int WBUF = 256;             // per-wave LDS record buffer (records)
constexpr int WFLUSH = 128;           // flush once this many are buffered
constexpr int REC_CHUNK = 512;        // records reserved per global atomic

__device__ __forceinline__ void top3_insert(uint64_t &a, uint64_t &b, uint64_t &c, uint64_t k) {
    if (k > a) { c = b; b = a; a = k; }
    else if (k > b) { c = b; b = k; }
    else if (k > c) { c = k; }
}

struct ProbeOut {
    uint32_t *__restrict__ rec_dst, *__restrict__ rec_src, *__restrict__ rec_ol;
    uint64_t rec_cap;
    uint32_t *__restrict__ indeg;
    int32_t dst_begin, dst_end;
    unsigned long long *__restrict__ counters;
};

__device__ __forceinline__ void store_record(const ProbeOut &o, uint64_t idx, uint32_t C, uint32_t B, uint32_t ol) {
    if (idx < o.rec_cap) { o.rec_dst[idx] = C; o.rec_src[idx] = B; o.rec_ol[idx] = ol; }
    if (o.indeg && (int) C >= o.dst_begin && (int) C < o.dst_end) atomicAdd(&o.indeg[(int) C - o.dst_begin], 1u);
}

// Convergent: all 64 lanes.  Moves the wave's LDS buffer to the record list.
__device__ __forceinline__ void flush_records(const ProbeOut &o, uint32_t *sC, uint32_t *sS, uint32_t *sO, uint32_t *sCnt,
                                              uint64_t &chunk_base, int &chunk_fill) {
    const int lane = lane_id();
    int n = (int) __builtin_amdgcn_readfirstlane((int) *sCnt);
    if (n > WBUF) n = WBUF;                      // the excess went out through the direct path
    if (n == 0) return;
    if (chunk_fill + n > REC_CHUNK) {
        // close the current chunk: invalid markers in its tail
        for (int i = chunk_fill + lane; i < REC_CHUNK; i += 64) {
            const uint64_t idx = chunk_base + (uint64_t) i;
            if (idx < o.rec_cap) o.rec_dst[idx] = REC_INVALID;
        }
        uint64_t base = 0;
        if (lane == 0) base = atomicAdd(&o.counters[CNT_RECORDS], (unsigned long long) REC_CHUNK);
        chunk_base = shfl_u64(base, 0);
        chunk_fill = 0;
    }
    for (int i = lane; i < n; i += 64) store_record(o, chunk_base + (uint64_t) (chunk_fill + i), sC[i], sS[i], sO[i]);
    chunk_fill += n;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (lane == 0) *sCnt = 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

template <bool STATS>
__global__ void __launch_bounds__(PROBE_WAVES * 64)
k_probe_sources(NodesDev nd, PrefSufCfg cfg, const unsigned long long *__restrict__ table, uint32_t mask,
                int32_t src_begin, int32_t src_end, ProbeOut o) {
    __shared__ uint32_t sB[PROBE_WAVES][STAGE_WORDS];
    __shared__ uint32_t sRecC[PROBE_WAVES][WBUF];
    __shared__ uint32_t sRecS[PROBE_WAVES][WBUF];
    __shared__ uint32_t sRecO[PROBE_WAVES][WBUF];
    __shared__ uint32_t sRecN[PROBE_WAVES];
    const int wave = (int) (threadIdx.x >> 6);
    const int lane = lane_id();
    uint32_t *sb = sB[wave];
    uint32_t *sC = sRecC[wave], *sS = sRecS[wave], *sO = sRecO[wave], *sCnt = &sRecN[wave];
    if (lane == 0) *sCnt = 0;
    uint64_t chunk_base = 0;
    int chunk_fill = REC_CHUNK;                            // "no chunk yet"
    uint64_t st_raw = 0, st_slots = 0, st_win = 0, st_rec = 0;
    const int64_t total_waves = (int64_t) gridDim.x * PROBE_WAVES;

    for (int64_t Bl = (int64_t) src_begin + (int64_t) blockIdx.x * PROBE_WAVES + wave; Bl < src_end; Bl += total_waves) {
        const int B = (int) Bl;
        const int lenB = nd.len[B];
        if (!(lenB >= cfg.Lmin && lenB > 0 && (!nd.from || nd.from[B]))) continue;      // wave-uniform
        // stage the last Lspan nucleotides of B (all an overlap of length <= Lcap can touch)
        const int Lspan = lenB < cfg.Lcap ? lenB : cfg.Lcap;
        const int w0 = (2 * (lenB - Lspan)) >> 5;         // first staged word of the row
        const int nwB = blocks_of(lenB) - w0;              // staged words (<= 33)
        {
            const uint32_t *row = nd.words + (size_t) B * nd.stride;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (lane < STAGE_WORDS) sb[lane] = lane < nwB ? row[w0 + lane] : 0u;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        const int nwin = Lspan - cfg.Lmin + 1;             // overlap lengths Lmin..Lspan
        uint64_t k0 = 0, k1 = 0, k2 = 0;                   // per-lane top-3 small overlaps, key=(L<<32)|C

        for (int base = 0; base < nwin; base += 64) {
            const int widx = base + lane;
            if (widx < nwin) {
                const int L = Lspan - widx;
                const int off = lenB - L;                  // == offset of the edge B -> C
                const int bit = 2 * off - 32 * w0;
                const int q = bit >> 5, r = bit & 31;
                uint64_t h = fp_init();
                for (int k = 0; k < cfg.seed_words; k++) {
                    uint32_t w = funnel(sb[q + k], sb[q + k + 1], r);
                    if (k == cfg.seed_words - 1) w &= cfg.seed_last_mask;
                    h = fp_step(h, w);
                }
                h = fp_final(h);
                const uint32_t tag = (uint32_t) (h >> 32);
                uint32_t slot = (uint32_t) h & mask;
                if (STATS) st_win++;
                const int nwL = (2 * L + 31) >> 5;
                const uint32_t lastmask = (2 * L & 31) ? ((1u << (2 * L & 31)) - 1u) : 0xFFFFFFFFu;
                for (;;) {
                    const unsigned long long e = table[slot];
                    if (STATS) st_slots++;
                    if (e == SEED_EMPTY) break;
                    slot = (slot + 1) & mask;
                    if ((uint32_t) (e >> 32) != tag) continue;
                    const int C = (int) (uint32_t) e;
                    if (C == B) continue;                                  // GraphCreatorPrefSuf.cpp:386
                    if (nd.len[C] < L) continue;                           // a prefix of length L must exist (:215)
                    // exact verification: B[off .. lenB) == C[0 .. L)
                    const uint32_t *rc = nd.words + (size_t) C * nd.stride;
                    bool ok = true;
                    for (int k = 0; k < nwL; k++) {
                        uint32_t x = funnel(sb[q + k], sb[q + k + 1], r) ^ rc[k];
                        if (k == nwL - 1) x &= lastmask;
                        if (x) { ok = false; break; }
                    }
                    if (!ok) continue;
                    if (STATS) st_raw++;
                    if (L < cfg.rsoemo) {
                        top3_insert(k0, k1, k2, ((uint64_t) (uint32_t) L << 32) | (uint32_t) C);   // :397-401
                    } else {
                        st_rec++;
                        const uint32_t i = atomicAdd(sCnt, 1u);            // LDS atomic
                        if (i < (uint32_t) WBUF) { sC[i] = (uint32_t) C; sS[i] = (uint32_t) B; sO[i] = ol_pack(off, L, false); }
                        else {                                             // buffer full (heavy repeats): direct, slow path
                            const uint64_t idx = atomicAdd(&o.counters[CNT_RECORDS], 1ull);
                            store_record(o, idx, (uint32_t) C, (uint32_t) B, ol_pack(off, L, false));
                        }
                    }
                }
            }
        }
        // per-source small-overlap cap: the reference keeps the LAST `SOES`=3 pushes in (L asc, C asc)
        // order (GraphCreatorPrefSuf.cpp:400-401) == the 3 largest (L, C) keys.
        uint64_t win[3] = {0, 0, 0};
        int nwon = 0;
#pragma unroll
        for (int rnd = 0; rnd < 3; rnd++) {
            const uint64_t m = wave_max_u64(k0);
            if (m == 0) break;
            if (k0 == m) { k0 = k1; k1 = k2; k2 = 0; }
            win[rnd] = m; nwon = rnd + 1;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        int nbuf = (int) __builtin_amdgcn_readfirstlane((int) *sCnt);
        if (nbuf > WBUF) nbuf = WBUF;
        if (nbuf + nwon > WBUF) {                           // make room (convergent)
            flush_records(o, sC, sS, sO, sCnt, chunk_base, chunk_fill);
            nbuf = 0;
        }
        if (lane < nwon) {
            const uint64_t m = lane == 0 ? win[0] : (lane == 1 ? win[1] : win[2]);
            const int L = (int) (m >> 32);
            sC[nbuf + lane] = (uint32_t) m; sS[nbuf + lane] = (uint32_t) B; sO[nbuf + lane] = ol_pack(lenB - L, L, true);
            st_rec++;
        }
        if (lane == 0) *sCnt = (uint32_t) (nbuf + nwon);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (nbuf + nwon >= WFLUSH) flush_records(o, sC, sS, sO, sCnt, chunk_base, chunk_fill);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    flush_records(o, sC, sS, sO, sCnt, chunk_base, chunk_fill);
    // invalid markers in the unused tail of the wave's last chunk
    if (chunk_fill < REC_CHUNK) {
        for (int i = chunk_fill + lane; i < REC_CHUNK; i += 64) {
            const uint64_t idx = chunk_base + (uint64_t) i;
            if (idx < o.rec_cap) o.rec_dst[idx] = REC_INVALID;
        }
    }
    st_rec = wave_sum_u64(st_rec);
    if (lane == 0 && st_rec) atomicAdd(&o.counters[CNT_VALID_RECORDS], (unsigned long long) st_rec);
    if (STATS) {
        st_raw = wave_sum_u64(st_raw); st_slots = wave_sum_u64(st_slots); st_win = wave_sum_u64(st_win);
        if (lane == 0) {
            atomicAdd(&o.counters[CNT_RAW], (unsigned long long) st_raw);
            atomicAdd(&o.counters[CNT_SLOTS], (unsigned long long) st_slots);
            atomicAdd(&o.counters[CNT_WINDOWS], (unsigned long long) st_win);
        }
    }
}

// ------------------------------------------------------------------------------------------
// in-degree histogram for records produced elsewhere (sharded reduce)
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_count_targets(const uint32_t *__restrict__ rec_dst, uint64_t n_rec,
                                                        int32_t dst_begin, int32_t dst_end, uint32_t *__restrict__ indeg) {
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n_rec; i += (uint64_t) gridDim.x * blockDim.x) {
        const uint32_t c = rec_dst[i];
        if (c == REC_INVALID) continue;
        int C = (int) c;
        if (C >= dst_begin && C < dst_end) atomicAdd(&indeg[C - dst_begin], 1u);
    }
}

// ------------------------------------------------------------------------------------------
// exclusive scan of uint32 (n up to 2^31): reduce tiles, scan tile sums, scan tiles
// ------------------------------------------------------------------------------------------
constexpr int SCAN_BLOCK = 256;
constexpr int SCAN_ITEMS = 8;
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
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        uint64_t i = base + (uint64_t) k * SCAN_BLOCK + threadIdx.x;
        if (i < n) s += in[i];
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

// out[i] = exclusive prefix (uint32; the host checks the 64-bit total fits); out[n] = total
__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_tiles(const uint32_t *__restrict__ in, uint64_t n,
                                                            const uint64_t *__restrict__ tile_sums, uint32_t *__restrict__ out) {
    __shared__ uint32_t lds[8];
    const uint64_t base = (uint64_t) blockIdx.x * SCAN_TILE + (uint64_t) threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) { uint64_t i = base + k; v[k] = i < n ? in[i] : 0u; s += v[k]; }
    uint32_t tot;
    uint32_t ex = block_exclusive_scan(s, &tot, lds) + (uint32_t) tile_sums[blockIdx.x];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) { uint64_t i = base + k; if (i < n) out[i] = ex; ex += v[k]; }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) out[n] = (uint32_t) tile_sums[gridDim.x];
}

// ------------------------------------------------------------------------------------------
// k_scatter_by_target : records -> segments; `cursor` starts as a copy of the in-degrees
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_scatter_by_target(const uint32_t *__restrict__ rec_dst, const uint32_t *__restrict__ rec_src,
                                                            const uint32_t *__restrict__ rec_ol, const unsigned long long *__restrict__ n_rec_ptr,
                                                            uint64_t n_rec_max, int32_t dst_begin, int32_t dst_end,
                                                            const uint32_t *__restrict__ rowptr, uint32_t *__restrict__ cursor,
                                                            uint32_t *__restrict__ seg_src, uint32_t *__restrict__ seg_ol) {
    uint64_t n_rec = n_rec_ptr ? (uint64_t) *n_rec_ptr : n_rec_max;
    if (n_rec > n_rec_max) n_rec = n_rec_max;
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n_rec; i += (uint64_t) gridDim.x * blockDim.x) {
        const uint32_t c = rec_dst[i];
        if (c == REC_INVALID) continue;
        int C = (int) c;
        if (C < dst_begin || C >= dst_end) continue;
        C -= dst_begin;
        uint32_t pos = rowptr[C] + (atomicSub(&cursor[C], 1u) - 1u);
        seg_src[pos] = rec_src[i];
        seg_ol[pos] = rec_ol[i];
    }
}

// ------------------------------------------------------------------------------------------
// k_reduce_targets : one thread per target C, in-place in C's segment
//   processing order == the reference's with --threads=1: small overlaps first (they exist before
//   the reversal at L == rsoemo, GraphCreatorPrefSuf.cpp:288-296), then big overlaps by
//   (L ascending, source id ascending) (:94-100, :369).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t seg_key(uint32_t src, uint32_t ol) {
    return ((uint64_t) (ol_small(ol) ? 0u : 1u) << 63) | ((uint64_t) (uint32_t) ol_len(ol) << 32) | src;
}

// A[d .. d+nnt) == B[0 .. nnt) on 2-bit packed rows; replaces the Bitset temp/shift/mismatchBounded
// sequence of GraphCreatorPrefSuf.cpp:434-451 (Bitset.cpp:116-163,879-909)
__device__ __forceinline__ bool eq_shifted(const uint32_t *__restrict__ rowA, const uint32_t *__restrict__ rowB,
                                           int d, int nnt, int stride) {
    const int bit = 2 * d, q = bit >> 5, r = bit & 31;
    const int nbits = 2 * nnt;
    const int nw = (nbits + 31) >> 5;
    for (int k = 0; k < nw; k++) {
        const uint32_t lo = rowA[q + k];
        const uint32_t hi = (r != 0 && q + k + 1 < stride) ? rowA[q + k + 1] : 0u;
        uint32_t x = funnel(lo, hi, r) ^ rowB[k];
        if (k == nw - 1 && (nbits & 31)) x &= (1u << (nbits & 31)) - 1u;
        if (x) return false;
    }
    return true;
}

template <bool STATS>
__global__ void __launch_bounds__(256)
k_reduce_targets(NodesDev nd, PrefSufCfg cfg, int32_t dst_begin, int32_t n_owned, const uint32_t *__restrict__ rowptr,
                 uint32_t *__restrict__ seg_src, uint32_t *__restrict__ seg_ol, uint32_t *__restrict__ out_cnt,
                 uint32_t *__restrict__ outdeg, unsigned long long *__restrict__ counters) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t st_listed = 0, st_cmp = 0, st_rem = 0;
    uint32_t nlive = 0, k = 0;
    if (t < n_owned) {
        const uint32_t beg = rowptr[t];
        k = rowptr[t + 1] - beg;
        uint32_t *ss = seg_src + beg, *so = seg_ol + beg;
        // insertion sort by processing order
        for (uint32_t i = 1; i < k; i++) {
            const uint32_t xs = ss[i], xo = so[i];
            const uint64_t kx = seg_key(xs, xo);
            uint32_t j = i;
            while (j > 0 && seg_key(ss[j - 1], so[j - 1]) > kx) { ss[j] = ss[j - 1]; so[j] = so[j - 1]; j--; }
            ss[j] = xs; so[j] = xo;
        }
        for (uint32_t i = 0; i < k; i++) {
            const uint32_t B = ss[i], ol = so[i];
            const int off = ol_off(ol), L = ol_len(ol);
            if (ol_small(ol)) {
                // Graph::retainOnlySmallestOffset after the reversal (Graph.cpp:348-387): one entry per source
                bool found = false;
                for (uint32_t u = 0; u < nlive; u++) {
                    if (ss[u] == B) { if (off < ol_off(so[u])) so[u] = ol; found = true; break; }
                }
                if (!found) { ss[nlive] = B; so[nlive] = ol; nlive++; }
            } else {
                if (STATS && off > 0) st_listed += nlive;
                const uint32_t *rowB = nd.words + (size_t) B * nd.stride;
                uint32_t w = 0;
                for (uint32_t u = 0; u < nlive; u++) {
                    const uint32_t A = ss[u], olA = so[u];
                    bool remove = (A == B);                                   // toRemove[suffId], :461-462
                    if (!remove && off > 0) {                                 // :406
                        const int d = ol_off(olA) - off;                      // offsetDiff, :417
                        if (d >= 0) {                                         // :420
                            if (STATS) st_cmp++;
                            // Read::getRightOffset(rA, rB, d) = |B| + d - |A| = L_B - L_A  (:429)
                            if (L - ol_len(olA) >= 0) {
                                const uint32_t *rowA = nd.words + (size_t) A * nd.stride;
                                if (eq_shifted(rowA, rowB, d, off, nd.stride)) { remove = true; if (STATS) st_rem++; }
                            }
                        }
                    }
                    if (!remove) { if (w != u) { ss[w] = A; so[w] = olA; } w++; }
                }
                ss[w] = B; so[w] = ol;                                        // pushDirectedEdge(prefId, suffId, offset), :477
                nlive = w + 1;
            }
        }
        out_cnt[t] = nlive;
        if (cfg.reversed) { if (nlive) outdeg[dst_begin + t] = nlive; }       // never-reversed quirk: rows are the targets
        else for (uint32_t u = 0; u < nlive; u++) atomicAdd(&outdeg[ss[u]], 1u);
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
                    const uint32_t *__restrict__ seg_src, const uint32_t *__restrict__ seg_ol, const uint32_t *__restrict__ out_cnt,
                    const uint32_t *__restrict__ out_rowptr, uint32_t *__restrict__ out_cursor, alga_edge_dev *__restrict__ edges) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_owned) return;
    const uint32_t beg = rowptr[t], cnt = out_cnt[t];
    const int C = dst_begin + t;
    for (uint32_t u = 0; u < cnt; u++) {
        const int A = (int) seg_src[beg + u];
        const int off = ol_off(seg_ol[beg + u]);
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
// launch wrappers (host)
// ------------------------------------------------------------------------------------------
static inline unsigned grid_for(uint64_t n, int block) { return (unsigned) ((n + (uint64_t) block - 1) / (uint64_t) block); }

void launch_node_stats(const NodesDev &nd, unsigned long long *counters, int *max_len, hipStream_t s) {
    if (nd.n <= 0) return;
    hipLaunchKernelGGL(k_node_stats, dim3(grid_for((uint64_t) nd.n, 256)), dim3(256), 0, s, nd, counters, max_len);
}

void launch_seed_build(const NodesDev &nd, const PrefSufCfg &cfg, unsigned long long *table, uint32_t mask, hipStream_t s) {
    if (nd.n <= 0) return;
    hipLaunchKernelGGL(k_seed_build, dim3(grid_for((uint64_t) nd.n, 256)), dim3(256), 0, s, nd, cfg, table, mask);
}

static uint64_t probe_blocks(int n_cu, uint64_t n_src) {
    // persistent grid: 8 workgroups of 4 waves per CU fill the 32 wave slots of a CU
    return std::max<uint64_t>(1, std::min<uint64_t>((n_src + PROBE_WAVES - 1) / PROBE_WAVES, (uint64_t) std::max(1, n_cu) * 8));
}

void launch_probe(const NodesDev &nd, const PrefSufCfg &cfg, const unsigned long long *table, uint32_t mask,
                  int32_t src_begin, int32_t src_end, uint32_t *rec_dst, uint32_t *rec_src, uint32_t *rec_ol, uint64_t rec_cap,
                  uint32_t *indeg, int32_t dst_begin, int32_t dst_end, unsigned long long *counters, int n_cu, hipStream_t s) {
    const int64_t ns = (int64_t) src_end - src_begin;
    if (ns <= 0) return;
    dim3 grid((unsigned) probe_blocks(n_cu, (uint64_t) ns)), block(PROBE_WAVES * 64);
    ProbeOut o{rec_dst, rec_src, rec_ol, rec_cap, indeg, dst_begin, dst_end, counters};
    if (cfg.stats) hipLaunchKernelGGL(k_probe_sources<true>, grid, block, 0, s, nd, cfg, table, mask, src_begin, src_end, o);
    else           hipLaunchKernelGGL(k_probe_sources<false>, grid, block, 0, s, nd, cfg, table, mask, src_begin, src_end, o);
}

uint64_t probe_record_slack(int n_cu, uint64_t n_src) {  // worst-case invalid padding of one launch
    return probe_blocks(n_cu, n_src) * PROBE_WAVES * REC_CHUNK;
}

void launch_count_targets(const uint32_t *rec_dst, uint64_t n_rec, int32_t dst_begin, int32_t dst_end, uint32_t *indeg, hipStream_t s) {
    if (n_rec == 0) return;
    unsigned g = grid_for(n_rec, 256); if (g > 8192) g = 8192;
    hipLaunchKernelGGL(k_count_targets, dim3(g), dim3(256), 0, s, rec_dst, n_rec, dst_begin, dst_end, indeg);
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

void launch_scatter_by_target(const uint32_t *rec_dst, const uint32_t *rec_src, const uint32_t *rec_ol,
                              const unsigned long long *n_rec_ptr, uint64_t n_rec_max, int32_t dst_begin, int32_t dst_end,
                              const uint32_t *rowptr, uint32_t *cursor, uint32_t *seg_src, uint32_t *seg_ol, hipStream_t s) {
    if (n_rec_max == 0) return;
    unsigned g = grid_for(n_rec_max, 256); if (g > 16384) g = 16384;
    hipLaunchKernelGGL(k_scatter_by_target, dim3(g), dim3(256), 0, s, rec_dst, rec_src, rec_ol, n_rec_ptr, n_rec_max, dst_begin, dst_end,
                       rowptr, cursor, seg_src, seg_ol);
}

void launch_reduce_targets(const NodesDev &nd, const PrefSufCfg &cfg, int32_t dst_begin, int32_t n_owned, const uint32_t *rowptr,
                           uint32_t *seg_src, uint32_t *seg_ol, uint32_t *out_cnt, uint32_t *outdeg, unsigned long long *counters, hipStream_t s) {
    if (n_owned <= 0) return;
    dim3 grid(grid_for((uint64_t) n_owned, 256)), block(256);
    if (cfg.stats)
        hipLaunchKernelGGL(k_reduce_targets<true>, grid, block, 0, s, nd, cfg, dst_begin, n_owned, rowptr, seg_src, seg_ol, out_cnt, outdeg, counters);
    else
        hipLaunchKernelGGL(k_reduce_targets<false>, grid, block, 0, s, nd, cfg, dst_begin, n_owned, rowptr, seg_src, seg_ol, out_cnt, outdeg, counters);
}

void launch_scatter_by_source(const PrefSufCfg &cfg, int32_t dst_begin, int32_t n_owned, const uint32_t *rowptr, const uint32_t *seg_src,
                              const uint32_t *seg_ol, const uint32_t *out_cnt, const uint32_t *out_rowptr, uint32_t *out_cursor,
                              alga_edge_dev *edges, hipStream_t s) {
    if (n_owned <= 0) return;
    hipLaunchKernelGGL(k_scatter_by_source, dim3(grid_for((uint64_t) n_owned, 256)), dim3(256), 0, s, cfg, dst_begin, n_owned, rowptr, seg_src,
                       seg_ol, out_cnt, out_rowptr, out_cursor, edges);
}

void launch_sort_rows(int32_t n, const uint32_t *out_rowptr, alga_edge_dev *edges, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_sort_rows, dim3(grid_for((uint64_t) n, 256)), dim3(256), 0, s, n, out_rowptr, edges);
}

} // namespace alga
