// alga_amd/csrc/prefsuf_device.h -- device-side building blocks shared by the probe kernels
// (prefsuf_kernels.hip: bucketised seed table; prefsuf_minimizer.hip: minimizer index).
#pragma once
#include <hip/hip_runtime.h>
#include "prefsuf_common.h"

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

// ---- DPP cross-lane steps (no LDS crossbar round trip, unlike ds_bpermute-based __shfl) ----------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint64_t dpp_or_zero_u64(uint64_t v) {   // lanes without a valid source read 0
    const uint32_t lo = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) (uint32_t) v, CTRL, ROW_MASK, 0xF, false);
    const uint32_t hi = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) (uint32_t) (v >> 32), CTRL, ROW_MASK, 0xF, false);
    return ((uint64_t) hi << 32) | lo;
}

// max over the 64 lanes, returned uniformly (scalar registers): row_shr 1,2,4,8 scan inside each 16-lane row,
// row_bcast15 / row_bcast31 carry the row totals upward, lane 63 holds the result
__device__ __forceinline__ uint64_t wave_max_u64_dpp(uint64_t v) {
    uint64_t t;
    t = dpp_or_zero_u64<0x111, 0xF>(v); v = t > v ? t : v;
    t = dpp_or_zero_u64<0x112, 0xF>(v); v = t > v ? t : v;
    t = dpp_or_zero_u64<0x114, 0xF>(v); v = t > v ? t : v;
    t = dpp_or_zero_u64<0x118, 0xF>(v); v = t > v ? t : v;
    t = dpp_or_zero_u64<0x142, 0xA>(v); v = t > v ? t : v;
    t = dpp_or_zero_u64<0x143, 0xC>(v); v = t > v ? t : v;
    const uint32_t lo = (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) v, 63);
    const uint32_t hi = (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) (v >> 32), 63);
    return ((uint64_t) hi << 32) | lo;
}

// OR over each group of four adjacent lanes (quad_perm [1,0,3,2] then [2,3,0,1])
__device__ __forceinline__ uint32_t quad_or(uint32_t v) {
    v |= (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0xB1, 0xF, 0xF, true);
    v |= (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x4E, 0xF, 0xF, true);
    return v;
}

// value of the lane N positions up / down inside the 16-lane row (0 past the row's end); all lanes involved must be active
template <int N> __device__ __forceinline__ int row_from_above(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x100 + N, 0xF, 0xF, true); }   // row_shl:N
template <int N> __device__ __forceinline__ int row_from_below(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x110 + N, 0xF, 0xF, true); }   // row_shr:N

// 32 bits of a bit string starting at bit r of lo (v_alignbit_b32)
__device__ __forceinline__ uint32_t funnel(uint32_t lo, uint32_t hi, int r) { return __funnelshift_r(lo, hi, r); }

__device__ __forceinline__ void wave_lds_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }

// ------------------------------------------------------------------------------------------
// k_probe_sources
// ------------------------------------------------------------------------------------------
constexpr int PROBE_WAVES = 4;        // waves per workgroup
constexpr int STAGE_WORDS = 52;       // staged tail (<= 33 words) + slack for unconditional wide compares
constexpr int CANDMAX = 128;          // per-wave candidate buffer
constexpr int WBUF = 256;             // per-wave LDS record buffer (records)
constexpr int WFLUSH = 128;           // flush once this many are buffered
constexpr int REC_CHUNK = 1024;       // records reserved per global atomic (per-target pipeline: ~11 records per source)
constexpr int REC_CHUNK_LOCAL = 128;  // same, source-side reduction (~1 record per source)
constexpr int WFLUSH_LOCAL = 64;
constexpr int WBUF_LOCAL = 128;       // per-wave LDS record buffer of the source-side form

// keep a >= b >= c = the three largest keys seen (branch-free: the three keys must stay in registers)
__device__ __forceinline__ void top3_insert(uint64_t &a, uint64_t &b, uint64_t &c, uint64_t k) {
    uint64_t t = k > a ? k : a; k = k > a ? a : k; a = t;
    t = k > b ? k : b; k = k > b ? b : k; b = t;
    c = k > c ? k : c;
}

// the three largest keys over the wave's per-lane (k0 >= k1 >= k2) triples, uniform; 0 = none.  Convergent.
__device__ __forceinline__ void wave_top3(uint64_t k0, uint64_t k1, uint64_t k2, uint64_t &win0, uint64_t &win1, uint64_t &win2) {
    win0 = win1 = win2 = 0;
    uint64_t m = wave_max_u64_dpp(k0);
    if (m) {
        if (k0 == m) { k0 = k1; k1 = k2; k2 = 0; }
        win0 = m;
        m = wave_max_u64_dpp(k0);
        if (m) {
            if (k0 == m) { k0 = k1; k1 = k2; k2 = 0; }
            win1 = m;
            win2 = wave_max_u64_dpp(k0);
        }
    }
}

// value of the quad's lane 0 / of the next lane of the quad (lane 3 reads itself); all lanes of the quad must be active
__device__ __forceinline__ int quad_bcast0(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x00, 0xF, 0xF, true); }
__device__ __forceinline__ uint32_t quad_next(uint32_t v) { return (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0xF9, 0xF, 0xF, true); }

struct ProbeOut {
    uint32_t *__restrict__ rec_dst;
    unsigned long long *__restrict__ rec_val;
    uint64_t rec_cap;
    unsigned long long *__restrict__ counters;
    // source-side form: per source of [src_base, ...) its out-degree and, for the one-edge fast path, the edge itself
    // ((dst << 32) | offset; LOCAL_FIRST_NONE = "the edges are in the record list")
    uint32_t *__restrict__ deg = nullptr;
    unsigned long long *__restrict__ first = nullptr;
    int32_t src_base = 0;
    unsigned long long *__restrict__ second = nullptr;     // pair kernel of the clustered probe: the other edge of a two-edge source ...
    uint32_t slot_stride = 0;                              // ... and, where != 0 (round 5), the third .. eighth at second[(k - 2) * slot_stride + i]
    // sources with more raw overlaps than a wave's LDS holds (repeats): the first pass lists them, a second pass (BIG
    // instantiation of the kernel) walks the list with the items in a global slice per wave
    int32_t *__restrict__ big_list = nullptr;
    uint32_t big_list_cap = 0, big_count = 0;
    uint32_t *__restrict__ bigC = nullptr;
    uint32_t *__restrict__ bigM = nullptr;
    uint4 *__restrict__ bigO = nullptr;
    uint32_t big_cap = 0;                                    // items per wave slice
};
constexpr unsigned long long LOCAL_FIRST_NONE = ~0ull;

__device__ __forceinline__ void store_record(const ProbeOut &o, uint64_t idx, uint32_t C, unsigned long long val) {
    if (idx < o.rec_cap) { o.rec_dst[idx] = C; o.rec_val[idx] = val; }
}

struct WaveLds {           // per-wave LDS views
    uint32_t *sb;          // staged tail of the source
    uint32_t *candC; uint32_t *candW; uint32_t *candN;
    uint32_t *recC; unsigned long long *recV; uint32_t *recN;
    uint32_t *actB = nullptr; uint32_t *actT = nullptr;      // compacted windows that passed the filter: bucket, tag | lane << 23
};

// Convergent: all 64 lanes.  Moves the wave's LDS record buffer to the global record list.
//   Chunks of the list are reserved with ONE global atomic each (a returning atomic on a single address
//   sustains only ~88 ops/us chip-wide: per-record or per-source reservations cost tens of ms).  A flush fills the
//   open chunk to the brim before it reserves the next one, so only the last chunk of a wave carries padding.
//   chunk_fill == CHUNK means "no chunk reserved yet".
template <int CHUNK, int WB = WBUF>
__device__ __forceinline__ void flush_records(const ProbeOut &o, const WaveLds &w, uint64_t &chunk_base, int &chunk_fill) {
    const int lane = lane_id();
    wave_lds_fence();
    int n = (int) __builtin_amdgcn_readfirstlane((int) *w.recN);
    if (n > WB) n = WB;                          // the excess went out through the direct path
    if (n == 0) return;
    int done = 0;
    while (done < n) {                           // wave-uniform
        if (chunk_fill == CHUNK) {
            uint64_t base = 0;
            if (lane == 0) base = atomicAdd(&o.counters[CNT_RECORDS], (unsigned long long) CHUNK);
            chunk_base = ((uint64_t) (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) (base >> 32)) << 32) |
                         (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) base);
            chunk_fill = 0;
        }
        const int take = (n - done) < (CHUNK - chunk_fill) ? (n - done) : (CHUNK - chunk_fill);
        for (int i = lane; i < take; i += 64) store_record(o, chunk_base + (uint64_t) (chunk_fill + i), w.recC[done + i], w.recV[done + i]);
        chunk_fill += take;
        done += take;
    }
    wave_lds_fence();
    if (lane == 0) *w.recN = 0;
    wave_lds_fence();
}

// Convergent: invalid markers in the unused tail of the wave's last chunk (end of the kernel).
template <int CHUNK>
__device__ __forceinline__ void close_chunk(const ProbeOut &o, uint64_t chunk_base, int chunk_fill) {
    for (int i = chunk_fill + lane_id(); i < CHUNK; i += 64) {
        const uint64_t idx = chunk_base + (uint64_t) i;
        if (idx < o.rec_cap) o.rec_dst[idx] = REC_INVALID;
    }
}

// NQ > 0: rows are 16-byte aligned, hold >= NQ uint4 and every compared prefix fits NQ uint4:
//         straight-line wide loads, all issued before the first use.
// NQ == 0: generic word loop (any stride / read length).
template <int NQ>
__device__ __forceinline__ bool verify_overlap(const NodesDev &nd, const uint32_t *sb, int C, int q, int r, int L) {
    const int nwL = (2 * L + 31) >> 5;
    const uint32_t lastmask = (2 * L & 31) ? ((1u << (2 * L & 31)) - 1u) : 0xFFFFFFFFu;
    uint32_t diff = 0;
    if constexpr (NQ > 0) {
        const uint4 *rc4 = reinterpret_cast<const uint4 *>(nd.words + (size_t) C * nd.stride);
        uint4 c[NQ];
#pragma unroll
        for (int k4 = 0; k4 < NQ; k4++) c[k4] = rc4[k4];
#pragma unroll
        for (int k4 = 0; k4 < NQ; k4++) {
            const uint32_t cw[4] = {c[k4].x, c[k4].y, c[k4].z, c[k4].w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int k = 4 * k4 + j;
                const uint32_t m = k < nwL - 1 ? 0xFFFFFFFFu : (k == nwL - 1 ? lastmask : 0u);
                diff |= (funnel(sb[q + k], sb[q + k + 1], r) ^ cw[j]) & m;
            }
        }
    } else {
        const uint32_t *rc = nd.words + (size_t) C * nd.stride;
        for (int k = 0; k < nwL; k++) {
            uint32_t x = funnel(sb[q + k], sb[q + k + 1], r) ^ rc[k];
            if (k == nwL - 1) x &= lastmask;
            diff |= x;
        }
    }
    return diff == 0;
}

// ------------------------------------------------------------------------------------------
// Source-side transitive reduction (executable statement: tests/source_side_rule.py)
//
// The reference decides per TARGET C which in-overlaps (A -> C) are implied by a big overlap (B -> C), replaying the
// pushes in (L, source) order (GraphCreatorPrefSuf.cpp:397-483).  Every such via B is itself a raw out-overlap of A
// (A[d_B..] == B[0..], |A| - d_B >= L_AC >= min_overlap), so the wave that has just verified ALL raw overlaps of A
// holds everything needed: with overhang(X) = X[L_AX ..) -- what X adds to the right of A's end --
//     via(B, C)  <=>  d_B < d_C, B != C, alignFrom[B], |B| - (d_C - d_B) >= max(rsoemo, Lmin),
//                     overhang(B) is a prefix of overhang(C), and (rho_B > 0 or B > A)   [push order at equal L]
// Items live in LDS: C, meta = d | len_C << 9 | alignFrom << 18, overhang (<= 63 nt: 128 bits, filled by the quad that verified
// the overlap from the row words it already holds).
// Fast path (error-free reads): one item per offset; each item is tested against its nearest predecessor only and
// anything that does not resolve that way sends the whole source down the generic all-pairs path.
// Host-checked preconditions (engine.hip local_ok): max_len - Lmin <= 63 (SW = 1) or <= 127 (SW = 2), max_len <= cap,
// alignFrom => alignTo, Lmin <= rsoemo <= Lcap.
// ------------------------------------------------------------------------------------------
constexpr int ITEMMAX = 128;              // raw overlaps of one source held in LDS; sources with more go to the second pass
constexpr int ITEM_BIG_MAX = 4096;        // largest per-wave global item slice of the second pass; beyond it: per-target pipeline
constexpr uint32_t ITEM_FROM = 1u << 18;
constexpr uint32_t ITEM_KEPT = 1u << 19;      // all-pairs evaluation: survives the per-source cap / is a big overlap
constexpr uint32_t ITEM_REMOVED = 1u << 20;   // all-pairs evaluation: implied by a via or superseded

struct ItemLds { uint32_t *C; uint32_t *M; uint4 *O; uint8_t *T; uint32_t *N; };

__device__ __forceinline__ uint64_t wave_or_u64_dpp(uint64_t v) {
    v |= dpp_or_zero_u64<0x111, 0xF>(v);
    v |= dpp_or_zero_u64<0x112, 0xF>(v);
    v |= dpp_or_zero_u64<0x114, 0xF>(v);
    v |= dpp_or_zero_u64<0x118, 0xF>(v);
    v |= dpp_or_zero_u64<0x142, 0xA>(v);
    v |= dpp_or_zero_u64<0x143, 0xC>(v);
    const uint32_t lo = (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) v, 63);
    const uint32_t hi = (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) (v >> 32), 63);
    return ((uint64_t) hi << 32) | lo;
}

__device__ __forceinline__ uint32_t low_bits32(int bits) { return bits >= 32 ? 0xFFFFFFFFu : (bits <= 0 ? 0u : ((1u << bits) - 1u)); }

// SW = 64-bit words of an offset mask = uint4 words of an overhang: 1 covers max_len - Lmin <= 63 (100-150 bp reads with the
// reference's default scale), 2 covers <= 127 (250 bp).
template <int SW> struct Ovh { uint32_t w[4 * SW]; };

template <int SW>
__device__ __forceinline__ Ovh<SW> load_ovh(const uint4 *O, int idx) {
    Ovh<SW> r;
#pragma unroll
    for (int q = 0; q < SW; q++) { const uint4 v = O[idx * SW + q]; r.w[4 * q] = v.x; r.w[4 * q + 1] = v.y; r.w[4 * q + 2] = v.z; r.w[4 * q + 3] = v.w; }
    return r;
}

// is item j (= B) a via that removes item i (= C) of source A ?
template <int SW>
__device__ __forceinline__ bool via_ok(int A, int lenA, int Lbig, uint32_t Cj, uint32_t mj, const Ovh<SW> &oj, uint32_t Ci, int di, int rho_i,
                                       const Ovh<SW> &oi) {
    const int dj = (int) (mj & 511u), lenj = (int) ((mj >> 9) & 511u);
    const int rho_j = lenj - (lenA - dj);
    const int Lv = lenj - (di - dj);                       // length of the overlap B -> C
    const bool ok = (mj & ITEM_FROM) != 0 && Cj != Ci && dj < di && Lv >= Lbig && rho_j <= rho_i && (rho_j > 0 || (int) Cj > A);
    const int nb = 2 * rho_j;
    uint32_t diff = 0;
#pragma unroll
    for (int k = 0; k < 4 * SW; k++) diff |= (oi.w[k] ^ oj.w[k]) & low_bits32(nb - 32 * k);
    return ok && diff == 0;
}

// overhang of item `slot` from X's row in global memory (cold paths of the probe; the wide verification fills it from registers).
// Bits past the overhang's own length are never compared (via_ok masks with the via's length <= the candidate's), so
// nothing here is masked.
template <int SW>
__device__ __forceinline__ void item_overhang_global(const NodesDev &nd, const ItemLds &it, int item_cap, int slot, int C, int L, int lenC) {
    if (slot < 0 || slot >= item_cap) return;
    const uint32_t *row = nd.words + (size_t) C * nd.stride;
    const int ws = (2 * L) >> 5, r = (2 * L) & 31, lastw = (2 * lenC - 1) >> 5;
    uint32_t x[4 * SW + 1];
#pragma unroll
    for (int k = 0; k <= 4 * SW; k++) x[k] = (ws + k <= lastw) ? row[ws + k] : 0u;
#pragma unroll
    for (int q = 0; q < SW; q++)
        it.O[slot * SW + q] = make_uint4(funnel(x[4 * q], x[4 * q + 1], r), funnel(x[4 * q + 1], x[4 * q + 2], r), funnel(x[4 * q + 2], x[4 * q + 3], r),
                                         funnel(x[4 * q + 3], x[4 * q + 4], r));
}

// set of offsets (bit d), SW x 64 bits
template <int SW> struct OffMask { uint64_t w[SW]; };
template <int SW> __device__ __forceinline__ OffMask<SW> offmask_below(const OffMask<SW> &m, int d) {      // bits < d
    OffMask<SW> r;
#pragma unroll
    for (int q = 0; q < SW; q++) {
        const int lo = d - 64 * q;
        r.w[q] = lo >= 64 ? m.w[q] : (lo <= 0 ? 0ull : (m.w[q] & ((1ull << lo) - 1ull)));
    }
    return r;
}
template <int SW> __device__ __forceinline__ OffMask<SW> offmask_from(const OffMask<SW> &m, int d) {       // bits >= d
    OffMask<SW> r;
#pragma unroll
    for (int q = 0; q < SW; q++) {
        const int lo = d - 64 * q;
        r.w[q] = lo >= 64 ? 0ull : (lo <= 0 ? m.w[q] : (m.w[q] & ~((1ull << lo) - 1ull)));
    }
    return r;
}
template <int SW> __device__ __forceinline__ int offmask_count(const OffMask<SW> &m) {
    int c = 0;
#pragma unroll
    for (int q = 0; q < SW; q++) c += __popcll(m.w[q]);
    return c;
}
template <int SW> __device__ __forceinline__ int offmask_highest(const OffMask<SW> &m) {                   // -1 = empty
    int r = -1;
#pragma unroll
    for (int q = 0; q < SW; q++) if (m.w[q]) r = 64 * q + 63 - __clzll((long long) m.w[q]);
    return r;
}

// Convergent.  n = items of source A in `it` (<= ITEMMAX), overhangs filled.
template <bool STATS, int WB, int SW>
__device__ __forceinline__ void local_reduce(const NodesDev &nd, const PrefSufCfg &cfg, const ItemLds &it, const WaveLds &w, const ProbeOut &o,
                                             int A, int lenA, int n, uint64_t &st_rec, uint64_t &st_cmp, uint64_t &st_generic) {
    const int lane = lane_id();
    const int Lbig = cfg.rsoemo > cfg.Lmin ? cfg.rsoemo : cfg.Lmin;
    auto push = [&](uint32_t C, int d) {
        const int L = lenA - d;
        const unsigned long long val = ((unsigned long long) ol_pack(d, L, L < cfg.rsoemo) << 32) | (uint32_t) A;
        const uint32_t i = atomicAdd(w.recN, 1u);
        if (i < (uint32_t) WB) { w.recC[i] = C; w.recV[i] = val; }
        else store_record(o, atomicAdd(&o.counters[CNT_RECORDS], 1ull), C, val);
        st_rec++;
    };
    wave_lds_fence();
    bool generic = n > 64;
    if (!generic) {
        const bool act = lane < n;
        uint32_t C = 0, m = 0;
        Ovh<SW> ov;
#pragma unroll
        for (int k = 0; k < 4 * SW; k++) ov.w[k] = 0u;
        int d = 0;
        if (act) { C = it.C[lane]; m = it.M[lane]; ov = load_ovh<SW>(it.O, lane); d = (int) (m & 511u); }
        OffMask<SW> occ;
#pragma unroll
        for (int q = 0; q < SW; q++) occ.w[q] = wave_or_u64_dpp((act && (d >> 6) == q) ? (1ull << (d & 63)) : 0ull);
        generic = offmask_count<SW>(occ) != n;             // two items at one offset
        if (!generic) {
            const OffMask<SW> below = offmask_below<SW>(occ, d);
            const int pd = act ? offmask_highest<SW>(below) : -1;
            const bool has_pred = pd >= 0;
            if (act) it.T[d] = (uint8_t) lane;
            wave_lds_fence();
            bool fail = false, removed = false;
            if (has_pred) {
                const int j = (int) it.T[pd];
                const uint32_t Cj = it.C[j], mj = it.M[j];
                const Ovh<SW> oj = load_ovh<SW>(it.O, j);
                const int rho = (int) ((m >> 9) & 511u) - (lenA - d);
                if (STATS) st_cmp++;
                removed = via_ok<SW>(A, lenA, Lbig, Cj, mj, oj, C, d, rho, ov);
                // Not removed by the nearest predecessor: if even the longest read placed there could not reach C with a big
                // overlap, no earlier item can (they all start further left) and the item stands; anything else is undecided.
                // (a lone predecessor that is no via settles it: nobody else sits before this item)
                fail = !removed && offmask_count<SW>(below) >= 2 && (cfg.Lcap - 1) - (d - (int) (mj & 511u)) >= Lbig;
            }
            generic = __ballot(fail) != 0ull;
            if (!generic) {
                uint64_t surv = __ballot(act && !removed);
                if ((surv & (surv - 1)) == 0ull) {
                    // One item stands: the one without a predecessor.  It has the longest overlap of the source, so it is big or
                    // the largest of the small ones and passes the cap of 3 (GraphCreatorPrefSuf.cpp:397-401).  One edge: it goes
                    // straight to the source's slot, no record list, no sort.
                    if ((surv >> lane) & 1ull) {
                        o.first[A - o.src_base] = ((unsigned long long) C << 32) | (uint32_t) d;
                        o.deg[A - o.src_base] = 1u;
                        st_rec++;
                    }
                } else {
                    // Several stand (gaps too long for a big via).  Per-source cap: with one item per offset the 3 largest small
                    // (L, C) keys are the 3 small items with the smallest offsets.
                    const int ds0 = lenA - cfg.rsoemo + 1;                           // first offset of a small overlap
                    const bool kept = act && (d < ds0 || offmask_count<SW>(offmask_from<SW>(below, ds0)) < 3);
                    surv &= __ballot(kept);
                    // the same target at a smaller offset supersedes (Graph.cpp:348-387, GraphCreatorPrefSuf.cpp:461-462), whatever
                    // became of that instance: test each survivor against all kept items
                    uint64_t todo = surv;
                    while (todo) {                                                   // uniform
                        const int sl = __builtin_ctzll(todo);
                        todo &= todo - 1;
                        const uint32_t Cs = (uint32_t) __builtin_amdgcn_readlane((int) C, sl);
                        const int ds = __builtin_amdgcn_readlane(d, sl);
                        if (__ballot(kept && C == Cs && d < ds) != 0ull) surv &= ~(1ull << sl);
                    }
                    if ((surv >> lane) & 1ull) push(C, d);
                    if (lane == 0) { o.deg[A - o.src_base] = (uint32_t) __popcll(surv); o.first[A - o.src_base] = LOCAL_FIRST_NONE; }
                }
            }
        }
    }
    if (generic) {
        // All pairs.  Lanes hold the possible vias j (64 at a time, their overhang pre-masked and its mask in registers), the
        // kept items i pass by one at a time (uniform LDS / global reads); a hit marks i removed in its meta word.
        if (STATS && lane == 0) st_generic++;
        uint64_t k0 = 0, k1 = 0, k2 = 0, win0, win1, win2;
        for (int i = lane; i < n; i += 64) {
            const int L = lenA - (int) (it.M[i] & 511u);
            if (L < cfg.rsoemo) top3_insert(k0, k1, k2, ((uint64_t) (uint32_t) L << 32) | it.C[i]);
        }
        wave_top3(k0, k1, k2, win0, win1, win2);
        for (int i = lane; i < n; i += 64) {               // kept: big, or among the source's three largest small (L, C)
            const uint32_t C = it.C[i], m = it.M[i];
            const int L = lenA - (int) (m & 511u);
            const uint64_t key = ((uint64_t) (uint32_t) L << 32) | C;
            it.M[i] = (m & ~(ITEM_KEPT | ITEM_REMOVED)) | ((L >= cfg.rsoemo || key == win0 || key == win1 || key == win2) ? ITEM_KEPT : 0u);
        }
        __threadfence_block();
        wave_lds_fence();
        // per-lane data of a possible via j: pre-masked overhang, its mask, reach
        auto via_fields = [&](int j, bool actj, uint32_t &Cj, Ovh<SW> &oj, Ovh<SW> &kj, int &dj, int &rho_j, int &dend_j, bool &ok_j, bool &kept_j) {
            uint32_t mj = 0;
            Cj = 0;
#pragma unroll
            for (int k = 0; k < 4 * SW; k++) { oj.w[k] = 0u; kj.w[k] = 0u; }
            if (actj) { Cj = it.C[j]; mj = it.M[j]; oj = load_ovh<SW>(it.O, j); }
            dj = (int) (mj & 511u);
            const int lenj = (int) ((mj >> 9) & 511u);
            rho_j = lenj - (lenA - dj);
            dend_j = dj + lenj - Lbig;                     // B -> C is big  <=>  d_C <= d_B + |B| - Lbig
            ok_j = actj && (mj & ITEM_FROM) != 0 && (rho_j > 0 || (int) Cj > A);
            kept_j = actj && (mj & ITEM_KEPT) != 0;
#pragma unroll
            for (int k = 0; k < 4 * SW; k++) { kj.w[k] = low_bits32(2 * rho_j - 32 * k); oj.w[k] &= kj.w[k]; }
        };
        if (n <= 64) {
            // Few items (the usual case with sequencing errors): the 64 lanes hold (item i, via j) PAIRS, 64 / npad items per step with
            // npad = n rounded up to a power of two, and a hit sets the item's removed bit with an LDS atomic.
            const int lg = n <= 2 ? 1 : 32 - __clz(n - 1);
            const int npad = 1 << lg, per_step = 64 >> lg;
            const int j = lane & (npad - 1), il = lane >> lg;
            const bool actj = j < n;
            uint32_t Cj; Ovh<SW> oj, kj; int dj, rho_j, dend_j; bool ok_j, kept_j;
            via_fields(j, actj, Cj, oj, kj, dj, rho_j, dend_j, ok_j, kept_j);
            for (int ib = 0; ib < n; ib += per_step) {
                const int i = ib + il;
                if (i < n && actj) {
                    const uint32_t mi = it.M[i];
                    if ((mi & ITEM_KEPT) != 0) {
                        const uint32_t Ci = it.C[i];
                        const Ovh<SW> oi = load_ovh<SW>(it.O, i);
                        const int di = (int) (mi & 511u);
                        const int rho_i = (int) ((mi >> 9) & 511u) - (lenA - di);
                        uint32_t diff = 0;
#pragma unroll
                        for (int k = 0; k < 4 * SW; k++) diff |= (oi.w[k] & kj.w[k]) ^ oj.w[k];
                        const bool via = ok_j && Cj != Ci && dj < di && di <= dend_j && rho_j <= rho_i && diff == 0;
                        const bool sup = kept_j && Cj == Ci && dj < di;
                        if (STATS && Cj != Ci && dj < di) st_cmp++;
                        if (via || sup) atomicOr(&it.M[i], ITEM_REMOVED);
                    }
                }
            }
            __threadfence_block();
            wave_lds_fence();
        } else
        for (int jb = 0; jb < n; jb += 64) {
            const int j = jb + lane;
            const bool actj = j < n;
            uint32_t Cj; Ovh<SW> oj, kj; int dj, rho_j, dend_j; bool ok_j, kept_j;
            via_fields(j, actj, Cj, oj, kj, dj, rho_j, dend_j, ok_j, kept_j);
            for (int i = 0; i < n; i++) {                  // uniform
                const uint32_t mi = (uint32_t) __builtin_amdgcn_readfirstlane((int) it.M[i]);     // scalar: the skip is a uniform branch
                if ((mi & (ITEM_KEPT | ITEM_REMOVED)) != ITEM_KEPT) continue;
                const uint32_t Ci = (uint32_t) __builtin_amdgcn_readfirstlane((int) it.C[i]);
                const Ovh<SW> oi = load_ovh<SW>(it.O, i);
                const int di = (int) (mi & 511u);
                const int rho_i = (int) ((mi >> 9) & 511u) - (lenA - di);
                uint32_t diff = 0;
#pragma unroll
                for (int k = 0; k < 4 * SW; k++) diff |= (oi.w[k] & kj.w[k]) ^ oj.w[k];
                const bool via = ok_j && Cj != Ci && dj < di && di <= dend_j && rho_j <= rho_i && diff == 0;
                const bool sup = kept_j && Cj == Ci && dj < di;      // same target at a smaller offset: Graph.cpp:348-387, :461-462
                if (STATS && actj && Cj != Ci && dj < di) st_cmp++;
                if (__ballot(via || sup) != 0ull) { if (lane == 0) it.M[i] = mi | ITEM_REMOVED; }
            }
            __threadfence_block();
            wave_lds_fence();
        }
        int n_out = 0;
        for (int base = 0; base < n; base += 64) {
            const int i = base + lane;
            bool out = false;
            if (i < n) {
                const uint32_t m = it.M[i];
                out = (m & (ITEM_KEPT | ITEM_REMOVED)) == ITEM_KEPT;
                if (out) push(it.C[i], (int) (m & 511u));
            }
            n_out += __popcll(__ballot(out));
        }
        if (lane == 0 && n_out) { o.deg[A - o.src_base] = (uint32_t) n_out; o.first[A - o.src_base] = LOCAL_FIRST_NONE; }
    }
    wave_lds_fence();
}

} // namespace alga
