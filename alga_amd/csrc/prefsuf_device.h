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
constexpr int REC_CHUNK = 1024;       // records reserved per global atomic

// keep a >= b >= c = the three largest keys seen (branch-free: the three keys must stay in registers)
__device__ __forceinline__ void top3_insert(uint64_t &a, uint64_t &b, uint64_t &c, uint64_t k) {
    uint64_t t = k > a ? k : a; k = k > a ? a : k; a = t;
    t = k > b ? k : b; k = k > b ? b : k; b = t;
    c = k > c ? k : c;
}

struct ProbeOut {
    uint32_t *__restrict__ rec_dst;
    unsigned long long *__restrict__ rec_val;
    uint64_t rec_cap;
    unsigned long long *__restrict__ counters;
};

__device__ __forceinline__ void store_record(const ProbeOut &o, uint64_t idx, uint32_t C, unsigned long long val) {
    if (idx < o.rec_cap) { o.rec_dst[idx] = C; o.rec_val[idx] = val; }
}

struct WaveLds {           // per-wave LDS views
    uint32_t *sb;          // staged tail of the source
    uint32_t *candC; uint32_t *candW; uint32_t *candN;
    uint32_t *recC; unsigned long long *recV; uint32_t *recN;
};

// Convergent: all 64 lanes.  Moves the wave's LDS record buffer to the global record list.
//   A chunk of the list is reserved with ONE global atomic (a returning atomic on a single address
//   sustains only ~88 ops/us chip-wide: per-record or per-source reservations cost tens of ms).
__device__ __forceinline__ void flush_records(const ProbeOut &o, const WaveLds &w, uint64_t &chunk_base, int &chunk_fill) {
    const int lane = lane_id();
    wave_lds_fence();
    int n = (int) __builtin_amdgcn_readfirstlane((int) *w.recN);
    if (n > WBUF) n = WBUF;                      // the excess went out through the direct path
    if (n == 0) return;
    if (chunk_fill + n > REC_CHUNK) {
        for (int i = chunk_fill + lane; i < REC_CHUNK; i += 64) {      // close the chunk: invalid markers in its tail
            const uint64_t idx = chunk_base + (uint64_t) i;
            if (idx < o.rec_cap) o.rec_dst[idx] = REC_INVALID;
        }
        uint64_t base = 0;
        if (lane == 0) base = atomicAdd(&o.counters[CNT_RECORDS], (unsigned long long) REC_CHUNK);
        chunk_base = ((uint64_t) (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) (base >> 32)) << 32) |
                     (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) base);
        chunk_fill = 0;
    }
    for (int i = lane; i < n; i += 64) store_record(o, chunk_base + (uint64_t) (chunk_fill + i), w.recC[i], w.recV[i]);
    chunk_fill += n;
    wave_lds_fence();
    if (lane == 0) *w.recN = 0;
    wave_lds_fence();
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


} // namespace alga
