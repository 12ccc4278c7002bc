// alga_amd/csrc/radix_sort.hip -- the engine's own stable LSD radix sort of (u32 key, u32 value) pairs for gfx950.
//
// What it replaces: the reference re-buckets every node once per overlap length (removeKmersFromBucketsJob / putKmersIntoBucketsJob,
// src/GraphCreators/GraphCreatorPrefSuf.cpp:317-332) and std::sorts its k-mer buckets (GraphCreatorKmerBased.cpp:94-106); this engine
// orders all nodes ONCE by their minimizer key (prefsuf_cluster.hip: launch_cluster_store) -- until round 4 with rocPRIM's onesweep sort
// (library code: 2.3 ms for 90.6 M pairs on 29 key bits, 0.25 of the HBM roofline).
//
// Design for CDNA4 (8 XCDs with a private L2 each, 160 KB of LDS per CU, wave64):
//   * passes of up to 10 bits; per pass THREE kernels and no spin-waiting anywhere (a decoupled look-back would chain every tile to
//     all tiles before it, which is exactly what the XCD-contiguous tile order below must not do):
//       k_rs_hist      a block walks a CHUNK of consecutive tiles, LDS histogram per tile, and writes per tile the exclusive prefix of its
//                      digit counts inside the chunk (one coalesced row of 2^bits words) + the chunk's totals;
//       k_rs_scan_*    exclusive scan of the chunk totals per digit (a block per 64 digits), then of the digit totals;
//       k_rs_scatter   a tile of 8192 pairs per block: stable rank of every item by WAVE-MATCH (the lanes of a wave that hold the same digit
//                      find each other through one ballot per digit bit; per-wave digit counters in LDS, touched by that wave alone: no
//                      atomics, no order dependence), a scan over waves and digits, then keys and values go THROUGH LDS in tile-sorted
//                      order, so that a block writes every digit's run as one contiguous piece.
//   * tiles are dealt to the XCDs in contiguous ranges (blockIdx % 8 = XCD under round-robin placement: block b takes tile
//     (b % 8) * per + b / 8).  The runs that consecutive tiles write for one digit are adjacent in memory; written from the same XCD at about
//     the same time they complete whole 128-byte lines in that XCD's L2 instead of leaving eight L2s with partial lines each.
//   * every decision is a function of the input order alone: the sort is stable and bit-for-bit reproducible.
// Algorithmic bytes per pass: 4 (histogram) + 8 + 8 per pair; the tile prefix rows add 4 * 2^bits per 8192 pairs (0.5 B per pair).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

#include "prefsuf_kernels.h"

namespace alga {

namespace {

constexpr int RS_THREADS = 512;                            // histogram and scan kernels
constexpr int RS_WAVES = RS_THREADS / 64;
constexpr int RS_IPT = 16;                                 // items per thread of the scatter kernel
constexpr int RS_TILE_MAX = 1024 * RS_IPT;
constexpr int RS_MAX_BITS = 10;
constexpr int RS_MAX_DIGITS = 1 << RS_MAX_BITS;
constexpr int RS_CHUNKS = 512;                             // blocks of the histogram pass (two per CU)

struct RsPlan {
    int passes = 0;
    int bits[5] = {0, 0, 0, 0, 0}, shift[5] = {0, 0, 0, 0, 0};
    uint32_t n_tiles = 0, chunks = 0, tiles_per_chunk = 0;
    int tile = 8192;
};

int g_rs_variant = 0;                                      // tuning only (engine option "rsort_variant"): 0 = tiles of 8192 pairs (512 threads), 1 = 16384 (1024 threads)

RsPlan rs_plan(uint64_t n, int begin_bit, int end_bit, int variant) {
    RsPlan p;
    p.tile = variant == 2 ? 4096 : (variant == 1 ? 16384 : 8192);          // (2: the 16-byte records of rsort_u64_pairs)
    const int RS_TILE = p.tile;
    const int total = end_bit - begin_bit;
    p.passes = (total + RS_MAX_BITS - 1) / RS_MAX_BITS;
    int at = begin_bit;
    for (int i = 0; i < p.passes; i++) {                   // as even as possible: 29 bits -> 10, 10, 9 (the wide digits first: fewer bits left for the last, widest-spread pass)
        const int left = end_bit - at, todo = p.passes - i;
        p.bits[i] = (left + todo - 1) / todo;
        p.shift[i] = at;
        at += p.bits[i];
    }
    p.n_tiles = (uint32_t) ((n + RS_TILE - 1) / RS_TILE);
    p.chunks = std::max<uint32_t>(1, std::min<uint32_t>(RS_CHUNKS, p.n_tiles));
    p.tiles_per_chunk = (p.n_tiles + p.chunks - 1) / std::max<uint32_t>(1, p.chunks);
    p.chunks = p.tiles_per_chunk ? (p.n_tiles + p.tiles_per_chunk - 1) / p.tiles_per_chunk : 1;
    return p;
}

// ---- histogram of a chunk of tiles -----------------------------------------------------------------------------------------------
// tile_pref[t][d] = number of items with digit d in the tiles of t's chunk before t;  chunk_tot[c][d] = items with digit d in chunk c
template <int TILE>
__global__ void __launch_bounds__(RS_THREADS) k_rs_hist(const uint32_t *__restrict__ keys, uint64_t n, int shift, int nbits, uint32_t n_tiles, uint32_t tiles_per_chunk,
                                                        uint32_t *__restrict__ tile_pref, uint32_t *__restrict__ chunk_tot, int aligned) {
    constexpr int Q = TILE / (4 * RS_THREADS);             // 16-byte loads per thread and tile
    __shared__ uint32_t h[RS_MAX_DIGITS], run[RS_MAX_DIGITS];
    const uint32_t D = 1u << nbits, mask = D - 1u;
    const uint32_t c = blockIdx.x, t0 = c * tiles_per_chunk, t1 = min(t0 + tiles_per_chunk, n_tiles);
    for (uint32_t d = threadIdx.x; d < D; d += RS_THREADS) { run[d] = 0; h[d] = 0; }
    __syncthreads();
    uint4 k[Q], kn[Q];
    // the keys of the NEXT tile are on their way while this one is counted (the two barriers per tile would otherwise expose a full round trip)
    auto fetch = [&](uint32_t t, uint4 *dst) {
        const uint64_t base = (uint64_t) t * TILE;
        if (t < t1 && aligned && base + TILE <= n) {
            const uint4 *p = (const uint4 *) (keys + base);
#pragma unroll
            for (int j = 0; j < Q; j++) dst[j] = p[j * RS_THREADS + threadIdx.x];
        }
    };
    fetch(t0, k);
    for (uint32_t t = t0; t < t1; t++) {
        const uint64_t base = (uint64_t) t * TILE;
        fetch(t + 1, kn);
        if (aligned && base + TILE <= n) {
#pragma unroll
            for (int j = 0; j < Q; j++) {
                atomicAdd(&h[(k[j].x >> shift) & mask], 1u); atomicAdd(&h[(k[j].y >> shift) & mask], 1u);
                atomicAdd(&h[(k[j].z >> shift) & mask], 1u); atomicAdd(&h[(k[j].w >> shift) & mask], 1u);
            }
        } else {
#pragma unroll 4
            for (int j = 0; j < TILE / RS_THREADS; j++) {
                const uint64_t i = base + (uint64_t) j * RS_THREADS + threadIdx.x;
                if (i < n) atomicAdd(&h[(keys[i] >> shift) & mask], 1u);
            }
        }
        __syncthreads();
        for (uint32_t d = threadIdx.x; d < D; d += RS_THREADS) {
            const uint32_t x = h[d], r = run[d];
            tile_pref[(size_t) t * D + d] = r;
            run[d] = r + x;
            h[d] = 0;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < Q; j++) k[j] = kn[j];
    }
    for (uint32_t d = threadIdx.x; d < D; d += RS_THREADS) chunk_tot[(size_t) c * D + d] = run[d];
}

// chunk_tot[c][d] -> exclusive prefix over the chunks (in place); digit_tot[d] = all items with digit d.  A block per 16 digits, 32 groups of
// chunks side by side (a block per 64 digits with eight groups walked 64 chunks per thread, twice: 18 us of a pass that should take five)
constexpr int RS_SC_DIGITS = 16, RS_SC_GROUPS = RS_THREADS / RS_SC_DIGITS;
__global__ void __launch_bounds__(RS_THREADS) k_rs_scan_chunks(uint32_t *__restrict__ chunk_tot, uint32_t chunks, int nbits, uint32_t *__restrict__ digit_tot) {
    __shared__ uint32_t gs[RS_SC_GROUPS][RS_SC_DIGITS];
    const uint32_t D = 1u << nbits;
    const uint32_t g = threadIdx.x / RS_SC_DIGITS, dl = threadIdx.x % RS_SC_DIGITS, d = blockIdx.x * RS_SC_DIGITS + dl;
    const uint32_t cpg = (chunks + RS_SC_GROUPS - 1) / RS_SC_GROUPS, c0 = min(g * cpg, chunks), c1 = min(c0 + cpg, chunks);
    uint32_t s = 0;
    if (d < D) for (uint32_t c = c0; c < c1; c++) s += chunk_tot[(size_t) c * D + d];
    gs[g][dl] = s;
    __syncthreads();
    uint32_t base = 0, total = 0;
#pragma unroll 8
    for (int q = 0; q < RS_SC_GROUPS; q++) { const uint32_t x = gs[q][dl]; base += (uint32_t) q < g ? x : 0u; total += x; }
    if (d < D) {
        for (uint32_t c = c0; c < c1; c++) { const uint32_t x = chunk_tot[(size_t) c * D + d]; chunk_tot[(size_t) c * D + d] = base; base += x; }
        if (g == 0) digit_tot[d] = total;
    }
}

__device__ inline uint32_t rs_wave_incl_scan(uint32_t x) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(x, o, 64); if ((int) (threadIdx.x & 63u) >= o) x += y; }
    return x;
}

// exclusive scan of a value per thread over the block (WAVES waves); wsum: WAVES words of LDS
template <int WAVES>
__device__ inline uint32_t rs_block_excl_scan(uint32_t x, uint32_t *wsum) {
    const uint32_t incl = rs_wave_incl_scan(x);
    if ((threadIdx.x & 63u) == 63u) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t base = 0;
#pragma unroll
    for (int q = 0; q < WAVES; q++) base += (uint32_t) q < (threadIdx.x >> 6) ? wsum[q] : 0u;
    __syncthreads();
    return base + incl - x;
}

// digit_base[d] = items with a smaller digit (one block; two digits per thread)
__global__ void __launch_bounds__(RS_THREADS) k_rs_scan_digits(const uint32_t *__restrict__ digit_tot, int nbits, uint32_t *__restrict__ digit_base) {
    __shared__ uint32_t wsum[RS_WAVES];
    const uint32_t D = 1u << nbits, d0 = 2u * threadIdx.x;
    const uint32_t a = d0 < D ? digit_tot[d0] : 0u, b = d0 + 1 < D ? digit_tot[d0 + 1] : 0u;
    const uint32_t ex = rs_block_excl_scan<RS_WAVES>(a + b, wsum);
    if (d0 < D) digit_base[d0] = ex;
    if (d0 + 1 < D) digit_base[d0 + 1] = ex + a;
}

// the lanes of the wave whose digit equals this lane's: one ballot per digit bit.  Written so that a bit costs four vector instructions:
// the bit as 0 / -1 (v_bfe_i32), the ballot (v_cmp), and one three-input v_bitop3_b32 per half of the mask: peers & ~(ballot ^ bit).
template <int NB>
__device__ inline void rs_match(uint32_t d, uint32_t &plo, uint32_t &phi) {
    plo = 0xFFFFFFFFu; phi = 0xFFFFFFFFu;
#pragma unroll
    for (int b = 0; b < NB; b++) {
        const uint32_t x = (uint32_t) (((int32_t) (d << (31 - b))) >> 31);
        const unsigned long long m = __ballot(x != 0u);
        plo &= ~((uint32_t) m ^ x);
        phi &= ~((uint32_t) (m >> 32) ^ x);
    }
}

// ---- one tile: stable scatter by digit ------------------------------------------------------------------------------------------
template <int NB, int THREADS, bool IOTA>
__global__ void __launch_bounds__(THREADS) k_rs_scatter(const uint32_t *__restrict__ kin, const uint32_t *__restrict__ vin, uint32_t *__restrict__ kout,
                                                           uint32_t *__restrict__ vout, uint64_t n, int shift, int nbits, uint32_t n_tiles, uint32_t tiles_per_chunk,
                                                           const uint32_t *__restrict__ tile_pref, const uint32_t *__restrict__ chunk_base,
                                                           const uint32_t *__restrict__ digit_base) {
    constexpr int DMAX = 1 << NB, WAVES = THREADS / 64, RS_TILE = THREADS * RS_IPT;
    __shared__ unsigned short cnt[WAVES][DMAX];         // per wave and digit: items so far, then the tile position where the wave's run starts
    __shared__ uint32_t goff[DMAX];                        // global position of the tile's first item of a digit, minus its tile position
    __shared__ uint32_t stage[RS_TILE];
    __shared__ uint32_t wsum[WAVES];
    // XCD-contiguous tile order (see the head of the file)
    const uint32_t per = (n_tiles + 7u) >> 3;
    const uint32_t t = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    if (t >= n_tiles) return;
    const uint32_t D = 1u << nbits, mask = D - 1u;
    const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    {
        uint32_t *z = (uint32_t *) &cnt[0][0];
        for (uint32_t i = threadIdx.x; i < WAVES * DMAX / 2; i += THREADS) z[i] = 0;
        const uint32_t c = t / tiles_per_chunk;
        for (uint32_t d = threadIdx.x; d < D; d += THREADS)
            goff[d] = tile_pref[(size_t) t * D + d] + chunk_base[(size_t) c * D + d] + digit_base[d];
    }
    const uint64_t tbase = (uint64_t) t * RS_TILE;
    const uint32_t nv = (uint32_t) min((uint64_t) RS_TILE, n - tbase);
    uint32_t k[RS_IPT], v[RS_IPT];
    // wave w holds the items [w * 64 * IPT, (w + 1) * 64 * IPT) of the tile, step j the 64 consecutive ones from j * 64 on: the order of
    // (wave, step, lane) is the memory order, which is what "stable" refers to
#pragma unroll
    for (int j = 0; j < RS_IPT; j++) {
        const uint32_t idx = w * (64u * RS_IPT) + (uint32_t) j * 64u + lane;
        const bool ok = idx < nv;
        k[j] = ok ? kin[tbase + idx] : 0xFFFFFFFFu;         // padding of the last tile: the largest digit, behind every item of the tile
        v[j] = !ok ? 0u : (IOTA ? (uint32_t) (tbase + idx) : vin[tbase + idx]);      // (IOTA: no value array, the values are the positions -- the first pass of an argsort)
    }
    __syncthreads();
    unsigned short pos[RS_IPT];
#pragma unroll
    for (int j = 0; j < RS_IPT; j++) {
        const uint32_t d = (k[j] >> shift) & mask;
        uint32_t plo, phi;
        rs_match<NB>(d, plo, phi);
        const uint32_t below = __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u));       // lanes of my group below me
        const uint32_t c0 = cnt[w][d];
        pos[j] = (unsigned short) (c0 + below);
        if (below == 0u) cnt[w][d] = (unsigned short) (c0 + (uint32_t) (__popc(plo) + __popc(phi)));       // the lowest lane of the group notes the group
    }
    __syncthreads();
    // per digit: exclusive scan over the waves; then over the digits (two consecutive digits per thread)
    {
        const uint32_t d0 = 2u * threadIdx.x;
        uint32_t a = 0, b = 0;
        if (d0 < D) {
#pragma unroll
            for (int q = 0; q < WAVES; q++) {
                uint32_t *pp = (uint32_t *) &cnt[q][d0];
                const uint32_t x = *pp;
                *pp = a | (b << 16);
                a += x & 0xFFFFu; b += x >> 16;
            }
        }
        const uint32_t ex = rs_block_excl_scan<WAVES>(a + b, wsum);          // tile position of digit d0's first item
        if (d0 < D) {
#pragma unroll
            for (int q = 0; q < WAVES; q++) {
                uint32_t *pp = (uint32_t *) &cnt[q][d0];
                const uint32_t x = *pp;
                *pp = ((x & 0xFFFFu) + ex) | (((x >> 16) + ex + a) << 16);
            }
            goff[d0] -= ex;
            if (d0 + 1 < D) goff[d0 + 1] -= ex + a;
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RS_IPT; j++) {
        const uint32_t d = (k[j] >> shift) & mask;
        pos[j] = (unsigned short) (pos[j] + cnt[w][d]);
        stage[pos[j]] = k[j];
    }
    __syncthreads();
    uint32_t ga[RS_IPT];
#pragma unroll
    for (int j = 0; j < RS_IPT; j++) {
        const uint32_t i = (uint32_t) j * THREADS + threadIdx.x;
        const uint32_t key = stage[i];
        ga[j] = goff[(key >> shift) & mask] + i;
        if (i < nv) kout[ga[j]] = key;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RS_IPT; j++) stage[pos[j]] = v[j];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RS_IPT; j++) {
        const uint32_t i = (uint32_t) j * THREADS + threadIdx.x;
        if (i < nv) vout[ga[j]] = stage[i];
    }
}

// ---- 16-byte records: (u64 key, u64 value), sorted on the key's low bits (the k-mer entries of the supplement: pkb_kernels.hip) ---------------------
// The same three kernels per pass; a tile is 4096 records (512 threads x 8), the staging buffer holds 8-byte words.
constexpr int RS_IPT64 = 8;
constexpr int RS_TILE64 = RS_THREADS * RS_IPT64;

template <typename K>
__global__ void __launch_bounds__(RS_THREADS) k_rs_hist64(const K *__restrict__ keys, uint64_t n, int shift, int nbits, uint32_t n_tiles,
                                                          uint32_t tiles_per_chunk, uint32_t *__restrict__ tile_pref, uint32_t *__restrict__ chunk_tot) {
    __shared__ uint32_t h[RS_MAX_DIGITS], run[RS_MAX_DIGITS];
    const uint32_t D = 1u << nbits, mask = D - 1u;
    const uint32_t c = blockIdx.x, t0 = c * tiles_per_chunk, t1 = min(t0 + tiles_per_chunk, n_tiles);
    for (uint32_t d = threadIdx.x; d < D; d += RS_THREADS) { run[d] = 0; h[d] = 0; }
    __syncthreads();
    K k[RS_IPT64], kn[RS_IPT64];
    auto fetch = [&](uint32_t t, K *dst) {
        const uint64_t base = (uint64_t) t * RS_TILE64;
        if (t < t1 && base + RS_TILE64 <= n) {
#pragma unroll
            for (int j = 0; j < RS_IPT64; j++) dst[j] = keys[base + (uint64_t) j * RS_THREADS + threadIdx.x];
        }
    };
    fetch(t0, k);
    for (uint32_t t = t0; t < t1; t++) {
        const uint64_t base = (uint64_t) t * RS_TILE64;
        fetch(t + 1, kn);
        if (base + RS_TILE64 <= n) {
#pragma unroll
            for (int j = 0; j < RS_IPT64; j++) atomicAdd(&h[(uint32_t) (k[j] >> shift) & mask], 1u);
        } else {
            for (int j = 0; j < RS_IPT64; j++) {
                const uint64_t i = base + (uint64_t) j * RS_THREADS + threadIdx.x;
                if (i < n) atomicAdd(&h[(uint32_t) (keys[i] >> shift) & mask], 1u);
            }
        }
        __syncthreads();
        for (uint32_t d = threadIdx.x; d < D; d += RS_THREADS) {
            const uint32_t x = h[d], r = run[d];
            tile_pref[(size_t) t * D + d] = r;
            run[d] = r + x;
            h[d] = 0;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < RS_IPT64; j++) k[j] = kn[j];
    }
    for (uint32_t d = threadIdx.x; d < D; d += RS_THREADS) chunk_tot[(size_t) c * D + d] = run[d];
}

template <typename K, int NB>
__global__ void __launch_bounds__(RS_THREADS) k_rs_scatter64(const K *__restrict__ kin, const unsigned long long *__restrict__ vin,
                                                             K *__restrict__ kout, unsigned long long *__restrict__ vout, uint64_t n, int shift,
                                                             int nbits, uint32_t n_tiles, uint32_t tiles_per_chunk, const uint32_t *__restrict__ tile_pref,
                                                             const uint32_t *__restrict__ chunk_base, const uint32_t *__restrict__ digit_base) {
    constexpr int DMAX = 1 << NB, WAVES = RS_THREADS / 64;
    __shared__ unsigned short cnt[WAVES][DMAX];
    __shared__ uint32_t goff[DMAX];
    __shared__ unsigned long long stage[RS_TILE64];
    __shared__ uint32_t wsum[WAVES];
    const uint32_t per = (n_tiles + 7u) >> 3;
    const uint32_t t = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    if (t >= n_tiles) return;
    const uint32_t D = 1u << nbits, mask = D - 1u;
    const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    {
        uint32_t *z = (uint32_t *) &cnt[0][0];
        for (uint32_t i = threadIdx.x; i < WAVES * DMAX / 2; i += RS_THREADS) z[i] = 0;
        const uint32_t c = t / tiles_per_chunk;
        for (uint32_t d = threadIdx.x; d < D; d += RS_THREADS)
            goff[d] = tile_pref[(size_t) t * D + d] + chunk_base[(size_t) c * D + d] + digit_base[d];
    }
    const uint64_t tbase = (uint64_t) t * RS_TILE64;
    const uint32_t nv = (uint32_t) min((uint64_t) RS_TILE64, n - tbase);
    K k[RS_IPT64];
    unsigned long long v[RS_IPT64];
#pragma unroll
    for (int j = 0; j < RS_IPT64; j++) {
        const uint32_t idx = w * (64u * RS_IPT64) + (uint32_t) j * 64u + lane;
        const bool ok = idx < nv;
        k[j] = ok ? kin[tbase + idx] : (K) ~(K) 0;          // padding of the last tile: the largest digit, behind every item of the tile
        v[j] = ok ? vin[tbase + idx] : 0ull;
    }
    __syncthreads();
    unsigned short pos[RS_IPT64];
#pragma unroll
    for (int j = 0; j < RS_IPT64; j++) {
        const uint32_t d = (uint32_t) (k[j] >> shift) & mask;
        uint32_t plo, phi;
        rs_match<NB>(d, plo, phi);
        const uint32_t below = __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u));
        const uint32_t c0 = cnt[w][d];
        pos[j] = (unsigned short) (c0 + below);
        if (below == 0u) cnt[w][d] = (unsigned short) (c0 + (uint32_t) (__popc(plo) + __popc(phi)));
    }
    __syncthreads();
    {
        const uint32_t d0 = 2u * threadIdx.x;
        uint32_t a = 0, b = 0;
        if (d0 < D) {
#pragma unroll
            for (int q = 0; q < WAVES; q++) {
                uint32_t *pp = (uint32_t *) &cnt[q][d0];
                const uint32_t x = *pp;
                *pp = a | (b << 16);
                a += x & 0xFFFFu; b += x >> 16;
            }
        }
        const uint32_t ex = rs_block_excl_scan<WAVES>(a + b, wsum);
        if (d0 < D) {
#pragma unroll
            for (int q = 0; q < WAVES; q++) {
                uint32_t *pp = (uint32_t *) &cnt[q][d0];
                const uint32_t x = *pp;
                *pp = ((x & 0xFFFFu) + ex) | (((x >> 16) + ex + a) << 16);
            }
            goff[d0] -= ex;
            if (d0 + 1 < D) goff[d0 + 1] -= ex + a;
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RS_IPT64; j++) {
        const uint32_t d = (uint32_t) (k[j] >> shift) & mask;
        pos[j] = (unsigned short) (pos[j] + cnt[w][d]);
        stage[pos[j]] = (unsigned long long) k[j];
    }
    __syncthreads();
    uint32_t ga[RS_IPT64];
#pragma unroll
    for (int j = 0; j < RS_IPT64; j++) {
        const uint32_t i = (uint32_t) j * RS_THREADS + threadIdx.x;
        const K key = (K) stage[i];
        ga[j] = goff[(uint32_t) (key >> shift) & mask] + i;
        if (i < nv) kout[ga[j]] = key;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RS_IPT64; j++) stage[pos[j]] = v[j];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RS_IPT64; j++) {
        const uint32_t i = (uint32_t) j * RS_THREADS + threadIdx.x;
        if (i < nv) vout[ga[j]] = stage[i];
    }
}

size_t rs_align(size_t x) { return (x + 255) & ~(size_t) 255; }

} // namespace

// temp: an intermediate (key, value) buffer for the passes that do not end in the output, the tile prefix rows, chunk and digit totals
size_t rsort_u32_pairs_temp_bytes(uint64_t n) {
    const RsPlan p = rs_plan(n, 0, 32, 0);                 // (the smaller tile: more prefix rows)
    return 2 * rs_align((size_t) (n + 4) * 4) + rs_align((size_t) p.n_tiles * RS_MAX_DIGITS * 4) + rs_align((size_t) RS_CHUNKS * RS_MAX_DIGITS * 4) +
           2 * rs_align(RS_MAX_DIGITS * 4);
}

void rsort_set_variant(int v) { g_rs_variant = v; }

// stable sort of (key, value) on the key bits [begin_bit, 32); keys_in / vals_in are left untouched.  vals_in == nullptr: the values are 0, 1, 2, ...
// (the sort of (key, id) pairs of the index build: 4 bytes per pair less to read in the first pass, and nobody has to write them)
hipError_t rsort_u32_pairs(void *temp, size_t temp_bytes, const uint32_t *keys_in, uint32_t *keys_out, const uint32_t *vals_in, uint32_t *vals_out, uint64_t n,
                           int begin_bit, hipStream_t s) {
    if (n == 0) return hipSuccess;
    if (begin_bit < 0 || begin_bit > 31 || n >= (1ull << 32) - RS_TILE_MAX) return hipErrorInvalidValue;
    if (temp_bytes < rsort_u32_pairs_temp_bytes(n)) return hipErrorInvalidValue;
    const int variant = g_rs_variant;
    const RsPlan p = rs_plan(n, begin_bit, 32, variant);
    char *at = (char *) temp;
    uint32_t *tk = (uint32_t *) at; at += rs_align((size_t) (n + 4) * 4);
    uint32_t *tv = (uint32_t *) at; at += rs_align((size_t) (n + 4) * 4);
    uint32_t *tile_pref = (uint32_t *) at; at += rs_align((size_t) rs_plan(n, 0, 32, 0).n_tiles * RS_MAX_DIGITS * 4);
    uint32_t *chunk_tot = (uint32_t *) at; at += rs_align((size_t) RS_CHUNKS * RS_MAX_DIGITS * 4);
    uint32_t *digit_tot = (uint32_t *) at; at += rs_align(RS_MAX_DIGITS * 4);
    uint32_t *digit_base = (uint32_t *) at;
    const uint32_t *ki = keys_in, *vi = vals_in;
    for (int i = 0; i < p.passes; i++) {
        // the last pass ends in the output; the ones before it alternate between the intermediate buffer and the output
        const bool to_out = ((p.passes - 1 - i) & 1) == 0;
        uint32_t *ko = to_out ? keys_out : tk, *vo = to_out ? vals_out : tv;
        const int nb = p.bits[i], sh = p.shift[i];
        const int aligned = (int) ((((uintptr_t) ki) & 15u) == 0);
        if (p.tile == 8192)
            hipLaunchKernelGGL((k_rs_hist<8192>), dim3(p.chunks), dim3(RS_THREADS), 0, s, ki, n, sh, nb, p.n_tiles, p.tiles_per_chunk, tile_pref, chunk_tot, aligned);
        else
            hipLaunchKernelGGL((k_rs_hist<16384>), dim3(p.chunks), dim3(RS_THREADS), 0, s, ki, n, sh, nb, p.n_tiles, p.tiles_per_chunk, tile_pref, chunk_tot, aligned);
        hipLaunchKernelGGL(k_rs_scan_chunks, dim3(((1u << nb) + RS_SC_DIGITS - 1u) / RS_SC_DIGITS), dim3(RS_THREADS), 0, s, chunk_tot, p.chunks, nb, digit_tot);
        hipLaunchKernelGGL(k_rs_scan_digits, dim3(1), dim3(RS_THREADS), 0, s, (const uint32_t *) digit_tot, nb, digit_base);
        const dim3 grid(8u * ((p.n_tiles + 7u) / 8u));
#define RS_SCATTER_(NB_, T_, I_) hipLaunchKernelGGL((k_rs_scatter<NB_, T_, I_>), grid, dim3(T_), 0, s, ki, vi, ko, vo, n, sh, nb, p.n_tiles, p.tiles_per_chunk, \
                                                   (const uint32_t *) tile_pref, (const uint32_t *) chunk_tot, (const uint32_t *) digit_base)
#define RS_SCATTER(NB_, T_) do { if (vi) RS_SCATTER_(NB_, T_, false); else RS_SCATTER_(NB_, T_, true); } while (0)
        if (p.tile == 8192) { if (nb <= 8) RS_SCATTER(8, 512); else RS_SCATTER(10, 512); }
        else                { if (nb <= 8) RS_SCATTER(8, 1024); else RS_SCATTER(10, 1024); }
#undef RS_SCATTER_
#undef RS_SCATTER
        ki = ko; vi = vo;
    }
    return hipGetLastError();
}

size_t rsort_u64_pairs_temp_bytes(uint64_t n) {
    const RsPlan p = rs_plan(n, 0, 32, 2);
    return 2 * rs_align((size_t) (n + 4) * 8) + rs_align((size_t) p.n_tiles * RS_MAX_DIGITS * 4) + rs_align((size_t) RS_CHUNKS * RS_MAX_DIGITS * 4) +
           2 * rs_align(RS_MAX_DIGITS * 4);
}

// stable sort of (u64 key, u64 value) records on the key bits [0, bits), 1 <= bits <= 50; the inputs are left untouched
hipError_t rsort_u64_pairs(void *temp, size_t temp_bytes, const unsigned long long *keys_in, unsigned long long *keys_out, const unsigned long long *vals_in,
                           unsigned long long *vals_out, uint64_t n, int bits, hipStream_t s) {
    if (n == 0) return hipSuccess;
    if (bits < 1 || bits > 50 || n >= (1ull << 32) - RS_TILE_MAX) return hipErrorInvalidValue;
    if (temp_bytes < rsort_u64_pairs_temp_bytes(n)) return hipErrorInvalidValue;
    const RsPlan p = rs_plan(n, 0, bits, 2);
    char *at = (char *) temp;
    unsigned long long *tk = (unsigned long long *) at; at += rs_align((size_t) (n + 4) * 8);
    unsigned long long *tv = (unsigned long long *) at; at += rs_align((size_t) (n + 4) * 8);
    uint32_t *tile_pref = (uint32_t *) at; at += rs_align((size_t) p.n_tiles * RS_MAX_DIGITS * 4);
    uint32_t *chunk_tot = (uint32_t *) at; at += rs_align((size_t) RS_CHUNKS * RS_MAX_DIGITS * 4);
    uint32_t *digit_tot = (uint32_t *) at; at += rs_align(RS_MAX_DIGITS * 4);
    uint32_t *digit_base = (uint32_t *) at;
    const unsigned long long *ki = keys_in, *vi = vals_in;
    for (int i = 0; i < p.passes; i++) {
        const bool to_out = ((p.passes - 1 - i) & 1) == 0;
        unsigned long long *ko = to_out ? keys_out : tk, *vo = to_out ? vals_out : tv;
        const int nb = p.bits[i], sh = p.shift[i];
        hipLaunchKernelGGL((k_rs_hist64<unsigned long long>), dim3(p.chunks), dim3(RS_THREADS), 0, s, ki, n, sh, nb, p.n_tiles, p.tiles_per_chunk, tile_pref, chunk_tot);
        hipLaunchKernelGGL(k_rs_scan_chunks, dim3(((1u << nb) + RS_SC_DIGITS - 1u) / RS_SC_DIGITS), dim3(RS_THREADS), 0, s, chunk_tot, p.chunks, nb, digit_tot);
        hipLaunchKernelGGL(k_rs_scan_digits, dim3(1), dim3(RS_THREADS), 0, s, (const uint32_t *) digit_tot, nb, digit_base);
        const dim3 grid(8u * ((p.n_tiles + 7u) / 8u));
        if (nb <= 8) hipLaunchKernelGGL((k_rs_scatter64<unsigned long long, 8>), grid, dim3(RS_THREADS), 0, s, ki, vi, ko, vo, n, sh, nb, p.n_tiles, p.tiles_per_chunk,
                                        (const uint32_t *) tile_pref, (const uint32_t *) chunk_tot, (const uint32_t *) digit_base);
        else hipLaunchKernelGGL((k_rs_scatter64<unsigned long long, 10>), grid, dim3(RS_THREADS), 0, s, ki, vi, ko, vo, n, sh, nb, p.n_tiles, p.tiles_per_chunk,
                                (const uint32_t *) tile_pref, (const uint32_t *) chunk_tot, (const uint32_t *) digit_base);
        ki = ko; vi = vo;
    }
    return hipGetLastError();
}

// stable sort of (u32 key, u64 value) records on the key bits [begin_bit, end_bit): the run descriptors of the bucket-sharded N-GPU build
// (prefsuf_shard.hip) and whatever else carries eight bytes behind a 32-bit key.  temp: rsort_u64_pairs_temp_bytes(n)
hipError_t rsort_u32_u64(void *temp, size_t temp_bytes, const uint32_t *keys_in, uint32_t *keys_out, const unsigned long long *vals_in, unsigned long long *vals_out,
                         uint64_t n, int begin_bit, int end_bit, hipStream_t s) {
    if (n == 0) return hipSuccess;
    if (begin_bit < 0 || end_bit > 32 || end_bit <= begin_bit || n >= (1ull << 32) - RS_TILE_MAX) return hipErrorInvalidValue;
    if (temp_bytes < rsort_u64_pairs_temp_bytes(n)) return hipErrorInvalidValue;
    const RsPlan p = rs_plan(n, begin_bit, end_bit, 2);
    char *at = (char *) temp;
    uint32_t *tk = (uint32_t *) at; at += rs_align((size_t) (n + 4) * 8);
    unsigned long long *tv = (unsigned long long *) at; at += rs_align((size_t) (n + 4) * 8);
    uint32_t *tile_pref = (uint32_t *) at; at += rs_align((size_t) p.n_tiles * RS_MAX_DIGITS * 4);
    uint32_t *chunk_tot = (uint32_t *) at; at += rs_align((size_t) RS_CHUNKS * RS_MAX_DIGITS * 4);
    uint32_t *digit_tot = (uint32_t *) at; at += rs_align(RS_MAX_DIGITS * 4);
    uint32_t *digit_base = (uint32_t *) at;
    const uint32_t *ki = keys_in;
    const unsigned long long *vi = vals_in;
    for (int i = 0; i < p.passes; i++) {
        const bool to_out = ((p.passes - 1 - i) & 1) == 0;
        uint32_t *ko = to_out ? keys_out : tk;
        unsigned long long *vo = to_out ? vals_out : tv;
        const int nb = p.bits[i], sh = p.shift[i];
        hipLaunchKernelGGL((k_rs_hist64<uint32_t>), dim3(p.chunks), dim3(RS_THREADS), 0, s, ki, n, sh, nb, p.n_tiles, p.tiles_per_chunk, tile_pref, chunk_tot);
        hipLaunchKernelGGL(k_rs_scan_chunks, dim3(((1u << nb) + RS_SC_DIGITS - 1u) / RS_SC_DIGITS), dim3(RS_THREADS), 0, s, chunk_tot, p.chunks, nb, digit_tot);
        hipLaunchKernelGGL(k_rs_scan_digits, dim3(1), dim3(RS_THREADS), 0, s, (const uint32_t *) digit_tot, nb, digit_base);
        const dim3 grid(8u * ((p.n_tiles + 7u) / 8u));
        if (nb <= 8) hipLaunchKernelGGL((k_rs_scatter64<uint32_t, 8>), grid, dim3(RS_THREADS), 0, s, ki, vi, ko, vo, n, sh, nb, p.n_tiles, p.tiles_per_chunk,
                                        (const uint32_t *) tile_pref, (const uint32_t *) chunk_tot, (const uint32_t *) digit_base);
        else hipLaunchKernelGGL((k_rs_scatter64<uint32_t, 10>), grid, dim3(RS_THREADS), 0, s, ki, vi, ko, vo, n, sh, nb, p.n_tiles, p.tiles_per_chunk,
                                (const uint32_t *) tile_pref, (const uint32_t *) chunk_tot, (const uint32_t *) digit_base);
        ki = ko; vi = vo;
    }
    return hipGetLastError();
}

} // namespace alga
