// alga_amd/csrc/prefsuf_minimizer.hip -- probe of the PrefSuf engine through a MINIMIZER INDEX.
//
// The bucketised seed table (prefsuf_kernels.hip) answers one probe per (source, overlap length): 63 random 64-byte
// lines per 150-bp read, 107 M at BASELINE configs[1], and that traffic to the Infinity Cache bounds the kernel.
// Here a target C is indexed once, under the minimizer of its min_overlap-long prefix (the k-mer with the smallest
// hash among its Lmin-k+1 k-mers, k <= 20), together with the minimizer's position m_C.  If suffix window s of a source
// B equals that prefix, the window has the same minimizer at position s + m_C; consecutive windows share minimizers
// (density 2/(w+1)), so a 150-bp source has ~3 distinct ones instead of 63 windows to look up.  Each lookup returns the
// short list of targets filed under that k-mer; `s = a - m_C` names the only window a target can match, which is then
// verified bit for bit exactly as before.  Directory (distinct k-mers) and lists stay L2 / Infinity-Cache resident.
//
//   k_index_targets     prefix minimizer of every target -> (k-mer, node | m_C | len)
//   (radix sort by k-mer)
//   k_index_count / k_index_directory   distinct k-mers -> open-addressing directory k-mer -> (start, count)
//   k_probe_min         persistent wavefronts, one source at a time: k-mer keys of the staged tail, sliding-window
//                       minimum by a sparse table in LDS, one directory lookup per distinct minimizer, candidates,
//                       4-lane row compare, per-source top-3, chunked record output (shared with the other probe)
#include <hip/hip_runtime.h>
#include <algorithm>
#include "prefsuf_common.h"
#include "prefsuf_kernels.h"
#include "prefsuf_device.h"

namespace alga {

constexpr int MIN_KMAX = 20;                 // minimizer k-mer length (<= 20 nt = 40 bits), capped by min_overlap
constexpr int MIN_SLOTS = 512;               // k-mer / window slots per wave (tail <= 501 nt)
constexpr unsigned long long DIR_EMPTY = ~0ull;

struct MinCfg {
    int32_t kk;          // k-mer length
    int32_t w;           // k-mers per window = Lmin - kk + 1
    uint64_t kmask;      // 2*kk low bits
};

// ordering hash (23 bits) of a k-mer value; the k-mer itself is the exact key of the directory
__device__ __forceinline__ uint32_t kmer_order(uint64_t v) { return (uint32_t) ((v * 0x9E3779B97F4A7C15ull) >> 41); }
__device__ __forceinline__ uint32_t dir_slot(uint64_t v, uint32_t dmask) { return (uint32_t) ((v * 0xD6E8FEB86659FD93ull) >> 32) & dmask; }

// 2*kk bits of a bit string held in 32-bit words, starting at bit `bit`
__device__ __forceinline__ uint64_t kmer_at(const uint32_t *w, int bit, uint64_t kmask) {
    const int q = bit >> 5, r = bit & 31;
    const uint32_t lo = funnel(w[q], w[q + 1], r), hi = funnel(w[q + 1], w[q + 2], r);
    return (((uint64_t) hi << 32) | lo) & kmask;
}

// ------------------------------------------------------------------------------------------
// index build
// ------------------------------------------------------------------------------------------
// one thread per target: minimizer of the prefix window C[0, Lmin)
//   keys[i] = k-mer value (all ones = not a target), vals[i] = node | m_C << 32 | min(len, 511) << 41
__global__ void __launch_bounds__(256) k_index_targets(NodesDev nd, PrefSufCfg cfg, MinCfg mc, unsigned long long *__restrict__ keys,
                                                        unsigned long long *__restrict__ vals) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nd.n) return;
    const int len = nd.len[i];
    unsigned long long key = ~0ull, val = 0;
    if (len > 0 && len >= cfg.Lmin && (!nd.to || nd.to[i])) {
        const uint32_t *row = nd.words + (size_t) i * nd.stride;
        const int nw = blocks_of(len);
        uint32_t best = 0xFFFFFFFFu;
        uint64_t bestv = 0;
        for (int j = 0; j < mc.w; j++) {
            const int bit = 2 * j, q = bit >> 5, r = bit & 31;
            const uint32_t w0 = row[q], w1 = q + 1 < nw ? row[q + 1] : 0u, w2 = q + 2 < nw ? row[q + 2] : 0u;
            const uint64_t v = ((((uint64_t) funnel(w1, w2, r)) << 32) | funnel(w0, w1, r)) & mc.kmask;
            const uint32_t k = (kmer_order(v) << 9) | (uint32_t) j;
            if (k < best) { best = k; bestv = v; }
        }
        key = bestv;
        val = (unsigned long long) (uint32_t) i | ((unsigned long long) (best & 511u) << 32) | ((unsigned long long) (len > 511 ? 511 : len) << 41);
    }
    keys[i] = key; vals[i] = val;
}

__global__ void __launch_bounds__(256) k_index_count(const unsigned long long *__restrict__ keys, uint64_t n, unsigned long long *__restrict__ out /* [0] distinct, [1] valid */) {
    __shared__ unsigned long long s[2][4];
    unsigned long long d = 0, v = 0;
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
        const unsigned long long k = keys[i];
        if (k == ~0ull) continue;
        v++;
        if (i == 0 || keys[i - 1] != k) d++;
    }
    d = wave_sum_u64(d); v = wave_sum_u64(v);
    if (lane_id() == 0) { s[0][threadIdx.x >> 6] = d; s[1][threadIdx.x >> 6] = v; }
    __syncthreads();
    if (threadIdx.x == 0) {
        d = s[0][0] + s[0][1] + s[0][2] + s[0][3]; v = s[1][0] + s[1][1] + s[1][2] + s[1][3];
        if (d) atomicAdd(&out[0], d);
        if (v) atomicAdd(&out[1], v);
    }
}

// directory slot = {key, start << 32 | count}; the thread of the first entry of a run inserts it
__global__ void __launch_bounds__(256) k_index_directory(const unsigned long long *__restrict__ keys, uint64_t n, unsigned long long *__restrict__ dir,
                                                          uint32_t dmask) {
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
        const unsigned long long k = keys[i];
        if (k == ~0ull || (i > 0 && keys[i - 1] == k)) continue;
        uint64_t e = i + 1;
        while (e < n && keys[e] == k) e++;
        uint32_t slot = dir_slot(k, dmask);
        for (;;) {
            if (atomicCAS(&dir[2 * (size_t) slot], DIR_EMPTY, k) == DIR_EMPTY) { dir[2 * (size_t) slot + 1] = ((unsigned long long) i << 32) | (e - i); break; }
            slot = (slot + 1) & dmask;
        }
    }
}

// ------------------------------------------------------------------------------------------
// k_probe_min
// ------------------------------------------------------------------------------------------
struct MinIndex {
    const unsigned long long *__restrict__ dir;   // 2 words per slot
    const unsigned long long *__restrict__ list;  // sorted vals
    uint32_t dmask;
};

template <bool STATS, int NQ>
__global__ void __launch_bounds__(PROBE_WAVES * 64, 4)
k_probe_min(NodesDev nd, PrefSufCfg cfg, MinCfg mc, MinIndex ix, int32_t src_begin, int32_t src_end, ProbeOut o) {
    __shared__ uint32_t sB[PROBE_WAVES][STAGE_WORDS];
    __shared__ uint32_t sK[PROBE_WAVES][2][MIN_SLOTS + 64];
    __shared__ uint32_t sRun[PROBE_WAVES][64];
    __shared__ uint32_t sRunA[PROBE_WAVES][64], sRunS[PROBE_WAVES][64], sRunP[PROBE_WAVES][65];
    __shared__ uint32_t sCandC[PROBE_WAVES][CANDMAX];
    __shared__ uint32_t sCandW[PROBE_WAVES][CANDMAX];
    __shared__ uint32_t sRecC[PROBE_WAVES][WBUF];
    __shared__ unsigned long long sRecV[PROBE_WAVES][WBUF];
    __shared__ uint32_t sCnt[PROBE_WAVES][4];
    const int wave = (int) (threadIdx.x >> 6);
    const int lane = lane_id();
    WaveLds w{sB[wave], sCandC[wave], sCandW[wave], &sCnt[wave][0], sRecC[wave], sRecV[wave], &sCnt[wave][1]};
    uint32_t *runN = &sCnt[wave][2];
    if (lane == 0) { *w.candN = 0; *w.recN = 0; *runN = 0; }
    uint64_t chunk_base = 0;
    int chunk_fill = REC_CHUNK;
    uint64_t st_raw = 0, st_slots = 0, st_win = 0, st_rec = 0;
    const int64_t total_waves = (int64_t) gridDim.x * PROBE_WAVES;
    const uint32_t *sb = w.sb;

    const int pre_words = nd.stride < STAGE_WORDS ? nd.stride : STAGE_WORDS;
    int64_t Bl = (int64_t) src_begin + (int64_t) blockIdx.x * PROBE_WAVES + wave;
    int n_len = 0; uint32_t n_word = 0; uint8_t n_from = 1;
    if (Bl < src_end) {
        n_len = nd.len[Bl];
        n_word = lane < pre_words ? nd.words[(size_t) Bl * nd.stride + lane] : 0u;
        if (nd.from) n_from = nd.from[Bl];
    }
    while (Bl < src_end) {
        const int B = (int) Bl;
        const int lenB = n_len;
        const uint32_t word0 = n_word;
        const bool from_ok = n_from != 0;
        Bl += total_waves;
        if (Bl < src_end) {                                 // software pipeline: next source's row is on its way
            n_len = nd.len[Bl];
            n_word = lane < pre_words ? nd.words[(size_t) Bl * nd.stride + lane] : 0u;
            if (nd.from) n_from = nd.from[Bl];
        }
        if (!(lenB >= cfg.Lmin && lenB > 0 && from_ok)) continue;                       // wave-uniform
        const int Lspan = lenB < cfg.Lcap ? lenB : cfg.Lcap;
        const int w0 = (2 * (lenB - Lspan)) >> 5;
        const int nwB = blocks_of(lenB) - w0;
        {
            wave_lds_fence();
            uint32_t x = word0;
            if (w0 != 0) x = (lane < nwB) ? nd.words[(size_t) B * nd.stride + w0 + lane] : 0u;
            if (lane < STAGE_WORDS) w.sb[lane] = lane < nwB ? x : 0u;
            wave_lds_fence();
        }
        const int nwin = Lspan - cfg.Lmin + 1;             // windows s = 0 .. nwin-1 <-> overlap length L = Lspan - s
        const int nk = Lspan - mc.kk + 1;                   // k-mers of the staged tail
        const int base_bit = 2 * (lenB - Lspan) - 32 * w0;  // bit of the first tail nucleotide inside the staged words
        uint64_t k0 = 0, k1 = 0, k2 = 0;

        auto classify = [&](int C, int L) {
            if (STATS) st_raw++;
            if (L < cfg.rsoemo) {
                top3_insert(k0, k1, k2, ((uint64_t) (uint32_t) L << 32) | (uint32_t) C);       // GraphCreatorPrefSuf.cpp:397-401
            } else {
                st_rec++;
                const unsigned long long val = ((unsigned long long) ol_pack(lenB - L, L, false) << 32) | (uint32_t) B;
                const uint32_t i = atomicAdd(w.recN, 1u);
                if (i < (uint32_t) WBUF) { w.recC[i] = (uint32_t) C; w.recV[i] = val; }
                else store_record(o, atomicAdd(&o.counters[CNT_RECORDS], 1ull), (uint32_t) C, val);
            }
        };
        // candidates -> exact 2-bit compare of C[0, L) with the window; four lanes per candidate (NQ > 0)
        auto verify_candidates = [&]() {
            wave_lds_fence();
            int ncand = (int) __builtin_amdgcn_readfirstlane((int) *w.candN);
            if (ncand > CANDMAX) ncand = CANDMAX;
            const int sub = lane & 3;
            if (NQ > 0) {
                for (int c0 = 0; c0 < ncand; c0 += 32) {
                    int Cc[2], Lc[2];
                    uint4 cc[2];
                    bool act[2];
#pragma unroll
                    for (int g = 0; g < 2; g++) {
                        const int ci = c0 + 16 * g + (lane >> 2);
                        act[g] = ci < ncand;
                        Cc[g] = 0; Lc[g] = Lspan;
                        cc[g] = make_uint4(0u, 0u, 0u, 0u);
                        if (act[g]) {
                            Cc[g] = (int) w.candC[ci];
                            Lc[g] = Lspan - (int) w.candW[ci];
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
#pragma unroll
                            for (int j = 0; j < 4; j++) {
                                const int k = 4 * sub + j;
                                const uint32_t m = k < nwL - 1 ? 0xFFFFFFFFu : (k == nwL - 1 ? lastmask : 0u);
                                diff |= (funnel(sb[q + k], sb[q + k + 1], r) ^ cw[j]) & m;
                            }
                        }
                        diff = quad_or(diff);
                        if (act[g] && sub == 0 && diff == 0) classify(Cc[g], Lc[g]);
                    }
                }
            } else {
                for (int c0 = 0; c0 < ncand; c0 += 64) {
                    const int ci = c0 + lane;
                    if (ci < ncand) {
                        const int C = (int) w.candC[ci];
                        const int L = Lspan - (int) w.candW[ci];
                        const int bit = 2 * (lenB - L) - 32 * w0;
                        if (verify_overlap<0>(nd, sb, C, bit >> 5, bit & 31, L)) classify(C, L);
                    }
                }
            }
            wave_lds_fence();
            if (lane == 0) *w.candN = 0;
            wave_lds_fence();
        };

#if defined(ABLATE) && ABLATE == 11
        continue;
#endif
        // ---- k-mer keys of the tail: (order hash << 9) | position ---------------------------------------------
        uint32_t *ka = sK[wave][0], *kb = sK[wave][1];
        for (int j = lane; j < nk; j += 64) ka[j] = (kmer_order(kmer_at(sb, base_bit + 2 * j, mc.kmask)) << 9) | (uint32_t) j;
        wave_lds_fence();
        // ---- sliding-window minimum over w k-mers: sparse table by doubling, two LDS buffers --------------------
        int span = 1;
        while (2 * span <= mc.w) {
            for (int j = lane; j < nk; j += 64) {
                const uint32_t x = ka[j], y = (j + span < nk) ? ka[j + span] : 0xFFFFFFFFu;
                kb[j] = x < y ? x : y;
            }
            wave_lds_fence();
            uint32_t *t = ka; ka = kb; kb = t;
            span *= 2;
        }
        // window s: min over k-mers [s, s + w) = min(table[s], table[s + w - span]); its low 9 bits = minimizer position
        for (int s = lane; s < nwin; s += 64) {
            const uint32_t x = ka[s], y = ka[s + mc.w - span];
            kb[s] = x < y ? x : y;
        }
        wave_lds_fence();
        if (STATS) st_win += (lane < 1) ? (uint64_t) nwin : 0;
        // ---- distinct minimizers: a window starts a run when its minimizer differs from the previous window's ----
        for (int s = lane; s < nwin; s += 64) {
            const uint32_t a = kb[s] & 511u;
            if (s == 0 || (kb[s - 1] & 511u) != a) { const uint32_t ri = atomicAdd(runN, 1u); if (ri < 64u) sRun[wave][ri] = a; }
        }
        wave_lds_fence();
        int nrun = (int) __builtin_amdgcn_readfirstlane((int) *runN);
#if defined(ABLATE) && ABLATE == 12
        if (lane == 0) *runN = 0;
        continue;
#endif
        if (nrun <= 64) {
            // ---- all distinct minimizers at once: lane r looks run r up in the directory (one round trip for the wave) ----
            uint32_t my_start = 0, my_count = 0;
            int my_a = 0;
            if (lane < nrun) {
                my_a = (int) sRun[wave][lane];
                const uint64_t kv = kmer_at(sb, base_bit + 2 * my_a, mc.kmask);
                uint32_t slot = dir_slot(kv, ix.dmask);
                for (;;) {
                    const unsigned long long dk = ix.dir[2 * (size_t) slot];
                    if (STATS) st_slots++;
                    if (dk == kv) { const unsigned long long hit = ix.dir[2 * (size_t) slot + 1]; my_start = (uint32_t) (hit >> 32); my_count = (uint32_t) hit; break; }
                    if (dk == DIR_EMPTY) break;
                    slot = (slot + 1) & ix.dmask;
                }
            }
            // exclusive prefix of the list lengths over the runs (nrun is small: ~3 for 150-bp reads)
            sRunA[wave][lane] = (uint32_t) my_a; sRunS[wave][lane] = my_start; sRunP[wave][lane] = my_count;
            wave_lds_fence();
            if (lane == 0) {
                uint32_t acc = 0;
                for (int r = 0; r < nrun; r++) { const uint32_t c = sRunP[wave][r]; sRunP[wave][r] = acc; acc += c; }
                sRunP[wave][nrun] = acc;
            }
            wave_lds_fence();
            const int total = (int) sRunP[wave][nrun];
            // ---- one list entry per lane (one round trip for the wave per 64 entries) -------------------------------
            for (int t0 = 0; t0 < total; t0 += 64) {
                const int t = t0 + lane;
                if (t < total) {
                    int r = 0;
                    while (r + 1 < nrun && (int) sRunP[wave][r + 1] <= t) r++;
                    const int a = (int) sRunA[wave][r];
                    const unsigned long long v = ix.list[(size_t) sRunS[wave][r] + (uint32_t) (t - (int) sRunP[wave][r])];
                    const int C = (int) (uint32_t) v;
                    const int mC = (int) ((v >> 32) & 511u), lenC = (int) ((v >> 41) & 511u);
                    const int s = a - mC;                                   // the only window C's prefix can match
                    // same minimizer, long enough for a prefix of length L (:215), not B itself (:386)
                    if (s >= 0 && s < nwin && (int) (kb[s] & 511u) == a && lenC >= Lspan - s && C != B) {
                        const uint32_t ci = atomicAdd(w.candN, 1u);
                        w.candC[ci] = (uint32_t) C; w.candW[ci] = (uint32_t) s;       // ci < CANDMAX: drained below before it can fill
                    }
                }
                wave_lds_fence();
                if ((int) __builtin_amdgcn_readfirstlane((int) *w.candN) > CANDMAX - 64) verify_candidates();
            }
        } else {
            // > 64 distinct minimizers in one tail (short windows on long reads): walk every k-mer position in turn
            for (int a = 0; a < nk; a++) {
                const uint64_t kv = kmer_at(sb, base_bit + 2 * a, mc.kmask);
                uint32_t slot = dir_slot(kv, ix.dmask);
                unsigned long long hit = 0;
                for (;;) {
                    const unsigned long long dk = ix.dir[2 * (size_t) slot];
                    if (STATS && lane == 0) st_slots++;
                    if (dk == kv) { hit = ix.dir[2 * (size_t) slot + 1]; break; }
                    if (dk == DIR_EMPTY) break;
                    slot = (slot + 1) & ix.dmask;
                }
                const uint32_t lstart = (uint32_t) (hit >> 32), lcount = (uint32_t) hit;
                for (uint32_t t0 = 0; t0 < lcount; t0 += 64) {
                    const uint32_t t = t0 + (uint32_t) lane;
                    if (t < lcount) {
                        const unsigned long long v = ix.list[(size_t) lstart + t];
                        const int C = (int) (uint32_t) v;
                        const int mC = (int) ((v >> 32) & 511u), lenC = (int) ((v >> 41) & 511u);
                        const int s = a - mC;
                        if (s >= 0 && s < nwin && (int) (kb[s] & 511u) == a && lenC >= Lspan - s && C != B) {
                            const uint32_t ci = atomicAdd(w.candN, 1u);
                            w.candC[ci] = (uint32_t) C; w.candW[ci] = (uint32_t) s;
                        }
                    }
                    wave_lds_fence();
                    if ((int) __builtin_amdgcn_readfirstlane((int) *w.candN) > CANDMAX - 64) verify_candidates();
                }
            }
        }
#if defined(ABLATE) && ABLATE == 13
        if (lane == 0) { *runN = 0; *w.candN = 0; }
        continue;
#endif
        verify_candidates();
        wave_lds_fence();
        if (lane == 0) *runN = 0;
#if defined(ABLATE) && ABLATE == 14
        continue;
#endif

        // per-source small-overlap cap: the reference keeps the LAST `SOES`=3 pushes in (L asc, C asc)
        // order (GraphCreatorPrefSuf.cpp:400-401) == the 3 largest (L, C) keys.
        uint64_t win0 = 0, win1 = 0, win2 = 0;
        int nwon = 0;
        {
            uint64_t m = wave_max_u64_dpp(k0);
            if (m) {
                if (k0 == m) { k0 = k1; k1 = k2; k2 = 0; }
                win0 = m; nwon = 1;
                m = wave_max_u64_dpp(k0);
                if (m) {
                    if (k0 == m) { k0 = k1; k1 = k2; k2 = 0; }
                    win1 = m; nwon = 2;
                    m = wave_max_u64_dpp(k0);
                    if (m) { win2 = m; nwon = 3; }
                }
            }
        }
        wave_lds_fence();
        int nbuf = (int) __builtin_amdgcn_readfirstlane((int) *w.recN);
        if (nbuf > WBUF) nbuf = WBUF;
        if (nbuf + nwon > WBUF) { flush_records<REC_CHUNK>(o, w, chunk_base, chunk_fill); nbuf = 0; }
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
        if (nbuf + nwon >= WFLUSH) flush_records<REC_CHUNK>(o, w, chunk_base, chunk_fill);
    }
    flush_records<REC_CHUNK>(o, w, chunk_base, chunk_fill);
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
// launchers
// ------------------------------------------------------------------------------------------
static MinCfg make_mincfg(const PrefSufCfg &cfg) {
    MinCfg mc;
    mc.kk = std::min(cfg.Lmin, MIN_KMAX);
    mc.w = cfg.Lmin - mc.kk + 1;
    mc.kmask = mc.kk >= 32 ? ~0ull : ((1ull << (2 * mc.kk)) - 1ull);
    return mc;
}

int minimizer_key_bits(const PrefSufCfg &cfg) { return 2 * std::min(cfg.Lmin, MIN_KMAX) + 1; }   // +1: the invalid key sorts last

void launch_index_targets(const NodesDev &nd, const PrefSufCfg &cfg, unsigned long long *keys, unsigned long long *vals, hipStream_t s) {
    if (nd.n <= 0) return;
    hipLaunchKernelGGL(k_index_targets, dim3((nd.n + 255) / 256), dim3(256), 0, s, nd, cfg, make_mincfg(cfg), keys, vals);
}

void launch_index_count(const unsigned long long *keys, uint64_t n, unsigned long long *out, hipStream_t s) {
    if (n == 0) return;
    unsigned g = (unsigned) std::min<uint64_t>((n + 255) / 256, 2048);
    hipLaunchKernelGGL(k_index_count, dim3(g), dim3(256), 0, s, keys, n, out);
}

uint32_t index_directory_slots(uint64_t distinct) {
    uint64_t s = 64;
    while (s < 2 * distinct) s <<= 1;
    return (uint32_t) std::min<uint64_t>(s, 1ull << 31);
}

void launch_index_directory(const unsigned long long *keys, uint64_t n, unsigned long long *dir, uint32_t slots, hipStream_t s) {
    if (n == 0) return;
    unsigned g = (unsigned) std::min<uint64_t>((n + 255) / 256, 8192);
    hipLaunchKernelGGL(k_index_directory, dim3(g), dim3(256), 0, s, keys, n, dir, slots - 1);
}

static uint64_t probe_blocks_min(int n_cu, uint64_t n_src) {
    return std::max<uint64_t>(1, std::min<uint64_t>((n_src + PROBE_WAVES - 1) / PROBE_WAVES, (uint64_t) std::max(1, n_cu) * 8));
}

template <int NQ>
static void launch_probe_min_nq(const NodesDev &nd, const PrefSufCfg &cfg, const MinCfg &mc, const MinIndex &ix, int32_t src_begin, int32_t src_end,
                                const ProbeOut &o, dim3 grid, dim3 block, hipStream_t s) {
    if (cfg.stats) hipLaunchKernelGGL((k_probe_min<true, NQ>), grid, block, 0, s, nd, cfg, mc, ix, src_begin, src_end, o);
    else           hipLaunchKernelGGL((k_probe_min<false, NQ>), grid, block, 0, s, nd, cfg, mc, ix, src_begin, src_end, o);
}

void launch_probe_min(const NodesDev &nd, const PrefSufCfg &cfg, const unsigned long long *dir, uint32_t dir_slots, const unsigned long long *list,
                      int32_t src_begin, int32_t src_end, uint32_t *rec_dst, unsigned long long *rec_val, uint64_t rec_cap,
                      unsigned long long *counters, int n_cu, hipStream_t s) {
    const int64_t ns = (int64_t) src_end - src_begin;
    if (ns <= 0) return;
    dim3 grid((unsigned) probe_blocks_min(n_cu, (uint64_t) ns)), block(PROBE_WAVES * 64);
    ProbeOut o{rec_dst, rec_val, rec_cap, counters};
    MinIndex ix{dir, list, dir_slots - 1};
    const MinCfg mc = make_mincfg(cfg);
    const int need_q = (((2 * cfg.Lcap + 31) >> 5) + 3) >> 2;
    const bool aligned = (nd.stride & 3) == 0 && ((uintptr_t) nd.words & 15u) == 0;
    const int row_q = nd.stride >> 2;
    if (aligned && need_q <= 2 && row_q >= 2)      launch_probe_min_nq<2>(nd, cfg, mc, ix, src_begin, src_end, o, grid, block, s);
    else if (aligned && need_q <= 3 && row_q >= 3) launch_probe_min_nq<3>(nd, cfg, mc, ix, src_begin, src_end, o, grid, block, s);
    else if (aligned && need_q <= 4 && row_q >= 4) launch_probe_min_nq<4>(nd, cfg, mc, ix, src_begin, src_end, o, grid, block, s);
    else                                           launch_probe_min_nq<0>(nd, cfg, mc, ix, src_begin, src_end, o, grid, block, s);
}

} // namespace alga
