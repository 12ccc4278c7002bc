// alga_amd/csrc/prefsuf_cluster.hip -- the probe of the PrefSuf engine as a CLUSTERED MINIMIZER JOIN (gfx950).
//
// Replaces the per-overlap-length hash join of src/GraphCreators/GraphCreatorPrefSuf.cpp:238-395 (reference paths relative
// to its root) for inputs past the on-die caches.  The seed-table probe (prefsuf_kernels.hip) touches, per 150-bp source,
// 63 random prefilter words, ~16 random 64-byte bucket lines and ~11 random 64-byte row lines; once table and rows leave
// the 256 MB Infinity Cache it waits on those lines.  Here the join is made local:
//
//   * a target C is filed under the MINIMIZER of its min_overlap-long prefix: the k-mer (k = Lmin - w + 1, w <= 64 k-mers
//     per window) with the smallest 24-bit order hash, ties to the left.  If the suffix window at offset p of a source B
//     equals that prefix, window p of B has the same minimizer, at position p + m_C;
//   * all targets are SORTED by the 32-bit hash of their minimizer k-mer and their rows are copied, in that order, into
//     one array of 16*EQ-byte entries {row words, node id, hash, m_C | len | alignFrom}: the targets a source can overlap
//     through one minimizer are CONTIGUOUS in HBM (one genomic locus: ~6 entries at 30x coverage), found through a
//     direct-address index on the top bits of the hash;
//   * a source has ~2 * 63 / (w + 1) + 1 = 3 distinct window minimizers instead of 63 windows to look up: three index
//     reads and three contiguous runs of entries, ONE LANE PER ENTRY (up to 64 entries per round): hash / offset / length
//     checks from the entry's own words, exact 2-bit compare of the entry's row against the source's staged tail,
//     verified overlaps become the items of the source-side transitive reduction (prefsuf_device.h local_reduce).
//
//   k_tgt_keys      prefix minimizer of every target -> sort key (hash), meta word
//   (radix sort by hash: rocPRIM)
//   k_tgt_gather    rows in hash order -> entry array;   k_tgt_index   first entry of every hash bucket
//   k_probe_clustered   persistent wavefronts, one source at a time (sliding-window minimum by doubling over ds_bpermute)
#include <hip/hip_runtime.h>
#include <algorithm>
#include "prefsuf_common.h"
#include "prefsuf_kernels.h"
#include "prefsuf_device.h"

namespace alga {

// order hash of a k-mer (lo: nucleotides 0..15, hi: 16..31, both masked to the k-mer's length): its top 24 bits rank the
// k-mers of a window
__device__ __forceinline__ uint32_t kmer_hash(uint32_t lo, uint32_t hi) {
    uint32_t x = (lo ^ (hi * 0x85EBCA6Bu)) * 0x9E3779B1u;
    x ^= x >> 16;
    x *= 0x2C1B3C6Du;
    return x;
}

// Cluster key of a minimizer = a second, bijective mix of its order hash.  The order hashes of MINIMIZERS are minima of w
// uniform values -- concentrated near zero -- so bucketing the entry array by their own top bits would put most clusters in
// 1/w of the buckets; the mix spreads them evenly.  0xFFFFFFFF is reserved for "not a target".
__device__ __forceinline__ uint32_t cluster_key(uint32_t h) {
    uint32_t k = h * 0x9E3779B1u;
    k ^= k >> 15;
    k *= 0x85EBCA77u;
    k ^= k >> 13;
    k *= 0xC2B2AE3Du;
    return k == 0xFFFFFFFFu ? 0xFFFFFFFEu : k;
}

// k-mer starting at nucleotide i of a 2-bit row (words readable up to index (2i >> 5) + 2): its hash and its packed
// order key (24-bit order | position); the smallest key of a window is the window's minimizer
__device__ __forceinline__ void kmer_key(const uint32_t *row, int i, bool valid, const ClusterCfg &cc, uint32_t &h, uint32_t &pk) {
    const int bit = 2 * i, q = bit >> 5, r = bit & 31;
    const uint32_t x0 = row[q], x1 = row[q + 1], x2 = row[q + 2];
    h = kmer_hash(funnel(x0, x1, r) & cc.lo_mask, funnel(x1, x2, r) & cc.hi_mask);
    pk = valid ? ((h & 0xFFFFFF00u) | (uint32_t) i) : 0xFFFFFFFFu;
}

// ------------------------------------------------------------------------------------------
// build: keys, gather, index
// ------------------------------------------------------------------------------------------
constexpr int TK_ROWS = 256;         // targets per workgroup of k_tgt_keys
constexpr int TK_WORDS = 16;         // row words staged per target (Lmin <= 208: the prefix window ends inside word 13)
constexpr int TK_STRIDE = TK_WORDS + 1;

// one thread per node: keys[i] = cluster key of the minimizer of C[0, Lmin) (all ones: not a target), vals[i] = i,
// meta[i] = m_C | len << 8 | alignFrom << 20.  Rows are staged through LDS (coalesced 64-byte pieces in, conflict-free out).
__global__ void __launch_bounds__(TK_ROWS) k_tgt_keys(NodesDev nd, PrefSufCfg cfg, ClusterCfg cc, uint32_t *__restrict__ keys,
                                                       uint32_t *__restrict__ vals, uint32_t *__restrict__ meta) {
    __shared__ uint32_t s[TK_ROWS][TK_STRIDE];
    const int base = blockIdx.x * TK_ROWS;
    const int nrows = min(TK_ROWS, nd.n - base);
    {
        const int c = (int) (threadIdx.x & 15u);
        for (int r = (int) (threadIdx.x >> 4); r < nrows; r += TK_ROWS / 16)
            s[r][c] = c < nd.stride ? nd.words[(size_t) (base + r) * nd.stride + c] : 0u;
    }
    __syncthreads();
    const int t = (int) threadIdx.x;
    if (t >= nrows) return;
    const int i = base + t;
    const int len = nd.len[i];
    uint32_t key = 0xFFFFFFFFu, m = 0u;
    if (len >= cfg.Lmin && len > 0 && (!nd.to || nd.to[i])) {
        uint32_t best = 0xFFFFFFFFu, besth = 0u;
        for (int j = 0; j < cc.w; j++) {
            uint32_t h, pk;
            kmer_key(s[t], j, true, cc, h, pk);
            if (pk < best) { best = pk; besth = h; }
        }
        key = cluster_key(besth);
        m = (best & 255u) | ((uint32_t) len << 8) | ((!nd.from || nd.from[i]) ? CL_META_FROM : 0u);
    }
    keys[i] = key; vals[i] = (uint32_t) i; meta[i] = m;
}

// entry j (hash order) = {row words 0 .. 4*EQ-4, node id, hash, meta}; one thread per 16-byte piece
template <int EQ>
__global__ void __launch_bounds__(256) k_tgt_gather(NodesDev nd, const uint32_t *__restrict__ keys, const uint32_t *__restrict__ vals,
                                                     const uint32_t *__restrict__ meta, uint4 *__restrict__ store) {
    const uint64_t t = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t j = t / EQ;
    const int c = (int) (t % EQ);
    if (j >= (uint64_t) nd.n) return;
    const uint32_t key = keys[j];
    if (key == 0xFFFFFFFFu) return;                        // not a target: behind the last entry
    const uint32_t id = vals[j];
    const uint32_t *row = nd.words + (size_t) id * nd.stride;
    uint4 v;
    if ((nd.stride & 3) == 0 && 4 * c + 3 < nd.stride && ((uintptr_t) nd.words & 15u) == 0) v = reinterpret_cast<const uint4 *>(row)[c];
    else {
        v.x = 4 * c + 0 < nd.stride ? row[4 * c + 0] : 0u; v.y = 4 * c + 1 < nd.stride ? row[4 * c + 1] : 0u;
        v.z = 4 * c + 2 < nd.stride ? row[4 * c + 2] : 0u; v.w = 4 * c + 3 < nd.stride ? row[4 * c + 3] : 0u;
    }
    if (c == EQ - 1) { v.y = id; v.z = key; v.w = meta[id]; }
    store[j * EQ + c] = v;
}

// idx[b] = first entry whose hash bucket is >= b, for b in [0, n_buckets]; non-targets (all-ones keys) count as bucket n_buckets
__global__ void __launch_bounds__(256) k_tgt_index(const uint32_t *__restrict__ keys, uint64_t n, int shift, uint32_t n_buckets,
                                                    uint32_t *__restrict__ idx) {
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += (uint64_t) gridDim.x * blockDim.x) {
        auto bucket = [&](uint64_t k) -> int64_t { const uint32_t x = keys[k]; return x == 0xFFFFFFFFu ? (int64_t) n_buckets : (int64_t) (x >> shift); };
        const int64_t lo = i == 0 ? 0 : bucket(i - 1) + 1;
        const int64_t hi = i == n ? (int64_t) n_buckets : bucket(i);
        for (int64_t b = lo; b <= hi && b <= (int64_t) n_buckets; b++) idx[b] = (uint32_t) i;
    }
}

// ------------------------------------------------------------------------------------------
// k_probe_clustered
// ------------------------------------------------------------------------------------------
#ifndef CL_OCC
#define CL_OCC 4
#endif
__device__ __forceinline__ uint32_t bperm(uint32_t v, int src_lane) { return (uint32_t) __builtin_amdgcn_ds_bpermute(src_lane << 2, (int) v); }

// Two-stage software pipeline over the sources of a wave: while the entry loads of source i are in flight the wave stages
// source i+1, computes its window minimizers (six dependent ds_bpermute steps) and issues its index loads; those land while
// source i is verified and reduced.  State of a source between the stages: its staged row, per-window minimizer positions and
// run list in LDS (two buffers), four uniform scalars.
template <bool STATS, int EQ, int KF>
__global__ void __launch_bounds__(PROBE_WAVES * 64, CL_OCC)
k_probe_clustered(NodesDev nd, PrefSufCfg cfg, ClusterCfg cc, const uint4 *__restrict__ store, const uint32_t *__restrict__ idx,
                  int32_t src_begin, int32_t src_end, ProbeOut o) {
    constexpr int WC = 4 * EQ - 3;                         // row words of an entry
    __shared__ uint32_t sB[PROBE_WAVES][2][STAGE_WORDS];
    __shared__ uint32_t sWm[PROBE_WAVES][2][64];           // minimizer position of window p
    __shared__ uint4 sRun[PROBE_WAVES][2][64];             // distinct minimizers of a source: position, cluster key, first entry, entries
    __shared__ uint32_t sRecC[PROBE_WAVES][WBUF_LOCAL];
    __shared__ unsigned long long sRecV[PROBE_WAVES][WBUF_LOCAL];
    __shared__ uint32_t sCnt[PROBE_WAVES][3];
    __shared__ uint32_t sItemC[PROBE_WAVES][ITEMMAX];
    __shared__ uint32_t sItemM[PROBE_WAVES][ITEMMAX];
    __shared__ uint4 sItemO[PROBE_WAVES][ITEMMAX];
    __shared__ uint8_t sItemT[PROBE_WAVES][64];
    const int wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));      // scalar: the source stream below is wave-uniform
    const int lane = lane_id();
    WaveLds w{sB[wave][0], nullptr, nullptr, &sCnt[wave][0], sRecC[wave], sRecV[wave], &sCnt[wave][1]};
    ItemLds it{sItemC[wave], sItemM[wave], sItemO[wave], sItemT[wave], &sCnt[wave][2]};
    if (lane == 0) { *w.recN = 0; *it.N = 0; }
    uint64_t chunk_base = 0;
    int chunk_fill = REC_CHUNK_LOCAL;                      // "no chunk yet"
    uint64_t st_raw = 0, st_slots = 0, st_win = 0, st_rec = 0, st_cmp = 0, st_generic = 0;
    const int total_waves = (int) gridDim.x * PROBE_WAVES;
    const int grp = lane >> 4, sl = lane & 15;
    const int kfull = KF ? KF : (2 * cfg.Lmin) >> 5;      // row words every overlap covers entirely

    // ---- source stream (scalar bookkeeping): the row, length and mask of the source after the one being staged are always in
    //      flight.  Node ids are < 2^31 (alga_nodes.n is an int32).
    const int pre_words = nd.stride < STAGE_WORDS ? nd.stride : STAGE_WORDS;
    int Bl = src_begin + (int) blockIdx.x * PROBE_WAVES + wave;
    int n_len = 0; uint32_t n_word = 0; int n_from = 1;
    auto fetch = [&]() {
        if (Bl < src_end) {                                // uniform
            n_len = nd.len[Bl];
            n_word = lane < pre_words ? nd.words[(size_t) Bl * nd.stride + lane] : 0u;
            if (nd.from) n_from = nd.from[Bl];
        }
    };
    fetch();
    // next source that takes part (long enough, alignFrom): uniform
    auto advance = [&](int &B, int &lenB, uint32_t &word0) -> bool {
        while (Bl < src_end) {
            B = Bl; lenB = __builtin_amdgcn_readfirstlane(n_len); word0 = n_word;
            const bool from_ok = __builtin_amdgcn_readfirstlane(n_from) != 0;
            Bl = total_waves <= src_end - Bl ? Bl + total_waves : src_end;      // no overflow near 2^31
            fetch();
            if (lenB >= cfg.Lmin && lenB > 0 && from_ok) return true;
        }
        return false;
    };
    // stage 1 of a source: row -> LDS, window minimizers, distinct runs, index loads issued (e0 / e1 are NOT waited for here)
    auto minimizers = [&](int buf, int lenB, uint32_t word0, int &nrun, bool &start, int &rank, uint32_t &q_out, uint32_t &key_out,
                          uint32_t &bucket) {
        uint32_t *sb = sB[wave][buf];
        const int nwB = blocks_of(lenB);
        wave_lds_fence();
        if (lane < STAGE_WORDS) sb[lane] = lane < nwB ? word0 : 0u;
        wave_lds_fence();
        const int nwin = lenB - cfg.Lmin + 1;              // overlap lengths Lmin..lenB <-> offsets p = 0..nwin-1 (<= 64)
        const int nk = lenB - cc.kk + 1;                   // k-mers of the source (<= 127)
        uint32_t h0, h1, a0, a1;
        kmer_key(sb, lane, lane < nk, cc, h0, a0);
        kmer_key(sb, lane + 64, lane + 64 < nk, cc, h1, a1);
        for (int j = 0, s = 1; j < cc.J; j++, s <<= 1) {   // a[i] = min key[i .. i + 2s - 1]
            const int src = (lane + s) & 63;
            const uint32_t t0 = bperm(a0, src), t1 = bperm(a1, src);
            const bool wrap = lane + s >= 64;
            const uint32_t u0 = wrap ? t1 : t0, u1 = wrap ? 0xFFFFFFFFu : t1;
            a0 = a0 < u0 ? a0 : u0;
            a1 = a1 < u1 ? a1 : u1;
        }
        uint32_t wm = a0;                                  // window p = lane: k-mers [p, p + w)
        if (cc.wrest) {
            const int src = (lane + cc.wrest) & 63;
            const uint32_t t0 = bperm(a0, src), t1 = bperm(a1, src);
            const uint32_t u = lane + cc.wrest >= 64 ? t1 : t0;
            wm = wm < u ? wm : u;
        }
        // distinct minimizers: a window starts a run when its minimizer differs from the previous window's
        const bool wv = lane < nwin;
        const uint32_t prev = bperm(wm, (lane + 63) & 63);
        start = wv && (lane == 0 || wm != prev);
        const uint64_t runmask = __ballot(start);
        nrun = __popcll(runmask);
        const int q = (int) (wm & 255u);                   // k-mer position of window p's minimizer
        const uint32_t g0 = bperm(h0, q & 63), g1 = bperm(h1, q & 63);
        sWm[wave][buf][lane] = wv ? (uint32_t) q : 0xFFFFu;
        q_out = (uint32_t) q;
        rank = (int) __builtin_amdgcn_mbcnt_hi((uint32_t) (runmask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) runmask, 0u));
        key_out = cluster_key(q >= 64 ? g1 : g0);
        bucket = start ? key_out >> cc.idx_shift : 0u;     // lanes that start no run read bucket 0 (one shared line)
    };
    // The index loads of stage 1 are issued by EVERY lane and outside any branch: a load under a branch leaves the number of
    // loads in flight unknown to the compiler, which then drains ALL of them where the entries are first used -- and that
    // serialises the two stages of the pipeline.
    auto index_loads = [&](uint32_t bucket, uint32_t &e0, uint32_t &e1) { e0 = idx[bucket]; e1 = idx[bucket + 1]; };
    auto finish_runs = [&](int buf, bool start, int rank, uint32_t q, uint32_t key, uint32_t e0, uint32_t e1) {
        if (start) sRun[wave][buf][rank] = make_uint4(q, key, e0, e1 - e0);
        wave_lds_fence();
    };
    // runs rb .. rb+3 of a source: 16 lanes each; uniform maximum of their entry counts
    auto load_runs = [&](int buf, int rb, int nrun, uint4 &rp, uint32_t &mc) {
        const int r = rb + grp;
        rp = make_uint4(0u, 0u, 0u, 0u);
        if (r < nrun) rp = sRun[wave][buf][r];
        mc = (uint32_t) __builtin_amdgcn_readlane((int) rp.w, 0);
        const uint32_t c1 = (uint32_t) __builtin_amdgcn_readlane((int) rp.w, 16), c2 = (uint32_t) __builtin_amdgcn_readlane((int) rp.w, 32),
                       c3 = (uint32_t) __builtin_amdgcn_readlane((int) rp.w, 48);
        mc = mc > c1 ? mc : c1; mc = mc > c2 ? mc : c2; mc = mc > c3 ? mc : c3;
    };
    // one entry per lane: entry k0 + (lane & 15) of the lane group's run
    auto load_entries = [&](const uint4 &rp, uint32_t k0, bool &ev, size_t &ei, uint32_t (&ew)[4 * EQ]) {
        const uint32_t j = k0 + (uint32_t) sl;
        ev = j < rp.w;
        ei = ev ? (size_t) rp.z + j : (size_t) 0;           // lanes without an entry read entry 0 (unconditional loads: see minimizers)
#pragma unroll
        for (int c = 0; c < EQ; c++) { const uint4 v = store[ei * EQ + c]; ew[4 * c] = v.x; ew[4 * c + 1] = v.y; ew[4 * c + 2] = v.z; ew[4 * c + 3] = v.w; }
    };

    int B = 0, lenB = 0, nrun = 0;
    uint32_t word0 = 0;
    bool have = advance(B, lenB, word0);
    int buf = 0;
    if (have) {
        bool start; int rank; uint32_t q, key, bk, e0, e1;
        minimizers(0, lenB, word0, nrun, start, rank, q, key, bk);
        index_loads(bk, e0, e1);
        finish_runs(0, start, rank, q, key, e0, e1);
    }
    while (have) {                                         // uniform
        const uint32_t *sb = sB[wave][buf];
        const uint32_t *wm_lds = sWm[wave][buf];
        const int nwin = lenB - cfg.Lmin + 1;
        if (STATS && lane == 0) st_win += (uint64_t) nwin;
        // ---- (1) pull the next source off the stream BEFORE any entry load is issued: its loop must not sit between the
        //          loads and their use (hipcc drains vmcnt at loop headers) ----
        int nB = 0, nlenB = 0, nnrun = 0, nrank = 0;
        uint32_t nword0 = 0, nq = 0, nkey = 0, nbk = 0, ne0, ne1;
        bool nstart = false;
        const bool have_next = advance(nB, nlenB, nword0);
        // ---- (2) first batch of this source's entries: loads issued ----
        int rb = 0;
        uint32_t k0 = 0, mc;
        uint4 rp;
        load_runs(buf, rb, nrun, rp, mc);
        bool ev; size_t ei; uint32_t ew[4 * EQ];
        load_entries(rp, k0, ev, ei, ew);
        // ---- (3) the next source: minimizers, index loads issued behind the entry loads ----
        if (have_next) minimizers(buf ^ 1, nlenB, nword0, nnrun, nstart, nrank, nq, nkey, nbk);
        index_loads(nbk, ne0, ne1);
        // ---- (4) verify: one entry per lane, four runs per batch ----
        int n_items = 0;                                   // verified overlaps of this source so far (uniform)
        auto verify = [&]() {
            if (STATS && ev) st_slots++;
            const uint32_t id = ew[4 * EQ - 3], eh = ew[4 * EQ - 2], meta = ew[4 * EQ - 1];
            const int lenC = (int) ((meta >> 8) & 0xFFFu);
            int p = (int) rp.x - (int) (meta & 255u);      // the only offset at which C's prefix can sit in B
            // same minimizer k-mer, an offset of B, not B itself (GraphCreatorPrefSuf.cpp:386)
            bool ok = ev && eh == rp.y && p >= 0 && p < nwin && (int) id != B;
            p = ok ? p : 0;
            // window p has THIS minimizer; C is long enough for a prefix of length L = |B| - p (:215)
            ok = ok && wm_lds[p] == rp.x && lenC >= lenB - p;
            const int L = lenB - p, nb = 2 * L;
            const int qw = (2 * p) >> 5, sh = (2 * p) & 31;
            uint32_t y[WC + 1];
#pragma unroll
            for (int k = 0; k <= WC; k++) y[k] = sb[qw + k];
            uint32_t diff = 0;
#pragma unroll
            for (int k = 0; k < WC; k++) {                 // exact compare C[0, L) == B[p, p + L)
                const uint32_t x = funnel(y[k], y[k + 1], sh) ^ ew[k];
                if (k < kfull) diff |= x;                  // uniform (compile time with KF)
                else diff |= x & low_bits32(nb - 32 * k);
            }
            const bool pass = ok && diff == 0;
            const uint64_t pm = __ballot(pass);
            if (pm != 0ull) {                              // uniform
                if (pass) {
                    if (STATS) st_raw++;
                    const int slot = n_items + (int) __builtin_amdgcn_mbcnt_hi((uint32_t) (pm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) pm, 0u));
                    if (slot < ITEMMAX) {
                        it.C[slot] = id;
                        it.M[slot] = (uint32_t) p | ((uint32_t) lenC << 9) | ((meta & CL_META_FROM) ? ITEM_FROM : 0u);
                        // overhang: what C adds to the right of B's end = C's row from bit 2L on; bits past C's own end are
                        // never compared (prefsuf_device.h via_ok), so the entry's trailing words may stand in for zeros
                        const int ws = nb >> 5, r2 = nb & 31;
                        uint32_t x[5];
                        if constexpr (KF > 0 && WC - KF <= 6) {
                            // word ws + k of the entry for ws in [KF, WC]: a select over registers
                            const int t = ws - KF;
#pragma unroll
                            for (int k = 0; k < 5; k++) {
                                uint32_t v = 0u;
#pragma unroll
                                for (int u = 0; u <= WC - KF; u++) { const int wi = KF + k + u; if (wi < 4 * EQ) v = t == u ? ew[wi] : v; }
                                x[k] = v;
                            }
                        } else {
                            const uint32_t *er = reinterpret_cast<const uint32_t *>(store + ei * EQ);      // re-read (an L1 hit)
#pragma unroll
                            for (int k = 0; k < 5; k++) x[k] = er[ws + k];
                        }
                        it.O[slot] = make_uint4(funnel(x[0], x[1], r2), funnel(x[1], x[2], r2), funnel(x[2], x[3], r2), funnel(x[3], x[4], r2));
                    }
                }
                n_items += __popcll(pm);
            }
        };
        verify();                                          // the first batch, peeled: its wait covers the entry loads only
        for (;;) {                                         // further batches: more than 16 entries in a run, more than four runs
            k0 += 16;
            if (k0 >= mc) {                                // uniform
                rb += 4;
                if (rb >= nrun) break;
                k0 = 0;
                load_runs(buf, rb, nrun, rp, mc);
            }
            load_entries(rp, k0, ev, ei, ew);
            verify();
        }
        wave_lds_fence();
        // ---- (5) transitive reduction at the source, edges out ----
        if (n_items > ITEMMAX) {
            // the source goes on the list of the second pass (k_probe_sources, BIG instantiation); a full list: per-target pipeline
            if (lane == 0) {
                const unsigned long long k = atomicAdd(&o.counters[CNT_LOCAL_OVERFLOW], 1ull);
                if (k < (unsigned long long) o.big_list_cap) o.big_list[k] = B;
                atomicMax(&o.counters[CNT_LOCAL_MAXITEMS], (unsigned long long) n_items);
                if (STATS) { st_raw -= (uint64_t) n_items; st_win -= (uint64_t) nwin; }     // the second pass counts this source
            }
        } else if (n_items > 0) {
            local_reduce<STATS, WBUF_LOCAL, 1>(nd, cfg, it, w, o, B, lenB, n_items, st_rec, st_cmp, st_generic);
        }
        const int nb2 = (int) __builtin_amdgcn_readfirstlane((int) *w.recN);
        if (nb2 >= WFLUSH_LOCAL) flush_records<REC_CHUNK_LOCAL, WBUF_LOCAL>(o, w, chunk_base, chunk_fill);
        // ---- (6) the next source's index loads have had the time of (4) and (5) to land ----
        if (have_next) finish_runs(buf ^ 1, nstart, nrank, nq, nkey, ne0, ne1);
        have = have_next; B = nB; lenB = nlenB; nrun = nnrun; buf ^= 1;
    }
    flush_records<REC_CHUNK_LOCAL, WBUF_LOCAL>(o, w, chunk_base, chunk_fill);
    close_chunk<REC_CHUNK_LOCAL>(o, chunk_base, chunk_fill);
    st_rec = wave_sum_u64(st_rec);
    if (lane == 0 && st_rec) atomicAdd(&o.counters[CNT_VALID_RECORDS], (unsigned long long) st_rec);
    if (STATS) {
        st_raw = wave_sum_u64(st_raw); st_slots = wave_sum_u64(st_slots); st_win = wave_sum_u64(st_win);
        st_cmp = wave_sum_u64(st_cmp); st_generic = wave_sum_u64(st_generic);
        if (lane == 0) {
            atomicAdd(&o.counters[CNT_RAW], (unsigned long long) st_raw);
            atomicAdd(&o.counters[CNT_SLOTS], (unsigned long long) st_slots);
            atomicAdd(&o.counters[CNT_WINDOWS], (unsigned long long) st_win);
            atomicAdd(&o.counters[CNT_TR_COMPARES], (unsigned long long) st_cmp);
            atomicAdd(&o.counters[CNT_LOCAL_GENERIC], (unsigned long long) st_generic);
        }
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
int cluster_entry_quads(int max_len) {                    // 16-byte pieces per entry: row words + 3; 0 = rows too long for this path
    const int W = blocks_of(max_len);
    const int eq = (W + 3 + 3) / 4;
    return eq < 2 ? 2 : (eq <= CL_MAX_EQ ? eq : 0);
}

ClusterCfg cluster_cfg(const PrefSufCfg &cfg, uint64_t live, int bucket_log2_bias) {
    ClusterCfg c;
    c.kk = std::max(cfg.Lmin - 63, std::min(cfg.Lmin, CL_KMIN));
    c.w = cfg.Lmin - c.kk + 1;
    c.J = 0;
    while ((2 << c.J) <= c.w) c.J++;
    c.wrest = c.w - (1 << c.J);
    c.lo_mask = c.kk >= 16 ? 0xFFFFFFFFu : ((1u << (2 * c.kk)) - 1u);
    c.hi_mask = c.kk <= 16 ? 0u : (c.kk >= 32 ? 0xFFFFFFFFu : ((1u << (2 * c.kk - 32)) - 1u));
    int bits = 4;
    while (bits < 28 && (1ull << bits) < live) bits++;     // ~one entry per bucket: a lookup returns its cluster and little else
    bits = std::max(4, std::min(30, bits + bucket_log2_bias));
    c.n_buckets = 1u << bits;
    c.idx_shift = 32 - bits;
    return c;
}

size_t cluster_sort_temp_bytes(uint64_t n) { return sort_u32_pairs_temp_bytes(n); }

hipError_t launch_cluster_build(const NodesDev &nd, const PrefSufCfg &cfg, const ClusterCfg &cc, int eq, uint32_t *keys, uint32_t *vals,
                                uint32_t *keys2, uint32_t *vals2, uint32_t *meta, void *sort_temp, size_t sort_temp_bytes, void *store,
                                uint32_t *idx, hipStream_t s) {
    if (nd.n <= 0) return hipSuccess;
    const uint64_t n = (uint64_t) nd.n;
    hipLaunchKernelGGL(k_tgt_keys, dim3((unsigned) ((n + TK_ROWS - 1) / TK_ROWS)), dim3(TK_ROWS), 0, s, nd, cfg, cc, keys, vals, meta);
    hipError_t err = sort_u32_pairs(sort_temp, sort_temp_bytes, keys, keys2, vals, vals2, n, s);
    if (err != hipSuccess) return err;
    const uint64_t pieces = n * (uint64_t) eq;
    const unsigned g = (unsigned) ((pieces + 255) / 256);
    if (eq == 2)      hipLaunchKernelGGL(k_tgt_gather<2>, dim3(g), dim3(256), 0, s, nd, (const uint32_t *) keys2, (const uint32_t *) vals2, (const uint32_t *) meta, (uint4 *) store);
    else if (eq == 3) hipLaunchKernelGGL(k_tgt_gather<3>, dim3(g), dim3(256), 0, s, nd, (const uint32_t *) keys2, (const uint32_t *) vals2, (const uint32_t *) meta, (uint4 *) store);
    else              hipLaunchKernelGGL(k_tgt_gather<4>, dim3(g), dim3(256), 0, s, nd, (const uint32_t *) keys2, (const uint32_t *) vals2, (const uint32_t *) meta, (uint4 *) store);
    hipLaunchKernelGGL(k_tgt_index, dim3((unsigned) std::min<uint64_t>((n + 256) / 256, 16384)), dim3(256), 0, s, (const uint32_t *) keys2, n, cc.idx_shift,
                       cc.n_buckets, idx);
    return hipGetLastError();
}

uint64_t cluster_probe_blocks(int n_cu, uint64_t n_src) {
    return std::max<uint64_t>(1, std::min<uint64_t>((n_src + PROBE_WAVES - 1) / PROBE_WAVES, (uint64_t) std::max(1, n_cu) * CL_OCC));
}

uint64_t cluster_record_slack(int n_cu, uint64_t n_src) { return cluster_probe_blocks(n_cu, n_src) * PROBE_WAVES * (uint64_t) REC_CHUNK_LOCAL; }

void launch_probe_clustered(const NodesDev &nd, const PrefSufCfg &cfg, const ClusterCfg &cc, int eq, const void *store, const uint32_t *idx,
                            int32_t src_begin, int32_t src_end, uint32_t *rec_dst, unsigned long long *rec_val, uint64_t rec_cap,
                            unsigned long long *counters, int n_cu, uint32_t *deg, unsigned long long *first, const ProbeBig *big, hipStream_t s) {
    const int64_t ns = (int64_t) src_end - src_begin;
    if (ns <= 0) return;
    dim3 grid((unsigned) cluster_probe_blocks(n_cu, (uint64_t) ns)), block(PROBE_WAVES * 64);
    ProbeOut o{rec_dst, rec_val, rec_cap, counters, deg, first, src_begin};
    if (big) { o.big_list = big->list; o.big_list_cap = big->list_cap; }
    const uint4 *st = (const uint4 *) store;
    // KF = (2 * Lmin) >> 5 as a compile-time constant for the shapes ALGA's defaults produce (150-bp reads: Lmin 82, rows of 9
    // words; 100-bp reads: Lmin 55, rows of 6 words); 0 = any shape
    const int kf = (2 * cfg.Lmin) >> 5;
#define CL_LAUNCH(ST, E, K) hipLaunchKernelGGL((k_probe_clustered<ST, E, K>), grid, block, 0, s, nd, cfg, cc, st, idx, src_begin, src_end, o)
#define CL_STATS(E, K) do { if (cfg.stats) CL_LAUNCH(true, E, K); else CL_LAUNCH(false, E, K); } while (0)
    if (eq == 3 && kf == 5)      CL_STATS(3, 5);
    else if (eq == 3 && kf == 3) CL_STATS(3, 3);
    else if (eq == 2 && kf == 3) CL_STATS(2, 3);
    else if (eq == 2)            CL_STATS(2, 0);
    else if (eq == 3)            CL_STATS(3, 0);
    else                         CL_STATS(4, 0);
#undef CL_STATS
#undef CL_LAUNCH
}

} // namespace alga
