// alga_amd/csrc/prefsuf_cluster_device.h -- device-side definitions of the clustered minimizer join shared by prefsuf_cluster.hip (one
// GPU: the index and the source-side probe kernels) and prefsuf_shard.hip (N GPUs: the bucket-sharded join): k-mer order, cluster key,
// sort key of a target, directory record -> entry slice.
#pragma once
#include <hip/hip_runtime.h>
#include "prefsuf_common.h"
#include "../../include/alga_amd.h"
#include "prefsuf_device.h"

namespace alga {

// order hash of a k-mer (lo: nucleotides 0..15, hi: 16..31, both masked to the k-mer's length)
__device__ __forceinline__ uint32_t kmer_hash(uint32_t lo, uint32_t hi) {
    // ONE multiply (32-bit integer multiplies run at quarter rate).  A second mixing round made no measurable difference to the
    // minimizers' statistics (runs per node, entries per source, sources the pair kernel finishes): what has to be well mixed is
    // the CLUSTER key, and that gets its own mix, once per run.
    return (lo ^ __funnelshift_l(hi, hi, 13) ^ (hi >> 7)) * 0x9E3779B1u;
}

// ORDER of the k-mers of a window: the smallest order key is the window's minimizer.
//   bit 31      content class: 0 for a k-mer that starts with A followed by C or G (one k-mer in eight), 1 for any other
//   bits 30..8  23 bits of the order hash
//   bits 7..0   position in the read (ties to the left; equal windows of a source and a target agree on it)
// The class bit is what makes k_node_runs cheap: a window of w = 64 k-mers holds a class-0 k-mer with probability 1 - (7/8)^64,
// so the minimizer of (nearly) every window is a class-0 k-mer, and those are found with a few word-parallel bit operations on
// the 2-bit rows -- the order hash is evaluated for one k-mer position in eight instead of all of them (13.6 G hashes and
// 11.4 ms per build at the north-star size in round 2).  It stays a function of the window's content alone, which is all the
// join needs: a source window and the equal target prefix choose the same k-mer.  Windows without a class-0 k-mer take the
// minimum over their class-1 k-mers (brute force: the prefix window inside k_node_runs, a source's other windows in
// k_probe_clustered's slow path).
// nucleotide codes: A = 0, C = 1, G = 2, T = 3 (include/Params.h:275-279); nucleotide j of a k-mer = bits (2j, 2j + 1) of lo
__device__ __forceinline__ bool kmer_class0(uint32_t lo) { return (lo & 3u) == 0u && ((((lo >> 2) ^ (lo >> 3)) & 1u) != 0u); }
__device__ __forceinline__ uint32_t order_key0(uint32_t h, int pos) { return ((h >> 1) & 0x7FFFFF00u) | (uint32_t) pos; }      // of a class-0 k-mer
__device__ __forceinline__ uint32_t order_key(uint32_t h, uint32_t lo, int pos) { return order_key0(h, pos) | (kmer_class0(lo) ? 0u : 0x80000000u); }
// class-0 positions among the 16 nucleotides of `cur` (bit 2j: nucleotide j); nxt = the following row word
__device__ __forceinline__ uint32_t class0_mask16(uint32_t cur, uint32_t nxt) {
    const uint32_t nx = __funnelshift_r(cur, nxt, 2);      // nucleotide j + 1 at bits (2j, 2j + 1)
    return ~(cur | (cur >> 1)) & (nx ^ (nx >> 1)) & 0x55555555u;
}
// bits 0, 2, 4, ... 30 of x (the others are zero) -> bits 0 .. 15
__device__ __forceinline__ uint32_t compress_even(uint32_t x) {
    x = (x | (x >> 1)) & 0x33333333u;
    x = (x | (x >> 2)) & 0x0F0F0F0Fu;
    x = (x | (x >> 4)) & 0x00FF00FFu;
    return (x | (x >> 8)) & 0x0000FFFFu;
}

// Cluster key of a minimizer = a second, bijective mix of its order hash.  The order hashes of MINIMIZERS are minima of w
// uniform values -- concentrated near zero -- so bucketing the entry array by their own top bits would put most clusters in
// 1/w of the buckets; the mix spreads them evenly.  The BUCKET bits (above the m_C field, tgt_sort_key below) are never all ones:
// the sort key of a target then never is 0xFFFFFFFF ("not a target") whatever its m_C, m_C can be read back from it, and a sort
// on the bucket bits alone already puts every target before every non-target (the last bucket stays empty).
__device__ __forceinline__ uint32_t cluster_key(uint32_t h, int fs) {
    uint32_t k = h * 0x9E3779B1u;                          // (one multiply instead of three: +2 % entries scanned, +28 % deferred sources)
    k ^= k >> 15;
    k *= 0x85EBCA77u;
    k ^= k >> 13;
    k *= 0xC2B2AE3Du;
    return (k >> (fs + CL_MBITS)) == (0xFFFFFFFFu >> (fs + CL_MBITS)) ? k ^ 0x80000000u : k;
}

// Sort key of a target: the cluster key with the CL_MBITS bits right below its bucket bits replaced by m_C, the position of the
// minimizer in the target's prefix (field at bit fs = idx_shift - CL_MBITS).  Inside a bucket the entries are therefore ordered
// by m_C >> 3 first (the sort stops there: launch_cluster_store), and the directory (k_tgt_dir) knows where every eighth of that order starts: a source run that covers the
// windows [p0, p1) with its minimizer at q can only match targets with q - p1 < m_C <= q - p0 and reads that slice of the bucket
// alone -- the other entries of the cluster are the reads of the same locus that start too far left or right of the run's
// windows (half of them at 30x coverage).  0xFFFFFFFF stays reserved for "not a target".
__device__ __forceinline__ uint32_t tgt_sort_key(uint32_t ckey, uint32_t m_c, int fs) {
    const uint32_t fm = ((1u << CL_MBITS) - 1u) << fs;
    return (ckey & ~fm) | ((m_c << fs) & fm);               // never all ones: cluster_key
}
__device__ __forceinline__ bool same_cluster(uint32_t entry_key, uint32_t ckey, int fs) {
    return ((entry_key ^ ckey) & ~(((1u << CL_MBITS) - 1u) << fs)) == 0u;
}

// directory record of a bucket: {first entry, entries, first entry (relative, saturating bytes) with m_C >> 3 >= 0..3, >= 4..7}
// -> the entries [e0, e0 + cnt) a run {q | p0 << 8 | p1 << 16} has to look at
// k_pile_build / k_pile_probe / k_probe_stream: more than one bucket in this many irregular and the pairwise kernels take the build (prefsuf_pile.hip)
constexpr unsigned long long PILE_IRREGULAR_ONE_IN = ALGA_PILE_IRREGULAR_ONE_IN;      // include/alga_amd.h
constexpr unsigned long long PILE_DECLINE_ONE_IN = ALGA_PILE_DECLINE_ONE_IN;
// What the sample of the key order says about a build (pile_cnt = {buckets, irregular buckets}; every kernel concerned reads the same two counters,
// so the form is decided on the device): DECLINED -- more than one bucket in PILE_DECLINE_ONE_IN irregular: the pairwise kernels take the build;
// MIXED (round 5) -- between the two thresholds: the pile kernels take it, and the sources they hand on go through k_probe_stream (list mode; it
// needs the entry array, so k_tgt_gather runs) before the general kernel sees what is left; else the pile kernels and the general kernel alone.
__device__ __forceinline__ bool pile_cnt_declines(const unsigned long long *c) { return c[1] * PILE_DECLINE_ONE_IN > c[0]; }
__device__ __forceinline__ bool pile_cnt_mixed(const unsigned long long *c) { return !pile_cnt_declines(c) && c[1] * PILE_IRREGULAR_ONE_IN > c[0]; }
__device__ __forceinline__ void run_slice(const uint4 &rec, uint32_t run_y, uint32_t &e0, uint32_t &cnt) {
    const int q = (int) (run_y & 255u), p0 = (int) ((run_y >> 8) & 255u), p1 = (int) ((run_y >> 16) & 255u);
    int mlo = q - p1 + 1, mhi = q - p0;
    mlo = mlo < 0 ? 0 : mlo; mhi = mhi > 63 ? 63 : mhi;
    e0 = rec.x; cnt = rec.y;
    if (mhi < mlo) { cnt = 0u; return; }
    if (rec.y > 255u) return;                              // offsets saturate: the whole bucket
    const int s0 = mlo >> 3, s1 = (mhi >> 3) + 1;
    const uint32_t b0 = ((s0 < 4 ? rec.z : rec.w) >> (8 * (s0 & 3))) & 255u;
    const uint32_t b1 = s1 >= 8 ? rec.y : (((s1 < 4 ? rec.z : rec.w) >> (8 * (s1 & 3))) & 255u);
    e0 = rec.x + b0; cnt = b1 - b0;
}

// k-mer starting at nucleotide i of a 2-bit row (words readable up to index (2i >> 5) + 2): its hash and its packed
// order key (class | 23-bit order | position); the smallest key of a window is the window's minimizer
__device__ __forceinline__ void kmer_key(const uint32_t *row, int i, bool valid, const ClusterCfg &cc, uint32_t &h, uint32_t &pk) {
    const int bit = 2 * i, q = bit >> 5, r = bit & 31;
    const uint32_t x0 = row[q], x1 = row[q + 1], x2 = row[q + 2];
    const uint32_t lo = funnel(x0, x1, r) & cc.lo_mask;
    h = kmer_hash(lo, funnel(x1, x2, r) & cc.hi_mask);
    pk = valid ? order_key(h, lo, i) : 0xFFFFFFFFu;
}

constexpr int TK_ROWS = 128;         // rows per workgroup of k_node_runs / k_pile_runs_consensus
constexpr int NR_STACK = 8;          // minimum records kept per row and block (a random block has ~2.7; more: the row is flagged)

// The window minimizers of ONE row (thread t of the workgroup; the row staged in LDS, `act`: the row takes part): steps (1) - (3) above.
// Leaves the runs in rbuf[0 .. min(nr, CL_RMAX))[t] as q | p0 << 8 (p1 = p0 of the run before; the last windows first), the minimum of
// block 0 of the first half (the row's minimizer as a TARGET) in cur0.  Shared by k_node_runs (a node's row) and k_pile_runs_consensus
// (prefsuf_pile.hip: the consensus of a pile on the pile's extent -- the windows of all its members at once).
// WIDE: more windows than one sweep takes (the two-block minimum needs no more windows than w): the windows go in PIECES of `step` (a multiple
// of 16 nucleotides = a whole row word, <= min(w, 64): 64 for the wide nodes cluster_plan admits), the last piece first, each by the same three
// steps on the row shifted by the piece's first window; the runs of the pieces are listed one after the other (a minimizer that spans a seam makes
// two runs: one more look-up for a node, merged again for a pile).
// RCAP: runs the list holds (rbuf has RCAP + 1 rows; CL_RMAX for a node's list, more for a pile's extent, whose pieces each add a run at their seam)
template <bool WIDE, int RCAP = CL_RMAX>
__device__ __forceinline__ void node_runs_core(const uint32_t *row, int nwin, bool act, const ClusterCfg &cc, uint32_t (*stk)[TK_ROWS], uint16_t (*rbuf)[TK_ROWS], int t,
                                               int &nr, bool &uncovered, bool &stack_ovf, uint32_t &cur0, int step = 64) {
    constexpr int S0 = NR_STACK + 1;                       // first row of block 0's records
    const int w = cc.w;
    nr = 0; uncovered = false; stack_ovf = false; cur0 = 0xFFFFFFFFu;
    for (int hb = (WIDE && nwin > step) ? ((nwin - 1) / step) * step : 0; hb >= 0; hb -= step) {       // first window of the piece (one pass with hb = 0 unless WIDE)
        const uint32_t *rowh = row + (hb >> 4);
        const int nwh = !WIDE ? nwin : (nwin - hb < step ? nwin - hb : step);
        const int nk = act ? nwh - 1 + w : 0;              // k-mer positions of the half (relative to hb): <= 127
        // ---- class-0 k-mer positions: bit p of the 128-bit mask (static row indices: registers, no scratch) ----
        uint64_t m_lo, m_hi;
        {
            uint32_t dm[4];
#pragma unroll
            for (int d = 0; d < 4; d++) {
                const uint32_t x0 = rowh[2 * d], x1 = rowh[2 * d + 1], x2 = rowh[2 * d + 2];
                dm[d] = compress_even(class0_mask16(x0, x1)) | (compress_even(class0_mask16(x1, x2)) << 16);
            }
            m_lo = (uint64_t) dm[0] | ((uint64_t) dm[1] << 32);
            m_hi = (uint64_t) dm[2] | ((uint64_t) dm[3] << 32);
            m_lo &= nk >= 64 ? ~0ull : (nk <= 0 ? 0ull : ((1ull << nk) - 1ull));       // positions below nk only (0 for a node that takes no part)
            m_hi &= nk <= 64 ? 0ull : ((1ull << (nk - 64)) - 1ull);
        }
        uint64_t b0 = w >= 64 ? m_lo : (m_lo & ((1ull << w) - 1ull));
        uint64_t b1 = w >= 64 ? m_hi : ((m_lo >> w) | (m_hi << (64 - w)));    // uniform branch; bit e = k-mer w + e
        auto key_at = [&](int pos) -> uint32_t {           // order key of the class-0 k-mer at `pos` (relative to the half)
            const int bit = 2 * pos, q = bit >> 5, r = bit & 31;
            const uint32_t x0 = rowh[q], x1 = rowh[q + 1], x2 = rowh[q + 2];
            return order_key0(kmer_hash(funnel(x0, x1, r) & cc.lo_mask, funnel(x1, x2, r) & cc.hi_mask), pos);
        };
        // ---- (1) block 1, left to right: prefix-minimum records ----
        uint32_t cur1 = 0xFFFFFFFFu;
        int sp1 = 0;
        while (b1 != 0ull) {
            const int e = __builtin_ctzll(b1);
            b1 &= b1 - 1ull;
            const uint32_t pk = key_at(w + e);
            // no branch: a lane that does not push writes the spare row
            const bool push = pk < cur1;
            stk[(push && sp1 < NR_STACK) ? sp1 : NR_STACK][t] = pk;
            cur1 = push ? pk : cur1;
            sp1 += push ? 1 : 0;
        }
        // ---- (2) block 0, right to left: suffix-minimum records ----
        uint32_t cur0h = 0xFFFFFFFFu;
        int sp0 = 0;
        while (b0 != 0ull) {
            const int e = 63 - __builtin_clzll(b0);
            b0 ^= 1ull << e;
            const uint32_t pk = key_at(e);
            const bool push = pk < cur0h;
            stk[S0 + ((push && sp0 < NR_STACK) ? sp0 : NR_STACK)][t] = pk;
            cur0h = push ? pk : cur0h;
            sp0 += push ? 1 : 0;
        }
        if (hb == 0) cur0 = cur0h;
        stack_ovf = stack_ovf || sp1 > NR_STACK || sp0 > NR_STACK;
        sp1 = sp1 > NR_STACK ? NR_STACK : sp1;
        sp0 = sp0 > NR_STACK ? NR_STACK : sp0;
        // ---- (3) the windows of the half, last to first: merge of the two record lists ----
        {
            const uint32_t hq = (uint32_t) hb | ((uint32_t) hb << 8);     // what makes q and p0 of a run absolute
            uint32_t top = sp1 > 0 ? stk[sp1 - 1][t] : 0xFFFFFFFFu;       // smallest record of block 1: in every window until it drops out
            uint32_t nxt0 = stk[S0][t];                                   // next record of block 0 (if i0 < sp0)
            uint32_t c0 = 0xFFFFFFFFu, win = top;
            int i0 = 0, p_hi = nwh - 1;
            int e0 = sp0 > 0 ? (int) (nxt0 & 255u) : -1;                  // the next block-0 record joins the windows p <= e0
            int e1 = sp1 > 0 ? (int) (top & 255u) - w : -1;               // block 1's smallest record is in no window p <= e1
            while ((e0 > e1 ? e0 : e1) >= 0) {
                const int pe = e0 > e1 ? e0 : e1;
                const bool take0 = e0 >= e1;
                // ONE stack read serves either move: the record after the block-0 record that joins, or the one below block 1's top
                const int row_i = take0 ? S0 + i0 + 1 : (sp1 >= 2 ? sp1 - 2 : NR_STACK);
                const uint32_t v = stk[row_i][t];
                i0 += take0 ? 1 : 0;
                sp1 -= take0 ? 0 : 1;
                c0 = take0 ? nxt0 : c0;
                nxt0 = take0 ? v : nxt0;
                top = take0 ? top : (sp1 >= 1 ? v : 0xFFFFFFFFu);
                e0 = take0 ? (i0 < sp0 ? (int) (v & 255u) : -1) : e0;
                e1 = take0 ? e1 : (sp1 >= 1 ? (int) (v & 255u) - w : -1);
                const uint32_t wn = c0 < top ? c0 : top;
                const int pc = pe < p_hi ? pe : p_hi;
                const bool em = wn != win && pc < p_hi;         // the windows (pc, p_hi] had `win`
                rbuf[(em && nr < RCAP) ? nr : RCAP][t] = (uint16_t) (((win & 255u) | ((uint32_t) (pc + 1) << 8)) + hq);
                uncovered = uncovered || (em && win == 0xFFFFFFFFu);
                nr += em ? 1 : 0;
                p_hi = em ? pc : p_hi;
                win = wn;
            }
            if (act) {                                         // the run of the half's first window
                if (nr < RCAP) rbuf[nr][t] = (uint16_t) ((win & 255u) + hq);
                uncovered = uncovered || win == 0xFFFFFFFFu;
                nr++;
            }
        }
    }
}

__device__ __forceinline__ uint32_t bperm(uint32_t v, int src_lane) { return (uint32_t) __builtin_amdgcn_ds_bpermute(src_lane << 2, (int) v); }

} // namespace alga
