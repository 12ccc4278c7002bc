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
//     reads and three contiguous runs of entries, ONE LANE PER ENTRY: hash / offset / length checks from the entry's own
//     words, exact 2-bit compare of the entry's row against the source's staged tail, verified overlaps become the items
//     of the source-side transitive reduction (prefsuf_device.h local_reduce).
//
//   k_node_runs        one thread per node: the runs of equal window minimizer {cluster key, k-mer position, windows [p0, p1)}
//                      of the node as a SOURCE, and the key / meta word it is filed under as a TARGET (run 0)
//   (radix sort of (key, id): rocPRIM)
//   k_tgt_gather       rows in key order -> entry array;   k_tgt_dir     directory record of every key bucket
//   k_probe_stream     entries of consecutive sources packed densely onto the lanes: finishes the regular sources, lists the others
//   k_probe_clustered  one source per wave, any shape: the listed sources (or all of them)
#include <hip/hip_runtime.h>
#include <algorithm>
#include "prefsuf_common.h"
#include "prefsuf_kernels.h"
#include "prefsuf_device.h"
#include "prefsuf_cluster_device.h"

namespace alga {

// ------------------------------------------------------------------------------------------
// build: minimizer runs of every node, keys, gather, index
// ------------------------------------------------------------------------------------------
// (TK_ROWS, NR_STACK and node_runs_core: prefsuf_cluster_device.h -- shared with prefsuf_pile.hip)

// One THREAD per node: the distinct minimizers of the suffix windows p = 0 .. len - Lmin of the node, as runs
// {cluster key, k-mer position q, windows [p0, p1)}; the minimizer of window 0 (the node's min_overlap-long prefix) is the key the
// node is filed under as a TARGET.
//   Window p covers the k-mers [p, p + w), nk = nwin - 1 + w <= 2w - 1 <= 127 k-mers in all.  Only the class-0 k-mers (order_key:
//   one position in eight) can be minimizers of a window that holds one; their positions are a 128-bit mask, built from the staged
//   row with word-parallel bit operations.  Block 0 = k-mers [0, w), block 1 = [w, nk):
//     min(window p) = min(suffix minimum of block 0 from p, prefix minimum of block 1 up to p + w - 1).
//   (1) block 1, class-0 positions left to right: the prefix-minimum RECORDS (a k-mer smaller than all before it) go on a
//       per-thread stack; (2) block 0, right to left: its suffix-minimum records on a second stack; the order hash is evaluated
//       in these two loops only, once per class-0 k-mer; (3) the sweep over the windows from the last to the first is a merge of
//       the two record lists -- a block-0 record at position c joins the windows p <= c, a block-1 record at c drops out of the
//       windows p <= c - w -- ~5 events per node, a run noted whenever the winner changes.
//   Ties go to the left on both the source and the target side (the position is part of the order key).
// keys[i] = sort key of the node as a target (all ones: not a target), vals[i] = i, meta[i] = m_C | len << 8 | alignFrom << 20,
// runs[i * CL_RMAX + k] = {cluster key, q | p0 << 8 | p1 << 16}, nruns[i] = number of runs (0: not a source; CL_RUNS_FLAGGED: a
// window without a class-0 k-mer, more than CL_RMAX runs or more records than a stack holds -- k_probe_clustered finds such a
// source's window minimizers by brute force over all k-mers, 1 source in ~10^3).
// TKW = row words staged per node: the row (<= TKW - 2 words) + the two words a k-mer read may run past it; odd: conflict-free.
// WIDE: nodes with up to 128 suffix windows (reads of up to ~270 nt at the reference's default scale: the two-word form of the source-side
// reduction).  The window minimum by two blocks needs nwin <= w <= 64, so the windows are taken in two halves of up to 64 -- [64, nwin) on
// the k-mers from position 64 on, then [0, 64) -- each by the very same three steps on the row shifted by 64 nucleotides (four words), and
// the runs of both halves are listed one after the other (a minimizer that spans the seam makes two runs: one more look-up, same overlaps).
// TARGETS ONLY (what_runs = false below): the key pass of a build whose sources get their run lists elsewhere (the pile path: a pile's run
// list comes from its consensus, prefsuf_pile.hip) needs the minimum of block 0 alone -- no stacks, no window sweep.
template <int TKW, bool WIDE, bool RUNS>
__global__ void __launch_bounds__(TK_ROWS) k_node_runs(NodesDev nd, PrefSufCfg cfg, ClusterCfg cc, int node_begin, int node_end,
                                                        uint32_t *__restrict__ keys, uint32_t *__restrict__ vals, uint32_t *__restrict__ meta,
                                                        uint2 *__restrict__ runs, uint8_t *__restrict__ nruns, const int32_t *__restrict__ id_list, const unsigned long long *__restrict__ id_count,
                                                        const unsigned long long *__restrict__ only_if_declined /* null, or the pile path's sample: the kernel leaves for a build that path keeps */) {
    if (only_if_declined && !pile_cnt_declines(only_if_declined)) return;
    __shared__ uint32_t s[TK_ROWS][TKW];
    // records, transposed (conflict-free): rows 0 .. NR_STACK - 1 block 1, NR_STACK the spare row that takes the stores of lanes
    // that do not push, NR_STACK + 1 .. 2 NR_STACK block 0, 2 NR_STACK + 1 its spare row
    __shared__ uint32_t stk[RUNS ? 2 * (NR_STACK + 1) : 1][TK_ROWS];
    __shared__ uint16_t rbuf[RUNS ? CL_RMAX + 1 : 1][TK_ROWS];        // runs as they are found: q | p0 << 8 (p1 = p0 of the run found before); last row: spare
    // the nodes node_begin .. node_end - 1 (a rank's share, or all of them), or -- id_list -- the nodes a list names (RUNS only: their run
    // lists and nothing else; the list's length is read from the device)
    const int list_n = id_list ? (int) min((unsigned long long) (node_end - node_begin), *id_count) : 0;
    __shared__ int sid[TK_ROWS];
    // (a list is walked with the grid's stride -- its length is only known on the device, the grid is sized for the chip, not for the cap; so is
    //  the node range of a launch that may leave at once: 700 000 workgroups that only look at two counters cost 0.15 ms)
    for (int blk = (int) blockIdx.x; id_list ? blk * TK_ROWS < list_n : node_begin + blk * TK_ROWS < node_end; blk += (int) gridDim.x) {
    const int base = node_begin + blk * TK_ROWS;
    const int nrows = id_list ? min(TK_ROWS, list_n - blk * TK_ROWS) : min(TK_ROWS, node_end - base);
    if (blk != (int) blockIdx.x) __syncthreads();          // (the rows of the chunk before are done with)
    if (id_list) {
        if ((int) threadIdx.x < nrows) sid[threadIdx.x] = min(max(id_list[blk * TK_ROWS + (int) threadIdx.x], 0), nd.n - 1);
        __syncthreads();
    }
    if ((nd.stride & 3) == 0 && 4 * ((TKW + 3) / 4) <= nd.stride && (reinterpret_cast<uintptr_t>(nd.words) & 15u) == 0) {
        // (round 5) the rows in 16-byte pieces: a quarter of the load instructions for the same bytes -- the word-by-word form below kept the
        // keys-only pass at 2.44 ms for 90.6 M nodes (2.8 TB/s, neither its arithmetic nor memory busy); 1.45 ms this way
        constexpr int PCS = (TKW + 3) / 4;
        for (int u = (int) threadIdx.x; u < nrows * PCS; u += TK_ROWS) {
            const int r = u / PCS, pc = u - r * PCS;
            const uint4 x = *reinterpret_cast<const uint4 *>(nd.words + (size_t) (id_list ? sid[r] : base + r) * nd.stride + 4 * pc);
            const uint32_t xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
            for (int q = 0; q < 4; q++) if (4 * pc + q < TKW) s[r][4 * pc + q] = xs[q];
        }
    } else
    for (int c = (int) (threadIdx.x & 15u); c < TKW; c += 16)
        for (int r = (int) (threadIdx.x >> 4); r < nrows; r += TK_ROWS / 16)
            s[r][c] = c < nd.stride ? nd.words[(size_t) (id_list ? sid[r] : base + r) * nd.stride + c] : 0u;
    __syncthreads();
    const int t = (int) threadIdx.x;
    const bool in = t < nrows;
    const int i = id_list ? sid[in ? t : 0] : base + (in ? t : 0);
    const int len = in ? nd.len[i] : 0;
    const bool act = in && len >= cfg.Lmin && len > 0;
    const bool is_tgt = act && (!nd.to || nd.to[i]);
    const bool is_src = act && (!nd.from || nd.from[i]);
    const uint32_t *row = s[in ? t : 0];
    const int w = cc.w;
    const int nwin = len - cfg.Lmin + 1;                   // <= 64 (one-word form of the source-side reduction); <= 128 with WIDE
    const int fs = cc.idx_shift - CL_MBITS;
    int nr = 0;
    bool uncovered = false;                                // some window holds no class-0 k-mer
    bool stack_ovf = false;
    uint32_t cur0 = 0xFFFFFFFFu;                           // minimum of block 0 of the FIRST half: the node's key as a target
    if (RUNS) {
        node_runs_core<WIDE>(row, nwin, act, cc, stk, rbuf, t, nr, uncovered, stack_ovf, cur0);
        const int nr_stored = nr < CL_RMAX ? nr : CL_RMAX;
        // what nruns[i] says, also in the top byte of run 0's second word (the quad kernel reads the run list and nothing else)
        const uint32_t nr_code = !is_src ? 0u : ((nr > CL_RMAX || stack_ovf || uncovered) ? (uint32_t) CL_RUNS_FLAGGED : (uint32_t) nr);
        int p1 = nwin;
        for (int r = 0; r < CL_RMAX; r++) {                // (skipped by the waves none of whose nodes has that many runs)
            if (r < nr_stored && is_src) {
                const uint32_t d = rbuf[r][t];
                const int q = (int) (d & 255u);
                uint32_t h, pk;
                kmer_key(row, q < (WIDE ? 192 : 128) ? q : 0, true, cc, h, pk);
                runs[(size_t) i * CL_RMAX + r] = make_uint2(cluster_key(h, fs), d | ((uint32_t) p1 << 16) | (r == 0 ? nr_code << 24 : 0u));
                p1 = (int) (d >> 8);
            }
        }
        if (in && !is_src) runs[(size_t) i * CL_RMAX] = make_uint2(0u, 0u);
        if (in) nruns[i] = (uint8_t) nr_code;
        if (id_list) continue;                             // (the listed nodes' keys are in place)
    } else {
        // the minimum of block 0 (k-mers [0, w)) over its class-0 positions
        const int nk = act ? min(nwin, WIDE ? 64 : nwin) - 1 + w : 0;
        uint64_t b0;
        {
            const uint32_t x0 = row[0], x1 = row[1], x2 = row[2], x3 = row[3], x4 = row[4];
            const uint32_t d0 = compress_even(class0_mask16(x0, x1)) | (compress_even(class0_mask16(x1, x2)) << 16);
            const uint32_t d1 = compress_even(class0_mask16(x2, x3)) | (compress_even(class0_mask16(x3, x4)) << 16);
            b0 = (uint64_t) d0 | ((uint64_t) d1 << 32);
            b0 &= nk >= 64 ? ~0ull : (nk <= 0 ? 0ull : ((1ull << nk) - 1ull));
            b0 &= w >= 64 ? ~0ull : ((1ull << w) - 1ull);
        }
        while (b0 != 0ull) {
            const int e = __builtin_ctzll(b0);
            b0 &= b0 - 1ull;
            const int bit = 2 * e, q = bit >> 5, r = bit & 31;
            const uint32_t x0 = row[q], x1 = row[q + 1], x2 = row[q + 2];
            const uint32_t pk = order_key0(kmer_hash(funnel(x0, x1, r) & cc.lo_mask, funnel(x1, x2, r) & cc.hi_mask), e);
            cur0 = pk < cur0 ? pk : cur0;
        }
    }
    // ---- the node as a target: the minimizer of window 0 = the minimum of block 0 (cur0: every class-0 k-mer of the block has
    //      been seen, whatever the stacks hold) ----
    uint32_t key = 0xFFFFFFFFu, m = act ? (((uint32_t) len << 8) | (is_src ? CL_META_FROM : 0u)) : 0u;
    if (is_tgt) {
        uint32_t pmin = cur0;
        if (pmin == 0xFFFFFFFFu) {                         // no class-0 k-mer in the prefix window (2 nodes in 10^4): all of its k-mers
            for (int k = 0; k < w; k++) { uint32_t h, pk; kmer_key(row, k, true, cc, h, pk); pmin = pk < pmin ? pk : pmin; }
        }
        uint32_t h, pk;
        kmer_key(row, (int) (pmin & 255u), true, cc, h, pk);
        key = tgt_sort_key(cluster_key(h, fs), pmin & 255u, fs);
        m = (pmin & 255u) | ((uint32_t) len << 8) | (is_src ? CL_META_FROM : 0u);
    }
    if (in) { keys[i] = key; if (vals) vals[i] = (uint32_t) i; meta[i] = m; }
    }
}

// entry j (key order) = {row words 0 .. 4*EQ-4, node id, sort key, meta}; one thread per 16-byte piece.  EVERY node gets its entry;
// the ones of the targets come first (their sort keys are below all ones) and are what the directory describes.
// UNIFORM (every live node has the same length and there is no alignFrom mask -- every BASELINE configuration): the meta word
// follows from the sort key (its m_C field) and the common length, and the one random 4-byte read per entry of meta[] goes away
// (4.6 -> 3.0 ms at 90.6 M nodes).
template <int EQ, bool UNIFORM>
__global__ void __launch_bounds__(256) k_tgt_gather(NodesDev nd, uint64_t count /* sorted (key, id) pairs: all nodes, or the targets of a rank's bucket range */,
                                                     const uint32_t *__restrict__ keys, const uint32_t *__restrict__ vals,
                                                     const uint32_t *__restrict__ meta, uint32_t uniform_meta, int fs, uint4 *__restrict__ store,
                                                     const unsigned long long *__restrict__ pile_cnt /* null, or the pile path's sample: a build it keeps has no use for the entry array */) {
    if (pile_cnt && pile_cnt[1] * PILE_IRREGULAR_ONE_IN <= pile_cnt[0]) return;
    for (uint64_t t = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; t < count * EQ; t += (uint64_t) gridDim.x * blockDim.x) {      // (a bounded grid: a million workgroups that only look at pile_cnt cost 0.3 ms)
    const uint64_t j = t / EQ;
    const int c = (int) (t % EQ);
    const uint32_t key = keys[j];                          // all ones: not a target -- behind the last bucket, never looked up; its entry
    const uint32_t id = vals[j];                           // serves the quad kernel, which walks the SOURCES in this order
    const uint32_t *row = nd.words + (size_t) id * nd.stride;
    uint4 v;
    if ((nd.stride & 3) == 0 && 4 * c + 3 < nd.stride && ((uintptr_t) nd.words & 15u) == 0) v = reinterpret_cast<const uint4 *>(row)[c];
    else {
        v.x = 4 * c + 0 < nd.stride ? row[4 * c + 0] : 0u; v.y = 4 * c + 1 < nd.stride ? row[4 * c + 1] : 0u;
        v.z = 4 * c + 2 < nd.stride ? row[4 * c + 2] : 0u; v.w = 4 * c + 3 < nd.stride ? row[4 * c + 3] : 0u;
    }
    if (c == EQ - 1) {
        v.y = id; v.z = key;
        if (!UNIFORM) v.w = meta[id];
        else if (key != 0xFFFFFFFFu) v.w = uniform_meta | ((key >> fs) & ((1u << CL_MBITS) - 1u));
        else v.w = (uint32_t) nd.len[id] == (uniform_meta >> 8 & 0xFFFu) ? uniform_meta : 0u;     // a removed node, or a source that is no target (alignTo)
    }
    store[j * EQ + c] = v;
    }
}

// dir[b] = {first entry of bucket b, entries of bucket b, byte offsets of the eight m_C >> 3 classes inside it} for the NON-EMPTY
// buckets b in [0, n_buckets]; non-targets (all-ones keys) count as bucket n_buckets.  The directory is zero-filled before this
// kernel (launch_cluster_store): an all-zero record reads as "no entries" (run_slice), and nine of ten buckets are empty at one
// bucket per node -- their records used to be written one by one from here (1.07 GB of 16-byte stores: 1.1 ms against 0.2 ms of fill
// + 0.2 ms for the non-empty ones).  One thread per entry of the sorted key array plus one past its end: the thread of the FIRST
// entry of a bucket makes its record.  One streaming pass over the keys, a tile of them (and what follows it) staged in
// LDS so that a bucket is walked at LDS latency (round 2: an index array first, then a second pass through global memory per
// bucket).  Measured and rejected: folding this pass into k_tgt_gather (the walk serialises behind that kernel's random row
// reads: 6.5 against 3.0 + 1.2 ms); collecting the records of a tile in LDS and writing them out as whole lines (1.4 against 1.2 ms).
constexpr int TD_TILE = 256, TD_HALO = 96;                 // entries per tile (512 and 1024 measure the same); entries staged past it
__global__ void __launch_bounds__(TD_TILE) k_tgt_dir(const uint32_t *__restrict__ keys, uint64_t n, int shift, uint32_t n_buckets /* of this directory */,
                                                     uint32_t bucket_base /* first bucket of this directory: 0, or the start of a rank's bucket range (every key lies in it) */, uint4 *__restrict__ dir,
                                                     unsigned long long *__restrict__ bad /* set when the keys are not in (bucket, m_C class) order: the build fails with ALGA_ERR_HIP */) {
    __shared__ uint32_t sb[TD_TILE + TD_HALO + 1];         // (bucket << 3 | m_C class) of the entries base - 1 .. base + TD_TILE + TD_HALO - 1
    const uint64_t base = (uint64_t) blockIdx.x * TD_TILE;
    const uint32_t nb3 = n_buckets << 3;
    auto tagged = [&](uint64_t k) -> uint32_t { if (k >= n) return nb3; const uint32_t x = keys[k]; return x == 0xFFFFFFFFu ? nb3 : (x >> (shift - 3)) - (bucket_base << 3); };
    for (int k = (int) threadIdx.x; k <= TD_TILE + TD_HALO; k += TD_TILE)
        sb[k] = (base == 0 && k == 0) ? 0xFFFFFFFFu : tagged(base + (uint64_t) k - 1u);     // "entry -1": a bucket no entry has
    __syncthreads();
    const int t = (int) threadIdx.x, lane = lane_id();
    const uint64_t j = base + (uint64_t) t;
    auto put = [&](int64_t b, const uint4 &rec) { dir[b] = rec; };
    // first[s] = entries of the bucket with class < s, s = 1 .. 7: seven byte counters in one 64-bit word (byte s; meaningful for
    // buckets of <= 255 entries only).  An entry of class c counts for every s > c: one shifted constant.
    auto tally = [](unsigned long long acc, uint32_t cls) { return acc + (0x0101010101010100ull << (8u * cls)); };
    // fail closed: a directory over keys that are not in order would send the probe to entries that are not there (the probe clamps its
    // reads, so nothing faults; the build reports the flag)
    if (j >= 1 && j < n && sb[t + 1] < sb[t]) atomicOr(bad, 1ull);
    const uint32_t b = sb[t + 1] >> 3;
    const bool start = j <= n && b != (sb[t] >> 3);
    if (start && b == n_buckets) put(b, make_uint4((uint32_t) j, 0u, 0u, 0u));
    const bool real = start && b != n_buckets;
    // A wave's time is its longest lane's: with one lane per bucket walking its entries, the 3 - 5 buckets that start in a wave's 64
    // entries (13 entries each at 30x coverage, hundreds in repeats) kept the other 60 lanes waiting -- 1.06 ms, VALU-bound.  Now a
    // lane walks its own bucket only if that ends within TD_SHORT entries; the longer ones are taken by 16-lane rows, four buckets side
    // by side (one after the other by the whole wave, the class counts from seven ballots per step: 0.82 ms).
    constexpr int TD_SHORT = 4;
    const bool is_short = real && (sb[t + 1 + TD_SHORT] >> 3) != b;       // (t + 1 + TD_SHORT <= TD_TILE + TD_HALO)
    if (is_short) {
        unsigned long long acc = 0ull;
        int u = 0;
#pragma unroll
        for (; u < TD_SHORT; u++) {
            const uint32_t x = sb[t + 1 + u];
            if ((x >> 3) != b) break;
            acc = tally(acc, x & 7u);
        }
        put(b, make_uint4((uint32_t) j, (uint32_t) u, (uint32_t) acc, (uint32_t) (acc >> 32)));
    }
    // the longer buckets, FOUR at a time, each by one 16-lane row of the wave: 16 entries per step, the eight byte counters of the
    // step summed over the row by a DPP scan (byte 0 counts the entries themselves; no byte overflows below 240 entries)
    unsigned long long longs = __ballot(real && !is_short);
    const int g = lane >> 4, gl = lane & 15;
    while (longs != 0ull) {                                // uniform: up to four bucket starts per pass
        int lsel = 64;                                     // the start lane this row takes (64: none)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int l = longs ? __builtin_ctzll(longs) : 64;
            longs = longs ? (longs & (longs - 1ull)) : 0ull;
            lsel = g == q ? l : lsel;
        }
        const bool has = lsel < 64;
        const int t0 = (t & ~63) + (has ? lsel : 0);       // the starting thread's index in the workgroup
        const uint32_t bb = bperm(b, has ? lsel : 0);
        unsigned long long acc = 0ull;                     // byte 0: entries, bytes 1 .. 7: first[s]
        bool open = has, past = false;
        for (int u = 0;; u += 16) {
            const int idx = t0 + 1 + u + gl;
            const bool staged = idx <= TD_TILE + TD_HALO;
            const uint32_t x = staged ? sb[idx] : 0xFFFFFFFFu;
            const bool inb = open && staged && (x >> 3) == bb;
            const unsigned long long c = inb ? ((0x0101010101010100ull << (8u * (x & 7u))) | 1ull) : 0ull;
            uint32_t lo = (uint32_t) c, hi = (uint32_t) (c >> 32);
            lo += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) lo, 0x111, 0xF, 0xF, true); hi += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) hi, 0x111, 0xF, 0xF, true);
            lo += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) lo, 0x112, 0xF, 0xF, true); hi += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) hi, 0x112, 0xF, 0xF, true);
            lo += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) lo, 0x114, 0xF, 0xF, true); hi += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) hi, 0x114, 0xF, 0xF, true);
            lo += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) lo, 0x118, 0xF, 0xF, true); hi += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) hi, 0x118, 0xF, 0xF, true);
            lo = bperm(lo, lane | 15); hi = bperm(hi, lane | 15);          // the row's total, in every lane of the row
            acc += ((unsigned long long) hi << 32) | lo;
            const uint32_t took = lo & 255u, cnt = (uint32_t) acc & 255u;
            if (open && took < 16u) { open = false; past = t0 + 1 + u + (int) took > TD_TILE + TD_HALO; }     // ended in this step -- or the staged entries did
            else if (open && cnt >= 240u) { open = false; past = true; }                                    // the byte counters stop here: on serially
            if (__ballot(open) == 0ull) break;             // uniform
        }
        if (has && gl == 0) {
            const uint64_t j0 = base + (uint64_t) t0;
            uint64_t k = j0 + (uint64_t) ((uint32_t) acc & 255u);
            acc &= ~255ull;
            if (past) {                                    // repeats: on through global memory, the end by bisection beyond 256 entries
                for (; k < j0 + 256u; k++) {
                    const uint32_t x = tagged(k);
                    if ((x >> 3) != bb) break;
                    acc = tally(acc, x & 7u);
                }
                if (k >= j0 + 256u && (tagged(k) >> 3) == bb) {
                    uint64_t a = k, z = n;                 // bucket(a) == bb, bucket(z) > bb (z == n: past the end)
                    while (z - a > 1) { const uint64_t mid = a + (z - a) / 2; if ((tagged(mid) >> 3) == bb) a = mid; else z = mid; }
                    k = z;
                }
            }
            const uint32_t c = (uint32_t) (k - j0);
            put(bb, make_uint4((uint32_t) j0, c, c <= 255u ? (uint32_t) acc : 0u, c <= 255u ? (uint32_t) (acc >> 32) : 0u));
        }
    }
}

// ------------------------------------------------------------------------------------------
// k_probe_clustered
// ------------------------------------------------------------------------------------------
#ifndef CL_OCC
#define CL_OCC 6                      // workgroups per CU of the persistent grid: 24 waves per CU (VGPR <= 80, LDS <= 26 KB, SGPR <= 112)
#endif

// Two-stage software pipeline over the sources of a wave: while the entry loads of source i are in flight the wave stages the
// row of source i+1 and issues the index loads of its runs; those land while source i is verified and reduced.  State of a
// source between the stages: its staged row and its resolved run list in LDS (two buffers), a few uniform scalars.
// SW = 64-bit words of an offset mask / uint4 words of an overhang in the source-side reduction (prefsuf_device.h): 1 for sources of up to
// 64 suffix windows, 2 for up to 128 (round 4: 250-bp reads; the items then always go through LDS and local_reduce<., ., 2>, their
// overhangs are read from the target's row, and a flagged source's windows are taken 64 at a time).
// BYID: there is no entry array -- a build the pile path (prefsuf_pile.hip) keeps does not make one; entry j of the key order is node sids[j] with
// sort key skeys[j], its row is read from the node array, its meta word follows from the key (one read length, no masks).  Whether the build
// has an entry array is decided on the device (pile_cnt: the pile path's sample): both forms are launched, the one that does not apply leaves.
struct ByIdEntries {
    const uint32_t *skeys, *sids;                          // the sorted (key, id) pairs
    uint32_t uniform_meta;                                 // len << 8 | CL_META_FROM
    const unsigned long long *pile_cnt;                    // null: the build has an entry array, no question
};
template <bool STATS, int EQ, int KF, int SW = 1, bool BYID = false>
__global__ void __launch_bounds__(PROBE_WAVES * 64, SW == 1 && EQ <= 4 ? CL_OCC : 4)
k_probe_clustered(NodesDev nd, PrefSufCfg cfg, ClusterCfg cc, const uint4 *__restrict__ store, const uint4 *__restrict__ dir,
                  const uint2 *__restrict__ runs, const uint8_t *__restrict__ nruns, int32_t src_begin, int32_t src_end, ProbeOut o,
                  const int32_t *__restrict__ src_list /* null: the sources are the ids src_begin .. src_end - 1; else src_list[src_begin .. src_end - 1] */,
                  const unsigned long long *__restrict__ list_count /* list mode, may be null: the list ends at min(src_end, *list_count) -- what the kernel before this one appended, no host round trip */,
                  ByIdEntries by) {
    constexpr int WC = 4 * EQ - 3;                         // row words of an entry
    if (by.pile_cnt) {
        const bool kept = by.pile_cnt[1] * PILE_IRREGULAR_ONE_IN <= by.pile_cnt[0];
        if (kept != BYID) return;
    }
    if (list_count) {
        const unsigned long long c = *list_count;
        if (c < (unsigned long long) src_end) src_end = (int32_t) c;
    }
    __shared__ uint32_t sB[PROBE_WAVES][2][STAGE_WORDS];
    __shared__ uint4 sRun[PROBE_WAVES][2][CL_RMAX];        // runs of a source: q | p0 << 8 | p1 << 16, cluster key, first entry, entries
    __shared__ uint32_t sRecC[PROBE_WAVES][WBUF_LOCAL];
    __shared__ unsigned long long sRecV[PROBE_WAVES][WBUF_LOCAL];
    __shared__ uint32_t sCnt[PROBE_WAVES][3];
    __shared__ uint32_t sItemC[PROBE_WAVES][ITEMMAX];
    __shared__ uint32_t sItemM[PROBE_WAVES][ITEMMAX];
    __shared__ uint4 sItemO[PROBE_WAVES][ITEMMAX * SW];
    __shared__ uint8_t sItemT[PROBE_WAVES][64 * SW];
    __shared__ uint4 sMask[KF > 0 ? 129 : 1];              // sMask[t] = masks of four consecutive words whose first holds t valid bits
    const int wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));      // scalar: the source stream below is wave-uniform
    const int lane = lane_id();
    WaveLds w{sB[wave][0], nullptr, nullptr, &sCnt[wave][0], sRecC[wave], sRecV[wave], &sCnt[wave][1]};
    ItemLds it{sItemC[wave], sItemM[wave], sItemO[wave], sItemT[wave], &sCnt[wave][2]};
    if (lane == 0) { *w.recN = 0; *it.N = 0; }
    if constexpr (KF > 0) {
        for (int t = (int) threadIdx.x; t <= 128; t += PROBE_WAVES * 64)
            sMask[t] = make_uint4(low_bits32(t), low_bits32(t - 32), low_bits32(t - 64), low_bits32(t - 96));
        __syncthreads();
    }
    uint64_t chunk_base = 0;
    int chunk_fill = REC_CHUNK_LOCAL;                      // "no chunk yet"
    uint64_t st_raw = 0, st_slots = 0, st_win = 0, st_rec = 0, st_cmp = 0, st_generic = 0;
    const int total_waves = (int) gridDim.x * PROBE_WAVES;
    const int kfull = KF ? KF : (2 * cfg.Lmin) >> 5;      // row words every overlap covers entirely

    // ---- source stream (scalar bookkeeping): row, length, run descriptors of the source after the one being staged are always
    //      in flight.  Node ids are < 2^31 (alga_nodes.n is an int32).
    const int pre_words = nd.stride < STAGE_WORDS ? nd.stride : STAGE_WORDS;
    int Bl = src_begin + (int) blockIdx.x * PROBE_WAVES + wave;
    int n_id = 0, n_len = 0, n_nr = 0; uint32_t n_word = 0; uint2 n_run = make_uint2(0u, 0u);
    auto fetch = [&]() {
        if (Bl < src_end) {                                // uniform
            const int b = src_list ? __builtin_amdgcn_readfirstlane(src_list[Bl]) : Bl;
            n_id = b;
            n_len = nd.len[b];
            n_nr = nruns[b];
            n_word = lane < pre_words ? nd.words[(size_t) b * nd.stride + lane] : 0u;
            if (lane < CL_RMAX) n_run = runs[(size_t) b * CL_RMAX + lane];
        }
    };
    fetch();
    // next source that takes part (nruns != 0: long enough, alignFrom): uniform
    auto advance = [&](int &B, int &lenB, int &nr, uint32_t &word0, uint2 &run) -> bool {
        while (Bl < src_end) {
            B = n_id; lenB = __builtin_amdgcn_readfirstlane(n_len); nr = __builtin_amdgcn_readfirstlane(n_nr); word0 = n_word; run = n_run;
            Bl = total_waves <= src_end - Bl ? Bl + total_waves : src_end;      // no overflow near 2^31
            fetch();
            if (nr != 0) return true;
        }
        return false;
    };
    // stage 1 of a source: row -> LDS; bucket of each run (lanes without a run read bucket 0: one shared line)
    auto stage = [&](int buf, int lenB, int nr, uint32_t word0, const uint2 &run) -> uint32_t {
        const int nwB = blocks_of(lenB);
        wave_lds_fence();
        if (lane < STAGE_WORDS) sB[wave][buf][lane] = lane < nwB ? word0 : 0u;
        return lane < nr ? run.x >> cc.idx_shift : 0u;
    };
    // The index loads of stage 1 are issued by EVERY lane and outside any branch: a load under a branch leaves the number of
    // loads in flight unknown to the compiler, which then drains ALL of them where the entries are first used -- and that
    // serialises the two stages of the pipeline.
    auto index_loads = [&](uint32_t bucket, uint4 &rec) { rec = dir[bucket]; };
    // resolved run list -> LDS; returns the largest entry count of the source's runs (uniform)
    auto finish_runs = [&](int buf, int nr, const uint2 &run, const uint4 &rec) -> uint32_t {
        uint32_t e0, cnt;
        run_slice(rec, run.y, e0, cnt);
        cnt = lane < nr ? cnt : 0u;
        if (lane < CL_RMAX) sRun[wave][buf][lane] = make_uint4(run.y, run.x, e0, cnt);
        uint32_t m = cnt, t;                               // max over lanes 0..7: row_shr 1, 2, 4 inside the first row
        t = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) m, 0x111, 0xF, 0xF, true); m = m > t ? m : t;
        t = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) m, 0x112, 0xF, 0xF, true); m = m > t ? m : t;
        t = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) m, 0x114, 0xF, 0xF, true); m = m > t ? m : t;
        wave_lds_fence();
        return (uint32_t) __builtin_amdgcn_readlane((int) m, CL_RMAX - 1);
    };
    // Lane layout of a source: 2^gs lanes per run (16 with up to four runs, 8 with up to eight), one entry per lane.
    auto load_runs = [&](int buf, int gs, uint4 &rp) { rp = sRun[wave][buf][(lane >> gs) & (CL_RMAX - 1)]; };
    auto load_entries = [&](const uint4 &rp, uint32_t k0, int gs, bool &ev, size_t &ei, uint32_t (&ew)[4 * EQ]) {
        const uint32_t j = k0 + ((uint32_t) lane & ((1u << gs) - 1u));
        ev = j < rp.w;
        ei = ev ? (size_t) min(rp.z + j, (uint32_t) nd.n - 1u) : (size_t) 0;   // lanes without an entry read entry 0 (unconditional loads: see index_loads); clamped: a corrupt directory must not fault
        if (BYID) {
            const uint32_t key = by.skeys[ei], id = min(by.sids[ei], (uint32_t) nd.n - 1u);
            const uint32_t *row = nd.words + (size_t) id * nd.stride;
#pragma unroll
            for (int k = 0; k < WC; k++) ew[k] = k < nd.stride ? row[k] : 0u;
            ew[4 * EQ - 3] = id; ew[4 * EQ - 2] = key;
            ew[4 * EQ - 1] = key != 0xFFFFFFFFu ? (by.uniform_meta | ((key >> (cc.idx_shift - CL_MBITS)) & ((1u << CL_MBITS) - 1u))) : 0u;
            return;
        }
#pragma unroll
        for (int c = 0; c < EQ; c++) { const uint4 v = store[ei * EQ + c]; ew[4 * c] = v.x; ew[4 * c + 1] = v.y; ew[4 * c + 2] = v.z; ew[4 * c + 3] = v.w; }
    };

    int B = 0, lenB = 0, nr = 0;
    uint32_t word0 = 0, mc = 0;
    uint2 run = make_uint2(0u, 0u);
    bool have = advance(B, lenB, nr, word0, run);
    int buf = 0;
    if (have) {
        uint4 rec0;
        const uint32_t bk = stage(0, lenB, nr == CL_RUNS_FLAGGED ? 0 : nr, word0, run);
        index_loads(bk, rec0);
        mc = finish_runs(0, nr == CL_RUNS_FLAGGED ? 0 : nr, run, rec0);
    }
    while (have) {                                         // uniform
        const uint32_t *sb = sB[wave][buf];
        const int nwin = lenB - cfg.Lmin + 1;
        // ---- (1) pull the next source off the stream BEFORE any entry load is issued: its loop must not sit between the
        //          loads and their use (hipcc drains vmcnt at loop headers) ----
        int nB = 0, nlenB = 0, nnr = 0;
        uint32_t nword0 = 0, nbk = 0;
        uint4 nrec;
        uint2 nrun = make_uint2(0u, 0u);
        const bool have_next = advance(nB, nlenB, nnr, nword0, nrun);
        const int nnr_eff = nnr == CL_RUNS_FLAGGED ? 0 : nnr;
        // ---- (2) first batch of this source's entries: loads issued ----
        const bool flagged = nr == CL_RUNS_FLAGGED;       // its run list is empty here: the slow path below finds its runs
        int gs = (nr > 4 || flagged) ? 3 : 4;
        uint32_t k0 = 0;
        uint4 rp;
        load_runs(buf, gs, rp);
        bool ev; size_t ei; uint32_t ew[4 * EQ];
        load_entries(rp, k0, gs, ev, ei, ew);
        // ---- (3) the next source: row staged, index loads issued behind the entry loads ----
        if (have_next) nbk = stage(buf ^ 1, nlenB, nnr_eff, nword0, nrun);
        index_loads(nbk, nrec);
        // ---- (4) verify: one entry per lane.  A verified overlap is an ITEM of the source-side reduction: target, offset | length |
        //      alignFrom, overhang.  With every entry of the source in this one batch (the rule) the items stay in the lanes'
        //      registers for the fused reduction below; otherwise they are appended to the item buffer in LDS. ----
        int n_items = 0;                                   // verified overlaps of this source so far (uniform)
        bool v_pass = false;
        uint32_t v_id = 0, v_m = 0;
        uint4 v_o = make_uint4(0u, 0u, 0u, 0u);
        uint64_t v_pm = 0;
        auto verify = [&](bool to_lds) {
            if (STATS && ev) st_slots++;
            const uint32_t id = ew[4 * EQ - 3], eh = ew[4 * EQ - 2], meta = ew[4 * EQ - 1];
            const int lenC = (int) ((meta >> 8) & 0xFFFu);
            int p = (int) (rp.x & 255u) - (int) (meta & 255u);         // the only offset at which C's prefix can sit in B
            // same minimizer k-mer, a window of THIS run (its minimizer is the run's), not B itself (GraphCreatorPrefSuf.cpp:386),
            // C long enough for a prefix of length L = |B| - p (:215)
            const bool ok = ev && same_cluster(eh, rp.y, cc.idx_shift - CL_MBITS) && p >= (int) ((rp.x >> 8) & 255u) && p < (int) ((rp.x >> 16) & 255u) && (int) id != B &&
                            lenC >= lenB - p;
            p = ok ? p : 0;
            const int L = lenB - p, nb = 2 * L;
            const int qw = (2 * p) >> 5, sh = (2 * p) & 31;
            uint32_t y[WC + 1];
#pragma unroll
            for (int k = 0; k <= WC; k++) y[k] = sb[qw + k];
            uint32_t diff = 0;
            uint32_t mk[4] = {0u, 0u, 0u, 0u};
            if constexpr (KF > 0) { const uint4 m4 = sMask[min(nb - 32 * KF, 128)]; mk[0] = m4.x; mk[1] = m4.y; mk[2] = m4.z; mk[3] = m4.w; }
#pragma unroll
            for (int k = 0; k < WC; k++) {                 // exact compare C[0, L) == B[p, p + L)
                const uint32_t x = funnel(y[k], y[k + 1], sh) ^ ew[k];
                if (k < kfull) diff |= x;                  // uniform (compile time with KF)
                else if (KF > 0 && k < KF + 4) diff |= x & mk[(k - KF) & 3];
                else diff |= x & low_bits32(nb - 32 * k);
            }
            const bool pass = ok && diff == 0;
            const uint64_t pm = __ballot(pass);
            v_pass = pass; v_pm = pm; v_id = id;
            v_m = (uint32_t) p | ((uint32_t) lenC << 9) | ((meta & CL_META_FROM) ? ITEM_FROM : 0u);
            if (pm != 0ull) {                              // uniform
                if (pass) {
                    if (STATS) st_raw++;
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
                    } else if constexpr (BYID) {
                        const uint32_t *er = nd.words + (size_t) id * nd.stride;                       // C's own row (no entry array)
#pragma unroll
                        for (int k = 0; k < 5; k++) x[k] = ws + k < nd.stride ? er[ws + k] : 0u;
                    } else {
                        const uint32_t *er = reinterpret_cast<const uint32_t *>(store + ei * EQ);      // re-read (an L1 hit)
#pragma unroll
                        for (int k = 0; k < 5; k++) x[k] = er[ws + k];
                    }
                    v_o = make_uint4(funnel(x[0], x[1], r2), funnel(x[1], x[2], r2), funnel(x[2], x[3], r2), funnel(x[3], x[4], r2));
                    if (to_lds) {
                        const int slot = n_items + (int) __builtin_amdgcn_mbcnt_hi((uint32_t) (pm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) pm, 0u));
                        if (slot < ITEMMAX) {
                            it.C[slot] = id; it.M[slot] = v_m;
                            if constexpr (SW == 1) it.O[slot] = v_o;
                            else item_overhang_global<SW>(nd, it, ITEMMAX, slot, (int) id, L, lenC);      // up to 127 nt: from C's own row
                        }
                    }
                }
                if (to_lds) n_items += __popcll(pm);
            }
        };
        // every entry of the source in this batch, run list complete: the items can stay in registers
        const bool one_batch = SW == 1 && !flagged && mc <= (1u << gs);
        verify(!one_batch);                                // the first batch, peeled: its wait covers the entry loads only
        bool reduced = false;
        if (one_batch) {
            n_items = __popcll(v_pm);
            if (n_items > 0) {
                // ---- fused source-side reduction (prefsuf_device.h local_reduce's fast path on lane-resident items): one item per
                // offset, every item but the first implied by its nearest predecessor -> the first one is the source's only edge.
                // Anything else (two items at one offset, a predecessor that implies nothing but could: sequencing errors, repeats,
                // several survivors) spills the items to LDS and takes local_reduce. ----
                const int Lbig = cfg.rsoemo > cfg.Lmin ? cfg.rsoemo : cfg.Lmin;
                uint8_t *T = it.T;
                T[lane] = 0xFFu;
                wave_lds_fence();
                const int d = (int) (v_m & 511u);
                if (v_pass) T[d] = (uint8_t) lane;
                wave_lds_fence();
                const uint64_t occ = __ballot(T[lane] != 0xFFu);
                if (__popcll(occ) == n_items) {            // uniform
                    const uint64_t below = v_pass ? (occ & ((1ull << d) - 1ull)) : 0ull;
                    const bool has_pred = below != 0ull;
                    const int j = has_pred ? (int) T[63 - __clzll((long long) below)] : lane;
                    const uint32_t Cj = bperm(v_id, j), mj = bperm(v_m, j);
                    Ovh<1> oj, oi;
                    oj.w[0] = bperm(v_o.x, j); oj.w[1] = bperm(v_o.y, j); oj.w[2] = bperm(v_o.z, j); oj.w[3] = bperm(v_o.w, j);
                    oi.w[0] = v_o.x; oi.w[1] = v_o.y; oi.w[2] = v_o.z; oi.w[3] = v_o.w;
                    const int rho = (int) ((v_m >> 9) & 511u) - (lenB - d);
                    const bool removed = has_pred && via_ok<1>(B, lenB, Lbig, Cj, mj, oj, v_id, d, rho, oi);
                    // not removed by the nearest predecessor although even the longest read placed there could reach C with a big
                    // overlap: undecided here -- unless that predecessor is the ONLY item before this one: then nobody else can be a via
                    // and the item stands (reads with sequencing errors: two overlaps whose overhangs disagree; round 4)
                    const bool fail = has_pred && !removed && (below & (below - 1ull)) != 0ull && (cfg.Lcap - 1) - (d - (int) (mj & 511u)) >= Lbig;
                    if (__ballot(fail) == 0ull) {          // uniform
                        const uint64_t surv = __ballot(v_pass && !removed);
                        if (__popcll(surv) == 1) {
                            if (v_pass && !removed) {
                                o.first[B - o.src_base] = ((unsigned long long) v_id << 32) | (uint32_t) d;
                                o.deg[B - o.src_base] = 1u;
                                st_rec++;
                            }
                            if (STATS && has_pred) st_cmp++;
                            reduced = true;
                        }
                    }
                }
                if (!reduced) {
                    if (v_pass) {
                        const int slot = (int) __builtin_amdgcn_mbcnt_hi((uint32_t) (v_pm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) v_pm, 0u));
                        it.C[slot] = v_id; it.M[slot] = v_m; it.O[slot] = v_o;
                    }
                }
            }
        } else {
            for (;;) {                                     // further batches: a run with more entries than its lane group
                k0 += 1u << gs;
                if (k0 >= mc) break;                       // uniform
                load_entries(rp, k0, gs, ev, ei, ew);
                verify(true);
            }
        }
        if (flagged) {
            // ---- slow path (one source in ~10^4): more runs than k_node_runs stores.  Window minimizers by brute force (lane p
            // scans the w k-mers of window p), runs by ballot, eight runs at a time through the same run list and verify(). ----
            gs = 3;
            for (int wb = 0; wb < nwin; wb += 64) {        // uniform: the windows 64 at a time (one pass unless SW == 2)
                const int wl = wb + lane;                  // this lane's window
                const bool wv = wl < nwin;
                uint32_t wm = 0xFFFFFFFFu;
                for (int k = 0; k < cc.w; k++) {           // uniform
                    uint32_t h, pk;
                    kmer_key(sb, (wv ? wl : 0) + k, true, cc, h, pk);
                    wm = pk < wm ? pk : wm;
                }
                const uint32_t prev = bperm(wm, (lane + 63) & 63);
                const bool start = wv && (lane == 0 || wm != prev);         // (a minimizer that spans two passes makes two runs: one more look-up)
                const uint64_t runmask = __ballot(start);
                const int nrun_all = __popcll(runmask);
                const uint64_t higher = lane >= 63 ? 0ull : runmask & ~((2ull << lane) - 1ull);
                const int wend = nwin - wb < 64 ? nwin - wb : 64;           // windows of this pass
                const int p1 = wb + (higher ? __builtin_ctzll(higher) : wend);
                const int rank = (int) __builtin_amdgcn_mbcnt_hi((uint32_t) (runmask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) runmask, 0u));
                uint32_t key = 0u;
                { uint32_t h, pk; kmer_key(sb, (int) (wm & 255u) < 192 ? (int) (wm & 255u) : 0, true, cc, h, pk); key = cluster_key(h, cc.idx_shift - CL_MBITS); }
                for (int rb = 0; rb < nrun_all; rb += CL_RMAX) {   // uniform
                    wave_lds_fence();
                    if (lane < CL_RMAX) sRun[wave][buf][lane] = make_uint4(0u, 0u, 0u, 0u);
                    wave_lds_fence();
                    uint32_t cnt = 0u;
                    if (start && rank >= rb && rank < rb + CL_RMAX) {
                        const uint4 rec = dir[key >> cc.idx_shift];
                        const uint32_t ry = (wm & 255u) | ((uint32_t) wl << 8) | ((uint32_t) p1 << 16);
                        uint32_t e0;
                        run_slice(rec, ry, e0, cnt);
                        sRun[wave][buf][rank - rb] = make_uint4(ry, key, e0, cnt);
                    }
                    wave_lds_fence();
                    const uint32_t mcs = (uint32_t) wave_max_u64_dpp((uint64_t) cnt);
                    load_runs(buf, gs, rp);
                    for (k0 = 0; k0 < mcs; k0 += 1u << gs) { load_entries(rp, k0, gs, ev, ei, ew); verify(true); }
                }
            }
        }
        wave_lds_fence();
        // ---- (5) transitive reduction at the source, edges out ----
        if (STATS && lane == 0) st_win += (uint64_t) nwin;
        if (n_items > ITEMMAX) {
            // more raw overlaps than the item buffer holds (repeats): second pass (k_probe_sources, BIG instantiation: items in
            // global memory, probing through the seed table); a full list: per-target pipeline
            if (lane == 0) {
                const unsigned long long k = atomicAdd(&o.counters[CNT_LOCAL_OVERFLOW], 1ull);
                if (k < (unsigned long long) o.big_list_cap) o.big_list[k] = B;
                atomicMax(&o.counters[CNT_LOCAL_MAXITEMS], (unsigned long long) n_items);
                if (STATS) { st_raw -= (uint64_t) n_items; st_win -= (uint64_t) nwin; }     // the second pass counts this source
            }
        } else if (n_items > 0 && !reduced) {
            local_reduce<STATS, WBUF_LOCAL, SW>(nd, cfg, it, w, o, B, lenB, n_items, st_rec, st_cmp, st_generic);
        }
        const int nb2 = (int) __builtin_amdgcn_readfirstlane((int) *w.recN);
        if (nb2 >= WFLUSH_LOCAL) flush_records<REC_CHUNK_LOCAL, WBUF_LOCAL>(o, w, chunk_base, chunk_fill);
        // ---- (6) the next source's index loads have had the time of (4) and (5) to land ----
        mc = finish_runs(buf ^ 1, nnr_eff, nrun, nrec);
        have = have_next; B = nB; lenB = nlenB; nr = nnr; buf ^= 1;
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
// k_probe_stream : the entries of consecutive sources packed densely onto the 64 lanes
// ------------------------------------------------------------------------------------------
// One source per wave (k_probe_clustered) leaves most lanes idle: a 150-bp source at 30x has ~16 entries to verify and ~11 items to
// reduce, and that is where the instructions go (the kernel is instruction-bound).  Round 2's pair kernel gave every source 32
// lanes (half of them idle); this kernel packs.
//   * The sources come four at a time (a QUAD): rows, run lists and directory look-ups are handled by the four 16-lane rows, one
//     source each (the DPP scans over the <= 8 runs stay inside a row).
//   * The ENTRIES of several sources are then laid out back to back on the 64 lanes -- source j occupies the lanes
//     [start_j, start_j + T_j) -- from a SLIDING WINDOW of two quads: eight consecutive sources, both quads staged, their run lists
//     resolved.  A round (= one iteration of the main loop) takes up to five sources from the cursor on for as long as they fit
//     the 64 lanes, across the quad boundary.  When the cursor has passed the older quad it retires (its unfinished sources go on
//     the defer list), the younger one takes its place and the quad staged during this round becomes the younger one: three LDS
//     buffers in rotation, at most one rotation per round (a round never takes the window's last source, so the quad that follows
//     is always ready when it is needed).  What is not consumed is fetched and staged again at the same addresses in the next
//     round: no load ever sits under a branch.  Measured at 30x: 3.45 sources and 55 of the 64 lanes per round (one quad per
//     round, the first version: 2.83 sources, 45 lanes).
//   * What is per source in the reduction -- offsets taken, two items at one offset, an undecided implication, the number of items
//     that stand -- is aggregated through a few LDS atomics on a per-source word instead of ballots over fixed parts of the wave,
//     so a source may sit on any run of lanes.
//   * Only REGULAR sources finish here (at most 8 runs, at most 64 entries, one item per offset, every item but one or two implied
//     by its nearest predecessor -- error-free data at moderate coverage): their one or two edges go straight to first[] /
//     second[] / deg[].  Any other source is appended to `defer_list` and taken by k_probe_clustered (list mode) afterwards:
//     nothing is decided twice, nothing is approximated.  A wave that has had to defer more than half of its first sources (reads
//     with sequencing errors) stops verifying and lists the rest of its share: the decision is taken from THIS build's data.
//   * The window's bookkeeping lives in few scalar registers (entry counts as packed bytes, flags as bit fields): scalars beyond
//     ~100 are spilled into VGPR lanes and cost a v_readlane per use inside the loop.
// Workgroups per CU.  Six (24 waves per CU) where the shape fits 80 VGPRs and a sixth of the LDS without scratch -- every 150-bp
// configuration: (EQ, KF) = (3, 5); the kernel is bound by VALU issue with ~75 % of the slots used, a sixth wave per SIMD fills some
// of the rest: 20.3 -> 19.8 ms -- five for the shapes that would spill ((3, 3): 100-bp reads; EQ = 4: rows of 10 - 13 words), and, by
// the same measurement, for the id-order traversal, which waits for HBM, not for issue slots (CLQ_OCC_IDORDER).
#ifndef CLQ_OCC_IDORDER
#define CLQ_OCC_IDORDER 5
#endif
constexpr int clq_occ(int eq, int kf, bool bykey) { return (eq == 4 || (eq == 3 && kf == 3)) ? 5 : (bykey ? 6 : CLQ_OCC_IDORDER); }
// LIST (round 5, the mixed form of a build the pile path keeps): the sources are the ids src_list[0 .. *src_count) -- what k_pile_probe handed
// on, in about the order of the entry array -- rows and lengths by id; what this kernel cannot finish either goes on defer_list (a SECOND list,
// counted in CNT_DEFERRED2).
template <bool STATS, int EQ, int KF, bool BYKEY, bool LIST = false>
__global__ void __launch_bounds__(PROBE_WAVES * 64, clq_occ(EQ, KF, BYKEY))
k_probe_stream(NodesDev nd, PrefSufCfg cfg, ClusterCfg cc, const uint4 *__restrict__ store, const uint4 *__restrict__ dir,
               const uint2 *__restrict__ runs, const uint8_t *__restrict__ nruns, int32_t src_begin_a, int32_t src_end_a, ProbeOut o,
               int32_t *__restrict__ defer_list, uint32_t defer_cap, const unsigned long long *__restrict__ pile_cnt,
               const int32_t *__restrict__ src_list, const unsigned long long *__restrict__ src_count) {
    static_assert(!(LIST && (BYKEY || STATS)), "list mode: ids from a list, no statistics build");
    // the pile kernel (prefsuf_pile.hip) was launched in front of this one and takes the build unless most buckets are irregular: the
    // same test on the same two counters, so exactly one of the two kernels does the work -- decided on the device, from this build's data
    if (!LIST && pile_cnt != nullptr && !pile_cnt_declines(pile_cnt)) return;
    if (LIST && (pile_cnt == nullptr || !pile_cnt_mixed(pile_cnt))) return;
    const int32_t src_begin = LIST ? 0 : src_begin_a;
    const int32_t src_end = LIST ? (int32_t) min(*src_count, (unsigned long long) src_end_a) : src_end_a;       // (list mode: src_end_a = the list's capacity)
    if (LIST && src_end <= 0) return;
    constexpr int CNT_DEFER_AT = LIST ? CNT_DEFERRED2 : CNT_DEFERRED;
    const uint32_t max_stand = o.slot_stride ? (uint32_t) LOCAL_SLOTS_MAX : 2u;    // standing items a source may have and still finish here: one slot in first[], the others in second[]
    constexpr int WC = 4 * EQ - 3;                         // row words of an entry
    constexpr int QW = 24;                                 // staged words per source: the row (<= 13) + the compare's slack, words 16.. stay zero
    constexpr int NS = 12;                                 // source slots: three quads
    __shared__ uint32_t sB[PROBE_WAVES][NS][QW];
    __shared__ uint4 sRun[PROBE_WAVES][NS][CL_RMAX];       // per run: q | p0 << 8 | p1 << 16, cluster key, first entry - slots before the run, entries
    __shared__ unsigned long long sIncl[PROBE_WAVES][NS];  // per source: inclusive prefix of the runs' entry counts, 8 x u8 (saturating)
    __shared__ uint2 sSrc[PROBE_WAVES][NS];                // per source: id, length
    __shared__ unsigned long long sOcc[PROBE_WAVES][NS];   // per source: offsets that hold an item
    __shared__ uint32_t sStat[PROBE_WAVES][NS];            // per source: bit 0 = irregular, bits 8.. = items that stand
    __shared__ uint8_t sT[PROBE_WAVES][8][64];             // per source of the window: lane of the item at offset d
    __shared__ int32_t sDefer[PROBE_WAVES][72];
    __shared__ uint4 sMask[KF > 0 ? 129 : 1];
    // per lane: the entry words KF .. 4 EQ - 1 and zeros up to word WC + 4, from which the overhang of a passing item is read at a
    // dynamic word offset (a register array can only be indexed through a chain of selects: 25 v_cndmask per round)
    constexpr int OVW = KF > 0 ? WC - KF + 5 : 1;          // words per lane (odd for every instantiated shape: conflict-free stride)
    constexpr int OVE = KF > 0 ? 4 * EQ - KF : 0;          // of which the entry supplies the first OVE
    __shared__ uint32_t sOv[PROBE_WAVES][KF > 0 ? 64 : 1][OVW];
    const int wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    const int lane = lane_id();
    const int g = lane >> 4, gl = lane & 15;
    for (int k = lane; k < NS * 8; k += 64) sB[wave][k >> 3][16 + (k & 7)] = 0u;       // the slack words, once
    if constexpr (KF > 0) { for (int k = OVE; k < OVW; k++) sOv[wave][lane][k] = 0u; }
    if constexpr (KF > 0) {
        for (int t = (int) threadIdx.x; t <= 128; t += PROBE_WAVES * 64)
            sMask[t] = make_uint4(low_bits32(t), low_bits32(t - 32), low_bits32(t - 64), low_bits32(t - 96));
        __syncthreads();
    }
    wave_lds_fence();
    uint64_t st_raw = 0, st_slots = 0, st_win = 0, st_rec = 0, st_cmp = 0, st_rounds = 0;
    int n_defer = 0;                                       // uniform: sources waiting in sDefer
    int n_seen = 0, n_deferred = 0;                        // uniform: sources of this wave that took part / that it deferred
    const int step = (int) gridDim.x * PROBE_WAVES * 4;
    const int kfull = KF ? KF : (2 * cfg.Lmin) >> 5;
    const int Lbig = cfg.rsoemo > cfg.Lmin ? cfg.rsoemo : cfg.Lmin;
    auto flush_defer = [&]() {                             // convergent
        if (n_defer == 0) return;
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(&o.counters[CNT_DEFER_AT], (unsigned long long) n_defer);
        base = ((unsigned long long) (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) (base >> 32)) << 32) | (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) base);
        for (int k = lane; k < n_defer; k += 64)
            if (base + (unsigned long long) k < (unsigned long long) defer_cap) defer_list[base + (unsigned long long) k] = sDefer[wave][k];
        wave_lds_fence();
        n_defer = 0;
    };
    auto defer_rows = [&](bool dfr, int id) {              // convergent: the leaders with `dfr` put their source on the list
        const uint64_t dm = __ballot(dfr);
        if (dm != 0ull) {                                  // uniform
            if (dfr) sDefer[wave][n_defer + (int) __builtin_amdgcn_mbcnt_hi((uint32_t) (dm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) dm, 0u))] = id;
            n_defer += __popcll(dm);
            n_deferred += __popcll(dm);
            wave_lds_fence();
            if (n_defer >= 60) flush_defer();
        }
    };

    // ---- source stream: quads of consecutive sources, one per 16-lane row; stage A (registers a_*) = identity, length and row word of
    //      the quad after the one stage B holds; stage B (b_*) = the quad that is staged next: its A data one iteration old plus its
    //      run list, which is loaded by id.  Every lane issues every load, unconditionally, at clamped addresses: a load under a
    //      branch makes hipcc wait for it right there.
    //      BYKEY = false: the sources are the ids src_begin .. src_end - 1, rows and lengths from the node set.
    //      BYKEY = true : the sources are taken in the order of the ENTRY ARRAY (positions src_begin .. src_end - 1 of it): id, length
    //      and row come from the source's own entry (48 sequential bytes).  Consecutive sources then share their prefix minimizer
    //      -- the same genomic locus -- so the sources of a round, and the waves of a workgroup, look up the same directory records
    //      and the same entries: one fetch serves several lanes and the rest hits the caches, where the id order touched ~1.1 KB
    //      of HBM per source in isolated lines (150 GB per launch at the north-star size against 64 GB). ----
    const int pre_words = BYKEY ? (4 * EQ - 3) : (nd.stride < 16 ? nd.stride : 16);
    const int last_src = src_end - 1, col = gl < pre_words ? gl : pre_words - 1;
    const uint32_t *run_w = reinterpret_cast<const uint32_t *>(runs);
    int posA = src_begin + 4 * ((int) blockIdx.x * PROBE_WAVES + wave), posB = posA;
    bool a_valid = false, b_valid = false;
    int a_id = 0, a_len = 0, b_id = 0, b_len = 0;
    uint32_t a_word = 0, b_word = 0, b_nrw = 0;
    uint2 b_run = make_uint2(0u, 0u);
    // list mode: the id of a position is itself a load, and the row is fetched BY that id -- it is read one stage earlier (a_lid: the quad at
    // posA, n_lid: the quad stage A moves to next), so that no load waits for another inside fetchA
    auto next_pos = [&](int p) { return step <= src_end - p ? p + step : src_end; };
    auto list_at = [&](int pos) -> int {
        const int j = pos + g, js = j < last_src ? j : last_src;
        return (int) min((uint32_t) src_list[js], (uint32_t) nd.n - 1u);
    };
    int a_lid = 0, n_lid = 0;
    if constexpr (LIST) { a_lid = list_at(posA); n_lid = list_at(next_pos(posA)); }
    auto fetchA = [&]() {
        const int j = posA + g;
        a_valid = j < src_end;
        const int js = j < last_src ? j : last_src;
        if constexpr (LIST) {
            a_id = a_lid;
            a_len = nd.len[a_lid];
            a_word = nd.words[(size_t) a_lid * nd.stride + col];
        } else
        if constexpr (BYKEY) {
            const uint32_t *ent = reinterpret_cast<const uint32_t *>(store) + (size_t) js * (4 * EQ);
            a_word = ent[col];
            a_id = (int) min(ent[4 * EQ - 3], (uint32_t) nd.n - 1u);      // (clamped: the run list is fetched by this id)
            a_len = (int) ((ent[4 * EQ - 1] >> 8) & 0xFFFu);
        } else {
            a_id = js;
            a_len = nd.len[js];
            a_word = nd.words[(size_t) js * nd.stride + col];
        }
    };
    auto fetchB = [&]() {                                  // (b_id is a valid node id even where b_valid is false: clamped positions)
        b_run = runs[(size_t) b_id * CL_RMAX + (gl & (CL_RMAX - 1))];
        b_nrw = run_w[(size_t) b_id * (2 * CL_RMAX) + 1];  // run 0, second word: nruns in its top byte
    };
    auto take = [&](int &id, int &len, int &nrn, uint32_t &word, uint2 &run) {   // the quad of stage B, masked
        id = b_id;
        len = b_valid ? b_len : 0; nrn = b_valid ? (int) (b_nrw >> 24) : 0;
        word = (b_valid && gl < pre_words) ? b_word : 0u;
        run = (b_valid && gl < CL_RMAX) ? b_run : make_uint2(0u, 0u);
    };
    auto advance = [&](bool adv) {                         // adv (uniform): stage B takes the quad of stage A, stage A moves on
        b_valid = adv ? a_valid : b_valid; b_id = adv ? a_id : b_id; b_len = adv ? a_len : b_len; b_word = adv ? a_word : b_word;
        posB = adv ? posA : posB;
        posA = adv ? (step <= src_end - posA ? posA + step : src_end) : posA;
        if constexpr (LIST) { a_lid = adv ? n_lid : a_lid; n_lid = list_at(next_pos(posA)); }
        fetchB();
        fetchA();
    };
    fetchA();
    // stage 1 of a quad into the buffer `qb` (slots 4 qb .. 4 qb + 3): rows -> LDS; bucket of each run
    auto stage = [&](int qb, int id, int lenB, int nr_eff, uint32_t word0, const uint2 &run) -> uint32_t {
        wave_lds_fence();
        sB[wave][4 * qb + g][gl] = gl < blocks_of(lenB) ? word0 : 0u;
        if (gl == 0) sSrc[wave][4 * qb + g] = make_uint2((uint32_t) id, (uint32_t) lenB);
        return gl < nr_eff ? run.x >> cc.idx_shift : 0u;
    };
    auto index_loads = [&](uint32_t bucket, uint4 &rec) { rec = dir[bucket]; };   // every lane, no branch
    auto finish_runs = [&](int qb, int nr_eff, const uint2 &run, const uint4 &rec) -> uint32_t {
        uint32_t e0, cnt;
        run_slice(rec, run.y, e0, cnt);
        cnt = gl < nr_eff ? cnt : 0u;                      // lanes gl >= 8 hold no run
        uint32_t inc = cnt, t;                             // inclusive scan over the runs (lanes 0..7 of the row)
        t = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) inc, 0x111, 0xF, 0xF, true); inc += t;
        t = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) inc, 0x112, 0xF, 0xF, true); inc += t;
        t = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) inc, 0x114, 0xF, 0xF, true); inc += t;
        if (gl < CL_RMAX) {
            sRun[wave][4 * qb + g][gl] = make_uint4(run.y, run.x, e0 - (inc - cnt), cnt);
            reinterpret_cast<uint8_t *>(&sIncl[wave][4 * qb + g])[gl] = (uint8_t) (inc > 255u ? 255u : inc);
        }
        if (gl == 0) { sOcc[wave][4 * qb + g] = 0ull; sStat[wave][4 * qb + g] = 0u; }
        wave_lds_fence();
        return inc;
    };
    // entries (one byte each, <= 64) / packed / takes-part of the four sources of a quad: wave-uniform, kept in few scalar registers
    // (the window's bookkeeping must not spill: every scalar beyond ~100 costs a v_readlane per use inside the loop)
    auto quad_plan = [&](uint32_t inc, int nrv, uint32_t &t4w, uint32_t &fl /* pk | tp << 4 */) {
        t4w = 0u; fl = 0u;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int T = __builtin_amdgcn_readlane((int) inc, q * 16 + CL_RMAX - 1), nrq = __builtin_amdgcn_readlane(nrv, q * 16);
            const bool p = nrq != 0 && nrq != CL_RUNS_FLAGGED && T <= 64;
            t4w |= (p ? (uint32_t) T : 0u) << (8 * q);
            fl |= (p ? 1u << q : 0u) | (nrq != 0 ? 16u << q : 0u);
        }
    };

    // ---- the window: quad 0 (buffer b0, the older one, consumed from `cur` on), quad 1 (buffer b1), quad 2 being staged into b2 ----
    uint64_t tw = 0ull;                                    // entries of the window's eight sources, one byte each
    uint32_t fl0 = 0u, fl1 = 0u;                           // quad 0 / quad 1: bit g = source g is packed, bit 4 + g = it takes part
    int b0 = 0, b1 = 1, b2 = 2, cur = 0;
    uint32_t red = 0u;                                     // leader lanes: bit 0 / 1 = the row's source of quad 0 / quad 1 is finished
    int pos0 = posA;                                       // stream position of quad 0
    bool have0 = posA < src_end;
    if (have0) {
        int id, len, nrv; uint32_t word0; uint2 run; uint4 rec;
        advance(true);                                     // stage B = the first quad
        take(id, len, nrv, word0, run);
        pos0 = posB;
        advance(true);
        {
            const int ne = nrv == CL_RUNS_FLAGGED ? 0 : nrv;
            const uint32_t bk = stage(0, id, len, ne, word0, run);
            index_loads(bk, rec);
            const uint32_t inc = finish_runs(0, ne, run, rec);
            uint32_t t4w;
            quad_plan(inc, nrv, t4w, fl0);
            tw = (uint64_t) t4w;
        }
        const bool have1 = posB < src_end;                 // the second quad (possibly past the end: then it is empty)
        take(id, len, nrv, word0, run);
        advance(have1);
        {
            const int ne = nrv == CL_RUNS_FLAGGED ? 0 : nrv;
            const uint32_t bk = stage(1, id, len, ne, word0, run);
            index_loads(bk, rec);
            const uint32_t inc = finish_runs(1, ne, run, rec);
            uint32_t t4w;
            quad_plan(inc, nrv, t4w, fl1);
            tw |= (uint64_t) t4w << 32;
        }
    }
    bool bail = false;
    while (have0 && !bail) {                               // uniform; one ROUND per iteration
        // ---- (0) what this round takes: up to five sources of the window from `cur` on, for as long as they fit the 64 lanes; never
        //      the window's last source (the quad after quad 1 is only ready at the end of this round).  Scalar packing and the
        //      lanes' look-up of their source in one pass. ----
        const uint64_t tsh = tw >> (8 * cur);
        int fill_r = 0, cntg = 0, a = 0, taken = 0;
        {
            bool open = true;
#pragma unroll
            for (int k = 0; k < 5; k++) {
                const int tk = (int) ((tsh >> (8 * k)) & 255ull);
                const bool in = open && fill_r + tk <= 64 && (k < 4 || cur < 3);
                open = in;
                const int lo = in ? fill_r : 64;           // first lane of the source in this round
                const bool c = lane >= lo;
                cntg += c ? 1 : 0;
                a = c ? lo : a;
                fill_r += in ? tk : 0;
                taken += in ? 1 : 0;
            }
        }
        const int e = cur + taken;
        const bool rot = e >= 4;                           // quad 0 retires at the end of this round
        // ---- (1) the quad that is staged next comes off the stream before any entry load is issued ----
        const bool have2 = posB < src_end;
        int nB, nlenB, nnr; uint32_t nword0; uint2 nrun;
        take(nB, nlenB, nnr, nword0, nrun);
        advance(rot && have2);                                         // past the end: clamped reads of the last source, never used
        const int nnr_eff = nnr == CL_RUNS_FLAGGED ? 0 : nnr;
        // ---- (2) lanes -> sources of this round, entries: loads issued ----
        const bool ev = lane < fill_r;
        const int s8 = ev ? cur + cntg - 1 : cur;          // window source of this lane (0 .. 6)
        const int slot = 4 * (s8 < 4 ? b0 : b1) + (s8 & 3);
        const int hl = ev ? lane - a : 0;                  // slot of this lane in its source
        const uint2 srec = sSrc[wave][slot];
        const int Bs = (int) srec.x, lenBs = (int) srec.y;
        const uint32_t *sb = sB[wave][slot];
        uint4 rp = make_uint4(0u, 0u, 0u, 0u);
        {
            // run of slot hl: the number of runs whose inclusive prefix is <= hl (binary search over 8 packed bytes)
            const unsigned long long inc8 = sIncl[wave][slot];
            const uint32_t lo = (uint32_t) inc8, hi = (uint32_t) (inc8 >> 32);
            int ri = (int) ((lo >> 24) & 255u) <= hl ? 4 : 0;
            { const uint32_t wv = ri ? hi : lo; ri += (int) ((wv >> 8) & 255u) <= hl ? 2 : 0; }
            { const uint32_t wv = ri >= 4 ? hi : lo; ri += (int) ((wv >> (8 * (ri & 3))) & 255u) <= hl ? 1 : 0; }
            if (ev) rp = sRun[wave][slot][ri & (CL_RMAX - 1)];
        }
        const size_t ei = ev ? (size_t) min(rp.z + (uint32_t) hl, (uint32_t) nd.n - 1u) : (size_t) 0;      // clamped: a corrupt directory must not fault
        uint32_t ew[4 * EQ];
#pragma unroll
        for (int c = 0; c < EQ; c++) { const uint4 v = store[ei * EQ + c]; ew[4 * c] = v.x; ew[4 * c + 1] = v.y; ew[4 * c + 2] = v.z; ew[4 * c + 3] = v.w; }
        // ---- (3) the quad after quad 1: rows staged into the free buffer, index loads issued behind the entry loads ----
        uint32_t nbk = 0;
        uint4 nrec;
        if (have2) nbk = stage(b2, nB, nlenB, nnr_eff, nword0, nrun);
        index_loads(nbk, nrec);
        // ---- (4) verify: one entry per lane ----
        const uint32_t id = ew[4 * EQ - 3], eh = ew[4 * EQ - 2], meta = ew[4 * EQ - 1];
        const int lenC = (int) ((meta >> 8) & 0xFFFu);
        int p = (int) (rp.x & 255u) - (int) (meta & 255u);
        // (bitwise &: one straight line of compares instead of a chain of exec-mask branches)
        const bool ok = ev & same_cluster(eh, rp.y, cc.idx_shift - CL_MBITS) & (p >= (int) ((rp.x >> 8) & 255u)) & (p < (int) ((rp.x >> 16) & 255u)) & ((int) id != Bs) &
                        (lenC >= lenBs - p);
        p = ok ? p : 0;
        const int L = lenBs - p, nb = 2 * L;
        bool pass;
        {
            const int qw = (2 * p) >> 5, sh = (2 * p) & 31;
            uint32_t y[WC + 1];
#pragma unroll
            for (int k = 0; k <= WC; k++) y[k] = sb[qw + k];
            uint32_t diff = 0;
            uint32_t mk[4] = {0u, 0u, 0u, 0u};
            if constexpr (KF > 0) { const uint4 m4 = sMask[min(max(nb - 32 * KF, 0), 128)]; mk[0] = m4.x; mk[1] = m4.y; mk[2] = m4.z; mk[3] = m4.w; }
#pragma unroll
            for (int k = 0; k < WC; k++) {
                const uint32_t x = funnel(y[k], y[k + 1], sh) ^ ew[k];
                if (k < kfull) diff |= x;
                else if (KF > 0 && k < KF + 4) diff |= x & mk[(k - KF) & 3];
                else diff |= x & low_bits32(nb - 32 * k);
            }
            pass = ok && diff == 0;
        }
        const uint64_t pm = __ballot(pass);
        const int d = p;
        const uint32_t v_m = (uint32_t) p | ((uint32_t) lenC << 9) | ((meta & CL_META_FROM) ? ITEM_FROM : 0u);
        uint32_t stv = 0u;                                 // status word of this lane's source after the reduction
        bool has_pred = false;
        int n_tried2 = 0;
        if (pm != 0ull) {                                  // uniform
            uint4 v_o = make_uint4(0u, 0u, 0u, 0u);
            if (pass) {                                    // overhang: C's row from bit 2L on (see k_probe_clustered)
                const int ws = nb >> 5, r2 = nb & 31;
                uint32_t x[5];
                if constexpr (KF > 0) {
                    uint32_t *ov = sOv[wave][lane];        // (this lane's own words: LDS operations of a wave complete in order)
#pragma unroll
                    for (int k = 0; k < OVE; k++) ov[k] = ew[KF + k];
                    const int t = min(max(ws - KF, 0), WC - KF);
#pragma unroll
                    for (int k = 0; k < 5; k++) x[k] = ov[t + k];
                } else {
                    const uint32_t *er = reinterpret_cast<const uint32_t *>(store + ei * EQ);
#pragma unroll
                    for (int k = 0; k < 5; k++) x[k] = er[ws + k];
                }
                v_o = make_uint4(funnel(x[0], x[1], r2), funnel(x[1], x[2], r2), funnel(x[2], x[3], r2), funnel(x[3], x[4], r2));
            }
            // ---- fused single-survivor reduction, every source of the round at once ----
            unsigned long long *occp = &sOcc[wave][slot];
            uint32_t *stp = &sStat[wave][slot];
            uint8_t *Tb = sT[wave][s8];
            if (pass) { atomicOr(occp, 1ull << d); Tb[d] = (uint8_t) lane; }
            wave_lds_fence();
            const uint64_t occ = *occp;                    // (plain LDS reads: a volatile access becomes a flat load and drains vmcnt)
            const bool clash = pass && Tb[d] != (uint8_t) lane;        // another item of this source sits at the same offset
            const uint64_t below = pass ? (occ & ((1ull << d) - 1ull)) : 0ull;
            has_pred = below != 0ull;
            const int j = has_pred ? (int) Tb[63 - __clzll((long long) below)] : lane;
            Ovh<1> oi;
            oi.w[0] = v_o.x; oi.w[1] = v_o.y; oi.w[2] = v_o.z; oi.w[3] = v_o.w;
            const int rho = lenC - (lenBs - d);
            // is the item on lane jj (of this lane's source, at a smaller offset) a via of this lane's item?  (convergent: cross-lane reads)
            auto via_of = [&](int jj, bool want) -> bool {
                const uint32_t Cj = bperm(id, jj), mjj = bperm(v_m, jj);
                Ovh<1> oj;
                oj.w[0] = bperm(v_o.x, jj); oj.w[1] = bperm(v_o.y, jj); oj.w[2] = bperm(v_o.z, jj); oj.w[3] = bperm(v_o.w, jj);
                if constexpr (KF > 0) {
                    // via_ok<1> (prefsuf_device.h) with the four word masks of the overhang compare from the table in LDS
                    const int dj = (int) (mjj & 511u), lenj = (int) ((mjj >> 9) & 511u);
                    const int rho_j = lenj - (lenBs - dj), Lv = lenj - (d - dj);
                    const bool vok = ((mjj & ITEM_FROM) != 0u) & (Cj != id) & (dj < d) & (Lv >= Lbig) & (rho_j <= rho) & ((rho_j > 0) | ((int) Cj > Bs));
                    const uint4 m4 = sMask[min(max(2 * rho_j, 0), 128)];
                    const uint32_t df = ((oi.w[0] ^ oj.w[0]) & m4.x) | ((oi.w[1] ^ oj.w[1]) & m4.y) | ((oi.w[2] ^ oj.w[2]) & m4.z) | ((oi.w[3] ^ oj.w[3]) & m4.w);
                    return want & vok & (df == 0u);
                } else return want && via_ok<1>(Bs, lenBs, Lbig, Cj, mjj, oj, id, d, rho, oi);
            };
            bool removed = via_of(j, has_pred);
            // (round 5) not removed by the nearest predecessor and there are others: they are tried one after the other, nearest first, as
            // far down as a via can sit at all (an item more than Lcap - 1 - Lbig offsets before this one cannot reach it with a big overlap
            // even at the longest read length).  Any item before this one may be its via (the all-pairs rule, prefsuf_device.h:
            // local_reduce), so the walk decides what the nearest predecessor alone left "undecided" -- and handed to the general kernel:
            // a read with a sequencing error in its overhang is no via of anything behind it and is itself implied by nobody; at LOW error
            // rates most sources are clean but meet one such target among their ~11 (0.2 % substitutions at the north-star size: 33.5 M
            // of 90.6 M sources handed on, 46 of the build's 85 ms in the general kernel).
            const int reach = (cfg.Lcap - 1) - Lbig;         // a via sits at most this many offsets before its item
            uint64_t cand = 0ull;
            if (pass && has_pred && !removed) {
                cand = below & ~(1ull << (63 - __clzll((long long) below)));                       // the nearest has been tried
                if (d - reach > 0) cand &= ~((1ull << (d - reach)) - 1ull);                          // (d <= 63)
            }
            while (__ballot(cand != 0ull) != 0ull) {       // uniform
                const bool need = cand != 0ull;
                const int top = need ? 63 - __clzll((long long) cand) : 0;
                const int j2 = need ? (int) Tb[top] : lane;
                const bool r2 = via_of(j2, need);
                removed = removed | r2;
                cand = (need && !r2) ? cand & ~(1ull << top) : 0ull;
                n_tried2 += need ? 1 : 0;
            }
            // every item that could be a via has been asked: nothing is left undecided (what still sends a source to the general kernel:
            // two items at one offset, more than LOCAL_SLOTS_MAX standing items, more than 64 entries, more than 8 runs, a flagged run list).
            // Tried and dropped (round 5): a second item per offset (two reads that start at one place and differ by a sequencing error: 68 % of
            // what is still handed on at 0.2 % substitutions).  It works -- 9.5 M -> 0.14 M sources handed on -- but as tables in LDS it costs the
            // kernel its sixth wave per SIMD (the whole build 48.3 ms instead of 49.4, error-free pairwise builds 7 ms slower), packed into one
            // word per source it spills registers (error-free pairwise builds 20.5 -> 22.5 ms).
            const bool keep = pass && !removed;
            if (clash) atomicOr(stp, 1u);
            if (keep) atomicAdd(stp, 0x100u);
            wave_lds_fence();
            stv = *stp;
            if (keep && stv == 0x100u) {                   // the only item that stands: the source's edge
                o.first[Bs - o.src_base] = ((unsigned long long) id << 32) | (uint32_t) d;
                o.deg[Bs - o.src_base] = 1u;
                st_rec++;
            }
            // Several items stand -- a coverage gap too long for any big via (1.7 % of the sources at 30x), or, with sequencing errors, targets with
            // an error in their overhang, which nobody implies (at 0.2 % substitutions most of what this kernel handed on: round 5) -- : the rest
            // of local_reduce's "several stand" branch for two to `max_stand` of them: the per-source cap (with one item per offset the three
            // largest small (L, C) are the three small items at the smallest offsets), "the same target at a smaller offset supersedes", and the
            // edges go to the source's slots: first[], then second[] with `slot_stride` entries per further slot.
            const uint32_t nst = stv >> 8;
            const bool seg_multi = ev && (stv & 255u) == 0u && nst >= 2u && nst <= max_stand;        // (the same in every lane of a source)
            if (__ballot(seg_multi) != 0ull) {             // uniform
                uint64_t segm = 0ull;                      // the lanes of this lane's source
#pragma unroll
                for (int q = 0; q < 8; q++) { const uint64_t mq = __ballot(ev && s8 == q); segm = s8 == q ? mq : segm; }
                const uint64_t km = __ballot(keep) & segm;                      // the standing items of this lane's source
                const int ds0 = lenBs - cfg.rsoemo + 1;                         // first offset of a small overlap
                const uint64_t lowm = ds0 <= 0 ? 0ull : (ds0 >= 64 ? ~0ull : ((1ull << ds0) - 1ull));
                const bool my_kept = pass && (d < ds0 || __popcll(below & ~lowm) < 3);
                bool fin_me = false;                       // this lane's item is an edge of its source
                uint64_t kmr = seg_multi ? km : 0ull;
                for (int it = 0; it < LOCAL_SLOTS_MAX; it++) {           // uniform: one standing item of every source per step
                    if (__ballot(kmr != 0ull) == 0ull) break;
                    const bool on = kmr != 0ull;
                    const int sl = on ? __builtin_ctzll(kmr) : lane;
                    kmr = on ? kmr & (kmr - 1ull) : 0ull;
                    const uint32_t Cs = bperm(id, sl);
                    const int dsl = (int) bperm((uint32_t) d, sl);
                    const bool ks = bperm(my_kept ? 1u : 0u, sl) != 0u;
                    const uint64_t supm = __ballot(on && my_kept && id == Cs && d < dsl) & segm;   // a kept item of the same target further left
                    if (on && lane == sl) fin_me = ks && supm == 0ull;
                }
                const uint64_t F = __ballot(fin_me) & segm;
                if (fin_me) {
                    const int r = __popcll(F & ((1ull << lane) - 1ull));
                    const unsigned long long mine = ((unsigned long long) id << 32) | (uint32_t) d;
                    if (r == 0) { o.first[Bs - o.src_base] = mine; o.deg[Bs - o.src_base] = (uint32_t) __popcll(F); }
                    else o.second[(size_t) (r - 1) * o.slot_stride + (size_t) (Bs - o.src_base)] = mine;
                    st_rec++;
                }
            }
        }
        if (STATS) {
            st_rounds++;
            const bool fin = ev && (stv & 255u) == 0u && (stv >> 8) <= max_stand;      // this lane's source finishes here
            st_slots += fin; st_raw += fin && pass; st_cmp += (fin && has_pred ? 1 : 0) + (fin ? n_tried2 : 0);
        }
        // ---- (5) the leader of a row: are its sources of quad 0 / quad 1 finished?  (status word of the round that packed them) ----
        if (gl == 0) {
            if (g >= cur && g < e) {
                const uint32_t sv = sStat[wave][4 * b0 + g];
                const bool ok = ((fl0 >> g) & 1u) != 0u && (sv & 255u) == 0u && (sv >> 8) <= max_stand;
                red = ok ? red | 1u : red & ~1u;
                if (STATS && ok) st_win += (uint64_t) ((int) sSrc[wave][4 * b0 + g].y - cfg.Lmin + 1);
            }
            if (4 + g >= cur && 4 + g < e) {
                const uint32_t sv = sStat[wave][4 * b1 + g];
                const bool ok = ((fl1 >> g) & 1u) != 0u && (sv & 255u) == 0u && (sv >> 8) <= max_stand;
                red = ok ? red | 2u : red & ~2u;
                if (STATS && ok) st_win += (uint64_t) ((int) sSrc[wave][4 * b1 + g].y - cfg.Lmin + 1);
            }
        }
        // ---- (6) the staged quad's index loads have had the time of (4) to land ----
        const uint32_t ninc = finish_runs(b2, nnr_eff, nrun, nrec);
        cur = e;
        if (rot) {                                         // uniform: quad 0 is done
            // its sources that did not finish here: to the general kernel
            n_seen += __popc(fl0 >> 4);
            defer_rows(gl == 0 && ((fl0 >> (4 + g)) & 1u) != 0u && (red & 1u) == 0u, (int) sSrc[wave][4 * b0 + g].x);
            // the window moves on by one quad
            uint32_t t4w, fl2;
            quad_plan(ninc, nnr, t4w, fl2);
            tw = (tw >> 32) | ((uint64_t) t4w << 32);
            fl0 = fl1; fl1 = fl2;
            red >>= 1;
            const int bt = b0; b0 = b1; b1 = b2; b2 = bt;
            pos0 = step <= src_end - pos0 ? pos0 + step : src_end;
            have0 = pos0 < src_end;
            cur -= 4;
            bail = n_seen >= 192 && 2 * n_deferred > n_seen;
        }
    }
    if (bail && have0) {
        // most of this wave's sources are irregular: what quad 0 has already had packed is settled like a retirement, the rest of the
        // wave's share goes to the general kernel unseen
        defer_rows(gl == 0 && g < cur && ((fl0 >> (4 + g)) & 1u) != 0u && (red & 1u) == 0u, (int) sSrc[wave][4 * b0 + g].x);
        bool first = true;
        for (int q = pos0; q < src_end; q = step <= src_end - q ? q + step : src_end) {     // uniform
            const int j = q + g < src_end ? q + g : last_src;
            const int b = BYKEY ? (int) reinterpret_cast<const uint32_t *>(store)[(size_t) j * (4 * EQ) + 4 * EQ - 3]
                                : (LIST ? (int) min((uint32_t) src_list[j], (uint32_t) nd.n - 1u) : j);
            const bool dfr = gl == 0 && q + g < src_end && !(first && g < cur) && (run_w[(size_t) b * (2 * CL_RMAX) + 1] >> 24) != 0u;
            defer_rows(dfr, b);
            first = false;
        }
    }
    flush_defer();
    st_rec = wave_sum_u64(st_rec);
    if (lane == 0 && st_rec) atomicAdd(&o.counters[CNT_VALID_RECORDS], (unsigned long long) st_rec);
    if (STATS) {
        st_raw = wave_sum_u64(st_raw); st_slots = wave_sum_u64(st_slots); st_win = wave_sum_u64(st_win); st_cmp = wave_sum_u64(st_cmp);
        if (lane == 0) {
            atomicAdd(&o.counters[CNT_RAW], (unsigned long long) st_raw);
            atomicAdd(&o.counters[CNT_SLOTS], (unsigned long long) st_slots);
            atomicAdd(&o.counters[CNT_WINDOWS], (unsigned long long) st_win);
            atomicAdd(&o.counters[CNT_TR_COMPARES], (unsigned long long) st_cmp);
            atomicAdd(&o.counters[CNT_ROUNDS], (unsigned long long) st_rounds);
        }
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
// Does the clustered probe take this input?  Fills the k-mer / index geometry and the entry size (16-byte pieces: row words + 3).
//   rows of up to 4 * CL_MAX_EQ - 3 words; at most 64 suffix windows (the one-word form of the source-side reduction);
//   w = Lmin - k + 1 k-mers per window with nwin <= w <= 64 (k_node_runs splits a window over two blocks of w k-mers) and
//   k >= CL_KMIN_HARD: k = max(Lmin - 63, min(Lmin, CL_KMIN)), lowered to 2 Lmin - max_len when the windows outnumber w.
bool cluster_plan(const PrefSufCfg &cfg, int max_len, uint64_t live, int bucket_log2_bias, ClusterCfg *c, int *eq) {
    const int W = blocks_of(max_len);
    int e = (W + 3 + 3) / 4;
    if (e < 2) e = 2;
    const int nwin = max_len - cfg.Lmin + 1;
    // up to 128 suffix windows: 64 take the one-word form of the source-side reduction (and k_probe_stream), 65 .. 128 the two-word form
    // (k_node_runs<., true>: the windows in two halves; k_probe_clustered<., ., ., 2>)
    if (e > CL_MAX_EQ || nwin < 1 || nwin > 128) return false;
    // The minimizer of a window is chosen among its first w = min(Lmin - k + 1, 64) k-mers (they all lie inside the window, so the choice
    // is a function of the window's content: a source window and the equal target prefix agree on it -- that is all the join needs).
    // k <= 32 (two hash words); the window minimum by two blocks needs the windows of one half (<= 64) not to outnumber w.
    int kk = std::min(32, std::max(cfg.Lmin - 63, std::min(cfg.Lmin, CL_KMIN)));
    const int nw_half = std::min(nwin, 64);
    if (nw_half > std::min(cfg.Lmin - kk + 1, 64)) kk = cfg.Lmin - nw_half + 1;
    if (kk < CL_KMIN_HARD || kk > 32 || kk > cfg.Lmin) return false;
    c->kk = kk;
    c->w = std::min(cfg.Lmin - kk + 1, 64);
    c->lo_mask = kk >= 16 ? 0xFFFFFFFFu : ((1u << (2 * kk)) - 1u);
    c->hi_mask = kk <= 16 ? 0u : (kk >= 32 ? 0xFFFFFFFFu : ((1u << (2 * kk - 32)) - 1u));
    int bits = 4;
    while (bits < 28 && (1ull << bits) < live) bits++;     // ~one entry per bucket: a lookup returns its cluster and little else
    bits = std::max(4, std::min(32 - CL_MBITS, bits + bucket_log2_bias));      // the low CL_MBITS key bits carry m_C, not the cluster
    c->n_buckets = 1u << bits;
    c->idx_shift = 32 - bits;
    *eq = e;
    return true;
}

size_t cluster_sort_temp_bytes(uint64_t n) { return sort_u32_pairs_temp_bytes(n); }

// minimizer keys, entry words and runs of the nodes node_begin .. node_end - 1 (per-node arrays, written at the node's own index).
// with_runs = false: the keys (and meta words) alone -- the key pass of a build whose run lists are made elsewhere (prefsuf_pile.hip).
void launch_cluster_keys(const NodesDev &nd, const PrefSufCfg &cfg, const ClusterCfg &cc, int32_t node_begin, int32_t node_end, uint32_t *keys, uint32_t *vals,
                         uint32_t *meta, void *runs, uint8_t *nruns, hipStream_t s, bool with_runs, const unsigned long long *only_if_declined) {
    if (node_end <= node_begin) return;
    const uint64_t m = (uint64_t) (node_end - node_begin);
    const unsigned long long *gate = only_if_declined;
    const dim3 grid((unsigned) (gate ? std::min<uint64_t>((m + TK_ROWS - 1) / TK_ROWS, 8192) : (m + TK_ROWS - 1) / TK_ROWS)), block(TK_ROWS);
    const int32_t *nol = nullptr; const unsigned long long *noc = nullptr;
    // rows of up to 9 words (every 100 - 150 bp configuration) stage 11 words per node, longer ones (<= 13 words) 17
    // ... and rows of up to 17 words or nodes with more than 64 suffix windows (250-bp reads) the two-halves form with 21
#define NR_LAUNCH(TKW, WIDE) do { if (with_runs) hipLaunchKernelGGL((k_node_runs<TKW, WIDE, true>), grid, block, 0, s, nd, cfg, cc, (int) node_begin, (int) node_end, keys, vals, meta, (uint2 *) runs, nruns, nol, noc, gate); \
                                  else hipLaunchKernelGGL((k_node_runs<TKW, WIDE, false>), grid, block, 0, s, nd, cfg, cc, (int) node_begin, (int) node_end, keys, vals, meta, (uint2 *) runs, nruns, nol, noc, gate); } while (0)
    if (blocks_of(cfg.Lcap - 1) > 13 || (cfg.Lcap - 1) - cfg.Lmin + 1 > 64) NR_LAUNCH(21, true);
    else if (blocks_of(cfg.Lcap - 1) <= 9) NR_LAUNCH(11, false);
    else NR_LAUNCH(17, false);
#undef NR_LAUNCH
}

// the run lists (and nothing else) of the nodes a device-side list names: ids[0 .. min(*count, cap))
void launch_cluster_runs_list(const NodesDev &nd, const PrefSufCfg &cfg, const ClusterCfg &cc, const int32_t *ids, const unsigned long long *count, uint32_t cap, uint32_t grid_blocks,
                              void *runs, uint8_t *nruns, hipStream_t s) {
    if (cap == 0 || nd.n <= 0) return;
    const dim3 grid(std::max<uint32_t>(1u, std::min<uint32_t>(grid_blocks, (cap + TK_ROWS - 1) / TK_ROWS))), block(TK_ROWS);
    uint32_t *nok = nullptr;
#define NR_LAUNCH(TKW, WIDE) hipLaunchKernelGGL((k_node_runs<TKW, WIDE, true>), grid, block, 0, s, nd, cfg, cc, 0, (int) cap, nok, nok, nok, (uint2 *) runs, nruns, ids, count, (const unsigned long long *) nullptr)
    if (blocks_of(cfg.Lcap - 1) > 13 || (cfg.Lcap - 1) - cfg.Lmin + 1 > 64) NR_LAUNCH(21, true);
    else if (blocks_of(cfg.Lcap - 1) <= 9) NR_LAUNCH(11, false);
    else NR_LAUNCH(17, false);
#undef NR_LAUNCH
}

__global__ void __launch_bounds__(256) k_iota(uint32_t *__restrict__ v, uint32_t n) {
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) v[i] = i;
}

// keys / meta of all n nodes -> the entry array in key order and its bucket directory.  fill_vals: the ids (sort payload) were not
// written by this engine's key pass for every node (keys gathered from other ranks): write them here.  uniform_len > 0: every
// live node has that length and there is no alignFrom mask (k_tgt_gather<., true>: meta[] is not read).
hipError_t launch_cluster_store(const NodesDev &nd, const ClusterCfg &cc, uint32_t *keys, uint32_t *vals, uint32_t *keys2, uint32_t *vals2,
                                void *sort_temp, size_t sort_temp_bytes, void *dir, bool fill_vals,
                                hipEvent_t ev_sorted, unsigned long long *bad_flag, bool test_skip_sort, hipStream_t s, bool own_sort) {
    if (nd.n <= 0) return hipSuccess;
    const uint64_t n = (uint64_t) nd.n;
    // (the sort payload is the node id = the position: the engine's own sort makes it up in its first pass; the library's wants the array)
    if (fill_vals && (!own_sort || test_skip_sort)) hipLaunchKernelGGL(k_iota, dim3((unsigned) std::min<uint64_t>((n + 255) / 256, 8192)), dim3(256), 0, s, vals, (uint32_t) n);
    // The order the directory and the probe need: bucket, then m_C >> 3 (the directory's eight classes).  The key bits below that --
    // the low three of m_C, the cluster bits under the field -- are compared entry by entry by the probe, never searched: they stay
    // unsorted (29 significant bits at the north-star size: three radix passes instead of four).
    hipError_t err = hipSuccess;
    if (test_skip_sort) {                                  // tests only (option "test_unsorted_index"): the keys go on as they are, the directory pass must notice
        err = hipMemcpyAsync(keys2, keys, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, s);
        if (err == hipSuccess) err = hipMemcpyAsync(vals2, vals, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, s);
    } else err = sort_u32_pairs(sort_temp, sort_temp_bytes, keys, keys2, own_sort ? (const uint32_t *) nullptr : vals, vals2, n, cc.idx_shift - 3, s, own_sort);
    if (err != hipSuccess) return err;
    if (ev_sorted) (void) hipEventRecord(ev_sorted, s);
    // (measured and rejected: zero-filling the directory as a side job of the VALU-bound k_node_runs -- that kernel got slower by what
    // the fill costs on its own, 0.24 ms)
    err = hipMemsetAsync(dir, 0, ((size_t) cc.n_buckets + 2) * 16, s);
    if (err != hipSuccess) return err;
    hipLaunchKernelGGL(k_tgt_dir, dim3((unsigned) ((n + 1 + TD_TILE - 1) / TD_TILE)), dim3(TD_TILE), 0, s, (const uint32_t *) keys2, n, cc.idx_shift, cc.n_buckets, 0u, (uint4 *) dir, bad_flag);
    return hipGetLastError();
}

// ... and the entry array itself: the rows in key order (what the pairwise probes walk).  uniform_len > 0: every live node has that length and
// there is no alignFrom mask (k_tgt_gather<., true>: meta[] is not read).  pile_cnt: the counters of the pile path's sample (prefsuf_pile.hip) --
// the kernel leaves at once for a build that path keeps.
hipError_t launch_cluster_gather(const NodesDev &nd, const ClusterCfg &cc, int eq, const uint32_t *keys2, const uint32_t *vals2, const uint32_t *meta, int uniform_len,
                                 void *store, const unsigned long long *pile_cnt, hipStream_t s) {
    if (nd.n <= 0) return hipSuccess;
    const uint64_t n = (uint64_t) nd.n;
    const uint64_t pieces = n * (uint64_t) eq;
    const unsigned g = (unsigned) std::min<uint64_t>((pieces + 255) / 256, 1u << 16);
    const int fs = cc.idx_shift - CL_MBITS;
    const uint32_t um = uniform_len > 0 ? (((uint32_t) uniform_len << 8) | CL_META_FROM) : 0u;
#define TG_LAUNCH(E, U) hipLaunchKernelGGL((k_tgt_gather<E, U>), dim3(g), dim3(256), 0, s, nd, n, keys2, vals2, meta, um, fs, (uint4 *) store, pile_cnt)
#define TG_EQ(E) do { if (uniform_len > 0) TG_LAUNCH(E, true); else TG_LAUNCH(E, false); } while (0)
    if (eq == 2)      TG_EQ(2);
    else if (eq == 3) TG_EQ(3);
    else if (eq == 4) TG_EQ(4);
    else              TG_EQ(5);
#undef TG_EQ
#undef TG_LAUNCH
    return hipGetLastError();
}

// The same for the targets of ONE RANK'S BUCKET RANGE (the seed-bucket-sharded N-GPU build, prefsuf_shard.hip): m (key, id) pairs whose
// bucket lies in [bucket_base, bucket_base + n_buckets_local) -> their entries in bucket order and a directory of that range alone
// (record b - bucket_base), ordered like the full build's (bucket, then m_C >> 3).
hipError_t launch_cluster_store_slice(const NodesDev &nd, const ClusterCfg &cc, int eq, uint64_t m, uint32_t bucket_base, uint32_t n_buckets_local, uint32_t *keys,
                                      uint32_t *vals, uint32_t *keys2, uint32_t *vals2, const uint32_t *meta, int uniform_len, void *sort_temp, size_t sort_temp_bytes,
                                      void *store, void *dir, unsigned long long *bad_flag, hipStream_t s, bool own_sort) {
    hipError_t err = hipMemsetAsync(dir, 0, ((size_t) n_buckets_local + 2) * 16, s);
    if (err != hipSuccess || m == 0) return err;
    err = sort_u32_pairs(sort_temp, sort_temp_bytes, keys, keys2, vals, vals2, m, cc.idx_shift - 3, s, own_sort);
    if (err != hipSuccess) return err;
    const uint64_t n = m, pieces = m * (uint64_t) eq;
    const unsigned g = (unsigned) ((pieces + 255) / 256);
    const int fs = cc.idx_shift - CL_MBITS;
    const uint32_t um = uniform_len > 0 ? (((uint32_t) uniform_len << 8) | CL_META_FROM) : 0u;
#define TG_LAUNCH(E, U) hipLaunchKernelGGL((k_tgt_gather<E, U>), dim3(g), dim3(256), 0, s, nd, n, (const uint32_t *) keys2, (const uint32_t *) vals2, meta, um, fs, (uint4 *) store, (const unsigned long long *) nullptr)
#define TG_EQ(E) do { if (uniform_len > 0) TG_LAUNCH(E, true); else TG_LAUNCH(E, false); } while (0)
    if (eq == 2)      TG_EQ(2);
    else if (eq == 3) TG_EQ(3);
    else if (eq == 4) TG_EQ(4);
    else              TG_EQ(5);
#undef TG_EQ
#undef TG_LAUNCH
    hipLaunchKernelGGL(k_tgt_dir, dim3((unsigned) ((n + 1 + TD_TILE - 1) / TD_TILE)), dim3(TD_TILE), 0, s, (const uint32_t *) keys2, n, cc.idx_shift, n_buckets_local, bucket_base,
                       (uint4 *) dir, bad_flag);
    return hipGetLastError();
}

uint64_t cluster_probe_blocks(int n_cu, uint64_t n_src) {
    return std::max<uint64_t>(1, std::min<uint64_t>((n_src + PROBE_WAVES - 1) / PROBE_WAVES, (uint64_t) std::max(1, n_cu) * CL_OCC));
}

uint64_t cluster_record_slack(int n_cu, uint64_t n_src) { return cluster_probe_blocks(n_cu, n_src) * PROBE_WAVES * (uint64_t) REC_CHUNK_LOCAL; }

// k_probe_stream over the sources src_begin .. src_end - 1: regular sources get their edges, the others go on defer_list.  by_key: the
// range is one of positions of the entry array (which holds an entry for every node), the sources are taken in that order
void launch_probe_stream(const NodesDev &nd, const PrefSufCfg &cfg, const ClusterCfg &cc, int eq, const void *store, const void *dir,
                         const void *runs, const uint8_t *nruns, int32_t src_begin, int32_t src_end, bool by_key, unsigned long long *counters, int n_cu,
                         uint32_t *deg, unsigned long long *first, unsigned long long *second, int32_t *defer_list, uint32_t defer_cap,
                         const unsigned long long *pile_cnt, hipStream_t s, uint32_t slot_stride) {
    const int64_t ns = (int64_t) src_end - src_begin;
    if (ns <= 0) return;
    const uint64_t quads = ((uint64_t) ns + 3) / 4;
    const int kf = (2 * cfg.Lmin) >> 5;
    const int kfs = (eq == 3 && (kf == 5 || kf == 3)) || (eq == 2 && kf == 3) ? kf : 0;       // the instantiated shape
    dim3 grid((unsigned) std::max<uint64_t>(1, std::min<uint64_t>((quads + PROBE_WAVES - 1) / PROBE_WAVES, (uint64_t) std::max(1, n_cu) * clq_occ(eq <= 3 ? eq : 4, kfs, by_key)))),
         block(PROBE_WAVES * 64);
    ProbeOut o{nullptr, nullptr, 0, counters, deg, first, by_key ? 0 : src_begin, second};
    o.slot_stride = slot_stride;
    const uint4 *st = (const uint4 *) store;
#define CLQ_LAUNCH(ST, E, K, BK) hipLaunchKernelGGL((k_probe_stream<ST, E, K, BK>), grid, block, 0, s, nd, cfg, cc, st, (const uint4 *) dir, (const uint2 *) runs, nruns, src_begin, src_end, o, defer_list, defer_cap, pile_cnt, \
                                                    (const int32_t *) nullptr, (const unsigned long long *) nullptr)
#define CLQ_ORDER(ST, E, K) do { if (by_key) CLQ_LAUNCH(ST, E, K, true); else CLQ_LAUNCH(ST, E, K, false); } while (0)
#define CLQ_STATS(E, K) do { if (cfg.stats) CLQ_ORDER(true, E, K); else CLQ_ORDER(false, E, K); } while (0)
    if (eq == 3 && kf == 5)      CLQ_STATS(3, 5);
    else if (eq == 3 && kf == 3) CLQ_STATS(3, 3);
    else if (eq == 2 && kf == 3) CLQ_STATS(2, 3);
    else if (eq == 2)            CLQ_STATS(2, 0);
    else if (eq == 3)            CLQ_STATS(3, 0);
    else                         CLQ_STATS(4, 0);
#undef CLQ_STATS
#undef CLQ_ORDER
#undef CLQ_LAUNCH
}

// The mixed form of a build the pile path keeps (prefsuf_cluster_device.h: pile_cnt_mixed): k_probe_stream over the sources k_pile_probe handed on
// (src_list[0 .. *src_count), the count on the device); what it cannot finish goes on defer2 (counted in CNT_DEFERRED2), and k_defer_swap puts that
// list and its count where the general kernel looks for its sources.  Both kernels leave at once for a build that is not of the mixed form.
__global__ void __launch_bounds__(256) k_defer_swap(const unsigned long long *__restrict__ pile_cnt, unsigned long long *__restrict__ counters, const int32_t *__restrict__ list2,
                                                     int32_t *__restrict__ list, uint32_t cap) {
    if (!pile_cnt_mixed(pile_cnt)) return;
    const uint64_t n2 = min(counters[CNT_DEFERRED2], (unsigned long long) cap);
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (uint64_t) gridDim.x * blockDim.x) list[i] = list2[i];
    if (blockIdx.x == 0 && threadIdx.x == 0) { counters[CNT_DEFERRED_PILE] = counters[CNT_DEFERRED]; counters[CNT_DEFERRED] = n2; }
}

void launch_probe_stream_list(const NodesDev &nd, const PrefSufCfg &cfg, const ClusterCfg &cc, int eq, const void *store, const void *dir, const void *runs, const uint8_t *nruns,
                              int32_t *src_list, uint32_t list_cap, unsigned long long *counters, int n_cu, uint32_t *deg, unsigned long long *first,
                              unsigned long long *second, int32_t *defer2, const unsigned long long *pile_cnt, hipStream_t s, uint32_t slot_stride, int32_t src_base) {
    if (eq != 3 || list_cap == 0) return;                  // (the pile path takes entries of three pieces only: pile_plan)
    const int kf = (2 * cfg.Lmin) >> 5;
    const int kfs = (kf == 5 || kf == 3) ? kf : 0;
    const dim3 grid((unsigned) std::max(1, n_cu) * clq_occ(3, kfs, false)), block(PROBE_WAVES * 64);
    ProbeOut o{nullptr, nullptr, 0, counters, deg, first, src_base, second};       // (src_base: the listed ids are those of a rank's range, the slots count from its first)
    o.slot_stride = slot_stride;
#define CLQ_LIST(K) hipLaunchKernelGGL((k_probe_stream<false, 3, K, false, true>), grid, block, 0, s, nd, cfg, cc, (const uint4 *) store, (const uint4 *) dir, (const uint2 *) runs, nruns, 0, \
                                       (int32_t) std::min<uint32_t>(list_cap, 0x7FFFFFFFu), o, defer2, list_cap, pile_cnt, (const int32_t *) src_list, (const unsigned long long *) (counters + CNT_DEFERRED))
    if (kf == 5) CLQ_LIST(5); else if (kf == 3) CLQ_LIST(3); else CLQ_LIST(0);
#undef CLQ_LIST
    // the list the general kernel reads: (list2, CNT_DEFERRED2) -> (list, CNT_DEFERRED)
    hipLaunchKernelGGL(k_defer_swap, dim3(1024), dim3(256), 0, s, pile_cnt, counters, (const int32_t *) defer2, src_list, list_cap);
}

// src_list == null: the sources are the ids src_begin .. src_end - 1; else the ids src_list[src_begin .. src_end - 1]
void launch_probe_clustered(const NodesDev &nd, const PrefSufCfg &cfg, const ClusterCfg &cc, int eq, const void *store, const void *dir,
                            const void *runs, const uint8_t *nruns, int32_t src_begin, int32_t src_end, const int32_t *src_list, int32_t src_base,
                            uint32_t *rec_dst, unsigned long long *rec_val, uint64_t rec_cap,
                            unsigned long long *counters, int n_cu, uint32_t *deg, unsigned long long *first, const ProbeBig *big,
                            const unsigned long long *list_count, int sw /* 1 | 2: words per offset mask of the source-side form */, hipStream_t s,
                            const uint32_t *skeys, const uint32_t *sids, int uniform_len, const unsigned long long *pile_cnt) {
    const int64_t ns = (int64_t) src_end - src_begin;
    if (ns <= 0) return;
    dim3 grid((unsigned) cluster_probe_blocks(n_cu, (uint64_t) ns)), block(PROBE_WAVES * 64);
    ProbeOut o{rec_dst, rec_val, rec_cap, counters, deg, first, src_base};
    if (big) { o.big_list = big->list; o.big_list_cap = big->list_cap; }
    const uint4 *st = (const uint4 *) store;
    // pile_cnt: the build may have no entry array (decided on the device): the form that reads the rows by id is launched beside the usual one
    const bool maybe_by_id = pile_cnt && skeys && sids && uniform_len > 0 && eq == 3 && sw == 1 && !cfg.stats;
    const ByIdEntries by{skeys, sids, uniform_len > 0 ? (((uint32_t) uniform_len << 8) | CL_META_FROM) : 0u, maybe_by_id ? pile_cnt : nullptr};
    // KF = (2 * Lmin) >> 5 as a compile-time constant for the shapes ALGA's defaults produce (150-bp reads: Lmin 82, rows of 9
    // words; 100-bp reads: Lmin 55, rows of 6 words); 0 = any shape
    const int kf = (2 * cfg.Lmin) >> 5;
#define CL_LAUNCH(ST, E, K) hipLaunchKernelGGL((k_probe_clustered<ST, E, K>), grid, block, 0, s, nd, cfg, cc, st, (const uint4 *) dir, (const uint2 *) runs, nruns, src_begin, src_end, o, src_list, list_count, by)
#define CL_BYID(K) hipLaunchKernelGGL((k_probe_clustered<false, 3, K, 1, true>), grid, block, 0, s, nd, cfg, cc, st, (const uint4 *) dir, (const uint2 *) runs, nruns, src_begin, src_end, o, src_list, list_count, by)
#define CL_STATS(E, K) do { if (cfg.stats) CL_LAUNCH(true, E, K); else CL_LAUNCH(false, E, K); } while (0)
#define CL_LAUNCH2(ST, E) hipLaunchKernelGGL((k_probe_clustered<ST, E, 0, 2>), grid, block, 0, s, nd, cfg, cc, st, (const uint4 *) dir, (const uint2 *) runs, nruns, src_begin, src_end, o, src_list, list_count, by)
#define CL_STATS2(E) do { if (cfg.stats) CL_LAUNCH2(true, E); else CL_LAUNCH2(false, E); } while (0)
    if (sw == 2) {                                         // more than 64 suffix windows (250-bp reads): the two-word form, any row length
        if (eq == 2)      CL_STATS2(2);
        else if (eq == 3) CL_STATS2(3);
        else if (eq == 4) CL_STATS2(4);
        else              CL_STATS2(5);
    }
    else if (eq == 3 && kf == 5) { CL_STATS(3, 5); if (maybe_by_id) CL_BYID(5); }
    else if (eq == 3 && kf == 3) { CL_STATS(3, 3); if (maybe_by_id) CL_BYID(3); }
    else if (eq == 2 && kf == 3) CL_STATS(2, 3);
    else if (eq == 2)            CL_STATS(2, 0);
    else if (eq == 3)            { CL_STATS(3, 0); if (maybe_by_id) CL_BYID(0); }
    else if (eq == 4)            CL_STATS(4, 0);
    else                         CL_STATS(5, 0);
#undef CL_STATS2
#undef CL_LAUNCH2
#undef CL_STATS
#undef CL_LAUNCH
#undef CL_BYID
}

} // namespace alga
