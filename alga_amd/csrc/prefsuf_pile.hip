// alga_amd/csrc/prefsuf_pile.hip -- the probe of the clustered minimizer join through PILES (gfx950).
//
// k_probe_stream (prefsuf_cluster.hip) verifies every (source, entry) pair with a 2-bit compare of the whole overlap and every item
// against its predecessor's overhang: ~16 row compares and ~11 overhang compares per 150-bp source at 30x, one lane each.  On
// error-free reads nearly all of them say what one compare per READ already says.  The targets filed under one minimizer k-mer (one
// bucket of the entry array, or one of the few k-mers that share it) all contain that k-mer, so they lie on ONE coordinate axis --
// the k-mer at [0, k), target C on [-m_C, -m_C + |C|) -- and if each of them equals one CONSENSUS string S on its whole extent (a
// PILE), then
//   * a source B that holds the k-mer at q lies at [-q, -q + |B|) of the same axis, and  B[p..] == C[..|B| - p]  at p = q - m_C  holds
//     for exactly the targets that start right of the LAST position where B differs from S: one compare of B against S per
//     (source, minimizer run) decides every entry of the pile at once, and the set of offsets that hold an item is the pile's set
//     of m_C, mirrored and shifted by q;
//   * two targets of one pile agree wherever both are defined: the overhang compare of the source-side reduction (the via B -> C,
//     src/GraphCreators/GraphCreatorPrefSuf.cpp:434-451) is implied inside a pile, and between the piles of two runs of a source it is
//     ONE compare of the two consensus strings past the source's end.
// With reads of one length and no alignFrom / alignTo mask (every BASELINE configuration) the reduction of a regular source is then a
// function of its 64-bit offset set alone (prefsuf_device.h local_reduce, fast path, with lenC == lenB everywhere): an item is removed
// iff another item sits at most G = len - max(rsoemo, Lmin) offsets before it, one or two items stand, the cap of three small overlaps
// is a population count.  Only the one or two targets that stand are looked up by id.
//
//   k_pile_build   one lane per ENTRY of the entry array: per bucket the groups of equal minimizer k-mer (<= PILE_MAXSUB), per group the
//                  consensus (leftmost-starting | rightmost-ending member), every member verified against it, the mirrored m_C set;
//                  the record of the bucket's first group in a table indexed by the bucket, the further groups' at their entry slots, a byte
//                  per entry (its group) and a 16-byte SIDE record (what k_pile_probe reads of the entry as a source).  A bucket where any
//                  of it fails (a member that differs from the consensus, two members at one m_C, more groups or entries than fit) is
//                  flagged: its sources go to the general kernel.  As <SAMPLE> on the first 1/32 of the entry array it only counts such
//                  buckets: the pile kernels leave a build of reads with errors to k_probe_stream.
//   k_pile_runs    one lane per entry, the lane of a pile's leftmost member works: the RUN LIST of the pile (the minimizer runs of all its
//                  members' windows on the pile's axis), into the second half of the bucket's record.
//   k_pile_probe   one lane per SOURCE (in the order of the entry array): per run the first 64 bytes of the bucket's record, loaded once
//                  per distinct bucket of a wave; regular sources get their edges, the others go on the defer list of k_probe_clustered.
//   k_pile_deg     a streaming pass that moves the out-degree k_pile_probe left in the source's slot to deg[].
// Nothing is approximated: every decision either follows from verified equalities or is handed to the pairwise kernels.
#include <hip/hip_runtime.h>
#include <algorithm>
#include "prefsuf_common.h"
#include "prefsuf_kernels.h"
#include "prefsuf_device.h"
#include "prefsuf_cluster_device.h"

namespace alga {

constexpr int PB_TILE = 192, PB_HALO = 64, PB_THREADS = PB_TILE + PB_HALO;      // buckets that START in the tile; the halo holds their tails.  (round 5: tiles of 512 -- three big blocks per CU, every one stalled behind its row fetches at the same time -- 4.3 ms; 256: 3.9; 192: 3.5; 128: 3.6 + a slower sample)
constexpr int PILE_SW = 13;                    // consensus words: coordinates -64 .. 143 (m_C <= 63, rows of up to 9 words)
constexpr int PILE_MAXSUB = 4;                 // k-mer groups of one bucket (two k-mers share one of 2^26 buckets for ~12 % of the non-empty buckets at the north-star size)
constexpr int PILE_EQ = 3;                     // entry size this path takes: rows of up to 9 words (reads of 100 - 150 bp)
// What a source reads per run is ONE 128-byte line: the bucket's record in a table indexed by the bucket itself (`tab`, 32 words per bucket)
//   w[0 .. 12]  consensus of the bucket's FIRST k-mer group, word k = coordinates -64 + 16 k ..
//   w[13], w[14] bit (63 - m) set: a member of that group with m_C == m
//   w[15]       tag of group 0 .. 3 (bits 0..4, 5..9, 10..14, 15..19) | groups - 1 (bits 20..21) | irregular (bit 22) | epoch of the build that wrote
//               the record (bits 23..31, never 0: the table is not cleared between builds, a record of another epoch is an empty bucket)
//   -- the first 64 bytes are all the run loop reads (the kernel is bound by the number of 64-byte requests that miss its L1) --
//   w[16]       entries (127: more than 64) | runs of the pile's run list << 8 (0: none -- its members read their own lists) | end of the last run's
//               windows (+ 64) << 16;  w[17] first entry of the bucket;  w[18], w[19] the directory's class offsets (k_tgt_dir): the look-up of a target
//   w[20 .. 31] the pile's run list (k_pile_runs_consensus): up to EIGHT runs, windows ascending and contiguous on the pile's axis: w[20 .. 27] their
//               cluster keys, w[28 .. 31] per run (k-mer position + 64) | (first window + 64) << 8 in 16 bits (a run ends where the next begins).
//               A pile's extent holds ~5.4 minimizers at 30x (one per 32 positions of ~174): with six slots one pile in four had no list
// tag = low five bits of the k-mer's cluster key, i.e. of the word a run carries (they lie below the bucket bits): which group a run wants
// without a second read.  The further groups of a bucket (another k-mer in the same bucket: 12 % of the non-empty buckets at the north-star
// size) have 16-word records {consensus, -, set} in `rec` at entry slot first + k, k in the order of the groups' first members.

// ------------------------------------------------------------------------------------------
// Is this a build for the pile path?  pile_cnt = {buckets, irregular buckets} of a SAMPLE of the entry array (its first 1 / 32, in hash order: loci
// from all over the genome): more than one bucket in PILE_IRREGULAR_ONE_IN (250) irregular -- reads with sequencing errors (one in four), or a
// genome so large that a 19-mer often sits at two loci (error-free reads of 1 Gb: 1.5 %) -- and the sources that would have to go to the general
// kernel (every source with a run in such a bucket: eight times the buckets' share, 2 - 3.5 ms per million) cost more than the pile path saves
// (~0.2 ms per million sources; measured: 363 M nodes of a 1 Gb genome took 252 ms this way against 160 ms through the pairwise kernels, 181 M
// nodes of 500 Mb -- 0.27 % irregular -- 57 against 69).  Every kernel concerned reads the same two counters: decided on the device.
// (round 5: between one irregular bucket in 250 and one in PILE_DECLINE_ONE_IN = 20 the pile path keeps the build in its MIXED form -- the sources
// handed on go through k_probe_stream in list mode, ~0.3 ms per million, before the general kernel: prefsuf_cluster_device.h)
__device__ __forceinline__ bool pile_declines(const unsigned long long *pile_cnt) { return pile_cnt_declines(pile_cnt); }

// the row of node `id`, up to nine words (the pile path takes rows of that size only), straight from the node array
__device__ __forceinline__ void load_row9(const NodesDev &nd, uint32_t id, uint32_t (&row)[9]) {
    const uint32_t *rp = nd.words + (size_t) id * nd.stride;
    if ((nd.stride & 3) == 0 && nd.stride >= 12 && ((uintptr_t) nd.words & 15u) == 0) {
        const uint4 v0 = reinterpret_cast<const uint4 *>(rp)[0], v1 = reinterpret_cast<const uint4 *>(rp)[1];
        row[0] = v0.x; row[1] = v0.y; row[2] = v0.z; row[3] = v0.w; row[4] = v1.x; row[5] = v1.y; row[6] = v1.z; row[7] = v1.w;
        row[8] = rp[8];
    } else {
#pragma unroll
        for (int k = 0; k < 9; k++) row[k] = k < nd.stride ? rp[k] : 0u;
    }
}

template <bool SAMPLE>
__global__ void __launch_bounds__(PB_THREADS, 6) k_pile_build(NodesDev nd, const uint32_t *__restrict__ skeys, const uint32_t *__restrict__ sids, uint64_t n_entries, const uint4 *__restrict__ dir, ClusterCfg cc, int U,
                                                           uint4 *__restrict__ rec, uint4 *__restrict__ tab, uint32_t epoch,
                                                           uint4 *__restrict__ side, unsigned long long *__restrict__ pile_cnt, uint32_t *__restrict__ own_mask) {
    if (!SAMPLE && pile_declines(pile_cnt)) return;
    const int idx_shift = cc.idx_shift, kk = cc.kk;
    __shared__ uint32_t sRow[PB_THREADS][PILE_SW];         // the entry's row on the pile's axis, masked to its extent (odd stride: conflict-free)
    __shared__ unsigned long long sKm[PB_THREADS];         // minimizer k-mer of the entry
    __shared__ unsigned long long sLead[PB_THREADS];       // per bucket (at its first entry): bit i = entry i leads a group
    __shared__ unsigned long long sRm[PB_THREADS];         // per leader: mirrored m_C set
    __shared__ uint32_t sMin[PB_THREADS], sMax[PB_THREADS];    // per leader: m_C << 16 | thread of the member with the smallest / largest m_C
    __shared__ uint32_t sBad[PB_THREADS];                  // per bucket
    __shared__ uint8_t sTag[PB_THREADS];                   // per leader
    __shared__ uint32_t sCount[3];                         // buckets, irregular buckets, entries of buckets that begin in the tile (sample only)
    const int t = (int) threadIdx.x;
    const uint64_t base = (uint64_t) blockIdx.x * PB_TILE;
    const uint64_t j = base + (uint64_t) t;
    const bool have = j < n_entries;
    const uint64_t jc = have ? j : (n_entries ? n_entries - 1 : 0);
    // entry j of the key order = node sids[j] with sort key skeys[j]; its row comes straight from the node array (the entry array of the
    // pairwise kernels -- the same rows copied into key order -- is not built for a build the pile path keeps: k_tgt_gather)
    const uint32_t key = skeys[jc], node_id = min(sids[jc], (uint32_t) nd.n - 1u);
    const uint32_t meta = key >> (cc.idx_shift - CL_MBITS);       // low bits: m_C (a target's sort key holds it under the bucket)
    const bool tgt = have && key != 0xFFFFFFFFu;
    uint4 drec = make_uint4(0u, 0u, 0u, 0u);
    if (tgt) drec = dir[key >> idx_shift];
    const uint64_t e0 = drec.x;
    const uint32_t cnt = drec.y;
    const bool owned = tgt && e0 >= base && e0 < base + PB_TILE && e0 <= j && j - e0 < (uint64_t) cnt;
    const int s = owned ? (int) (e0 - base) : 0;           // thread of the bucket's first entry
    const int i = owned ? (int) (j - e0) : 0;              // index of this entry in its bucket
    const bool part = owned && cnt <= 64u;
    // (only the thread that WORKS on an entry reads its row: the entries of a bucket that began in the tile before are that tile's halo, and a
    // halo thread whose bucket begins behind the tile has nothing to do -- one random row read per entry, none for the overlap of the tiles)
    uint32_t row[9] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    if (part) load_row9(nd, node_id, row);
    sLead[t] = 0ull; sRm[t] = 0ull; sMin[t] = 0xFFFFFFFFu; sMax[t] = 0u; sBad[t] = 0u;
    if (t < 3) sCount[t] = 0u;
    __syncthreads();
    // the row on the pile's axis: consensus index x (coordinate x - 64) = nucleotide x - 64 + m of the row = bit 2 x + 2 m of the row padded
    // with four zero words in front.  The word offset (2 m) >> 5 is one of 0 .. 3: two selects per word, static register indices.
    const int m = (int) (meta & 63u);
    uint32_t A[PILE_SW];
    {
        const int w0 = (2 * m) >> 5, sh = (2 * m) & 31;
        const int lo = 2 * (64 - m), hi = lo + 2 * U;      // bits of the axis this entry covers
        uint32_t Rp[18];
#pragma unroll
        for (int k = 0; k < 18; k++) Rp[k] = 0u;
#pragma unroll
        for (int k = 0; k < 9; k++) Rp[4 + k] = row[k];
        // (bit masks the compiler cannot see through: written as selects, hipcc turns the two stages into ONE dynamically indexed array in scratch)
        uint32_t m1 = 0u - (uint32_t) (w0 & 1), m2 = 0u - (uint32_t) ((w0 >> 1) & 1);
        asm volatile("" : "+v"(m1), "+v"(m2));
        uint32_t y1[PILE_SW + 3], y[PILE_SW + 1];
#pragma unroll
        for (int k = 0; k < PILE_SW + 3; k++) y1[k] = (Rp[k + 1] & m1) | (Rp[k] & ~m1);
#pragma unroll
        for (int k = 0; k <= PILE_SW; k++) y[k] = (y1[k + 2] & m2) | (y1[k] & ~m2);
#pragma unroll
        for (int k = 0; k < PILE_SW; k++) A[k] = funnel(y[k], y[k + 1], sh) & low_bits32(hi - 32 * k) & ~low_bits32(lo - 32 * k);
    }
    const unsigned long long kmask = kk >= 32 ? ~0ull : ((1ull << (2 * kk)) - 1ull);
    const unsigned long long kmer = (((unsigned long long) A[5] << 32) | A[4]) & kmask;       // coordinates 0 .. k - 1
    sKm[t] = kmer;
#pragma unroll
    for (int k = 0; k < PILE_SW; k++) sRow[t][k] = A[k];
    __syncthreads();
    int L = t;                                             // leader: the first entry of the bucket with this k-mer
    if (part) {
        for (int u = 0; u < i; u++) if (sKm[s + u] == kmer) { L = s + u; break; }
        if (L == t) {
            atomicOr(&sLead[s], 1ull << i);
            sTag[t] = (uint8_t) (cluster_key(kmer_hash((uint32_t) kmer & cc.lo_mask, (uint32_t) (kmer >> 32) & cc.hi_mask), idx_shift - CL_MBITS) & 31u);
        }
        atomicMin(&sMin[L], ((uint32_t) m << 16) | (uint32_t) t);
        atomicMax(&sMax[L], ((uint32_t) m << 16) | (uint32_t) t);
        const unsigned long long bit = 1ull << (63 - m);
        if (atomicOr(&sRm[L], bit) & bit) atomicOr(&sBad[s], 1u);          // two members start at the same coordinate
    }
    __syncthreads();
    // The members of a group in the order of their m_C, densely inside the bucket's own stretch of a table (the k-mers have done their job:
    // their words hold it): slot = bucket start + members of the groups before mine + members of my group with a smaller m_C.  The member
    // that starts next to a member's right is then the one slot below it -- no search.
    uint32_t *sSlot = reinterpret_cast<uint32_t *>(sKm);
    int my_slot = -1, my_rank = 0;
    if (part) {
        const unsigned long long lm = sLead[s];
        const int kgrp = __popcll(lm & ((1ull << (L - s)) - 1ull));
        if (kgrp < PILE_MAXSUB) {
            int before = 0;
            unsigned long long rest = lm;
#pragma unroll
            for (int g = 0; g < PILE_MAXSUB - 1; g++) {
                if (g < kgrp) { before += __popcll(sRm[s + __builtin_ctzll(rest)]); rest &= rest - 1ull; }
            }
            my_rank = m == 0 ? 0 : __popcll(sRm[L] >> (64 - m));                     // members of my group that start right of me
            my_slot = s + before + my_rank;
            if (my_slot < s + (int) cnt) sSlot[my_slot] = node_id; else my_slot = -1;   // (beyond the stretch: two members at one m_C -- an irregular bucket)
        }
    }
    uint32_t S[PILE_SW];
#pragma unroll
    for (int k = 0; k < PILE_SW; k++) S[k] = 0u;
    if (part) {
        const int t0 = (int) (sMax[L] & 0xFFFFu), t1 = (int) (sMin[L] & 0xFFFFu);      // leftmost start; rightmost end (one length)
        const int lo = 2 * (64 - m), hi = lo + 2 * U;
        uint32_t diff = 0u;
#pragma unroll
        for (int k = 0; k < PILE_SW; k++) {
            S[k] = sRow[t0][k] | sRow[t1][k];
            diff |= (A[k] ^ S[k]) & low_bits32(hi - 32 * k) & ~low_bits32(lo - 32 * k);
        }
        if (diff != 0u) atomicOr(&sBad[s], 1u);
    }
    __syncthreads();
    if (owned) {
        uint4 *line = tab + (size_t) (key >> idx_shift) * 8;                        // the bucket's 128-byte record
        if (!part) {
            if (i == 0) {                                  // more than 64 entries: not for this path
                if (!SAMPLE) { line[3] = make_uint4(0u, 0u, 0u, (1u << 22) | (epoch << 23)); line[4] = make_uint4(127u, drec.x, drec.z, drec.w); }
                atomicAdd(&sCount[0], 1u); atomicAdd(&sCount[1], 1u);
            }
        } else {
            const unsigned long long lm = sLead[s];
            const int nsub = __popcll(lm);
            const int iL = L - s;                          // index of my group's first member in the bucket
            const int k = __popcll(lm & ((1ull << iL) - 1ull));                        // my group's number
            if (!SAMPLE) {
                // the member of my group that starts next to my right (the largest m_C below mine): what a source of this pile keeps when that
                // member lies within its home run's windows -- k_pile_probe then needs no look-up at all
                const unsigned long long right = m == 0 ? 0ull : (sRm[L] >> (64 - m));    // bit b: a member with m_C == m - 1 - b
                uint32_t succ_id = 0xFFFFFFFFu, delta = 0u;
                if (right != 0ull && my_slot > s && my_rank >= 1) { succ_id = sSlot[my_slot - 1]; delta = 1u + (uint32_t) __builtin_ctzll(right); }
                // ... and what k_pile_probe reads of the entry AS A SOURCE: a member of a regular bucket's first group equals that group's consensus on
                // its whole extent (verified above), so the probe takes its row from the bucket's record, which it reads anyway, not from the entry
                // (round 5, own_mask != null: the members of EVERY group of a regular bucket -- a further group's consensus sits in `rec` at slot first + k,
                //  its run list in `rec2` at the same slot (k_pile_runs_consensus); round 4's form knows the first group only)
                const bool first_group = (k == 0 || own_mask != nullptr) && k < PILE_MAXSUB && sBad[s] == 0u && nsub <= PILE_MAXSUB;
                const bool leftmost = first_group && t == (int) (sMax[L] & 0xFFFFu);          // T0 of its group: the lane the group's run list is made from
                // the last word: the bucket (a first group's record is the bucket's line of `tab`), or the slot of a further group's records
                side[j] = make_uint4(node_id, succ_id, delta | ((uint32_t) m << 8) | ((uint32_t) k << 16) | (leftmost ? 0x40000000u : 0u) | (first_group ? 0x80000000u : 0u),
                                     k == 0 ? key >> idx_shift : (uint32_t) (e0 + (uint64_t) k));
                // an entry of no pile reads its OWN run list as a source (k_pile_probe): bit j of the mask the list-driven key pass works from
                if (!first_group && own_mask) atomicOr(&own_mask[j >> 5], 1u << (j & 31u));
            }
            if (!SAMPLE && L == t && k >= 1 && k < PILE_MAXSUB) {
                const unsigned long long rm = sRm[t];
                const uint64_t slot = e0 + (uint64_t) k;
                rec[slot * 4 + 0] = make_uint4(S[0], S[1], S[2], S[3]);
                rec[slot * 4 + 1] = make_uint4(S[4], S[5], S[6], S[7]);
                rec[slot * 4 + 2] = make_uint4(S[8], S[9], S[10], S[11]);
                rec[slot * 4 + 3] = make_uint4(S[12], 0u, (uint32_t) rm, (uint32_t) (rm >> 32));
            }
            if (i == 0) {
                // the bucket's first entry leads group 0: its consensus, its set, and which group a run wants (tags of the groups' leaders)
                uint32_t tg[PILE_MAXSUB] = {0u, 0u, 0u, 0u};
                unsigned long long rest = lm;
#pragma unroll
                for (int g = 0; g < PILE_MAXSUB; g++) {
                    if (rest) { tg[g] = sTag[s + __builtin_ctzll(rest)]; rest &= rest - 1ull; }
                }
                const bool irregular = sBad[s] != 0u || nsub > PILE_MAXSUB;
                const uint32_t ns1 = (uint32_t) (nsub < PILE_MAXSUB ? nsub : PILE_MAXSUB) - 1u;
                if (!SAMPLE) {
                    const unsigned long long rm = sRm[t];
                    line[0] = make_uint4(S[0], S[1], S[2], S[3]);
                    line[1] = make_uint4(S[4], S[5], S[6], S[7]);
                    line[2] = make_uint4(S[8], S[9], S[10], S[11]);
                    line[3] = make_uint4(S[12], (uint32_t) rm, (uint32_t) (rm >> 32), tg[0] | (tg[1] << 5) | (tg[2] << 10) | (tg[3] << 15) | (ns1 << 20) | (irregular ? 1u << 22 : 0u) | (epoch << 23));
                }
                atomicAdd(&sCount[0], 1u);
                if (irregular) atomicAdd(&sCount[1], 1u);
            }
            if (!SAMPLE && L == s && t == (int) (sMax[s] & 0xFFFFu)) {
                // T0 of the first group (the member that starts leftmost): the second half of the bucket's record -- no run list yet (k_pile_runs), and
                // who T1 is (the member that starts rightmost: the lowest slot of the group's dense stretch): its id and its m_C
                line[4] = make_uint4(cnt, drec.x, drec.z, drec.w);
                line[5] = make_uint4(sSlot[s], sMin[s] >> 16, 0u, 0u);
            }
        }
    }
    // entries of no regular bucket (non-targets, buckets of more than 64 entries): written by the thread of the tile proper
    if (!SAMPLE && have && t < PB_TILE && !(tgt && cnt <= 64u)) {
        side[j] = make_uint4(node_id, 0xFFFFFFFFu, 0u, 0u);
        if (own_mask) atomicOr(&own_mask[j >> 5], 1u << (j & 31u));
    }
    if (SAMPLE && owned) atomicAdd(&sCount[2], 1u);
    __syncthreads();
    if (SAMPLE && t < 3 && sCount[t]) atomicAdd(&pile_cnt[t], (unsigned long long) sCount[t]);      // (the sample is a few thousand workgroups)
}

// ------------------------------------------------------------------------------------------
// The RUN LIST of a pile (the first group of a regular bucket).  The windows of the pile's members overlap, and a window's minimizer is a matter
// of its content: the member that starts leftmost (T0) and the one that starts rightmost (T1) between them hold every window of every member,
// so their two run lists, joined where T0's windows end, are the run list of the whole pile on the pile's axis -- and a member's own run list is
// that list clipped to its windows.  k_pile_probe reads it from the bucket's record (one read shared by the pile's members, who sit side by side
// in the entry order) instead of every source's own list by id: a random read per source, 3 of that kernel's 10 ms; here it is two random reads
// per PILE.  One lane per entry; the lanes of the T0s work (k_pile_build flagged them and left T1's id in the record).
constexpr int PILE_RUNS = 8;                               // runs a pile's list holds
// the list into the second half of the bucket's record (n in 1 .. PILE_RUNS; kc[a] = kpos | c0 << 8), or "none"
__device__ __forceinline__ void pile_list_store(uint4 *line, uint32_t entries, bool ok, int n, const uint32_t (&key)[PILE_RUNS], const uint32_t (&kc)[PILE_RUNS], uint32_t c1_last) {
    uint32_t *lw = reinterpret_cast<uint32_t *>(line);
    if (ok) {
        line[5] = make_uint4(key[0], key[1], key[2], key[3]);
        line[6] = make_uint4(key[4], key[5], key[6], key[7]);
        line[7] = make_uint4(kc[0] | (kc[1] << 16), kc[2] | (kc[3] << 16), kc[4] | (kc[5] << 16), kc[6] | (kc[7] << 16));
        // the same cluster key twice in the list (one k-mer at two places of the extent: a tandem repeat -- a target could be an item at two offsets;
        // or two k-mers that share all 32 bits): noted once per PILE, bit 31, instead of compared pair by pair by every member
        bool dup = false;
#pragma unroll
        for (int a = 0; a < PILE_RUNS; a++)
#pragma unroll
            for (int b = a + 1; b < PILE_RUNS; b++) dup = dup || (b < n && key[a] == key[b]);
        lw[16] = (entries & 255u) | ((uint32_t) n << 8) | ((c1_last & 255u) << 16) | (dup ? 0x80000000u : 0u);
    } else lw[16] = entries & 255u;
}

__global__ void __launch_bounds__(256) k_pile_runs(const uint4 *__restrict__ side, uint64_t n_entries, uint32_t n_buckets, uint4 *__restrict__ tab, const uint2 *__restrict__ runs,
                                                   int n_nodes, int nwin, const unsigned long long *__restrict__ pile_cnt) {
    if (pile_declines(pile_cnt)) return;
    const uint64_t j = (uint64_t) blockIdx.x * 256 + threadIdx.x;
    if (j >= n_entries) return;
    const uint4 sd = side[j];
    if (((sd.z >> 30) & 1u) == 0u) return;                 // not the leftmost member of a pile
    uint4 *line = tab + (size_t) min(sd.w, n_buckets) * 8;
    const uint4 l4 = line[4], l5 = line[5];
    const uint4 *r0 = reinterpret_cast<const uint4 *>(runs + (size_t) min(sd.x, (uint32_t) n_nodes - 1u) * CL_RMAX);
    const uint4 *r1 = reinterpret_cast<const uint4 *>(runs + (size_t) min(l5.x, (uint32_t) n_nodes - 1u) * CL_RMAX);
    const uint4 a0 = r0[0], a1 = r0[1], a2 = r0[2], a3 = r0[3], b0 = r1[0], b1 = r1[1], b2 = r1[2];
    // windows ascending, on the pile's axis (coordinate + 64 in a byte): T0's runs from its first window to its last (its run 0 holds the last ones),
    // then what T1's first six runs hold beyond T0's last window; at most six runs, else the members read their own lists
    const int s0 = -(int) ((sd.z >> 8) & 63u), s1 = -(int) (l5.y & 63u), seam = s0 + nwin;
    const uint32_t k0[CL_RMAX] = {a0.x, a0.z, a1.x, a1.z, a2.x, a2.z, a3.x, a3.z}, y0[CL_RMAX] = {a0.y, a0.w, a1.y, a1.w, a2.y, a2.w, a3.y, a3.w};
    const uint32_t k1[6] = {b0.x, b0.z, b1.x, b1.z, b2.x, b2.z}, y1[6] = {b0.y, b0.w, b1.y, b1.w, b2.y, b2.w};
    const int nr0 = (int) (y0[0] >> 24), nr1 = (int) (y1[0] >> 24);
    bool ok = nr0 >= 1 && nr0 <= CL_RMAX && nr1 >= 1 && nr1 <= 6 && s1 <= seam;
    uint32_t ok_[PILE_RUNS], kc_[PILE_RUNS];
#pragma unroll
    for (int k = 0; k < PILE_RUNS; k++) { ok_[k] = 0u; kc_[k] = 0u; }
    int n = 0;
    uint32_t last_key = 0u, last_pos = 0xFFFFFFFFu, c1_last = 0u;
#pragma unroll
    for (int r = CL_RMAX - 1; r >= 0; r--) {
        if (ok && r < nr0) {
            const uint32_t kpos = (uint32_t) (s0 + (int) (y0[r] & 255u) + 64), c0 = (uint32_t) (s0 + (int) ((y0[r] >> 8) & 255u) + 64), c1 = (uint32_t) (s0 + (int) ((y0[r] >> 16) & 255u) + 64);
#pragma unroll
            for (int k = 0; k < PILE_RUNS; k++) if (k == n) { ok_[k] = k0[r]; kc_[k] = kpos | (c0 << 8); }
            last_key = k0[r]; last_pos = kpos; c1_last = c1;
            n++;
        }
    }
#pragma unroll
    for (int r = 5; r >= 0; r--) {
        if (ok && r < nr1) {
            const int c1i = s1 + (int) ((y1[r] >> 16) & 255u);
            if (c1i > seam) {
                const int c0i = max(s1 + (int) ((y1[r] >> 8) & 255u), seam);
                const uint32_t kpos = (uint32_t) (s1 + (int) (y1[r] & 255u) + 64), c1 = (uint32_t) (c1i + 64);
                if (n >= 1 && k1[r] == last_key && kpos == last_pos) c1_last = c1;             // the same minimizer on both sides of the seam: one run
                else {
#pragma unroll
                    for (int k = 0; k < PILE_RUNS; k++) if (k == n) { ok_[k] = k1[r]; kc_[k] = kpos | ((uint32_t) (c0i + 64) << 8); }
                    last_key = k1[r]; last_pos = kpos; c1_last = c1;
                    n++;
                }
            }
        }
    }
    ok = ok && n >= 1 && n <= PILE_RUNS;
    pile_list_store(line, l4.x, ok, n, ok_, kc_, c1_last);
}

// ------------------------------------------------------------------------------------------
// The run list of a pile FROM ITS CONSENSUS (round 5).  A window's minimizer is a function of the window's content, and the windows of all
// members of a pile are windows of its consensus: the pile's extent [-m_max, -m_min + U) is treated as ONE long row and goes through the very
// code that lists a node's runs (node_runs_core, two halves of up to 64 windows: a pile spans up to 126) -- once per PILE instead of once per
// member (six at 30x), and no member of a first group needs a run list of its own any more: the key pass of a build the pile path keeps
// computes the TARGET keys alone (k_node_runs<., ., false>), and own lists only for the entries outside a first group (own_mask) and for
// the sources k_pile_probe hands to the general kernel.  A member's own list is the pile's list clipped to its windows, bit for bit
// (tests/test_gpu_pile.py: test_consensus_run_lists_equal_the_members_own; engine option "pile_check").
// A block takes a tile of entries, lists the leftmost members (T0, flagged by k_pile_build) in LDS and gives every thread one pile.
// A pile whose list cannot be made (a window without a class-0 k-mer, more than eight runs, more records than a stack holds) stays without one:
// its members probe with their own lists (own_mask), and a member whose own list is flagged as well goes to the general kernel.
constexpr int PR_TILE = 3072;                              // entries per block of k_pile_runs_consensus (~500 piles at 30x: four rounds of 128)
constexpr int PR_TKW = 21;
constexpr int PR_RCAP = 14;                                // raw runs of a pile's extent (the eight a list holds after the seams are merged + a run per seam)
__global__ void __launch_bounds__(TK_ROWS) k_pile_runs_consensus(const uint4 *__restrict__ side, uint64_t n_entries, uint32_t n_buckets, uint4 *__restrict__ tab, const uint4 *__restrict__ rec,
                                                                 uint4 *__restrict__ rec2, ClusterCfg cc, int U, int Lmin,
                                                                 const unsigned long long *__restrict__ pile_cnt, uint32_t *__restrict__ own_mask) {
    if (pile_declines(pile_cnt)) return;
    __shared__ uint32_t s[TK_ROWS][PR_TKW];
    __shared__ uint32_t stk[2 * (NR_STACK + 1)][TK_ROWS];
    __shared__ uint16_t rbuf[PR_RCAP + 1][TK_ROWS];
    __shared__ uint32_t sList[PR_TILE];                    // buckets of the tile's piles
    __shared__ uint32_t sN;
    const int t = (int) threadIdx.x;
    if (t == 0) sN = 0u;
    __syncthreads();
    const uint64_t base = (uint64_t) blockIdx.x * PR_TILE;
#pragma unroll
    for (int q = 0; q < PR_TILE / TK_ROWS; q++) {
        const uint64_t j = base + (uint64_t) q * TK_ROWS + (uint64_t) t;
        if (j < n_entries) {
            const uint4 sd = side[j];
            // (the order of the piles in the list is free: each writes its own record).  A first group: its bucket; a further group: its slot | bit 31
            if ((sd.z >> 30) & 1u) sList[atomicAdd(&sN, 1u)] = ((sd.z >> 16) & 3u) == 0u ? min(sd.w, n_buckets) : (uint32_t) min((uint64_t) sd.w, n_entries) | 0x80000000u;
        }
    }
    __syncthreads();
    const int np = (int) sN;
    const int fs = cc.idx_shift - CL_MBITS;
    const int step = max(16, min(64, cc.w) & ~15);         // windows per piece of the sweep: no more than w, whole row words (w >= 16 for every shape pile_plan takes)
    for (int c0 = 0; c0 < np; c0 += TK_ROWS) {             // uniform
        const bool in = c0 + t < np;
        const uint32_t who = sList[in ? c0 + t : 0];
        const bool further = (who >> 31) != 0u;            // a further group of its bucket: consensus in rec, list into rec2
        const uint32_t slot = who & 0x7FFFFFFFu;
        uint4 *line = tab + (size_t) (further ? 0u : who) * 8;
        const uint4 *src = further ? rec + (size_t) slot * 4 : (const uint4 *) line;
        uint32_t S[PILE_SW + 1];
        unsigned long long rm = 0ull;
        {
            const uint4 l0 = src[0], l1 = src[1], l2 = src[2], l3 = src[3];
            S[0] = l0.x; S[1] = l0.y; S[2] = l0.z; S[3] = l0.w; S[4] = l1.x; S[5] = l1.y; S[6] = l1.z; S[7] = l1.w;
            S[8] = l2.x; S[9] = l2.y; S[10] = l2.z; S[11] = l2.w; S[12] = l3.x; S[13] = 0u;
            rm = !in ? 0ull : (further ? (((unsigned long long) l3.w << 32) | l3.z) : (((unsigned long long) l3.z << 32) | l3.y));
        }
        const bool act = in && rm != 0ull;
        const int m_min = act ? __clzll((long long) rm) : 0, m_max = act ? 63 - __builtin_ctzll(rm) : 0;
        const int len_v = U + m_max - m_min;               // the pile's extent: [-m_max, -m_min + U)
        const int nwin = len_v - Lmin + 1;                 // <= 126
        // the extent as a row of its own: consensus index 64 - m_max onwards (the consensus is zero outside the extent, like the tail of a row)
        {
            const int ob = 2 * (64 - m_max), w0 = ob >> 5, sh = ob & 31;          // w0 in 0 .. 4 (4: every member starts at the k-mer)
            uint32_t m1 = 0u - (uint32_t) (w0 & 1), m2 = 0u - (uint32_t) ((w0 >> 1) & 1), m4 = 0u - (uint32_t) ((w0 >> 2) & 1);
            asm volatile("" : "+v"(m1), "+v"(m2), "+v"(m4));          // (opaque masks: written as selects the stages become a dynamically indexed array in scratch)
            uint32_t y1[PILE_SW + 1], y2[PILE_SW + 1], y[PILE_SW + 1];
#pragma unroll
            for (int k = 0; k <= PILE_SW; k++) y1[k] = ((k + 1 <= PILE_SW ? S[k + 1] : 0u) & m1) | (S[k] & ~m1);
#pragma unroll
            for (int k = 0; k <= PILE_SW; k++) y2[k] = ((k + 2 <= PILE_SW ? y1[k + 2] : 0u) & m2) | (y1[k] & ~m2);
#pragma unroll
            for (int k = 0; k <= PILE_SW; k++) y[k] = ((k + 4 <= PILE_SW ? y2[k + 4] : 0u) & m4) | (y2[k] & ~m4);
#pragma unroll
            for (int k = 0; k < PR_TKW; k++) s[t][k] = k < PILE_SW ? funnel(y[k], y[k + 1], sh) : 0u;
        }
        // (each thread reads the row it wrote: no barrier)
        int nr = 0;
        bool uncovered = false, stack_ovf = false;
        uint32_t cur0 = 0u;
        node_runs_core<true, PR_RCAP>(s[t], nwin, act, cc, stk, rbuf, t, nr, uncovered, stack_ovf, cur0, step);
        bool ok = act && nr >= 1 && nr <= PR_RCAP && !uncovered && !stack_ovf;
        // runs as they were found: the last windows first, q | p0 << 8 in the extent's own coordinates, p1 = p0 of the run before.  The record
        // wants them ascending on the pile's axis (coordinate + 64 in a byte); the two halves of the window range meet at window 64: a minimizer
        // on both sides of that seam is ONE run (k_pile_probe takes "the same minimizer twice" for a tandem repeat)
        uint32_t key_[PILE_RUNS], kc_[PILE_RUNS];
#pragma unroll
        for (int k = 0; k < PILE_RUNS; k++) { key_[k] = 0u; kc_[k] = 0u; }
        int n = 0;
        uint32_t last_q = 0xFFFFFFFFu;
        const int nrs = nr < PR_RCAP ? nr : PR_RCAP;
        for (int r = PR_RCAP - 1; r >= 0; r--) {
            if (ok && r < nrs) {
                const uint32_t d = rbuf[r][t];
                const int q = (int) (d & 255u), p0 = (int) (d >> 8);
                if ((uint32_t) q != last_q) {              // (the same k-mer as the run before: the seam of two pieces -- one run)
                    uint32_t h, pk;
                    kmer_key(s[t], q < 192 ? q : 0, true, cc, h, pk);
                    const uint32_t kv = cluster_key(h, fs), cv = (uint32_t) (q - m_max + 64) | ((uint32_t) (p0 - m_max + 64) << 8);
#pragma unroll
                    for (int k = 0; k < PILE_RUNS; k++) if (k == n) { key_[k] = kv; kc_[k] = cv; }
                    n++;
                    last_q = (uint32_t) q;
                }
            }
        }
        ok = ok && n >= 1 && n <= PILE_RUNS;
        if (in && further) {
            // a further group: its list in rec2 at the group's slot, laid out like the second half of a bucket's line (word 0 = runs << 8 | end << 16)
            uint4 *l2 = rec2 + (size_t) slot * 4 - 4;      // (pile_list_store writes line[4 .. 7])
            pile_list_store(l2, 0u, ok, n, key_, kc_, (uint32_t) (nwin - m_max + 64));
            if (!ok) {
                // its members probe with their own lists: the entries of the bucket (it starts k slots back) that carry this group's slot
                // (the bucket begins k <= 3 slots before the group's slot)
                const uint64_t jb = slot >= 3u ? (uint64_t) slot - 3u : 0ull;
                for (uint64_t j2 = jb; j2 < jb + 64u + 3u && j2 < n_entries; j2++) {
                    const uint4 s2 = side[j2];
                    if ((s2.z >> 31) && ((s2.z >> 16) & 3u) != 0u && s2.w == slot) atomicOr(&own_mask[j2 >> 5], 1u << (j2 & 31u));
                }
            }
        } else if (in) {
            uint32_t *lw = reinterpret_cast<uint32_t *>(line);
            pile_list_store(line, lw[16], ok, n, key_, kc_, (uint32_t) (nwin - m_max + 64));
            if (!ok) {
                // no list (3 piles in 1000: a window without a class-0 k-mer somewhere on the extent, mostly): the members probe with their OWN
                // lists, as the entries outside a first group do -- noted in own_mask for the list-driven key pass that follows.  (Marking the
                // bucket irregular instead sent every source with a run in it to the general kernel: 1.7 M more sources at the north-star size.)
                lw[16] = lw[16] & 255u;
                const uint32_t cntb = min(lw[16] & 255u, 64u), e0b = lw[17];
                for (uint32_t i2 = 0; i2 < cntb; i2++) {
                    const uint64_t j2 = (uint64_t) e0b + i2;
                    if (j2 < n_entries) { const uint4 s2 = side[j2]; if ((s2.z >> 31) && ((s2.z >> 16) & 3u) == 0u) atomicOr(&own_mask[j2 >> 5], 1u << (j2 & 31u)); }
                }
            }
        }
    }
}

// own_mask (bit j: entry j of the key order reads its own run list) -> the ids of those entries, densely (the order is free).  A block takes
// 32 768 entries: one global atomic per block.
__global__ void __launch_bounds__(256) k_pile_own_ids(const uint32_t *__restrict__ own_mask, const uint32_t *__restrict__ sids, uint64_t n_entries, int32_t *__restrict__ out, uint32_t cap,
                                                      unsigned long long *__restrict__ count, const unsigned long long *__restrict__ pile_cnt) {
    if (pile_declines(pile_cnt)) return;
    __shared__ uint32_t sTot, sBase;
    const uint64_t words = (n_entries + 31) / 32;
    const uint64_t w0 = (uint64_t) blockIdx.x * 1024 + threadIdx.x;
    uint32_t mw[4];
    uint32_t mine = 0u;
#pragma unroll
    for (int q = 0; q < 4; q++) { const uint64_t w = w0 + (uint64_t) q * 256; mw[q] = w < words ? own_mask[w] : 0u; mine += (uint32_t) __popc(mw[q]); }
    if (threadIdx.x == 0) sTot = 0u;
    __syncthreads();
    const uint32_t at = atomicAdd(&sTot, mine);
    __syncthreads();
    if (threadIdx.x == 0) sBase = (uint32_t) atomicAdd(count, (unsigned long long) sTot);
    __syncthreads();
    uint32_t o = sBase + at;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint64_t w = w0 + (uint64_t) q * 256;
        uint32_t m = mw[q];
        while (m) {
            const int b = __builtin_ctz(m);
            m &= m - 1u;
            const uint64_t j = w * 32 + (uint64_t) b;
            if (j < n_entries && o < cap) out[o] = (int32_t) sids[j];
            o++;
        }
    }
}

// test harness (engine option "pile_check"; every node has its own run list): a first-group member's own list against its pile's list clipped
// to its windows -- count of the members for which the two differ
__global__ void __launch_bounds__(256) k_pile_list_check(const uint4 *__restrict__ side, uint64_t n_entries, uint32_t n_buckets, const uint4 *__restrict__ tab, const uint4 *__restrict__ rec2,
                                                         uint32_t epoch, const uint2 *__restrict__ runs, int n_nodes, int nwin, unsigned long long *__restrict__ out /* [0] members checked, [1] mismatches */,
                                                         const unsigned long long *__restrict__ pile_cnt) {
    if (pile_declines(pile_cnt)) return;
    const uint64_t j = (uint64_t) blockIdx.x * 256 + threadIdx.x;
    if (j >= n_entries) return;
    const uint4 sd = side[j];
    if ((sd.z >> 31) == 0u) return;
    const bool further = ((sd.z >> 16) & 3u) != 0u;        // (its list: rec2 at the group's slot, laid out like words 16 .. 31 of a bucket's line)
    const uint4 *line = further ? rec2 + (size_t) min((uint64_t) sd.w, n_entries - 1) * 4 - 4 : tab + (size_t) min(sd.w, n_buckets) * 8;
    const uint4 l4 = line[4];
    if (!further) { const uint4 l3 = line[3]; if ((l3.w >> 23) != epoch || ((l3.w >> 22) & 1u)) return; }      // an irregular bucket: nothing to compare
    const int npr = (int) ((l4.x >> 8) & 15u);
    if (npr == 0) return;
    const uint32_t *lw = reinterpret_cast<const uint32_t *>(line);
    const uint2 *own = runs + (size_t) min(sd.x, (uint32_t) n_nodes - 1u) * CL_RMAX;
    const int code = (int) (own[0].y >> 24);
    if (code == 0 || code == CL_RUNS_FLAGGED) return;      // (a flagged member: its list says nothing)
    const int nown = code < CL_RMAX ? code : CL_RMAX;
    const int sB = 64 - (int) ((sd.z >> 8) & 63u);
    bool bad = false;
    int r = nown - 1;                                      // own runs: the last windows first -> walk them backwards
    for (int a = 0; a < npr && a < PILE_RUNS; a++) {
        const uint32_t vx = lw[20 + a], pq = (lw[28 + (a >> 1)] >> (16 * (a & 1))) & 0xFFFFu;
        const uint32_t pqn = a + 1 < npr ? (lw[28 + ((a + 1) >> 1)] >> (16 * ((a + 1) & 1))) & 0xFFFFu : 0u;
        const int kpos = (int) (pq & 255u), c0 = (int) (pq >> 8), c1 = a + 1 < npr ? (int) (pqn >> 8) : (int) ((l4.x >> 16) & 255u);
        const int p0 = max(c0, sB) - sB, p1 = min(c1, sB + nwin) - sB;
        if (p0 >= p1) continue;
        if (r < 0) { bad = true; break; }
        const uint2 o = own[r];
        const int oq = (int) (o.y & 255u), op0 = (int) ((o.y >> 8) & 255u), op1 = (int) ((o.y >> 16) & 255u);
        bad = bad || o.x != vx || oq != kpos - sB || op0 != p0 || op1 != p1;
        r--;
    }
    bad = bad || r >= 0;
    atomicAdd(&out[0], 1ull);
    if (bad) atomicAdd(&out[1], 1ull);
}

// ------------------------------------------------------------------------------------------
// OR of x << 1 .. x << g
__device__ __forceinline__ unsigned long long smear_up(unsigned long long x, int g /* uniform, 0 .. 63 */) {
    unsigned long long acc = 0ull, pw = x;                 // pw = OR of x << 0 .. x << (2^b - 1)
    int done = 0;
#pragma unroll
    for (int b = 0; b < 6; b++) {
        if ((g >> b) & 1) { acc |= pw << (done + 1); done += 1 << b; }       // uniform; done + 1 <= 63
        pw |= pw << (1 << b);
    }
    return acc;
}

#ifndef PP_WAVES_N
#define PP_WAVES_N 4
#endif
constexpr int PP_WAVES = PP_WAVES_N;
constexpr int PP_LSTRIDE = 20;
#ifndef PP_OCC
#define PP_OCC 4
#endif
// One lane per source.  What bounds this kernel is neither arithmetic (its vector ALUs are busy ~55 % of the time) nor bytes (2.7 TB/s) but
// the waves' waiting for memory -- the chain side record -> run list -> bucket records -> the targets that stand -- at four waves per SIMD
// (five spill: 9.9 against 7.9 ms): side record and run list are read a tile ahead, the records of the next run are on their way while
// the current one is compared (of the next two: no faster), the run loop is unrolled over the eight slots (static registers; a wave skips the slots none of its lanes uses).
__global__ void __launch_bounds__(PP_WAVES * 64, PP_OCC) k_pile_probe(PrefSufCfg cfg, ClusterCfg cc, int U, NodesDev nd, uint64_t n_entries, uint64_t n_mine, int n_nodes,
                                                                 const uint4 *__restrict__ tab, uint32_t epoch, const uint4 *__restrict__ rec, const uint4 *__restrict__ rec2,
                                                                 const uint4 *__restrict__ side, const uint4 *__restrict__ side_src, const uint2 *__restrict__ runs, ProbeOut o,
                                                                 int32_t *__restrict__ defer_list, uint32_t defer_cap, const unsigned long long *__restrict__ pile_cnt) {
    // the records of the slot at hand, ONE copy per distinct bucket of the wave (the members of a pile sit side by side and want the same
    // records): PP_LSTRIDE words per record (16 used; 20: the lanes of sixteen records read conflict-free), four words of slack in front and
    // twelve behind -- a compare reads up to four words before and six behind a record, and what it finds there is masked
    __shared__ __attribute__((aligned(16))) uint32_t sL[PP_WAVES][4 + 64 * PP_LSTRIDE + 12];
    __shared__ uint32_t sBk[PP_WAVES][64];                 // the distinct buckets of the slot whose records are on their way, in lane order
    __shared__ int32_t sDefer[PP_WAVES][128];              // sources of this wave that wait for the defer list
    if (blockIdx.x == 0 && threadIdx.x == 0) { o.counters[CNT_PILE_BUCKETS] = pile_cnt[0]; o.counters[CNT_PILE_IRREGULAR] = pile_cnt[1]; o.counters[CNT_PILE_OWN] = pile_cnt[3]; }
    // a build whose buckets are mostly irregular (reads with sequencing errors) is k_probe_stream's: this kernel leaves at once
    if (pile_declines(pile_cnt)) return;
    const int wave = (int) (threadIdx.x >> 6), lane = lane_id();
    uint32_t *sl = sL[wave] + 4;                           // word 0 of record 0
    for (int k = lane; k < 4 + 64 * PP_LSTRIDE + 12; k += 64) sL[wave][k] = 0u;
    wave_lds_fence();
    // A persistent grid: one global atomic per WAVE for the edge count and one per ~64 deferred sources -- as one workgroup per 256 sources
    // the kernel made 1.4 M same-address atomics per build, which serialise in the L2 at ~90 per microsecond (6 ms of its 19).
    uint64_t st_rec = 0;
    int n_defer = 0;                                       // uniform
    auto flush_defer = [&]() {                             // convergent
        if (n_defer == 0) return;
        unsigned long long basep = 0;
        if (lane == 0) basep = atomicAdd(&o.counters[CNT_DEFERRED], (unsigned long long) n_defer);
        basep = ((unsigned long long) (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) (basep >> 32)) << 32) | (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) basep);
        for (int k = lane; k < n_defer; k += 64)
            if (basep + (unsigned long long) k < (unsigned long long) defer_cap) defer_list[basep + (unsigned long long) k] = sDefer[wave][k];
        wave_lds_fence();
        n_defer = 0;
    };
    // side_src, n_mine: the side records this launch WALKS -- side itself, all n_entries of them, or (a rank's share: k_pile_side_range's output)
    // those of the sources of an id range; every other index (entries, group slots, the side records of TARGETS) is one of the whole entry array
    const uint64_t n_tiles = (n_mine + PP_WAVES * 64 - 1) / (PP_WAVES * 64);
    const uint64_t last = n_entries - 1, last_mine = n_mine - 1;
    // What a lane reads of its source is a 16-byte SIDE record (k_pile_build: id, the right neighbour in its pile, its place in the pile), a
    // tile ahead, and the run list by that id while this tile's targets are looked up.  The source's ROW is not read at all where the source is
    // a member of its home bucket's first group (94 % of them): it equals that group's consensus on its whole extent -- k_pile_build verified
    // it -- and the consensus arrives with the home run's record anyway.  (The entry's line visited twice, a tile apart, was 1.5 of the kernel's
    // ~4.6 L1 misses per source; the other sources read their row from the entry array.)
    auto side_of = [&](uint64_t t) -> uint4 {              // side record of this lane's source in tile t (clamped: the last entry)
        const uint64_t j = t * (PP_WAVES * 64) + threadIdx.x;
        uint4 v = side_src[j < n_mine ? j : last_mine];
        v.x = min(v.x, (uint32_t) n_nodes - 1u);
        return v;
    };
    // Where a lane's run list comes from: the second half of its home bucket's record (the PILE's run list, k_pile_build: one 64-byte read
    // shared by the pile's members, who sit side by side) for a member of the bucket's first group; its own list by id (a random read) otherwise.
    // (a member of a further group of its bucket: the group's list in rec2 at the group's slot, laid out like the second half of a bucket's line)
    auto run_list_of = [&](const uint4 &sd) -> const uint4 * {
        if ((sd.z >> 31) == 0u) return reinterpret_cast<const uint4 *>(runs + (size_t) sd.x * CL_RMAX);
        return ((sd.z >> 16) & 3u) == 0u ? tab + (size_t) min(sd.w, cc.n_buckets) * 8 + 4 : rec2 + (size_t) min((uint64_t) sd.w, last) * 4;
    };
    const int nwin = U - cfg.Lmin + 1;
    uint4 next_side = side_of(blockIdx.x < n_tiles ? blockIdx.x : 0);
    uint4 Qr[CL_RMAX / 2];
    {
        const uint4 *rp = run_list_of(next_side);
#pragma unroll
        for (int c = 0; c < CL_RMAX / 2; c++) Qr[c] = rp[c];
    }
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {      // uniform
    const uint64_t j = tile * (PP_WAVES * 64) + threadIdx.x;
    const bool have = j < n_mine;
    const uint4 my = next_side;                            // {id, right neighbour's id, its offset | m_C << 8 | group << 16 | first-group member << 31, -}
    const int Bs = (int) my.x;
    const bool first_group = (my.z >> 31) != 0u;
    const uint32_t npr = first_group ? (Qr[0].x >> 8) & 15u : 0u;         // runs of my pile's list (0: it has none -- more than six, or a member's own list is flagged)
    if (first_group && npr == 0u) {                        // (rare: a pile without a list -- its members probe with their own, k_pile_runs_consensus saw to them)
        const uint4 *rp = reinterpret_cast<const uint4 *>(runs + (size_t) Bs * CL_RMAX);
#pragma unroll
        for (int c = 0; c < CL_RMAX / 2; c++) Qr[c] = rp[c];
    }
    // my row = my home bucket's first consensus on [64 - m_C, 64 - m_C + len): with the pile's run list the home run is the first slot, so the
    // row is there before any other run is compared
    const bool row_from_pile = npr != 0u;
    uint32_t rk[CL_RMAX], ry[CL_RMAX];
    uint32_t vmask = 0u;                                   // bit a: slot a holds a run of this source
    {
        // (a) my own list: slot a = run a, the last windows first
        uint32_t ok[CL_RMAX], oy[CL_RMAX];
#pragma unroll
        for (int c = 0; c < CL_RMAX / 2; c++) { ok[2 * c] = Qr[c].x; oy[2 * c] = Qr[c].y; ok[2 * c + 1] = Qr[c].z; oy[2 * c + 1] = Qr[c].w; }
        // (b) the pile's list, windows ascending on the pile's axis (coordinate + 64): clipped to my windows [-m_C, -m_C + nwin)
        const uint32_t pk[PILE_RUNS] = {Qr[1].x, Qr[1].y, Qr[1].z, Qr[1].w, Qr[2].x, Qr[2].y, Qr[2].z, Qr[2].w};
        const uint32_t pq[PILE_RUNS] = {Qr[3].x & 0xFFFFu, Qr[3].x >> 16, Qr[3].y & 0xFFFFu, Qr[3].y >> 16, Qr[3].z & 0xFFFFu, Qr[3].z >> 16, Qr[3].w & 0xFFFFu, Qr[3].w >> 16};
        const int c1_last = (int) ((Qr[0].x >> 16) & 255u);
        const int sB = 64 - (int) ((my.z >> 8) & 63u);     // my first window, + 64
#pragma unroll
        for (int a = 0; a < CL_RMAX; a++) {
            uint32_t k = ok[a], y = oy[a] & 0x00FFFFFFu;
            bool v = false;
            {
                const int kpos = (int) (pq[a] & 255u), c0 = (int) (pq[a] >> 8), c1 = ((uint32_t) a + 1u < npr && a + 1 < PILE_RUNS) ? (int) (pq[a + 1 < PILE_RUNS ? a + 1 : a] >> 8) : c1_last;
                const int p0 = max(c0, sB) - sB, p1 = min(c1, sB + nwin) - sB;
                const bool pv = row_from_pile && (uint32_t) a < npr && p0 < p1 && kpos >= sB;
                k = row_from_pile ? pk[a] : k;
                y = row_from_pile ? (uint32_t) (kpos - sB) | ((uint32_t) max(p0, 0) << 8) | ((uint32_t) max(p1, 0) << 16) : y;
                v = pv;
            }
            rk[a] = k; ry[a] = y;
            vmask |= v ? 1u << a : 0u;
        }
        if (!row_from_pile) {
            const int code = have ? (int) (oy[0] >> 24) : 0;
            vmask = (code == 0 || code == CL_RUNS_FLAGGED) ? 0u : (code >= CL_RMAX ? 0xFFu : ((1u << code) - 1u));
            ry[0] |= (uint32_t) code << 24;
        }
    }
    uint32_t B[9] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    if (!row_from_pile) load_row9(nd, (uint32_t) Bs, B);      // (one source in sixteen: its row by id)
    next_side = side_of(tile + gridDim.x < n_tiles ? tile + gridDim.x : tile);
    const int nr_code = !have ? 0 : (row_from_pile ? 1 : (int) (ry[0] >> 24));       // (a pile's member is a source: one length, no masks)
    bool dfr = nr_code == CL_RUNS_FLAGGED;                 // runs k_node_runs could not list: the general kernel finds them by brute force
    const bool active = nr_code != 0 && !dfr;
    vmask = active ? vmask : 0u;
#ifndef PP_NO_COMPACT
    // A member's runs are a stretch of its pile's eight slots (the pile's list clipped to the member's windows); the slot loop below runs over the
    // UNION of the wave's slots -- ten piles, each with its own stretch: nearly all eight.  Shifted down to slot 0 the loop ends with the longest
    // list of the wave instead (three stages of selects: by four, two, one slot).
    {
        const int sh = vmask ? __builtin_ctz(vmask) : 0;
        const bool s4 = (sh & 4) != 0, s2 = (sh & 2) != 0, s1 = (sh & 1) != 0;
#pragma unroll
        for (int a = 0; a < CL_RMAX; a++) { rk[a] = s4 ? (a + 4 < CL_RMAX ? rk[a + 4] : 0u) : rk[a]; ry[a] = s4 ? (a + 4 < CL_RMAX ? ry[a + 4] : 0u) : ry[a]; }
#pragma unroll
        for (int a = 0; a < CL_RMAX; a++) { rk[a] = s2 ? (a + 2 < CL_RMAX ? rk[a + 2] : 0u) : rk[a]; ry[a] = s2 ? (a + 2 < CL_RMAX ? ry[a + 2] : 0u) : ry[a]; }
#pragma unroll
        for (int a = 0; a < CL_RMAX; a++) { rk[a] = s1 ? (a + 1 < CL_RMAX ? rk[a + 1] : 0u) : rk[a]; ry[a] = s1 ? (a + 1 < CL_RMAX ? ry[a + 1] : 0u) : ry[a]; }
        vmask >>= sh;
    }
#endif
    uint32_t um = 0u;                                      // uniform: slots in which any lane of the wave has a run
#pragma unroll
    for (int a = 0; a < CL_RMAX; a++) um |= __ballot(((vmask >> a) & 1u) != 0u) != 0ull ? 1u << a : 0u;
    // the same minimizer twice in one source (a tandem repeat): a target could be an item at two offsets.  A pile's list carries the answer
    // (pile_list_store, bit 31 of its first word); the lanes that probe with a list of their own (one entry in 400) compare pair by pair
    dfr = dfr || (row_from_pile && (Qr[0].x >> 31) != 0u);
    if (__ballot(active && !row_from_pile) != 0ull) {      // uniform
#pragma unroll
        for (int a = 0; a < CL_RMAX; a++)
#pragma unroll
            for (int b = a + 1; b < CL_RMAX; b++) dfr = dfr || (((vmask >> a) & (vmask >> b) & 1u) != 0u && rk[a] == rk[b]);
    }
    const int Lbig = cfg.rsoemo > cfg.Lmin ? cfg.rsoemo : cfg.Lmin;
    const int G = min(max(U - Lbig, 0), 63);               // an item is removed iff another sits 1 .. G offsets before it
    unsigned long long occ = 0ull;                         // offsets that hold an item
    uint32_t grp = 0u;                                     // 2 bits per slot: the group that gave the slot's items
    uint32_t Ea[4] = {0u, 0u, 0u, 0u};                     // consensus past the source's end: the longest one seen, valid for `ea_len` positions
    int ea_len = 0;
    // one record against the source: the items it gives.  The consensus strings of two runs must agree past the source's end wherever both
    // are defined -- that is what makes "the overhangs of two items agree" (the via compare of the reduction) hold across runs.
    // `lb`: where the record's word 0 is (in sl).  Consensus word k = coordinates -64 + 16 k ..; the words the compare touches outside 0 .. 12
    // (a source that reaches left of coordinate -64, or the words behind the consensus) belong to positions no item of the record can use:
    // a mismatch there lies left of every offset the record's set holds, and the consensus past the source's end is masked to its length.
    auto take_record = [&](int lb, unsigned long long rm, int slot_a, uint32_t group, uint32_t y, bool on, bool home) {
        const int q = (int) (y & 255u), p0 = (int) ((y >> 8) & 255u), p1 = (int) ((y >> 16) & 255u);
        const uint32_t *ss = sl + lb - 4;                  // (index 4 = word 0 of the record)
        // position t of the source = consensus index 64 - q + t = bit 2 (128 - q + t) of the consensus padded with four words in front
        const int ob = 2 * (128 - min(q, 127)), w0 = ob >> 5, sh = ob & 31;
        uint32_t x[10];
#pragma unroll
        for (int k = 0; k < 10; k++) x[k] = ss[w0 + k];
        uint32_t dv = 0u;                                  // last word in which the source differs from the consensus, and which
        int dk = 0;
#pragma unroll
        for (int k = 0; k < 9; k++) {
            const uint32_t sw = funnel(x[k], x[k + 1], sh) & low_bits32(2 * U - 32 * k);
            B[k] = home ? sw : B[k];                       // the home run of a first-group member: this IS the source's row (and the first record a lane sees)
            const uint32_t dd = (sw ^ B[k]) & low_bits32(2 * U - 32 * k);
            dk = dd ? k : dk;
            dv = dd ? dd : dv;
        }
        const int mism = dv ? 16 * dk + ((31 - __clz((int) dv)) >> 1) : -1;
        unsigned long long oc = q <= 63 ? (rm >> (63 - q)) : (q - 63 >= 64 ? 0ull : (rm << (q - 63)));            // offset d = q - m
        oc &= (p1 >= 64 ? ~0ull : ((1ull << p1) - 1ull)) & ~((1ull << (p0 & 63)) - 1ull) & ~1ull;                 // the run's windows; offset 0 is the source itself
        if (mism >= 0) oc &= mism >= 63 ? 0ull : ~((2ull << mism) - 1ull);
        oc = (on && !dfr) ? oc : 0ull;
        if (oc != 0ull) {
            uint32_t E[4];                                 // the consensus from the source's end on: 64 positions, defined up to the group's rightmost member
            {
                const int oe = ob + 2 * U, we = oe >> 5, she = oe & 31;
                uint32_t xe[5];
#pragma unroll
                for (int k = 0; k < 5; k++) xe[k] = ss[we + k];
#pragma unroll
                for (int k = 0; k < 4; k++) E[k] = funnel(xe[k], xe[k + 1], she);
            }
            const int len = min(64, max(0, q - __clzll((long long) rm)));       // (smallest m_C of the group = clz of its mirrored set)
            if (occ != 0ull) {
                const int nb = 2 * min(len, ea_len);
                uint32_t df = 0u;
#pragma unroll
                for (int k = 0; k < 4; k++) df |= (E[k] ^ Ea[k]) & low_bits32(nb - 32 * k);
                if (df != 0u) dfr = true;
            }
            if (occ == 0ull || len > ea_len) {
#pragma unroll
                for (int k = 0; k < 4; k++) Ea[k] = E[k];
                ea_len = len;
            }
            occ |= oc;
            grp = (grp & ~(3u << (2 * slot_a))) | (group << (2 * slot_a));
        }
    };
    // Runs from slot 0 up.  Which k-mer group of the bucket a run wants is
    // read off the record: the groups whose tag equals the low five bits of the run's cluster key -- one, except where two k-mers of a
    // bucket share all 32 bits of their hash (3 % of the buckets at the north-star size: 19-mers do not fit 32 bits).  Group 0 is in the
    // record itself; the others are taken in a short loop behind this one (the order of the records is free: what a source keeps is
    // decided from the complete offset set).
    auto bucket_of = [&](int a) -> uint32_t { return ((vmask >> a) & 1u) ? min(rk[a] >> cc.idx_shift, cc.n_buckets) : cc.n_buckets; };
    uint32_t more = 0u;                                    // 4 bits per slot: further groups to take
    bool got_row = !row_from_pile;                         // the source's row is in B
    // The records of a slot, loaded ONCE per distinct bucket of the wave: lanes in entry order -> equal buckets in adjacent lanes -> the first
    // lane of each stretch leads; the leaders' buckets go through sBk, then four lanes load the 64 bytes of one record (sixteen records per
    // load instruction) and park them in LDS, where every lane of the stretch reads them.  (As four 16-byte loads per LANE the same ~15
    // distinct lines per slot were looked up 4 x 64 times in the L1, the kernel's address path was busy half of the time, and each lane
    // staged its own copy through LDS: 7.9 -> 7.7 ms on one box.)  The loads of slot a + 1 are in flight while slot a is compared.
    uint4 V0 = make_uint4(0u, 0u, 0u, 0u), V1 = V0;        // the records on their way: sixteen per register
    int li_next = 0, nl_next = 0, li = 0;
    const int l0 = lane >> 2, c4 = lane & 3;
    uint32_t *bk = sBk[wave];
    auto issue = [&](int a) {                              // convergent
        const uint32_t b = bucket_of(a);
        const uint32_t bp = (uint32_t) __builtin_amdgcn_update_dpp((int) ~b, (int) b, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
        const bool lead = b != bp;                         // (lane 0 reads ~b)
        const uint64_t lm = __ballot(lead);
        li_next = (int) __builtin_amdgcn_mbcnt_hi((uint32_t) (lm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) lm, 0u)) + (lead ? 1 : 0) - 1;
        nl_next = __popcll(lm);                            // uniform, >= 1
        if (lead) bk[li_next] = b;
        wave_lds_fence();
        V0 = tab[(size_t) bk[min(l0, nl_next - 1)] * 8 + c4];
        if (nl_next > 16) V1 = tab[(size_t) bk[min(16 + l0, nl_next - 1)] * 8 + c4];
    };
    auto commit = [&]() {                                  // convergent: the records `issue` asked for, into LDS
        li = li_next;
        *reinterpret_cast<uint4 *>(sl + l0 * PP_LSTRIDE + 4 * c4) = V0;
        if (nl_next > 16) *reinterpret_cast<uint4 *>(sl + (16 + l0) * PP_LSTRIDE + 4 * c4) = V1;
        for (int r = 32; r < nl_next; r += 16) {           // (more than 32 distinct buckets in a wave: rare, not overlapped)
            const uint4 v = tab[(size_t) bk[min(r + l0, nl_next - 1)] * 8 + c4];
            *reinterpret_cast<uint4 *>(sl + (r + l0) * PP_LSTRIDE + 4 * c4) = v;
        }
        wave_lds_fence();
    };
    if (um & 1u) issue(0);
    // A member of a FURTHER group of its bucket (another k-mer in the same bucket: one entry in ten): its row is its group's consensus on its own
    // extent as well -- read from the group's record (rec, shared by the group's members: no read by id), through this lane's LDS stretch,
    // which the slot loop has not touched yet (the records of slot 0 are still in registers)
    {
        const bool fur = row_from_pile && ((my.z >> 16) & 3u) != 0u;
        if (__ballot(fur) != 0ull) {                       // uniform
            const uint64_t se = min((uint64_t) my.w, last);
            uint4 Q0 = make_uint4(0u, 0u, 0u, 0u), Q1 = Q0, Q2 = Q0, Q3 = Q0;
            if (fur) { Q0 = rec[se * 4]; Q1 = rec[se * 4 + 1]; Q2 = rec[se * 4 + 2]; Q3 = rec[se * 4 + 3]; }
            uint4 *mine = reinterpret_cast<uint4 *>(sl + lane * PP_LSTRIDE);
            mine[0] = Q0; mine[1] = Q1; mine[2] = Q2; mine[3] = make_uint4(Q3.x, 0u, 0u, 0u);
            wave_lds_fence();
            // my row = the consensus from index 64 - m_C on (bit 2 (128 - m_C) of the record padded with four words in front)
            const int mC = (int) ((my.z >> 8) & 63u);
            const uint32_t *ss = sl + lane * PP_LSTRIDE - 4;
            const int ob = 2 * (128 - mC), w0 = ob >> 5, sh = ob & 31;
            uint32_t x[10];
#pragma unroll
            for (int k = 0; k < 10; k++) x[k] = ss[w0 + k];
#pragma unroll
            for (int k = 0; k < 9; k++) { const uint32_t sw = funnel(x[k], x[k + 1], sh) & low_bits32(2 * U - 32 * k); B[k] = fur ? sw : B[k]; }
            got_row = got_row || fur;
            wave_lds_fence();
        }
    }
#pragma unroll
    for (int a = 0; a < CL_RMAX; a++) {                    // (slot 0 first: the home run of a lane that takes its row from the pile)
        const bool use = ((um >> a) & 1u) != 0u;           // uniform: a lane of the wave has a run in this slot
        if (use) commit();
        if (a + 1 < CL_RMAX && ((um >> (a + 1)) & 1u)) issue(a + 1);
        if (!use) continue;
        const uint4 R3 = *reinterpret_cast<const uint4 *>(sl + li * PP_LSTRIDE + 12);      // {consensus word 12, set, set, tags | flags | epoch}
        const uint32_t y = R3.w, tag = rk[a] & 31u;
        const bool live = ((vmask >> a) & 1u) != 0u && (y >> 23) == epoch;    // a record of another epoch: no entry in this bucket
        dfr = dfr || (live && ((y >> 22) & 1u) != 0u);     // a bucket k_pile_build found irregular (or of more than 64 entries)
        const uint32_t ns = ((y >> 20) & 3u) + 1u;
        uint32_t mt = ((y & 31u) == tag ? 1u : 0u) | ((ns >= 2u && ((y >> 5) & 31u) == tag) ? 2u : 0u) | ((ns >= 3u && ((y >> 10) & 31u) == tag) ? 4u : 0u) |
                      ((ns >= 4u && ((y >> 15) & 31u) == tag) ? 8u : 0u);
        mt = (live && !dfr) ? mt : 0u;
        more |= (mt & ~1u) << (4 * a);
        const bool on = (mt & 1u) != 0u;
        if (__ballot(on) == 0ull) continue;                // uniform
        const bool home = on && row_from_pile && !got_row && ((ry[a] >> 8) & 255u) == 0u;      // the run of window 0
        got_row = got_row || home;
        take_record(li * PP_LSTRIDE, ((unsigned long long) R3.z << 32) | R3.y, a, 0u, ry[a], on, home);
    }
    {                                                      // the next tile's run lists: on their way while this tile's targets are looked up
        const uint4 *rp = run_list_of(next_side);
#pragma unroll
        for (int c = 0; c < CL_RMAX / 2; c++) Qr[c] = rp[c];
    }
    dfr = dfr || (active && !got_row);                     // (cannot happen: a first-group member's home run wants its bucket's first group)
    // the further groups that share a run's tag (another k-mer of the bucket with the same tag, or the run's own k-mer when it is not the
    // bucket's first): at most one of a slot's groups holds the source's k-mer
    more = dfr ? 0u : more;
    while (__ballot(more != 0u) != 0ull) {                 // uniform
        const bool on = more != 0u;
        const int bit = on ? __builtin_ctz(more) : 0;
        more &= more - 1u;
        const int a = bit >> 2;
        const uint32_t g = (uint32_t) (bit & 3);
        uint32_t ya = ry[0], ka = rk[0];
#pragma unroll
        for (int k = 1; k < CL_RMAX; k++) { ya = a == k ? ry[k] : ya; ka = a == k ? rk[k] : ka; }
        const uint32_t xa = reinterpret_cast<const uint32_t *>(tab)[(size_t) min(ka >> cc.idx_shift, cc.n_buckets) * 32 + 17];      // first entry of the bucket
        const uint64_t se = min((uint64_t) xa + g, last);  // (clamped: a corrupt record must not fault)
        const uint4 Q0 = rec[se * 4], Q1 = rec[se * 4 + 1], Q2 = rec[se * 4 + 2], Q3 = rec[se * 4 + 3];
        uint4 *mine = reinterpret_cast<uint4 *>(sl + lane * PP_LSTRIDE);              // (the slot loop is done with the records in LDS)
        mine[0] = Q0; mine[1] = Q1; mine[2] = Q2; mine[3] = Q3;
        wave_lds_fence();
        take_record(lane * PP_LSTRIDE, ((unsigned long long) Q3.w << 32) | Q3.z, a, g, ya, on, false);
        wave_lds_fence();
    }
    // ---- what the source keeps, from the complete offset set; the one or two targets by id ----
    const unsigned long long kept = occ & ~smear_up(occ, G);
    const int nkept = __popcll(kept);
    if (nkept > 2) dfr = true;
    auto lookup = [&](int d) -> uint32_t {
        // the run whose windows hold offset d, its bucket, the group that gave the items; the member with m_C == q - d among the entries of its m_C >> 3 class
        uint32_t ya = 0u, ka = 0u, ga = 0u;
#pragma unroll
        for (int k = 0; k < CL_RMAX; k++) {
            const bool in = ((vmask >> k) & 1u) != 0u && d >= (int) ((ry[k] >> 8) & 255u) && d < (int) ((ry[k] >> 16) & 255u);
            ya = in ? ry[k] : ya; ka = in ? rk[k] : ka; ga = in ? (grp >> (2 * k)) & 3u : ga;
        }
        const uint4 dd = tab[(size_t) min(ka >> cc.idx_shift, cc.n_buckets) * 8 + 4];       // {entries | ..., first entry, class offsets}
        const int mm = (int) (ya & 255u) - d;
        const int cl = (mm >> 3) & 7;
        const uint32_t b0 = ((cl < 4 ? dd.z : dd.w) >> (8 * (cl & 3))) & 255u;
        const uint32_t b1 = cl >= 7 ? (dd.x & 127u) : ((((cl + 1) < 4 ? dd.z : dd.w) >> (8 * ((cl + 1) & 3))) & 255u);
        uint32_t id = 0xFFFFFFFFu;
        for (uint32_t e = b0; e < b1 && e < 64u; e++) {         // (the entries' side records: id, m_C, group)
            const uint64_t ei = min((uint64_t) dd.y + e, last);
            const uint4 sd = side[ei];
            if ((int) ((sd.z >> 8) & 63u) == mm && ((sd.z >> 16) & 3u) == ga) id = sd.x;
        }
        return id;
    };
    // every source position writes its slot exactly once (no fill in front of this kernel): the edge(s), or "none" -- also for a source that
    // goes to the general kernel, which overwrites it if the source has edges
    unsigned long long slot_val = LOCAL_FIRST_NONE;
    if (active && !dfr && nkept > 0) {
        const int d1 = __builtin_ctzll(kept);
        // the first item lies within the windows of the home run (the run of window 0: the source's own pile) for nine sources in ten: it is the
        // member k_pile_build noted for this entry, no look-up (a bucket line's second half, an entry tail and a group byte: three more L1 misses)
        int p1_home = 0;
#pragma unroll
        for (int k = 0; k < CL_RMAX; k++) p1_home = (((vmask >> k) & 1u) != 0u && ((ry[k] >> 8) & 255u) == 0u) ? (int) ((ry[k] >> 16) & 255u) : p1_home;
        const bool by_succ = d1 < p1_home && (int) (my.z & 255u) == d1 && my.y != 0xFFFFFFFFu;
#ifdef PP_LK
        const int d2 = nkept == 2 ? 63 - __clzll((long long) kept) : -1;
        const int dA = by_succ ? d2 : d1, dB = by_succ ? -1 : d2;
        uint32_t idA = 0xFFFFFFFFu, idB = 0xFFFFFFFFu;
        if (dA >= 0) idA = lookup(dA);
        if (dB >= 0) idB = lookup(dB);
        const uint32_t id1 = by_succ ? my.y : idA;
        const uint32_t id2 = by_succ ? idA : idB;
        bool two = false;
        if (nkept == 2) {
#else
        const uint32_t id1 = by_succ ? my.y : lookup(d1);
        uint32_t id2 = 0u;
        int d2 = 0;
        bool two = false;
        if (nkept == 2) {
            d2 = 63 - __clzll((long long) kept);
            id2 = lookup(d2);
#endif
            // the cap of three small overlaps per source: the second item stands if it is big or fewer than three small items lie before it
            const int ds0 = U - cfg.rsoemo + 1;
            const unsigned long long lowm = ds0 <= 0 ? 0ull : (ds0 >= 64 ? ~0ull : ((1ull << ds0) - 1ull));
            two = d2 < ds0 || __popcll(occ & ((1ull << d2) - 1ull) & ~lowm) < 3;
        }
        if (id1 == 0xFFFFFFFFu || (nkept == 2 && id2 == 0xFFFFFFFFu)) dfr = true;       // (cannot happen: the set said a member sits there)
        else {
            // ONE scattered store per source: the out-degree rides in bit 8 of the slot and k_pile_deg, a streaming pass, moves it to deg[]
            // (a second scattered 4-byte store per source costs this kernel 2.8 ms at the north-star size, the streaming pass 0.3)
            slot_val = ((unsigned long long) id1 << 32) | (uint32_t) d1 | (two ? 0x100u : 0u);
            if (two) o.second[Bs - o.src_base] = ((unsigned long long) id2 << 32) | (uint32_t) d2;
            st_rec += two ? 2 : 1;
        }
    }
    if (have) o.first[Bs - o.src_base] = slot_val;
    // ---- the others: to the general kernel ----
    dfr = dfr && nr_code != 0;
    const uint64_t dm = __ballot(dfr);
    if (dm != 0ull) {                                      // uniform
        if (dfr) sDefer[wave][n_defer + (int) __builtin_amdgcn_mbcnt_hi((uint32_t) (dm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) dm, 0u))] = Bs;
        n_defer += __popcll(dm);
        wave_lds_fence();
        if (n_defer >= 64) flush_defer();
    }
    }                                                      // tiles
    flush_defer();
    st_rec = wave_sum_u64(st_rec);
    if (lane == 0 && st_rec) atomicAdd(&o.counters[CNT_VALID_RECORDS], (unsigned long long) st_rec);
}

// out-degrees of the sources k_pile_probe finished (it left them in bit 8 of the source's slot, and "none" in every other slot); nothing
// to do where that kernel declined the build (the slots are then what the pairwise kernels made of them)
__global__ void __launch_bounds__(256) k_pile_deg(int32_t n, unsigned long long *__restrict__ first, uint32_t *__restrict__ deg, const unsigned long long *__restrict__ pile_cnt) {
    if (pile_declines(pile_cnt)) return;
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned long long f = first[i];
    if (f == LOCAL_FIRST_NONE || deg[i] != 0u) return;     // no edge, or a source the pairwise kernels finished (deg and slot are theirs)
    deg[i] = 1u + (uint32_t) ((f >> 8) & 1ull);
    if (f & 0x100ull) first[i] = f & ~0x100ull;
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
// Does the pile path take this input?  Entries of three 16-byte pieces (rows of up to 9 words: reads of up to 144 nt after ALGA's
// trimming), ONE read length and no alignFrom / alignTo mask (the reduction is then a function of the offset set), one-word offset sets.
// A build that collects the work counters (cfg.stats) takes the pairwise kernels: the counters are defined by what those do.
bool pile_plan(const PrefSufCfg &cfg, const ClusterCfg &cc, int eq, int uniform_len, bool masks) {
    if (eq != PILE_EQ || uniform_len <= 0 || masks || cfg.stats) return false;
    if (uniform_len > 144 || blocks_of(uniform_len) > 9) return false;
    if (uniform_len - cfg.Lmin + 1 > 64 || cc.w > 64 || cc.w < 16 || cc.kk > 32 || cc.kk < 1) return false;      // (w >= 16: k_pile_runs_consensus sweeps a pile's windows in pieces of whole row words)
    return true;
}
size_t pile_record_bytes(uint64_t n) { return (size_t) (n + 2) * 64; }

size_t pile_table_bytes(uint32_t n_buckets) { return ((size_t) n_buckets + 2) * 128; }

// What a bucket of the sample stands for grows with the coverage: the pairwise kernels pay per ENTRY of a bucket, the pile path per bucket.  The
// sample's bucket count is therefore raised to an eighth of its entries where that is more (at 30x a bucket holds ~7.6 entries: unchanged; at
// 160x the threshold of irregular buckets is three times as generous) -- every kernel that reads the two counters then needs no third one.
__global__ void k_pile_sample_close(unsigned long long *__restrict__ pile_cnt) {
    const unsigned long long by_entries = pile_cnt[2] / 8ull;
    if (by_entries > pile_cnt[0]) pile_cnt[0] = by_entries;
}

// The SAMPLE (k_pile_build<true> on the first 1/32 of the key order) comes before k_tgt_gather: a build the pile path keeps needs no entry
// array -- its kernels take the rows by id -- and that kernel, like the pairwise probes, reads the two counters and leaves at once.
__global__ void k_pile_sample_force(unsigned long long *__restrict__ pile_cnt, unsigned long long buckets, unsigned long long irregular) { pile_cnt[0] = buckets; pile_cnt[1] = irregular; }

void launch_pile_sample(const NodesDev &nd, const ClusterCfg &cc, int uniform_len, const uint32_t *skeys, const uint32_t *sids, const void *dir, unsigned long long *pile_cnt,
                        int no_sample, hipStream_t s) {
    (void) hipMemsetAsync(pile_cnt, 0, PILE_CNT_WORDS * sizeof(unsigned long long), s);
    const uint64_t n_entries = nd.n > 0 ? (uint64_t) nd.n : 0;
    // (no_sample -- tests only: 1: the two counters stay zero and the pile kernels take the build whatever its buckets look like; 2: counters that say "mixed form")
    if (no_sample == 2) hipLaunchKernelGGL(k_pile_sample_force, dim3(1), dim3(1), 0, s, pile_cnt, 100ull, 1ull);
    if (n_entries == 0 || no_sample) return;
    const uint64_t tiles = (n_entries + PB_TILE - 1) / PB_TILE;
    const dim3 sample((unsigned) std::max<uint64_t>(1, std::min<uint64_t>(tiles, std::max<uint64_t>(64, tiles / 32)))), block(PB_THREADS);
    hipLaunchKernelGGL((k_pile_build<true>), sample, block, 0, s, nd, skeys, sids, n_entries, (const uint4 *) dir, cc, uniform_len, (uint4 *) nullptr, (uint4 *) nullptr, 0u, (uint4 *) nullptr, pile_cnt,
                       (uint32_t *) nullptr);
    hipLaunchKernelGGL(k_pile_sample_close, dim3(1), dim3(1), 0, s, pile_cnt);
}

// own_mask != null: the run lists come from the consensus (k_pile_runs_consensus) and the entries that read a list of their own are noted
// in own_mask (zeroed here); null: round 4's form -- the pile's list joined from the own lists of its two outer members (k_pile_runs: every node
// must have its run list then)
void launch_pile_build(const NodesDev &nd, const PrefSufCfg &cfg, const ClusterCfg &cc, int uniform_len, const uint32_t *skeys, const uint32_t *sids, const void *dir, void *rec, void *rec2,
                       void *tab, uint32_t epoch, void *side, const void *runs, int nwin, const unsigned long long *pile_cnt, uint32_t *own_mask, hipStream_t s) {
    const uint64_t n_entries = nd.n > 0 ? (uint64_t) nd.n : 0;
    if (n_entries == 0) return;
    const uint64_t tiles = (n_entries + PB_TILE - 1) / PB_TILE;
    if (own_mask) (void) hipMemsetAsync(own_mask, 0, pile_own_mask_bytes(n_entries), s);
    hipLaunchKernelGGL((k_pile_build<false>), dim3((unsigned) tiles), dim3(PB_THREADS), 0, s, nd, skeys, sids, n_entries, (const uint4 *) dir, cc, uniform_len, (uint4 *) rec, (uint4 *) tab, epoch,
                       (uint4 *) side, const_cast<unsigned long long *>(pile_cnt), own_mask);
    if (own_mask)
        hipLaunchKernelGGL(k_pile_runs_consensus, dim3((unsigned) ((n_entries + PR_TILE - 1) / PR_TILE)), dim3(TK_ROWS), 0, s, (const uint4 *) side, n_entries, cc.n_buckets, (uint4 *) tab,
                           (const uint4 *) rec, (uint4 *) rec2, cc, uniform_len, cfg.Lmin, pile_cnt, own_mask);
    else
        hipLaunchKernelGGL(k_pile_runs, dim3((unsigned) ((n_entries + 255) / 256)), dim3(256), 0, s, (const uint4 *) side, n_entries, cc.n_buckets, (uint4 *) tab, (const uint2 *) runs, nd.n, nwin, pile_cnt);
}

size_t pile_own_mask_bytes(uint64_t n) { return (size_t) ((n + 31) / 32 + 1024) * 4; }

// the ids of the entries own_mask names -> list (dense; *count = how many: pile_cnt + 3, zeroed by launch_pile_sample)
void launch_pile_own_ids(const uint32_t *own_mask, const uint32_t *sids, uint64_t n_entries, int32_t *list, uint32_t cap, unsigned long long *pile_cnt, hipStream_t s) {
    if (n_entries == 0) return;
    const uint64_t words = (n_entries + 31) / 32;
    hipLaunchKernelGGL(k_pile_own_ids, dim3((unsigned) ((words + 1023) / 1024)), dim3(256), 0, s, own_mask, sids, n_entries, list, cap, pile_cnt + 3, (const unsigned long long *) pile_cnt);
}

// test harness: members checked / members whose own list differs from their pile's clipped list -> pile_cnt[4], pile_cnt[5]
void launch_pile_check(const void *side, uint64_t n_entries, uint32_t n_buckets, const void *tab, const void *rec2, uint32_t epoch, const void *runs, int n_nodes, int nwin,
                       unsigned long long *pile_cnt, hipStream_t s) {
    if (n_entries == 0) return;
    hipLaunchKernelGGL(k_pile_list_check, dim3((unsigned) ((n_entries + 255) / 256)), dim3(256), 0, s, (const uint4 *) side, n_entries, n_buckets, (const uint4 *) tab, (const uint4 *) rec2, epoch,
                       (const uint2 *) runs,
                       n_nodes, nwin, pile_cnt + 4, (const unsigned long long *) pile_cnt);
}

// A rank's share of the sources (the strong-scaling N-GPU build: every rank holds the index, the sources are dealt out by id range): the side records
// of the entries whose id lies in [src_begin, src_end), compacted block by block -- a block of 1024 entries keeps its order, so the members of a
// pile that fall into the range still sit side by side and share the wave's copy of their bucket's record; the order of the blocks is that of
// their atomics (the probe's results do not depend on the order of its sources).  Every node has exactly one entry: src_end - src_begin records.
constexpr int PSR_IPT = 4;
__global__ void __launch_bounds__(256) k_pile_side_range(const uint4 *__restrict__ side, uint64_t n_entries, uint32_t src_begin, uint32_t src_end, uint4 *__restrict__ out,
                                                         uint64_t out_cap, unsigned long long *__restrict__ cursor, const unsigned long long *__restrict__ pile_cnt) {
    if (pile_declines(pile_cnt)) return;
    __shared__ uint32_t cnt[PSR_IPT * 4];
    __shared__ unsigned long long base;
    const int wave = (int) (threadIdx.x >> 6);
    const uint64_t b0 = (uint64_t) blockIdx.x * (256 * PSR_IPT);
    uint4 v[PSR_IPT];
    uint32_t before[PSR_IPT];
    bool in[PSR_IPT];
#pragma unroll
    for (int k = 0; k < PSR_IPT; k++) {
        const uint64_t j = b0 + (uint64_t) k * 256 + threadIdx.x;
        v[k] = j < n_entries ? side[j] : make_uint4(0xFFFFFFFFu, 0u, 0u, 0u);
        in[k] = j < n_entries && v[k].x >= src_begin && v[k].x < src_end;
        const uint64_t m = __ballot(in[k]);
        before[k] = (uint32_t) __builtin_amdgcn_mbcnt_hi((uint32_t) (m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) m, 0u));
        if ((threadIdx.x & 63u) == 0u) cnt[k * 4 + wave] = (uint32_t) __popcll(m);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (int i = 0; i < PSR_IPT * 4; i++) tot += cnt[i];
        base = tot ? atomicAdd(cursor, (unsigned long long) tot) : 0ull;
    }
    __syncthreads();
    uint32_t run = 0;
#pragma unroll
    for (int k = 0; k < PSR_IPT; k++) {
        uint32_t mine = run;
        for (int w = 0; w < 4; w++) { if (w < wave) mine += cnt[k * 4 + w]; run += cnt[k * 4 + w]; }
        const unsigned long long at = base + mine + before[k];
        if (in[k] && at < out_cap) out[at] = v[k];
    }
}

// side_range / cursor: scratch for a source range that is not all nodes (src_end - src_begin + 64 records of 16 B; one 64-bit word), else unused
void launch_pile_probe(const NodesDev &nd, const PrefSufCfg &cfg, const ClusterCfg &cc, int uniform_len, const void *tab, uint32_t epoch, const void *rec, const void *rec2,
                       const void *side, const void *runs, unsigned long long *counters, uint32_t *deg, unsigned long long *first, unsigned long long *second,
                       int32_t *defer_list, uint32_t defer_cap, const unsigned long long *pile_cnt, int n_cu, hipStream_t s, int32_t src_begin, int32_t src_end,
                       void *side_range, unsigned long long *cursor) {
    const uint64_t n_entries = nd.n > 0 ? (uint64_t) nd.n : 0;
    if (n_entries == 0 || src_end <= src_begin) return;
    ProbeOut o{};
    o.counters = counters; o.deg = deg; o.first = first; o.second = second; o.src_base = src_begin;
    uint64_t n_mine = n_entries;
    const void *side_src = side;
    if (src_begin != 0 || src_end != nd.n) {
        n_mine = (uint64_t) (src_end - src_begin);
        (void) hipMemsetAsync(cursor, 0, sizeof(unsigned long long), s);
        hipLaunchKernelGGL(k_pile_side_range, dim3((unsigned) ((n_entries + 256 * PSR_IPT - 1) / (256 * PSR_IPT))), dim3(256), 0, s, (const uint4 *) side, n_entries,
                           (uint32_t) src_begin, (uint32_t) src_end, (uint4 *) side_range, n_mine, cursor, pile_cnt);
        side_src = side_range;
    }
    const uint64_t tiles = (n_mine + PP_WAVES * 64 - 1) / (PP_WAVES * 64);
    const dim3 grid((unsigned) std::max<uint64_t>(1, std::min<uint64_t>(tiles, (uint64_t) std::max(1, n_cu) * (PP_OCC * 4 / PP_WAVES)))), block(PP_WAVES * 64);      // PP_OCC waves per SIMD, four SIMDs per CU
    hipLaunchKernelGGL(k_pile_probe, grid, block, 0, s, cfg, cc, uniform_len, nd, n_entries, n_mine, nd.n, (const uint4 *) tab, epoch, (const uint4 *) rec, (const uint4 *) rec2,
                       (const uint4 *) side, (const uint4 *) side_src, (const uint2 *) runs, o, defer_list, defer_cap, pile_cnt);
}

void launch_pile_deg(int32_t n, unsigned long long *first, uint32_t *deg, const unsigned long long *pile_cnt, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_pile_deg, dim3((unsigned) (((int64_t) n + 255) / 256)), dim3(256), 0, s, n, first, deg, pile_cnt);
}

} // namespace alga
