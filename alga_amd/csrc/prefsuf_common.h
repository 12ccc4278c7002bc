// alga_amd/csrc/prefsuf_common.h -- shared host/device definitions of the PrefSuf overlap engine.
//
// The engine computes the graph of the reference's GraphCreatorPrefSuf
// (src/GraphCreators/GraphCreatorPrefSuf.cpp:73-488 + src/main.cpp:291) with a different
// algorithm shape: one seed table over a prefix of every target, one probe per
// (source, overlap length) suffix window verified by an exact 2-bit compare, and the transitive
// reduction either at the source, inside the probing wave (the default whenever it is exact:
// prefsuf_device.h local_reduce), or as a per-target sequential replay of
// the reference's insertion order (any input).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define ALGA_HD __host__ __device__
#else
#define ALGA_HD
#endif

namespace alga {

// ---- overlap record: one verified suffix(B)==prefix(C) pair that survived the per-source cap ----
//   ol = offset | (overlap_len << 22) | (small << 31)      overlap_len <= 501 (9 bits); offset = |source| - overlap_len: 22 bits,
//   so nodes of up to OL_MAX_NODE_LEN nucleotides (the reference's second call runs on contigs, src/main.cpp:633-656)
constexpr uint32_t OL_OFF_MASK = 0x3FFFFFu;
constexpr int      OL_LEN_SHIFT = 22;
constexpr uint32_t OL_LEN_MASK = 0x1FFu;
constexpr int      OL_MAX_NODE_LEN = (1 << 22) - 1;
constexpr uint32_t OL_SMALL = 0x80000000u;

ALGA_HD inline uint32_t ol_pack(int off, int L, bool small) {
    return (uint32_t) off | ((uint32_t) L << OL_LEN_SHIFT) | (small ? OL_SMALL : 0u);
}
ALGA_HD inline int  ol_off(uint32_t ol) { return (int) (ol & OL_OFF_MASK); }
ALGA_HD inline int  ol_len(uint32_t ol) { return (int) ((ol >> OL_LEN_SHIFT) & OL_LEN_MASK); }
ALGA_HD inline bool ol_small(uint32_t ol) { return (ol & OL_SMALL) != 0; }

// ---- seed fingerprint of a 2*min_overlap-bit window, fed 32 bits at a time --------------------
// Only a filter: every candidate is verified bit for bit afterwards, so the fingerprint never
// decides an edge (the reference decides on two modular hashes, GraphCreatorPrefSuf.cpp:386-387).
// Step = multiply-with-carry on 32-bit halves: one v_mad_u64_u32 per word (a full 64-bit multiply is four quarter-rate
// instructions on CDNA).  The seed covers the first min(min_overlap, SEED_MAX_NT) nucleotides of a window.
constexpr int SEED_MAX_NT = 64;
constexpr int SEED_MAX_WORDS = SEED_MAX_NT / 16;
ALGA_HD inline uint64_t fp_init() { return 0x243F6A8885A308D3ull; }
ALGA_HD inline uint64_t fp_step(uint64_t h, uint32_t w) {
    return (uint64_t) ((uint32_t) h ^ w) * 0x9E3779B1u + (h >> 32);
}
ALGA_HD inline uint64_t fp_final(uint64_t h) {
    h ^= h >> 31;
    h *= 0xc4ceb9fe1a85ec53ull;
    return h ^ (h >> 29);
}

constexpr uint32_t REC_INVALID = 0xFFFFFFFFu; // rec_dst marker: unused slot of a record chunk
constexpr uint64_t SEED_EMPTY = ~0ull; // seed-table slot: ((tag23 << 9 | len9) << 32) | node id ; empty = all ones
constexpr int LOCAL_MAX_SPAN = 127;    // source-side reduction: max_len - Lmin; offsets and overhangs fit one (<= 63) or two 64-bit mask words
constexpr int SEED_BUCKET = 8;         // slots per bucket: 8 x 8 B = one 64-byte line per probe

// number of uint32 blocks that hold `len_nt` nucleotides (Bitset::blocks(), Bitset.h:206)
ALGA_HD inline int blocks_of(int len_nt) { return len_nt <= 0 ? 0 : ((2 * len_nt - 1) >> 5) + 1; }

// Row stride (uint32 words) of the engine's own HBM layout for rows of `w` used words: 16-, 32- or 64-byte rows never
// straddle a 64-byte line (a 48-byte row does, every other time); longer rows are whole lines.
ALGA_HD inline int hbm_row_stride(int w) { return w <= 4 ? 4 : (w <= 8 ? 8 : ((w + 15) & ~15)); }

// device-side counters; index = enum below
enum Counter {
    CNT_RECORDS = 0,       // record-list cursor (incl. invalid padding; may exceed capacity -> retry)
    CNT_RAW,               // verified raw overlaps
    CNT_WINDOWS,           // windows probed
    CNT_SLOTS,             // seed-table slots read
    CNT_TR_LISTED,         // bitsetChecksCount
    CNT_TR_COMPARES,       // goodBitsetChecksCount
    CNT_TR_REMOVED,        // bitsetCheckEdgesRemoved
    CNT_EDGES,             // final edges
    CNT_MAX_IN,            // max records per target
    CNT_LIVE_NODES,
    CNT_VALID_RECORDS,     // records that carry an overlap
    CNT_SORT_VALID,        // records whose target lies in the owned range (k_make_keys)
    CNT_MASK_ASYM,         // live nodes with alignFrom but not alignTo (source-side reduction needs none)
    CNT_LOCAL_OVERFLOW,    // sources with more raw overlaps than the source-side reduction holds (LDS; in the second pass: its global slice)
    CNT_LOCAL_GENERIC,     // sources that took the all-pairs path of the source-side reduction
    CNT_LOCAL_MAXITEMS,    // largest number of raw overlaps of one source seen by the source-side reduction
    CNT_DEFERRED,          // clustered probe, pair kernel: sources handed to the general kernel
    CNT_ROUNDS,            // clustered probe, quad kernel: rounds (wave iterations), statistics builds only
    CNT_PILE_BUCKETS,      // pile path: non-empty buckets of the entry array / those it does not take (copied from k_pile_build's counters by k_pile_probe)
    CNT_PILE_IRREGULAR,
    CNT_PILE_OWN,          // pile path: entries that read a run list of their own (the list-driven key pass behind k_pile_runs_consensus)
    CNT_DEFERRED2,         // mixed form: sources k_probe_stream (list mode) handed on; k_defer_swap moves that list and count to the first list's place
    CNT_DEFERRED_PILE,     // mixed form: sources k_pile_probe handed on (CNT_DEFERRED before the swap)
    CNT_TOTAL = 24
};
constexpr int LOCAL_SLOTS_MAX = 4;                         // edges of a source k_probe_stream writes to slots itself (first[] + up to three in second[]: ProbeOut::slot_stride; eight measured no better)


struct NodesDev {
    const uint32_t *words;
    const int32_t  *len;
    const uint8_t  *from; // may be null
    const uint8_t  *to;   // may be null
    int32_t n;
    int32_t stride;       // uint32 per row
};

// clustered minimizer join (prefsuf_cluster.hip): targets filed under the minimizer of their min_overlap-long prefix
constexpr int      CL_KMIN = 16;              // preferred shortest minimizer k-mer; k = max(Lmin - 63, min(Lmin, CL_KMIN)), w = Lmin - k + 1 <= 64
constexpr int      CL_KMIN_HARD = 8;          // below this the clustered probe declines (the seed-table probe takes the input)
constexpr int      CL_MBITS = 6;              // low bits of a target's sort key: m_C, the minimizer's position in its prefix (w <= 64)
constexpr int      CL_RMAX = 8;               // minimizer runs stored per node (a 150-bp read has 2.9 on average)
constexpr int      CL_RUNS_FLAGGED = 0xFF;    // nruns marker: more runs / records than k_node_runs stores
constexpr int      CL_MAX_EQ = 5;             // 16-byte pieces per entry: rows of up to 4 * CL_MAX_EQ - 3 words (272 nt)
constexpr uint32_t CL_META_FROM = 1u << 20;   // entry meta word: m_C | len << 8 | alignFrom << 20
struct ClusterCfg {
    int32_t  kk;         // minimizer k-mer length
    int32_t  w;          // k-mers per window
    uint32_t lo_mask;    // k-mer bits in the first / second 32-bit word
    uint32_t hi_mask;
    int32_t  idx_shift;  // hash bucket = hash >> idx_shift
    uint32_t n_buckets;
};

struct PrefSufCfg {
    int32_t Lmin;        // MIN_OVERLAP_PREF_SUF
    int32_t rsoemo;      // REMOVE_SMALL_OVERLAP_EDGES_MIN_OVERLAP
    int32_t Lcap;        // last overlap length iterated: min(maxReadLength, cap) + 1
    int32_t soes;        // 3
    int32_t seed_words;  // ceil(2*seed_nt/32), seed_nt = min(Lmin, SEED_MAX_NT)
    uint32_t seed_last_mask;
    int32_t reversed;    // reference quirk: rsoemo beyond the last iteration -> graph comes out reversed
    int32_t stats;
};

} // namespace alga
