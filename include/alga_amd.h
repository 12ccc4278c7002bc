/*
 * include/alga_amd.h -- C ABI of the MI355X (gfx950) overlap-graph engine for the ALGA assembler.
 *
 * This is the drop-in boundary for ONE path of swacisko/ALGA: the overlap-graph construction that
 * sits behind `class GraphCreator` (reference include/GraphCreators/GraphCreator.h:12-62) and is
 * selected in src/main.cpp:246-250.  Plain pointers and sizes only; no C++/torch types.
 * The reference-side bindings (GraphCreator subclasses that marshal to these calls) are
 * the headers under alga_amd/host/adapter/, described in INTEGRATION.md.  Paths below are relative to the reference root.
 *
 * Conventions
 *   - every call returns 0 on success or a negative alga_status; alga_last_error() gives the text.
 *     (the reference's convention is cerr + exit(1), e.g. src/DataStructures/Read.cpp:146-149; the
 *     adapter maps a non-zero status to that.)
 *   - the engine borrows caller memory and never frees it (GraphCreator::~GraphCreator only nulls
 *     its pointers, src/GraphCreators/GraphCreator.cpp:15-18); buffers the engine returns are
 *     released with alga_free_edges().
 *   - calls block until the result is complete (the reference joins its worker threads before
 *     returning, src/GraphCreators/GraphCreatorPrefSuf.cpp:147-161); one engine handle must not be
 *     used from two threads at once, different handles are independent.
 *   - there is NO CPU fallback: without a usable HIP device every compute call fails.
 */
#ifndef ALGA_AMD_H
#define ALGA_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ALGA_AMD_ABI_VERSION 7

typedef enum {
    ALGA_OK = 0,
    ALGA_ERR_INVALID_ARGUMENT = -1,
    ALGA_ERR_NO_DEVICE = -2,        /* no HIP device / HIP runtime failure at start-up        */
    ALGA_ERR_HIP = -3,              /* a HIP call or kernel failed                            */
    ALGA_ERR_OUT_OF_MEMORY = -4,
    ALGA_ERR_CAPACITY = -5,         /* an internal 32-bit index space would overflow; shard   */
    ALGA_ERR_IO = -6,
    ALGA_ERR_UNSUPPORTED = -7       /* the requested reduction form is not exact for this input: use the other one */
} alga_status;

/* Where the transitive reduction of GraphCreatorPrefSuf.cpp:397-483 is evaluated (same result either way):
 *   PER_TARGET  : the literal replay of the reference's push order per target node (any input);
 *   SOURCE_SIDE : per source node inside the probing wave, from the source's own raw overlaps;
 *                 exact when max_len <= max_len_cap, max_len - min_overlap <= 127, min_overlap <= rsoe_min_overlap <=
 *                 min(max_len, max_len_cap) + 1 and no live node has alignFrom without alignTo -- everything ALGA's
 *                 command line can produce for reads up to ~280 nt after trimming with the default scale.
 *   AUTO        : SOURCE_SIDE when those conditions hold (checked on the device), else PER_TARGET. */
typedef enum { ALGA_REDUCTION_AUTO = 0, ALGA_REDUCTION_PER_TARGET = 1, ALGA_REDUCTION_SOURCE_SIDE = 2 } alga_reduction;

/* Which probe finds the raw overlaps of the SOURCE_SIDE form (same result either way):
 *   TABLE   : bucketised seed table, one probe per (source, overlap length) -- takes any input;
 *   CLUSTER : clustered minimizer join -- targets sorted by the minimizer of their min_overlap-long prefix, ~3 contiguous
 *             lookups per source; takes max_len - min_overlap <= 127 and reads of up to 272 nt (round 4: 250-bp reads too, in the
 *             two-word form of the source-side reduction through its general kernel; up to 63 / 208 nt through k_probe_stream),
 *             anything else uses TABLE;
 *   AUTO    : CLUSTER whenever it takes the input (1.8x faster at 1.7 M nodes, 2.9x at 90 M) -- for the wide shapes (span > 63 or
 *             reads > 208 nt) from 4 M live nodes on (below, table and rows are cache-resident and TABLE is faster) --, else TABLE. */
typedef enum { ALGA_PROBE_AUTO = 0, ALGA_PROBE_TABLE = 1, ALGA_PROBE_CLUSTER = 2 } alga_probe;

typedef struct alga_engine alga_engine; /* opaque */

/* Directed overlap edge src -> dst: "dst starts at position `offset` of src"
 * == one (neighbor, offset) pair of Graph::V[src] (include/DataStructures/Graph.h:54,97). */
typedef struct { int32_t src, dst, offset; } alga_edge;

/* Node set = what GraphCreator's constructor receives as `vector<Read*>* reads`:
 * 2-bit packed reads, nucleotide i in bits (2i, 2i+1) of a little-endian bit string held in
 * uint32 blocks, A0 C1 G2 T3, unused tail bits zero (include/DataStructures/Bitset.h:41-50,
 * src/DataStructures/Read.cpp:40-68).  len[i] == 0 means READS[i] == nullptr. */
typedef struct {
    const uint32_t *words;   /* n rows of `stride_words` uint32                                  */
    int32_t         stride_words; /* >= ceil(2*max_len/32)                                       */
    const int32_t  *len;     /* n lengths in nucleotides                                         */
    int32_t         n;       /* number of nodes == Graph::size()                                 */
    const uint8_t  *align_from; /* n bytes or NULL (= all 1): GraphCreator::alignFrom            */
    const uint8_t  *align_to;   /* n bytes or NULL (= all 1): GraphCreator::alignTo              */
} alga_nodes;

/* Parameters GraphCreatorPrefSuf reads from Params:: globals. */
typedef struct {
    int32_t min_overlap;      /* Params::MIN_OVERLAP_PREF_SUF                                    */
    int32_t rsoe_min_overlap; /* Params::REMOVE_SMALL_OVERLAP_EDGES_MIN_OVERLAP                   */
    int32_t soes;             /* small-overlap edges kept per source; the reference hard-codes 3
                                 (include/GraphCreators/GraphCreatorPrefSuf.h:62)                 */
    int32_t max_len_cap;      /* 500 (src/GraphCreators/GraphCreatorPrefSuf.cpp:92)               */
    int32_t collect_stats;    /* != 0: fill the work counters of alga_prefsuf_stats               */
    int32_t reduction;        /* alga_reduction; SOURCE_SIDE fails with ALGA_ERR_UNSUPPORTED when not exact */
    int32_t keys_shared;      /* sharded form only.  1: the per-node keys were made by alga_prefsuf_keys_device and
                                 all-gathered by the caller; the build skips its own key pass.  2: this build follows
                                 another build of the SAME node set on this engine (nothing else in between) and reuses
                                 its index: only the probe of [src_begin, src_end) runs.  A previous build that the pile path
                                 KEPT (reads of one length, no masks, the sample found the buckets regular: option "pile")
                                 has no entry array: the piece goes through its piles (k_pile_probe over the piece's ids;
                                 with option "pile_range" 0 it is refused with ALGA_ERR_INVALID_ARGUMENT, as until round 5);
                                 after a build the pile path declined the entry array serves the piece as before        */
    int32_t twin_rows;        /* host entry points only.  1: nodes->words holds the rows of the ODD nodes alone (row k = node 2k + 1,
                                 n / 2 rows): node 2k is the reverse complement of node 2k + 1 -- ALGA's layout (src/IO/InputReader.cpp:
                                 78-80,363-377; the duplicate removal deletes twins together, src/main.cpp:150-232) -- and its row is
                                 rebuilt on the device from len[2k] (0 = removed, else == len[2k + 1]); half of the PCIe upload.  len and
                                 the masks keep all n entries                                                                           */
} alga_prefsuf_params;

/* Work counters; the first four mirror GATHER_STATISTICS of the reference
 * (include/GraphCreators/GraphCreatorPrefSuf.h:112-118). */
typedef struct {
    uint64_t raw_overlaps;        /* goodPrefSufChecks: suffix==prefix pairs found               */
    uint64_t transitive_listed;   /* bitsetChecksCount: in-list entries visited by big overlaps  */
    uint64_t transitive_compares; /* goodBitsetChecksCount: 2-bit compares started               */
    uint64_t transitive_removed;  /* bitsetCheckEdgesRemoved                                     */
    uint64_t windows_probed;      /* (node, overlap length) seed-table probes                    */
    uint64_t slots_scanned;       /* seed-table slots read while probing                         */
    uint64_t records;             /* overlap records kept after the small-overlap cap            */
    uint64_t edges;               /* edges in the result                                         */
    uint64_t table_slots;         /* seed-table capacity                                         */
    uint64_t max_in_records;      /* largest per-target record list                              */
    double   ms_total;            /* device time of the last call, HIP events on the engine's stream */
    double   ms_seed, ms_probe, ms_group, ms_reduce, ms_emit;
    uint64_t nodes_live;          /* nodes with len>0                                            */
    uint64_t reduction_used;      /* alga_reduction of the last build (1 or 2)                   */
    uint64_t generic_sources;     /* SOURCE_SIDE: sources that needed the all-pairs path (collect_stats) */
    uint64_t big_sources;         /* SOURCE_SIDE: sources with more raw overlaps than a wave's LDS holds (second pass) */
    uint64_t probe_used;          /* alga_probe of the last build (1 or 2)                       */
    uint64_t deferred_sources;    /* CLUSTER probe: sources the first kernel handed to the general kernel (all of them when it was skipped) */
    double   ms_probe_pairs;      /* CLUSTER probe: the first kernel's (k_probe_stream) part of ms_probe (0: not run) */
    double   ms_keys, ms_sort, ms_gather, ms_dir; /* CLUSTER probe: the parts of ms_seed -- k_node_runs, radix sort of (key, id),
                                     k_tgt_gather, k_tgt_dir (0 for a build that reused them: keys_shared)         */
    uint64_t probe_rounds;        /* CLUSTER probe, k_probe_stream: rounds = wave iterations (collect_stats); sources / rounds = sources packed per round */
    double   ms_pile;             /* CLUSTER probe, option pile: k_pile_build (consensus records of the entry array), part of ms_seed; 0: not run  */
    uint64_t pile_buckets, pile_irregular;   /* ... non-empty buckets in a SAMPLE of the key order (its first 1/32; an eighth of the sample's entries where that
                                     is more: a bucket of a high-coverage read set stands for more pairwise work saved) / those of them the pile path does
                                     not take (a source with a run in such a bucket goes to the general kernel); more than 1 in
                                     ALGA_PILE_IRREGULAR_ONE_IN irregular: the pairwise kernel k_probe_stream took the build instead of k_pile_probe */
    uint64_t pile_list_checked, pile_list_mismatch;   /* option "pile_check" (tests): first-group members whose own run list was compared with their pile's
                                     consensus-derived list clipped to their windows / those for which the two differ (must be 0) */
    uint64_t pile_own_lists;      /* pile path (option pile_runs): entries of the key order that read a run list of their OWN -- outside a first group, or members of a
                                     pile whose consensus gave no list -- and got it from the list-driven key pass (0: every node's list was made up front) */
    double   host_ms_check, host_ms_upload, host_ms_build, host_ms_download;   /* host entry points (alga_prefsuf_build_host*): wall time of the argument / length
                                     checks, of the upload (staging, re-stride / twin expansion included), of the build, of the download of the edges */
    uint64_t pile_mixed;          /* 1: the pile path kept a build with more than 1 irregular bucket in ALGA_PILE_IRREGULAR_ONE_IN (but not more than 1 in
                                     ALGA_PILE_DECLINE_ONE_IN): the sources k_pile_probe handed on went through k_probe_stream first                       */
    uint64_t pile_deferred;       /* sources k_pile_probe handed on (deferred_sources: what reached the general kernel in the end)                          */
} alga_prefsuf_stats;
/* The pile path keeps a build iff  pile_irregular * ALGA_PILE_IRREGULAR_ONE_IN <= pile_buckets  (decided on the device; pile_buckets as reported: raised to
 * an eighth of the sample's entries at high coverage).  Every source with a run in an
 * irregular bucket goes to the general kernel -- eight times the buckets' share of the sources, at ~20 times the cost per source: above ~0.5 % of
 * irregular buckets the pairwise kernels are faster (reads with sequencing errors: 25 %; error-free reads of a genome of 1 Gb: 1.5 %, 250 Mb: 0.07 %). */
#define ALGA_PILE_IRREGULAR_ONE_IN 250
/* Round 5: above that share the pile path still keeps the build as long as  pile_irregular * ALGA_PILE_DECLINE_ONE_IN <= pile_buckets  -- the sources
 * it hands on then go through the pairwise stream kernel (k_probe_stream over the list; the entry array is built for it) and only what that cannot
 * finish reaches the general kernel: ~0.3 instead of ~2 ms per million sources handed on.  Beyond it (reads with sequencing errors) the pairwise
 * kernels take the whole build as before.  alga_prefsuf_stats.pile_mixed tells which form ran. */
#define ALGA_PILE_DECLINE_ONE_IN 20

/* ---- lifetime --------------------------------------------------------------------------- */
int         alga_abi_version(void);
int         alga_engine_create(int hip_device, alga_engine **out);
void        alga_engine_destroy(alga_engine *e);
const char *alga_last_error(const alga_engine *e);      /* valid until the next call on `e`     */
int         alga_engine_device_name(const alga_engine *e, char *buf, size_t buflen);

/* Engine switches.  None changes a result, only how it is computed; there are no environment variables.
 *   "probe"                      alga_probe (default AUTO)
 *   "cluster_bucket_bias"        -8..8: log2 factor on the bucket count of the CLUSTER probe's index (default 0: ~1 entry per bucket)
 *   "cluster_pairs"              0: the CLUSTER probe runs its general kernel only (one source per wave); default 1: k_probe_stream first
 *                                (the entries of consecutive sources packed densely onto the lanes), the general kernel on what it defers
 *   "pile"                       default 1: reads of one length without masks take the probe through PILES (alga_amd/csrc/prefsuf_pile.hip): one compare
 *                                of a source against the consensus of a minimizer's targets instead of one per target; 0: always the pairwise kernels;
 *                                2 (tests only): without the sample that leaves reads with errors to the pairwise kernels; 3 (tests only): without it, in
 *                                the MIXED form (ALGA_PILE_DECLINE_ONE_IN below): what the pile kernel hands on goes through k_probe_stream by list
 *   "pile_runs"                  default 1: the run list of a pile (what its members probe with) is computed from the pile's CONSENSUS, once per pile
 *                                (k_pile_runs_consensus), and the key pass of a build the pile path keeps makes the target keys alone -- own run lists
 *                                only for the entries outside a first group and the sources handed to the general kernel; 0: round 4's form (every
 *                                node its own list, a pile's list joined from its two outer members')
 *   "pile_check"                 tests only.  != 0: every node gets its own run list as well and every first-group member's is compared with its pile's list
 *                                clipped to the member's windows (alga_prefsuf_stats.pile_list_checked / pile_list_mismatch)
 *   "pile_skip_gather"           default 1: a build the pile path keeps has no entry array (the rows in key order, 48 bytes per node: its kernels
 *                                read the rows by id, and the copy alone costs 3 ms per 90 M nodes); 0: the entry array is always built
 *   "cluster_order"              default 1: k_probe_stream takes the sources in the order of the entry array (sources of one locus together:
 *                                shared look-ups, cache hits); 0: in id order (what a range of ids always gets)
 *   "local_big_max"              largest per-wave item slice of the SOURCE_SIDE second pass (default -1 = built-in 4096); beyond it
 *                                the build takes PER_TARGET
 *   "auto_reduction_per_target"  != 0: alga_prefsuf_params.reduction == AUTO resolves to PER_TARGET
 *   "shard_bucket_max"           1..4096 (default 4096): run descriptors of ONE bucket the bucket-sharded join (alga_shard_join_device) takes; a
 *                                bucket with more makes the call answer ALGA_ERR_UNSUPPORTED (tests lower it to exercise that)
 *   "own_sort"                   default 1: the (key, id) sort of the index build and the descriptor sort of the bucket-sharded form are the engine's own
 *                                radix sort (alga_amd/csrc/radix_sort.hip: stable LSD, wave-match ranking, XCD-contiguous tiles); 0: rocPRIM's onesweep
 *                                sort (what rounds 2-4 used; kept for A/B and tests)
 *   "stream_slots"               default 4: k_probe_stream writes the edges of a source with up to four standing items itself (slots beside the source's
 *                                first: 16 B per node more; eight measured no better); 2: two, any other source goes to the general kernel (the form until
 *                                round 4; A/B and tests)
 *   "pile_range"                 default 1: the pile path also takes a build of a source id range (a rank's share of the strong-scaling N-GPU build: the
 *                                index as for all sources, the range's sources compacted for k_pile_probe); 0: all sources only (until round 5; A/B and tests)
 *   "pkb_legacy"                 default 0; A/B and tests: one bit per piece of the approximate supplement that round 5 reworked, set = round 4's form of it:
 *                                1 groups of 8..16 k-mers a wave each (now four per wave), 2 the library's sort of the k-mer entries (now the engine's),
 *                                4 group heads in three kernels (now one), 8 the replay of the 8..16 groups inside the pair kernel, 16 the library's sort /
 *                                unique of the additions and a row-pointer pass, 32 the k-mer walk on a 128-bit value, 64 every tip record's snapshot
 *                                half rewritten every round, 128 a k-mer walk per round (now all rounds in one), 256 a device-to-host copy per count the
 *                                host waits for (now one small kernel per wait writes them into the pinned block), 512 no look-ahead (now the next round's
 *                                k-mer entries are sorted and their groups listed on a second stream beside the round's group joins and merge)
 *   "test_presort_oom"           tests only.  != 0: the allocation of the supplement's look-ahead buffers (a second set of sorted k-mer entries and sort scratch,
 *                                ~50 B per entry) answers ALGA_ERR_OUT_OF_MEMORY: the sequence must give them back and run its rounds one after the other
 *   "test_pile_oom"              tests only.  != 0: the allocation of the pile path's own buffers (~180 B per node) answers ALGA_ERR_OUT_OF_MEMORY: the build
 *                                must give them back and finish on the pairwise kernels (what a real out-of-memory there does)
 *   "test_unsorted_index"        tests only.  != 0: the CLUSTER probe's entry directory is built over UNSORTED keys; the directory pass
 *                                flags it, the probe's reads are clamped to the entry array, and the build returns ALGA_ERR_HIP (no GPU fault) */
int         alga_engine_set_option(alga_engine *e, const char *name, int64_t value);

void        alga_prefsuf_default_params(alga_prefsuf_params *p);

/* ---- the drop-in: GraphCreatorPrefSuf ---------------------------------------------------- */
/* Host buffers in, edges out.  Replaces, for the caller at src/main.cpp:246-291,
 *   new GraphCreatorPrefSuf(READS, G, false); setAlignFrom/To(...); startAlignmentGraphCreation();
 *   G->retainOnlySmallestOffset();
 * The result is the graph the caller would hold at src/main.cpp:293: edges grouped by src,
 * each adjacency list sorted by (dst, offset) (src/DataStructures/Graph.cpp:367-387).
 * *edges is engine-owned host memory; release with alga_free_edges() on the SAME engine and BEFORE alga_engine_destroy():
 * the engine keeps track of the lists it handed out (one released list is kept as a spare for the next call), destroy frees
 * whatever is still outstanding, and a list must not be touched afterwards. */
int  alga_prefsuf_build_host(alga_engine *e, const alga_nodes *nodes, const alga_prefsuf_params *p,
                             alga_edge **edges, uint64_t *n_edges);
void alga_free_edges(alga_engine *e, alga_edge *edges);

/* The same build with the graph handed back in COMPACT form -- what crosses PCIe on the way down is 5.1 bytes per edge instead of 12 (the host
 * entry point is PCIe-bound: 0.56 GB instead of 1.1 GB at the north-star size).  Lists in node order: node i owns the next degree[i] entries of
 * dst[] / offset[] (each list sorted by (dst, offset), as in alga_edge form: src/DataStructures/Graph.cpp:367-387).  ALGA_ERR_UNSUPPORTED where an
 * out-degree or an offset does not fit a byte (reads longer than min_overlap + 255): take alga_prefsuf_build_host.  One host block, released by
 * alga_free_compact_edges; alga_adapter::fill_graph_compact turns it into Graph::V. */
typedef struct {
    int32_t         n_nodes;
    uint64_t        n_edges;
    const uint8_t  *degree;   /* n_nodes */
    const uint32_t *dst;      /* n_edges */
    const uint8_t  *offset;   /* n_edges */
} alga_compact_edges;
int         alga_prefsuf_build_host_compact(alga_engine *e, const alga_nodes *nodes, const alga_prefsuf_params *p, alga_compact_edges *out);
/* ... and any edge list on the device (a build's result, the supplement's) brought down in that form: what alga_download_edges is to alga_edge triples */
int         alga_download_edges_compact(alga_engine *e, int32_t n_nodes, const alga_edge *d_edges, uint64_t n_edges, alga_compact_edges *out);
void        alga_free_compact_edges(alga_engine *e, alga_compact_edges *c);

/* Pinned host memory (hipHostMalloc): node arrays allocated here are read by the DMA engines as they are -- alga_prefsuf_build_host* and
 * alga_upload_*nodes then skip the copy through the engine's staging buffers (the pointer is recognised, nothing else changes).  Allocation is slow
 * (pages are mapped and locked): ask once, while the reads are still being parsed.  NULL on failure (alga_last_error). */
void       *alga_host_alloc(alga_engine *e, size_t bytes);
void        alga_host_free(alga_engine *e, void *p);

/* Allocates, ahead of time, every device buffer a build of the exact path needs for a node set of `n_nodes` rows of up to
 * `max_len` nucleotides (and, n_edges_hint > 0, for that many edges; 0 = one per node), and runs a miniature build of the same shape
 * (4096 random reads) so that the HIP runtime loads the kernels' code objects now: an assembler builds its graph ONCE, and what the
 * first build of a process pays on top of a warm one is mostly that loading (~20 ms; the allocations themselves ~1 ms at 90 M nodes).
 * Safe to call from a second host thread while the caller is still parsing or uploading the reads -- but not concurrently with
 * another call on the same engine.  Later builds allocate only what turns out larger. */
int  alga_engine_reserve(alga_engine *e, int32_t n_nodes, int32_t max_len, int32_t min_overlap, uint64_t n_edges_hint);

/* The two halves of alga_prefsuf_build_host for callers that keep the node set resident across several stages (exact graph ->
 * supplement -> ... on ONE upload): host node set -> the engine's upload buffers in the engine's row layout, `*dev` describes the
 * resident copy (valid until the next upload on this engine); device edge list -> engine-owned host list (alga_free_edges). */
int  alga_upload_nodes(alga_engine *e, const alga_nodes *nodes, alga_nodes *dev);
/* the same for a node set in ALGA's twin layout, given by the rows of its ODD nodes alone (alga_prefsuf_params.twin_rows): half the upload */
int  alga_upload_twin_nodes(alga_engine *e, const alga_nodes *nodes /* words: n / 2 rows */, alga_nodes *dev);
int  alga_download_edges(alga_engine *e, const alga_edge *d_edges, uint64_t n_edges, alga_edge **edges);

/* Same computation with the node set already resident in HBM (all pointers in `nodes` are device
 * pointers on the engine's device).  Work is enqueued on `hip_stream` (a hipStream_t).  NULL = the engine's own stream, a
 * NON-BLOCKING stream that orders with no other stream (not even the null stream): with NULL every input must be complete
 * before the call (synchronise the producing stream first); to chain behind work of another stream pass that stream.
 * The same holds for every call below that takes a `hip_stream`.  The edge list stays on the device:
 *   *d_edges : device pointer to alga_edge[*n_edges], owned by the engine, valid until the next
 *              build call on this engine or alga_engine_destroy(). */
int  alga_prefsuf_build_device(alga_engine *e, const alga_nodes *nodes, const alga_prefsuf_params *p,
                               void *hip_stream, const alga_edge **d_edges, uint64_t *n_edges);

/* Copies `bytes` of engine-owned device memory (an edge list, a device node set) into host memory: for callers that
 * are plain C/C++ without the HIP headers. */
int  alga_copy_to_host(alga_engine *e, void *dst, const void *d_src, size_t bytes);

/* Plain device buffers for callers without the HIP headers (e.g. alignFrom / alignTo masks next to a device-resident node set). */
int  alga_device_alloc(alga_engine *e, size_t bytes, void **d_out);
void alga_device_free(alga_engine *e, void *d_ptr);
int  alga_copy_to_device(alga_engine *e, void *d_dst, const void *src, size_t bytes);

/* Counters and per-phase device times of the last build call on `e`. */
int  alga_prefsuf_last_stats(const alga_engine *e, alga_prefsuf_stats *out);

/* ---- sharded form (one process per GPU; the exchange between the two calls is the caller's) ----
 * Phase 1, on every rank: discover the overlaps whose SOURCE node id is in [src_begin, src_end)
 * against the full (replicated) node set, apply the per-source small-overlap cap (it is per source,
 * so it needs no exchange: GraphCreatorPrefSuf.cpp:397-401), and return the overlap records as two
 * device arrays of *n_records slots (engine-owned, valid until the next call on `e`):
 *   d_dst[i] : target node id, or 0xFFFFFFFF for an unused slot (skip it)
 *   d_val[i] : (ol << 32) | source node id,  ol = offset | (overlap_len << 22) | (small << 31)
 * Phase 2, on the rank that owns the target ids [dst_begin, dst_end): reduce records (any order,
 * every record of an owned target present) to edges.  Slots with d_dst outside the range are ignored. */
int  alga_prefsuf_discover_device(alga_engine *e, const alga_nodes *nodes, const alga_prefsuf_params *p,
                                  int32_t src_begin, int32_t src_end, void *hip_stream,
                                  const uint32_t **d_dst, const uint64_t **d_val, uint64_t *n_records);
int  alga_prefsuf_reduce_device(alga_engine *e, const alga_nodes *nodes, const alga_prefsuf_params *p,
                                const uint32_t *d_dst, const uint64_t *d_val, uint64_t n_records,
                                int32_t dst_begin, int32_t dst_end, void *hip_stream,
                                const alga_edge **d_edges, uint64_t *n_edges);

/* Sharded form without an exchange: the final edges whose SOURCE node id is in [src_begin, src_end), by the source-side
 * reduction (every rank holds the full node set; a source's edges depend on nothing another rank computes).  Returns
 * ALGA_ERR_UNSUPPORTED when that form is not exact for the input (see alga_reduction) -- every rank gets the same answer
 * for the same node set, except for the capacity case (more than 65 536 sources with over 160 raw overlaps each, or one with
 * over 4 096: repeat-rich input), so ranks agree on the
 * fallback with one flag all-reduce.  *d_edges: engine-owned, sorted by (src, dst), valid until the next call on `e`. */
int  alga_prefsuf_build_range_device(alga_engine *e, const alga_nodes *nodes, const alga_prefsuf_params *p,
                                     int32_t src_begin, int32_t src_end, void *hip_stream,
                                     const alga_edge **d_edges, uint64_t *n_edges);

/* Sharding of the build step itself (CLUSTER probe).  Without it every rank computes the minimizer keys of ALL nodes before it
 * probes its own sources -- the part of a build that does not shrink with the rank count.  With it, rank r
 *   1. alga_prefsuf_keys_device(node range of r)   keys + runs of its own nodes (the nodes whose sources it will probe);
 *   2. all-gathers, IN PLACE, the two per-node arrays the call returns (uint32 d_keys[n], d_meta[n]: rank q's slice is
 *      [node_begin_q, node_end_q)) -- 8 bytes per node over RCCL;
 *   3. alga_prefsuf_build_range_device(params.keys_shared = 1, src range inside its node range): sorts the gathered keys into
 *      the bucket order, builds the entry array and probes.
 *   A rank may cut step 3 into pieces (first piece keys_shared = 1, the following ones keys_shared = 2 on consecutive source
 *   sub-ranges) so that the transfer of one piece's edges overlaps the probe of the next.
 *   Up to four ranks the drivers (engine_multi.hip, alga_amd.multigpu) skip steps 1-2: every rank's build (keys_shared = 0, its id range) computes
 *   the target keys of all nodes itself -- the key pass of a build the PILE path keeps makes no run lists, 1.5 ms at the north-star size -- and
 *   probes its range through the piles; the shared key pass pays from five ranks on.
 * out->eligible == 0: the CLUSTER probe does not take this input (every rank gets the same answer for the same node set and
 * options); skip steps 2-3's flag and call the build as before.  The arrays are engine-owned, valid until the next build, and
 * have room for n + ALGA_KEY_ARRAY_SLACK entries, so that equal-sized slices (ceil(n / ranks), the last one running past n) can be
 * gathered in place. */
#define ALGA_KEY_ARRAY_SLACK 1024
typedef struct { uint32_t *d_keys; uint32_t *d_meta; int32_t n; int32_t eligible;
                 int32_t meta_needed; /* 0: every live node has the same length and there is no alignFrom mask -- the build does not read d_meta, it need not be shared */
                 int32_t reserved; } alga_node_keys;
int  alga_prefsuf_keys_device(alga_engine *e, const alga_nodes *nodes, const alga_prefsuf_params *p,
                              int32_t node_begin, int32_t node_end, void *hip_stream, alga_node_keys *out);

/* ---- sharded form with the INDEX sharded by seed bucket (alga_amd/csrc/prefsuf_shard.hip, engine_shard.hip) --------------------
 * The reference's one bucket table per overlap length (src/GraphCreators/GraphCreatorPrefSuf.cpp:247-280,317-332: every thread reads every
 * bucket) divided over N ranks: rank g owns the targets whose minimizer bucket lies in its 1 / N of the bucket space, builds THEIR
 * entries and directory only, receives the run descriptors (12 bytes: cluster key, source id, minimizer position and window range)
 * of every rank's sources that fall into its buckets, and decides the transitive reduction per TARGET from the target's complete
 * candidate list (tests/bucket_side_rule.py: the reference's per-target rule, GraphCreatorPrefSuf.cpp:403-483).  The per-source cap
 * of three small overlaps (:397-401) is the one decision that needs a source's other overlaps: surviving small overlaps are PENDING
 * until the top-3 small keys of their sources have been collected from the bucket owners.  Per build, on every rank r of N:
 *   1. alga_prefsuf_keys_device(own node range [b_r, b_r+1))          keys + runs of the rank's nodes
 *   2. all-gather, in place, of the key array (and meta if meta_needed) -- as for the replicated form above
 *   3. alga_shard_index_device       my slice of the entry array + directory; the descriptors of MY sources, grouped by owner rank
 *   4. all-to-all of the descriptors (12-byte records; counts / offsets per destination from step 3)
 *   5. alga_shard_join_device        verification + per-target reduction in my buckets; the sources of my pending edges
 *   6. all-gather of the pending source ids (u32, variable length)
 *   7. alga_shard_small_keys_device  {source, L, C} (3 x u32) of the small overlaps my descriptors of those sources saw (top 3 per run)
 *   8. all-gather of those lists
 *   9. alga_shard_resolve_device     pending edges that fail their source's cap are dropped; final edges grouped by the rank that owns
 *                                    the SOURCE id (ranges of alga_amd/multigpu.py: shard_chunk -- the same as for the key all-gather)
 *  10. all-to-all of the edges (12-byte alga_edge)
 *  11. alga_shard_place_device       adjacency lists of my source range, (src, dst)-ordered: ready for the gather to rank 0
 * A call answers ALGA_ERR_UNSUPPORTED when the form does not take the input (what the clustered probe declines; a bucket with more than
 * 4096 descriptors): all ranks then take the replicated form.  Every result is engine-owned and valid until the next shard call. */
typedef struct {
    uint64_t targets_owned, descriptors_out, descriptors_in, flagged_sources;   /* index phase / join phase                          */
    uint64_t records, pending, pending_sources, small_keys_out, small_keys_in, dropped;
    uint64_t edges_out, edges_in, edges;
    uint64_t join_passes, join_passes_serial;   /* wave passes of k_shard_join (whole buckets packed onto 64 lanes); those decided target by target */
    double   ms_index, ms_export, ms_sort, ms_join, ms_cap, ms_edges_out, ms_place;   /* device time of each phase (HIP events)  */
} alga_shard_stats;
/* desc_counts[q] descriptors for rank q start at descriptor desc_offsets[q] of *d_desc (3 x uint32 each) */
int  alga_shard_index_device(alga_engine *e, const alga_nodes *nodes, const alga_prefsuf_params *p, int32_t rank, int32_t n_ranks, void *hip_stream,
                             const uint32_t **d_desc, uint64_t *desc_counts /* n_ranks */, uint64_t *desc_offsets /* n_ranks */);
int  alga_shard_join_device(alga_engine *e, const alga_nodes *nodes, const uint32_t *d_desc_in, uint64_t n_desc, void *hip_stream,
                            const uint32_t **d_pending_src, uint64_t *n_pending_src);
int  alga_shard_small_keys_device(alga_engine *e, const uint32_t *d_pending_src_all, uint64_t n_all, void *hip_stream,
                                  const uint32_t **d_small /* 3 x uint32: source, L, C */, uint64_t *n_small);
int  alga_shard_resolve_device(alga_engine *e, const uint32_t *d_small_all, uint64_t n_small_all, void *hip_stream,
                               const alga_edge **d_edges_out, uint64_t *edge_counts /* n_ranks */, uint64_t *edge_offsets /* n_ranks */);
int  alga_shard_place_device(alga_engine *e, const alga_edge *d_edges_in, uint64_t n_in, int32_t src_begin, int32_t src_end, void *hip_stream,
                             const alga_edge **d_edges, uint64_t *n_edges);
int  alga_shard_last_stats(const alga_engine *e, alga_shard_stats *out);

/* ---- the N GPUs of one node behind one handle (alga_amd/csrc/engine_multi.hip) ------------------------------------
 * The reference's parallelism is --threads (src/Params.cpp:237-294; worker threads inside GraphCreatorPrefSuf,
 * src/GraphCreators/GraphCreatorPrefSuf.cpp:150-161); the counterpart: ONE process, one host thread and one engine per GPU.  Rank r
 * computes the minimizer keys of its node range, the key arrays are all-gathered in place (RCCL over xGMI), every rank builds the final
 * edges of its own sources (steps 1-3 above) and the lists are gathered on rank 0's GPU with their exact lengths (grouped
 * ncclSend / ncclRecv) -- the concatenation is the single-GPU byte order.  A rank declining the source-side form makes rank 0 build
 * the whole graph alone: the result never depends on the number of ranks.
 * transport: RCCL (dlopen of librccl.so.1; one GPU per rank) or COPY (hipMemcpyPeerAsync + host barriers; also takes several ranks on
 * ONE device, which is how the driver is tested on a one-GPU box); AUTO = RCCL when every rank has its own GPU and there is more
 * than one, else COPY.  Not yet run on more than one GPU: in particular the variable-length exchanges of the
 * BUCKET_SHARDED form have only ever run over COPY and gloo, never over RCCL.  A post inside an RCCL group that fails on one rank aborts
 * every communicator of the handle (ncclCommAbort: the peers' matching halves would otherwise wait for ever), the build returns an
 * error on all ranks, and the handle refuses further RCCL collectives: destroy it and create a new one. */
typedef struct alga_multi alga_multi; /* opaque */
typedef enum { ALGA_TRANSPORT_AUTO = 0, ALGA_TRANSPORT_RCCL = 1, ALGA_TRANSPORT_COPY = 2 } alga_transport;
/* How the N ranks divide a build (alga_multi_set_option "form"; same graph either way):
 *   REPLICATED      every rank builds the whole bucket-ordered entry array from the all-gathered keys and probes its own source ids
 *                   (round 3: nothing but keys and edges travels, but the index build does not shrink with N);
 *   BUCKET_SHARDED  the index itself is sharded by seed bucket (alga_shard_* above): 1 / N of the entry array per rank, run descriptors
 *                   travel to the bucket's owner, the reduction is decided per target there, edges return to the source's owner;
 *   AUTO            = REPLICATED: measured per rank on one GPU at the north-star size (tools/emulate_shard.py, tools/emulate_rank.py), the
 *                   sharded form costs a rank more device time than the replicated one up to eight ranks.
 *   A build the sharded form declines (ALGA_ERR_UNSUPPORTED on any rank) continues in the replicated form. */
typedef enum { ALGA_MULTI_FORM_AUTO = 0, ALGA_MULTI_FORM_REPLICATED = 1, ALGA_MULTI_FORM_BUCKET_SHARDED = 2 } alga_multi_form;
typedef struct {
    int32_t  n_ranks, transport;          /* alga_transport actually used                                        */
    int32_t  fell_back_to_one_gpu;        /* a rank declined the source-side form: rank 0 built the whole graph  */
    int32_t  form;                        /* alga_multi_form the last build ended in (1 or 2)                     */
    uint64_t edges;
    double   ms_upload, ms_download;      /* host entry point only: node set to every GPU (side by side), edges from the GPUs */
    double   ms_keys, ms_share, ms_build, ms_gather, ms_total;   /* rank 0's host clock: key pass of its nodes, key all-gather, build of its
                                             sources (waits for the slowest rank at its end), gather of the edge lists, all of it */
    /* what rank 0 SENT to other ranks in each exchange of the last build, bytes (the all-gathers: its own slice times N - 1) */
    uint64_t xbytes_keys, xbytes_descriptors, xbytes_pending, xbytes_small_keys, xbytes_edges, xbytes_gather;
    double   ms_shard_index, ms_shard_exchange, ms_shard_join, ms_shard_cap, ms_shard_place;   /* BUCKET_SHARDED, rank 0's host clock per phase */
} alga_multi_stats;
/* "form": alga_multi_form */
int         alga_multi_set_option(alga_multi *m, const char *name, int64_t value);
int         alga_multi_create(const int32_t *hip_devices, int32_t n_ranks, int32_t transport, alga_multi **out);
void        alga_multi_destroy(alga_multi *m);
const char *alga_multi_last_error(const alga_multi *m);
alga_engine *alga_multi_engine(alga_multi *m, int32_t rank);   /* rank's engine: input stage (alga_ingest_device) before, further stages (supplement ...) after, on rank 0's */
/* nodes_per_rank[r]: the node set resident on rank r's GPU (the same nodes on every rank).  *d_edges: the complete graph on rank 0's
 * GPU, owned by the handle, valid until its next build. */
int         alga_multi_prefsuf_build_device(alga_multi *m, const alga_nodes *nodes_per_rank, const alga_prefsuf_params *p,
                                            const alga_edge **d_edges, uint64_t *n_edges);
/* Host buffers in, edges out -- alga_prefsuf_build_host on N GPUs; release *edges with alga_multi_free_edges before alga_multi_destroy. */
int         alga_multi_prefsuf_build_host(alga_multi *m, const alga_nodes *nodes, const alga_prefsuf_params *p, alga_edge **edges, uint64_t *n_edges);
void        alga_multi_free_edges(alga_multi *m, alga_edge *edges);
int         alga_multi_last_stats(const alga_multi *m, alga_multi_stats *out, alga_prefsuf_stats *per_rank /* n_ranks entries, or NULL */);

/* Exchange helpers of the sharded form (device in, device out, engine-owned results):
 *   alga_sort_records_device  orders record slots by target id and drops the padding: the first *n_valid
 *                             entries of the outputs are the records, so the slice that belongs to the rank
 *                             owning targets [a, b) is contiguous (lower_bound of a and b in d_dst_sorted).
 *   alga_sort_edges_device    orders a gathered edge list by (src, dst): the byte order of the single-GPU result. */
int  alga_sort_records_device(alga_engine *e, const uint32_t *d_dst, const uint64_t *d_val, uint64_t n_records,
                              int32_t n_nodes, void *hip_stream, const uint32_t **d_dst_sorted,
                              const uint64_t **d_val_sorted, uint64_t *n_valid);
/* The (u32 key, u32 value) sort of the index build on its own (tests and tools): stable on the key bits [begin_bit, 32).  own != 0: the
 * engine's radix sort (alga_amd/csrc/radix_sort.hip), 0: rocPRIM's.  For n < 2^22 the library path sorts on all 32 bits (its merge-sort path
 * compares the wrong bits for a partial key); the engine's own sort looks at [begin_bit, 32) whatever n is.  The sorted arrays are engine-owned
 * (valid until the next call on e); *ms_best = the fastest of `repeat` runs by HIP events. */
int  alga_sort_u32_pairs_device(alga_engine *e, const uint32_t *d_keys, const uint32_t *d_vals, uint64_t n, int32_t begin_bit, int32_t own, int32_t repeat,
                                void *hip_stream, const uint32_t **d_keys_sorted, const uint32_t **d_vals_sorted, double *ms_best);
/* The (u64 key, u64 value) sort of the supplement's k-mer entries on its own (tests and tools): stable on the key bits [0, bits); own != 0: the
 * engine's radix sort (bits <= 50), 0: rocPRIM's.  Replaces the std::sort of the reference's k-mer buckets (src/GraphCreators/GraphCreatorKmerBased.cpp:94-106). */
int  alga_sort_u64_pairs_device(alga_engine *e, const uint64_t *d_keys, const uint64_t *d_vals, uint64_t n, int32_t bits, int32_t own, int32_t repeat,
                                void *hip_stream, const uint64_t **d_keys_sorted, const uint64_t **d_vals_sorted, double *ms_best);
int  alga_sort_edges_device(alga_engine *e, const alga_edge *d_edges, uint64_t n_edges, int32_t n_nodes,
                            void *hip_stream, const alga_edge **d_sorted);

/* ---- approximate supplement (error_rate > 0.01): GraphCreatorLI ---------------------------------
 * Replaces, for the caller at src/main.cpp:300-347,
 *   new GraphCreatorLI(READS, G); setAlignFrom/To from the degrees of G; startAlignmentGraphCreation();
 *   G->retainOnlySmallestOffset();
 * i.e. four rounds (rotated alphabet priorities, src/GraphCreators/GraphCreatorLI.cpp:18-28) of: LI minimizer
 * k-mers of the tip nodes (src/DataStructures/Read.cpp:145-226), groups of equal k-mer
 * (src/GraphCreators/GraphCreatorKmerBased.cpp:28-136), pairwise join with branch markers
 * (src/GraphCreators/GraphCreatorPairwiseKmerBranch.cpp:16-97) and the mismatch-budget check
 * AlignmentControllerHybrid::canAlign (src/AlignmentControllers/AlignmentControllerLowErrorRate.cpp:15-49).
 * Order dependence: the reference walks the groups of a round one after the other (and races between its threads when
 * --threads > 1) and leaves the order of equal k-mers to std::sort; the engine gives every group the graph as it was
 * when the round started and orders equal k-mers by node id.  DESIGN.md states the measured difference. */
typedef struct {
    int32_t min_overlap_area;   /* Params::MIN_OVERLAP_AREA        = int((1 + SCALE) * avg_len / 2)   (src/main.cpp:333) */
    int32_t max_offset_pct;     /* Params::MAX_OFFSET_CONSIDERED_FOR_ALIGNMENT = int((1 - SCALE) * avg_len / 2), %      */
    int32_t min_identity_pct;   /* Params::MINIMAL_OVERLAP_FOR_LCS_LOW_ERROR = 99 - int(100 * error_rate)             */
    int32_t same_ends;          /* Params::ALIGNMENT_CONTROLLER_SAME_ENDS_LENGTH, 3                                    */
    int32_t li_k;               /* Params::LI_KMER_LENGTH, 35 (src/main.cpp:340)                                       */
    int32_t li_intervals;       /* Params::LI_KMER_INTERVALS, 6 (src/main.cpp:339)                                     */
    int32_t rounds;             /* min(4, Params::LI_PRIORITIES_TO_CONSIDER) = 4                                       */
    int32_t kmer_length_bucket; /* Params::KMER_LENGTH_BUCKET = min(2L/3, 60): shorter reads give no k-mers            */
} alga_pkb_params;

typedef struct {
    uint64_t kmers[4];          /* k-mers per round                                                                    */
    uint64_t groups[4];         /* groups of >= 2 equal k-mers per round                                               */
    uint64_t can_align_calls[4];
    uint64_t edges_after[4];    /* edges in the graph after each round                                                 */
    uint64_t max_group;
    double   ms_total;
    uint64_t group_hist[4][8];  /* per round, groups of 2, 3, 4, 5-7, 8-15, 16-31, 32-64 and > 64 equal k-mers (this rank's)     */
} alga_pkb_stats;

/* (1 + SCALE) / (1 - SCALE) arithmetic of src/main.cpp:332-336 in float, truncating; error_rate as on the command line */
void alga_pkb_derive_params(double avg_len, float scale, double error_rate, int32_t kmer_length_bucket, alga_pkb_params *p);

/* canAlign on n (r1, r2, offset) int32 triples -> n bytes (host buffers in and out) */
int  alga_can_align_batch_host(alga_engine *e, const alga_nodes *nodes, const alga_pkb_params *p, const int32_t *triples,
                               uint64_t n, uint8_t *out);
/* LI k-mers of every node under the alphabet permutation prio[4]: hash[n*li_intervals], ind[n*li_intervals], count[n] */
int  alga_li_kmers_host(alga_engine *e, const alga_nodes *nodes, const alga_pkb_params *p, const int32_t prio[4],
                        uint64_t *hash, int32_t *ind, int32_t *count);
/* The supplement: edges_in = the graph of the exact path (sorted by (src, dst), one edge per pair), result in the same
 * form; host buffers (`nodes` as for alga_prefsuf_build_host).  Release *edges_out with alga_free_edges(). */
int  alga_pkb_supplement_host(alga_engine *e, const alga_nodes *nodes, const alga_pkb_params *p, const alga_edge *edges_in,
                              uint64_t n_edges_in, alga_edge **edges_out, uint64_t *n_edges_out);
/* Same with node set and edge list resident in HBM; the result stays on the device (engine-owned). */
int  alga_pkb_supplement_device(alga_engine *e, const alga_nodes *nodes, const alga_pkb_params *p, const alga_edge *d_edges_in,
                                uint64_t n_edges_in, void *hip_stream, const alga_edge **d_edges_out, uint64_t *n_edges_out);
int  alga_pkb_last_stats(const alga_engine *e, alga_pkb_stats *out);

/* The supplement on N ranks (one engine per GPU; SURVEY.md section 8(e): groups are keyed by k-mer hash, the reference spreads its k-mer buckets
 * over worker threads, src/GraphCreators/GraphCreatorKmerBased.cpp:108-136).  Every rank holds the node set and the COMPLETE exact graph; a group of
 * equal k-mers belongs to rank mix(k-mer key) mod n_ranks.  Per round (params->rounds of them) each rank joins its own groups against the graph as
 * it stood when the round began -> its additions (edge keys src << 36 | dst << 9 | offset, unsorted, engine-owned until the merge); the CALLER
 * brings the additions of all ranks together on every rank (an all-gather of variable length: RCCL, torch.distributed, peer copies) and every rank
 * merges them ALL: the graphs stay identical, and identical to the one-GPU supplement's -- every group of a round sees the round's start graph and
 * the merge orders by key, so nothing depends on how the groups were dealt out (tests/test_gpu_pkb.py: 2, 3 and 5 ranks on one GPU).
 *   begin -> { round -> [exchange] -> merge } x rounds -> end.   alga_pkb_supplement_device is exactly this with one rank.
 * k-mers and their sort are computed by every rank (the pairwise join and canAlign are what is shared out). */
int  alga_pkb_shard_begin(alga_engine *e, const alga_nodes *nodes, const alga_pkb_params *p, const alga_edge *d_edges_in, uint64_t n_edges_in, int32_t rank,
                          int32_t n_ranks, void *hip_stream);
int  alga_pkb_shard_round(alga_engine *e, void *hip_stream, const uint64_t **d_additions, uint64_t *n_additions);
int  alga_pkb_shard_merge(alga_engine *e, const uint64_t *d_all_additions, uint64_t n_all, void *hip_stream);
int  alga_pkb_shard_end(alga_engine *e, void *hip_stream, const alga_edge **d_edges_out, uint64_t *n_edges_out);
/* The approximate supplement (alga_pkb_shard_*) on the handle's N ranks: d_edges_rank0 = the exact graph on rank 0's GPU (what
 * alga_multi_prefsuf_build_device returned); it is sent to every rank, each round's additions are all-gathered over the handle's transport, and
 * rank 0's copy of the result -- identical on all ranks, and to the one-GPU supplement's -- is handed back (engine-owned, rank 0's GPU). */
int         alga_multi_pkb_supplement_device(alga_multi *m, const alga_nodes *nodes_per_rank, const alga_pkb_params *p, const alga_edge *d_edges_rank0,
                                             uint64_t n_edges, const alga_edge **d_edges_out, uint64_t *n_edges_out);

/* ---- input stages (host, multithreaded C++; no GPU involved) ---------------------------------
 * What the reference does between its command line and the GraphCreator constructor, in its
 * --threads=1 order: record parsing, end trimming, N / STR filters, 2-bit packing, reverse-complement
 * twins, pair interleave (src/IO/InputReader.cpp:44-139,272-391), parameter derivation
 * (src/main.cpp:93-115), duplicate / prefix read removal (src/IO/ReadPreprocess.cpp:13-152), id
 * compaction and removal of too-short reads (src/main.cpp:150-232,253-266). */
typedef struct {
    int32_t trim_left, trim_right;   /* Params::READ_END_TRIM_LEFT/RIGHT, default 3 / 3           */
    int32_t remove_reads_with_n;     /* default 1                                                  */
    int32_t rna;                     /* default 0                                                  */
    float   scale;                   /* Params::SCALE, default 0.55                                */
    int32_t min_overlap;             /* -l ; -1 = derive from the mean read length                 */
    int32_t rsoemo;                  /* --rsoemo ; -1 = derive                                     */
    int32_t remove_pref_reads;       /* 1 duplicates, 2 all prefix reads (default), 3 none         */
    int32_t threads;
} alga_ingest_params;

typedef struct {
    int32_t   n, stride_words;       /* stride_words is a multiple of 4 (16-byte aligned rows)     */
    uint32_t *words;                 /* n * stride_words                                           */
    int32_t  *len;                   /* 0 = removed node                                           */
    uint8_t  *pair_off;              /* Global::pairedReadOffset                                   */
    int32_t   LEN, min_overlap, rsoemo, li_kmer_length;
    int64_t   records;
    int32_t   removed_n, removed_str, removed_prefix, removed_short;
    double    avg_len;
} alga_node_set;

void alga_ingest_default_params(alga_ingest_params *p);
int  alga_ingest_files(const char *file1, const char *file2 /* may be NULL */, const alga_ingest_params *p,
                       alga_node_set *out, char *errbuf, size_t errlen);
void alga_free_node_set(alga_node_set *ns);

/* Stage 1 alone: files -> every record's two nodes in the reference's node order, before the removals that depend on other
 * reads (src/IO/InputReader.cpp:44-139,272-391, parameters of src/main.cpp:93-115).  Release with alga_free_parsed_reads. */
typedef struct {
    int64_t   n_nodes;               /* 2 x records                                                                     */
    int32_t   stride_words;
    uint32_t *rows;                  /* n_nodes x stride_words, node 2k = reverse complement, 2k+1 = forward of read k  */
    int32_t  *len;                   /* -1 = removed (N / STR)                                                          */
    int32_t   paired;
    int64_t   records;
    int32_t   removed_n, removed_str;
    int32_t   LEN, min_overlap, rsoemo, li_kmer_length;
    double    avg_len;
    void     *owner;                 /* internal                                                                        */
} alga_parsed_reads;

int  alga_parse_files(const char *file1, const char *file2 /* may be NULL */, const alga_ingest_params *p,
                      alga_parsed_reads *out, char *errbuf, size_t errlen);
void alga_free_parsed_reads(alga_parsed_reads *pr);

/* ---- duplicate / prefix-read removal and id compaction on the GPU ----------------------------
 * The stage between the parser and the GraphCreator constructor (src/IO/ReadPreprocess.cpp:13-152, src/main.cpp:150-232,
 * 253-266): every record's two nodes in the reference's node order come in from the host (as alga_amd/host/ingest.cpp
 * `parse` leaves them; len -1 = removed by the N / STR filters), the surviving node set stays on the device, ready for
 * alga_prefsuf_build_device.  Same result as the host statement of the stage behind alga_ingest_files. */
typedef struct {
    const uint32_t *rows;            /* host: n_nodes rows of stride_words uint32, zero padded, node 2k = reverse complement,
                                        2k+1 = forward strand of read k                                               */
    int32_t         stride_words;
    const int32_t  *len;             /* host: n_nodes lengths, -1 = removed                                             */
    int64_t         n_nodes;         /* even                                                                            */
    int32_t         remove_pref_reads; /* 1 duplicates, 2 all prefix reads (the reference's default), 3 none           */
    int32_t         min_keep_len;    /* nodes shorter than this are emptied (len 0): 3 + li_kmer_length                */
} alga_preprocess_input;

typedef struct {
    const uint32_t *d_words;         /* device, engine-owned until the next alga_preprocess_nodes call on the engine   */
    const int32_t  *d_len;
    const uint8_t  *d_pair_off;      /* Global::pairedReadOffset                                                        */
    int32_t         n, stride_words;
    int32_t         removed_prefix, removed_short, max_len;
    double          ms_device;       /* upload excluded: sort + mark + compaction                                       */
} alga_device_node_set;

int  alga_preprocess_nodes(alga_engine *e, const alga_preprocess_input *in, alga_device_node_set *out);

/* ---- the whole input stage on the GPU -----------------------------------------------------------
 * Files -> node set resident in HBM.  The host maps the files and moves their bytes; everything InputReader does per record
 * (src/IO/InputReader.cpp:142-180,272-391: sequence line, blanks, end trimming, letter check, N / STR filters, 2-bit packing,
 * reverse complement, [rc, r] node order, pair interleave) and the stage behind alga_preprocess_nodes run as HIP kernels.
 * Takes .fasta (two lines per record) and .fastq / .fq (four) with remove_reads_with_n = 1, the reference's default; anything
 * else answers ALGA_ERR_UNSUPPORTED (use alga_parse_files + alga_preprocess_nodes).  Same node set as alga_ingest_files. */
typedef struct {
    int64_t records;                 /* records read (both files)                                                       */
    int32_t removed_n, removed_str;  /* nodes removed by the N / STR filters                                             */
    int32_t LEN, min_overlap, rsoemo, li_kmer_length;   /* src/main.cpp:93-115                                          */
    int32_t paired;
    double  avg_len;
    double  ms_parse, ms_preprocess; /* wall: map + upload + parse kernels; duplicate / prefix removal + compaction     */
    double  ms_upload;               /* part of ms_parse: device buffer + file bytes to HBM                             */
} alga_ingest_info;

int  alga_ingest_device(alga_engine *e, const char *file1, const char *file2 /* may be NULL */, const alga_ingest_params *p,
                        alga_device_node_set *out, alga_ingest_info *info);

/* ---- first step of the graph simplifier ---------------------------------------------------------
 * What the caller does first with the graph on the PrefSuf path (GraphSimplifier::simplifyGraphOld,
 * src/GraphSimplifiers/GraphSimplifier.cpp:90-125): Graph::sortEdgesByIncreasingOffset (src/DataStructures/Graph.cpp:584-614) and
 * GraphSimplifier::cutNonAndWeaklyMetricTriangles (src/GraphSimplifiers/GraphSimplifier.cpp:228-348): an edge i -> b of weight
 * w <= max_offset_parallel_paths (Params::MAX_OFFSET_PARALLEL_PATHS = max(250, int(1.75 * LEN)), src/main.cpp:95) goes when the
 * shortest two-edge path i -> a -> b weighs exactly w.  In: edges grouped by src, lists sorted by (dst, offset) -- what the
 * builds above return.  Out: grouped by src, every list in the order the reference leaves it in (sorted by (offset, dst), then
 * Graph::removeDirectedEdge's swap-with-last removals, src/DataStructures/Graph.cpp:96-119), so a Graph::V filled from it is the
 * reference's graph after that step, entry for entry.  *d_edges_out: engine-owned, valid until the next call of this function. */
int  alga_cut_triangles_device(alga_engine *e, int32_t n_nodes, const alga_edge *d_edges, uint64_t n_edges, int32_t max_offset_parallel_paths,
                               void *hip_stream, const alga_edge **d_edges_out, uint64_t *n_edges_out, uint64_t *n_removed /* may be NULL */);
int  alga_cut_triangles_host(alga_engine *e, int32_t n_nodes, const alga_edge *edges, uint64_t n_edges, int32_t max_offset_parallel_paths,
                             alga_edge **edges_out, uint64_t *n_edges_out);     /* release with alga_free_edges() */

/* ---- contig trimming: the second use of the PrefSuf creator ---------------------------------------
 * src/main.cpp:633-725: the contigs and their reverse complements become the "reads" of one more GraphCreatorPrefSuf run with
 * MIN_OVERLAP_PREF_SUF = REMOVE_SMALL_OVERLAP_EDGES_MIN_OVERLAP = 25 (overlap lengths stop at 501 as always); the longest
 * overlap of an edge between two forward contigs that ends in contig d is cut off the left end of d (:683-712).  This call
 * replaces :636-697: contigs in (2-bit rows as everywhere, any length up to 4 194 303 nt), trim_left[n_contigs] out; the string
 * surgery of :700-712 stays with the caller.  `threshold` = 25 in the reference. */
int  alga_contig_trim_host(alga_engine *e, const uint32_t *words, int32_t stride_words, const int32_t *len, int32_t n_contigs,
                           int32_t threshold, int32_t *trim_left);

/* ---- graph dump: the reference's own checkpoint format ------------------------------------ */
/* Graph::serializeGraph (src/DataStructures/Graph.cpp:269-297): u32 n; n x {i32 id; i32 deg;
 * deg x {i32 neighbour; i32 offset}}, native endian.  Stock ALGA loads it with
 * --deserialize_graph=1 (src/main.cpp:242).  `edges` must be sorted by (src, dst, offset). */
int  alga_write_graph(const char *path, int32_t n_nodes, const alga_edge *edges, uint64_t n_edges);

#ifdef __cplusplus
}
#endif
#endif /* ALGA_AMD_H */
