// alga_amd/csrc/engine_internal.h -- engine state and small helpers shared by engine.hip and engine_pkb.hip
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <algorithm>
#include <string>
#include <vector>
#include <unordered_map>

#include "../../include/alga_amd.h"
#include "prefsuf_common.h"
#include "prefsuf_kernels.h"
#include "pkb_kernels.h"

struct DevBuf {
    void  *p = nullptr;
    size_t cap = 0;
};

enum { EV_START = 0, EV_SEED, EV_PROBE, EV_GROUP, EV_REDUCE, EV_EMIT, EV_PAIRS, EV_KEYS, EV_SORT, EV_GATHER, EV_DIR, EV_COUNT };
constexpr int ALGA_STAGE_THREADS = 8;      // worker threads (pinned buffer pairs, streams) of the staged host <-> HBM copies

struct alga_engine {
    std::vector<DevBuf *> owned;              // the members below that hold an allocation (alga_ensure)
    int         device = -1;
    hipStream_t own_stream = nullptr;
    hipStream_t side_stream = nullptr;         // supplement: the next round's sort beside this round's groups (pkb_presort)
    hipEvent_t  ev_side = nullptr;             // ... its end
    std::string err;
    char        dev_name[256] = {0};
    int         n_cu = 256;
    int         seed_fill_x10 = 20;           // average seed-table bucket fill x10
    // alga_engine_set_option (include/alga_amd.h): every switch that changes how (never what) the engine computes
    int         opt_probe = 0;                // ALGA_PROBE_AUTO / _TABLE / _CLUSTER
    int         opt_cluster_bucket_bias = 0;  // log2 factor on the bucket count of the clustered probe's index
    int         opt_force_per_target = 0;     // AUTO reduction resolves to PER_TARGET
    int         opt_shard_dmax = 4096;        // bucket-sharded join: descriptors of one bucket it takes (option "shard_bucket_max"; tests lower it to see the decline)
    int         opt_test_unsorted_index = 0;  // tests only: skip the sort of the clustered index (the directory pass must flag it, the build must fail)
    hipEvent_t  ev[EV_COUNT] = {};
    // device buffers, grown on demand and kept between calls
    DevBuf table, filter, counters, rowptr, rec_dst, rec_val, keys, seg_key, seg_val, heads, sort_temp, out_cnt, outdeg, out_rowptr, edges, scan_scratch;
    DevBuf cl_defer;                                        // sources the pair kernel hands to the general kernel
    uint32_t h_first_hkey = 255;                            // supplement: sort key of the longest group of a round (255 - min(D, 255))
    // alga_prefsuf_keys_device: the node range whose keys / runs this engine computed last (n < 0: none), consumed by a build with
    // params.keys_shared
    int32_t keyed_n = -1, keyed_begin = 0, keyed_end = 0;
    bool defer_list_valid = false;     // the last discover ran k_probe_stream first: cl_defer / counters[CNT_DEFERRED] list every source whose row came from records
    const void *keyed_words = nullptr;
    // the entry array / index / directory / runs left by the last clustered build (n < 0: none): reusable with params.keys_shared = 2
    int32_t store_n = -1, store_run_begin = 0, store_run_end = 0, store_eq = 0;
    uint32_t store_buckets = 0;
    const void *store_words = nullptr;
    // node statistics of the last prepare() (k_node_stats): reused by the further pieces of a build (params.keys_shared = 2)
    int stat_max_len = 0, stat_min_len = 0; uint64_t stat_live = 0; unsigned long long stat_mask_asym = 0;
    const void *stat_len = nullptr, *stat_from = nullptr, *stat_to = nullptr;
    bool   warmed = false;                                  // alga_engine_reserve has run its miniature build (kernel code objects loaded)
    bool   pairs_timed = false;                             // EV_PAIRS was recorded in the last discovery
    bool   store_timed = false;                             // EV_KEYS / EV_SORT / EV_GATHER were recorded in the last discovery
    bool   pile_keys_only = false, pile_kept_pure = false;  // of the build that made the piles at hand: its key pass made no run lists / it kept the build in the pure pile form (no entry array)
    int    opt_pile_range = 1;                              // option "pile_range": the pile path also for a source id range (a rank's share of the N-GPU build); 0: all sources only, as until round 5
    int    opt_pile_skip_gather = 1;                        // option "pile_skip_gather": no entry array for a build the pile path keeps
    int    opt_pile = 1;                                    // option "pile": the probe through piles (prefsuf_pile.hip) where the input allows it
    int    opt_pile_runs = 1;                               // option "pile_runs": 1 = a pile's run list from its consensus (k_pile_runs_consensus; the key pass of a kept build makes target keys only), 0 = from its outer members' own lists (round 4)
    int    opt_pile_check = 0;                              // option "pile_check" (tests): every node gets its own run list and every first-group member's is compared with its pile's clipped list (stats.pile_list_*)
    bool   expect_pairwise = false;                         // the pile path declined the build before this one: the next key pass makes every run list up front
    DevBuf cl_pile_side_r, cl_pile_cursor;                  // the side records of a source id range (k_pile_side_range), its block cursor
    DevBuf cl_pile_own;                                     // bit j: entry j of the key order reads a run list of its own
    int    opt_pkb_legacy = 0;                              // option "pkb_legacy" (A/B and tests): bit 0 groups of 8 .. 16 k-mers a wave each, bit 1 the library's k-mer sort, bit 2 head list in three kernels, bit 3 groups of 8 .. 16 replayed inside the pair kernel, bit 4 the library's unique + a row-pointer pass after the merge, bit 5 the k-mer walk on a 128-bit value, bit 6 every tip record's snapshot half rewritten every round, bit 7 a k-mer walk per round
    int    opt_own_sort = 1;                                // option "own_sort": the (key, id) sort of the index build is the engine's own radix sort (radix_sort.hip); 0: rocPRIM's
    int    opt_test_presort_oom = 0;                        // tests only: the allocation of the supplement's look-ahead buffers reports out of memory (the rounds must go on one after the other)
    int    opt_test_pile_oom = 0;                           // tests only: the pile path's allocation reports out of memory (the build must continue on the pairwise kernels)
    bool   pile_timed = false;                              // EV_DIR was recorded in the last discovery (k_pile_build ran behind it)
    DevBuf cl_pile_succ;                                    // per entry (16 B): its id, the member of its own pile that starts next to its right, its place in the pile (k_pile_probe reads this, not the entry)
    DevBuf cl_pile_rec2;                                    // run lists of the further k-mer groups of a bucket (64 B at the group's slot, laid out like the second half of a bucket's line)
    DevBuf cl_pile_rec, cl_pile_cnt, cl_pile_tab;   // records of further k-mer groups (64 B per entry slot), group of every entry, {buckets, irregular buckets} of the sample, bucket records (128 B per bucket)
    uint32_t pile_epoch = 0;                                // of the last k_pile_build: what makes a record of cl_pile_tab valid (the table is cleared when it is allocated, and when this wraps)
    int32_t pile_n = -1; const void *pile_words = nullptr;  // the node set the pile records describe (n < 0: none)
    int    opt_cluster_order = 1;                           // option "cluster_order": k_probe_stream walks all sources in entry-array (key) order (0: id order)
    int    opt_cluster_pairs = 1;                           // option "cluster_pairs": 0 = general kernel only, 1 = k_probe_stream first
    DevBuf cl_keys[2], cl_vals[2], cl_meta, cl_runs, cl_nruns, cl_store, cl_dir;   // clustered minimizer join: sort buffers, per-node minimizer runs, entry array, bucket directory
    DevBuf loc_second;                                      // ... the other edge of a two-edge source the pair kernel finished (clustered probe)
    bool   loc_second_used = false;                         // the last discovery wrote loc_second
    uint32_t loc_slot_stride = 0;                           // ... with three slots per source, this many entries apart (0: one slot)
    int    opt_stream_slots = 4;                            // option "stream_slots": standing items of a source k_probe_stream writes to slots itself (4; 2: round 4's form)
    DevBuf loc_first, loc_big_list, loc_big_items;          // source-side form: one-edge slots; second pass over repeat-rich sources
    int    big_limit = -1;                                  // largest per-wave item slice of that pass; -1 = built-in (option "local_big_max": tests)
    DevBuf edge_keys, edge_keys2, edge_vals, edge_vals2, edges_sorted, xs_dst, xs_val;
    DevBuf up_words, up_len, up_from, up_to;   // uploads of the host-buffer entry points
    DevBuf up_len_narrow;                      // ... the lengths as they cross PCIe (one or two bytes per node), widened on the device
    DevBuf cp_deg, cp_dst, cp_off;             // compact edge list (alga_prefsuf_build_host_compact): degree bytes, neighbours, offset bytes
    // approximate supplement (engine_pkb.hip)
    DevBuf cl_defer2;                           // mixed form of a pile-path build: the sources k_probe_stream (list mode) hands on
    DevBuf pk_keys, pk_keys2, pk_vals, pk_vals2, pk_marks, pk_big, pk_add, pk_ekeys, pk_ekeys2, pk_flag, pk_pos, pk_edges[2], pk_rowptr, pk_deg,
           pk_mask, pk_cnt, pk_io, pk_io2, pk_tips, pk_heads, pk_g[2], pk_addk, pk_addk2, pk_merged, pk_hsz, pk_hsz2, pk_heads2, pk_nadd, pk_koff,
           pk_gsz, pk_fixlist, pk_bounds, pk_tiprec, pk_tipidx, pk_keys_all, pk_vals_all;
    // the supplement's look-ahead (engine_pkb.hip: pkb_presort): the NEXT round's sorted entries and group heads, made on side_stream while this round's groups and merge run
    DevBuf pk_keys2b, pk_vals2b, pk_headsb, pk_hszb, sort_temp2, pk_fixlist2, pk_cnt2;
    // duplicate / prefix-read removal (engine_ingest.hip)
    DevBuf pp_rows, pp_len, pp_perm[2], pp_keys[2], pp_mark, pp_keep, pp_pos, pp_out_rows, pp_out_len, pp_out_pair, pp_tally;
    // staged host <-> HBM copies (staging.hip)
    bool        stage_ready = false;
    void       *stage_pin[ALGA_STAGE_THREADS][2] = {};
    hipEvent_t  stage_ev[ALGA_STAGE_THREADS][2] = {};
    hipStream_t stage_stream[ALGA_STAGE_THREADS] = {};
    DevBuf      up_raw;                        // the caller's rows at the caller's stride, before the device re-stride
    // host edge lists handed out by the *_host entry points: capacity of every live one; one released list is kept for the
    // next call (its pages are already mapped: a fresh 1 GB malloc costs more in page faults than the copy into it)
    std::unordered_map<void *, size_t> host_lists;
    void       *host_spare = nullptr;
    size_t      host_spare_cap = 0;
    DevBuf      in_bytes[2], in_nl[2], in_tiles, in_tile_off;   // device ingest: file bytes, line ends, newline counts per tile
    DevBuf      sp_rowptr, sp_sorted, sp_list, sp_cnt, sp_orow, sp_out, sp_in;   // first simplifier step (engine_simplify.hip)
    // seed-bucket-sharded N-GPU build (engine_shard.hip): state between its phases (the exchanges in between are the caller's)
    DevBuf      sh_keys[2], sh_vals[2], sh_store, sh_dir, sh_desc_out, sh_dkey[2], sh_dval[2], sh_small_top, sh_pending, sh_bitmap, sh_small_out,
                sh_ssrc[2], sh_skey[2], sh_edges_out, sh_deg, sh_rowptr, sh_cursor, sh_edges, sh_flagged, sh_cnt, sh_gflag, sh_gpos, sh_gstart;
    struct {
        int      phase = 0;                    // 0 none, 1 indexed, 2 joined, 3 small keys listed, 4 resolved
        int      rank = 0, n_ranks = 1, eq = 0, uniform_len = 0;
        int32_t  n = 0;
        const void *words = nullptr;
        alga::PrefSufCfg cfg{};
        alga::ClusterCfg cc{};
        uint32_t bucket_base = 0, bpr = 0;
        uint64_t n_targets = 0, n_desc = 0, n_rec = 0;
        uint32_t *d_keys_sorted = nullptr;     // descriptors of the last join, sorted by bucket
        unsigned long long *d_vals_sorted = nullptr;
    } sh;
    alga_shard_stats shard_stats{};
    // the approximate supplement between its phases (engine_pkb.hip: begin / round / merge / end)
    struct {
        int      phase = 0;                    // 0 none, 1 between rounds, 2 a round's additions are out, their merge pending
        int      rounds = 0, round = 0, rank = 0, n_ranks = 1, cur = 0, key_bits = 0;
        int32_t  prio[4] = {0, 1, 2, 3};
        uint64_t E = 0, nk = 0;
        uint32_t n_tips = 0;
        int      pre_set = 0, cur_set = 0;     // which of the two sets (engine_pkb.hip: PkbSet) the look-ahead fills / the round at hand works on
        bool     no_look_ahead = false;        // the look-ahead's buffers did not fit: this sequence runs its rounds one after the other
        int      pre_round = -1;               // the round whose sort -> repair -> heads were sent ahead on side_stream (their counts: h_counters + H_PRE), -1 none
        bool     kmers_all = false;            // the k-mer entries of every round are in pk_keys_all / pk_vals_all (made in round 0), kmers_stride entries apart
        size_t   kmers_stride = 0;
        int      kmers_sort_bits = 0;
        alga_nodes dn{};
        alga::PkbCfg cfg{};
    } pkb;
    unsigned long long *h_counters = nullptr;  // pinned, CNT_TOTAL + 2 entries + H_EXTRA more (supplement: the class bounds of a round, 257 x u32)
    static constexpr int H_EXTRA = 132 + 16, H_PRE = alga::CNT_TOTAL + 2 + 132;       // (H_PRE: 13 counts of the look-ahead, in 64-bit words from the block's start)
    uint32_t *h_counters_dev = nullptr;         // the same block by its device address (launch_mail writes it from a kernel)
    uint64_t    rec_cap_hint = 0, rec_cap_hint_local = 0;
    alga_prefsuf_stats stats;
    double      stats_host[2] = {0.0, 0.0};   // upload_nodes_impl: wall ms of its checks / of the upload (the host entry points copy them into stats)
    alga_pkb_stats pkb_stats;
};

// what prepare() (engine.hip) derives from a build's arguments and the node statistics
struct AlgaPrepared {
    alga::NodesDev   nd;
    alga::PrefSufCfg cfg;
    int        max_len = 0;
    int        uniform_len = 0;      // > 0: every live node has this length and there is no alignFrom mask
    uint64_t   live = 0;
    bool       local_ok = false;     // the source-side reduction is exact for this input
    int        local_sw = 1;         // ... with one or two 64-bit words per offset mask / uint4 per overhang
    int        cluster_eq = 0;       // clustered minimizer probe: 16-byte pieces per entry (0 = that probe does not take this input)
    alga::ClusterCfg cluster{};
    int        reduction = ALGA_REDUCTION_AUTO;
    int        keys_shared = 0;      // 1: the per-node keys come from alga_prefsuf_keys_device + the caller's all-gather; 2: the whole
                                     // entry array of the previous build of this node set is reused
};

// the upload of a host node set in two phases (engine.hip: upload_nodes_impl; mode 1 = my slice of the rows + all lengths, 2 = finish once the raw
// buffer is complete): what alga_multi_prefsuf_build_host shards over the ranks
extern "C" int alga_upload_nodes_phase(alga_engine *e, const alga_nodes *nodes, bool twin_rows, int mode, uint64_t row_begin, uint64_t row_end, uint64_t raw_rows_cap, alga_nodes *dev,
                            uint32_t **d_raw);
int alga_prepare(alga_engine *e, const alga_nodes *nodes, const alga_prefsuf_params *p, hipStream_t s, AlgaPrepared &out);
int alga_cluster_alloc(alga_engine *e, const AlgaPrepared &pp);

inline int alga_fail(alga_engine *e, int code, const char *what, hipError_t herr = hipSuccess) {
    char buf[512];
    if (herr != hipSuccess) snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(herr));
    else snprintf(buf, sizeof(buf), "%s", what);
    e->err = buf;
    return code;
}

#define HIP_TRY(e, call)                                                                     \
    do {                                                                                     \
        hipError_t _err = (call);                                                            \
        if (_err != hipSuccess) return alga_fail((e), _err == hipErrorOutOfMemory ? ALGA_ERR_OUT_OF_MEMORY : ALGA_ERR_HIP, #call, _err); \
    } while (0)

// Every device buffer of an engine is allocated here, and remembered: alga_engine_destroy releases what this function handed out
// (a hand-written list of the members missed the buffers later rounds added).
inline int alga_ensure(alga_engine *e, DevBuf &b, size_t bytes) {
    if (bytes == 0) bytes = 16;
    if (b.cap >= bytes) return ALGA_OK;
    if (b.p) { HIP_TRY(e, hipFree(b.p)); b.p = nullptr; b.cap = 0; }
    else if (std::find(e->owned.begin(), e->owned.end(), &b) == e->owned.end()) e->owned.push_back(&b);
    HIP_TRY(e, hipMalloc(&b.p, bytes));
    b.cap = bytes;
    return ALGA_OK;
}

// Engine-owned node storage (uploads, the output of the device input stage) is about to be rewritten: nothing derived from the
// node set that lived there -- key pass, entry array, node statistics -- may be reused by a later build (they are matched on
// addresses and sizes, which a rewrite does not change).
inline void alga_forget_node_set(alga_engine *e) {
    e->keyed_n = -1; e->store_n = -1; e->pile_n = -1;
    e->stat_len = nullptr; e->stat_from = nullptr; e->stat_to = nullptr;
}

inline void alga_release(DevBuf &b) {
    if (b.p) (void) hipFree(b.p);
    b.p = nullptr; b.cap = 0;
}

#include <functional>
typedef std::function<void(void * /* pinned chunk */, size_t /* byte offset of the chunk */, size_t /* bytes */)> AlgaStageFill;
int  alga_staged_h2d(alga_engine *e, void *d_dst, const void *h_src, size_t bytes);   // staging.hip: blocking
int  alga_staged_h2d_fill(alga_engine *e, void *d_dst, size_t bytes, const AlgaStageFill &fill);
int  alga_staged_d2h(alga_engine *e, void *h_dst, const void *d_src, size_t bytes);
void alga_staging_release(alga_engine *e);
void *alga_host_list_take(alga_engine *e, size_t bytes);
void alga_host_list_give(alga_engine *e, void *p);

inline int alga_check_launch(alga_engine *e, const char *what) {
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return alga_fail(e, ALGA_ERR_HIP, what, err);
    return ALGA_OK;
}
