// alga_amd/csrc/prefsuf_kernels.h -- host-callable launchers of the kernels in prefsuf_kernels.hip / sort_records.hip
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include "prefsuf_common.h"

namespace alga {

struct alga_edge_dev { int32_t src, dst, offset; }; // layout == alga_edge of include/alga_amd.h

void launch_restride(const uint32_t *in, int stride_in, uint32_t *out, int stride_out, uint64_t n, hipStream_t s);
void launch_widen_len(const void *in, int elem_bytes /* 1 or 2 */, int32_t *out, uint64_t n, hipStream_t s);
void launch_compact_edges(const alga_edge_dev *edges, int32_t n, uint64_t n_edges, uint8_t *deg, uint32_t *dst, uint8_t *off, unsigned long long *bad, hipStream_t s);
// in: rows of the odd nodes (n_pairs rows); out: 2 n_pairs rows, the even ones = reverse complements over len[2k] nucleotides
void launch_expand_twins(const uint32_t *in, int stride_in, const int32_t *len /* device, 2 n_pairs */, uint32_t *out, int stride_out, uint64_t n_pairs, hipStream_t s);
void launch_node_stats(const NodesDev &nd, unsigned long long *counters, int *max_len /* [0] = max length, [1] = INT32_MAX - min length of the live nodes */, hipStream_t s);

uint32_t seed_buckets_for(uint64_t live_nodes, int fill_x10);
uint32_t seed_filter_bits_for(uint64_t live_nodes);   // 0 = prefilter off
void launch_seed_build(const NodesDev &nd, const PrefSufCfg &cfg, unsigned long long *table, uint32_t n_buckets, uint32_t *filter,
                       uint32_t filter_bits, hipStream_t s);

// source-side form, sources with more raw overlaps than a wave's LDS holds: the first pass appends them to `list` (counter
// CNT_LOCAL_OVERFLOW), the second pass (count > 0) probes exactly those with `item_cap` items per wave in `items`
struct ProbeBig { int32_t *list; uint32_t list_cap; uint32_t count; void *items; uint32_t item_cap; };
int    local_item_capacity();    // items of one source a wave holds in LDS
int    local_big_limit();        // largest per-wave item slice the engine allocates for the second pass
size_t probe_big_bytes(int n_cu, uint32_t count, int local, uint32_t item_cap);

// overlap record = rec_dst[i] (target id, REC_INVALID for chunk padding) + rec_val[i] ((ol << 32) | source id)
void launch_probe(const NodesDev &nd, const PrefSufCfg &cfg, const unsigned long long *table, uint32_t n_buckets,
                  const uint32_t *filter, uint32_t filter_bits, int32_t src_begin, int32_t src_end, uint32_t *rec_dst, unsigned long long *rec_val, uint64_t rec_cap,
                  unsigned long long *counters, int n_cu, int local /* 0, or the mask width 1 | 2 of the source-side form */, uint32_t *deg /* local: zeroed, n_src */, unsigned long long *first /* local: n_src */,
                  const ProbeBig *big /* local: may be null */, hipStream_t s);
uint64_t probe_record_slack(int n_cu, uint64_t n_src, bool local);
// source-side reduction: adjacency lists from the probe's out-degrees (rowptr = their scan), one-edge slots and record list
// record_sources (device, may be null: every row is sorted): the ids of the only sources whose rows were filled from records, their number at
// *record_sources_count (device), at most record_sources_cap
void launch_local_emit(int32_t src_base, int32_t n_src, const uint32_t *deg, const unsigned long long *first, const unsigned long long *second,
                       const uint32_t *rec_dst, const unsigned long long *rec_val, uint64_t n_rec, const uint32_t *rowptr, uint32_t *cursor,
                       alga_edge_dev *edges, const int32_t *record_sources, const unsigned long long *record_sources_count, uint32_t record_sources_cap, hipStream_t s, uint32_t slot_stride = 0);

// clustered minimizer join (prefsuf_cluster.hip): source-side form with one-word offset masks (max_len - Lmin <= 63)
bool       cluster_plan(const PrefSufCfg &cfg, int max_len, uint64_t live, int bucket_log2_bias, ClusterCfg *c, int *eq);   // false: this probe does not take the input
size_t     cluster_sort_temp_bytes(uint64_t n);
// keys/vals/keys2/vals2/meta: n uint32 each; runs: n * CL_RMAX * 8 bytes; nruns: n bytes; store: (n + 2) * 16 * eq bytes; dir: (n_buckets + 2) * 16 bytes
void       launch_cluster_keys(const NodesDev &nd, const PrefSufCfg &cfg, const ClusterCfg &cc, int32_t node_begin, int32_t node_end, uint32_t *keys,
                               uint32_t *vals, uint32_t *meta, void *runs, uint8_t *nruns, hipStream_t s, bool with_runs = true /* false: keys and meta words alone */,
                               const unsigned long long *only_if_declined = nullptr /* the pile path's sample counters: the kernel leaves for a build that path keeps */);
void       launch_cluster_runs_list(const NodesDev &nd, const PrefSufCfg &cfg, const ClusterCfg &cc, const int32_t *ids, const unsigned long long *count, uint32_t cap,
                                    uint32_t grid_blocks, void *runs, uint8_t *nruns, hipStream_t s);      // run lists of the listed nodes (list length read on the device)
hipError_t launch_cluster_store(const NodesDev &nd, const ClusterCfg &cc, uint32_t *keys, uint32_t *vals, uint32_t *keys2 /* out: sorted */, uint32_t *vals2, void *sort_temp,
                                size_t sort_temp_bytes, void *dir, bool fill_vals, hipEvent_t ev_sorted /* may be null */,
                                unsigned long long *bad_flag /* device: set when the sorted keys are not in order */,
                                bool test_skip_sort /* tests: build the index over UNSORTED keys */, hipStream_t s, bool own_sort = true);      // sort + directory
hipError_t launch_cluster_gather(const NodesDev &nd, const ClusterCfg &cc, int eq, const uint32_t *keys2, const uint32_t *vals2, const uint32_t *meta,
                                 int uniform_len /* > 0: all live nodes have this length, no alignFrom mask */, void *store,
                                 const unsigned long long *pile_cnt /* null, or the pile path's sample counters: no entry array for a build it keeps */, hipStream_t s);
uint64_t   cluster_record_slack(int n_cu, uint64_t n_src);
void       launch_probe_clustered(const NodesDev &nd, const PrefSufCfg &cfg, const ClusterCfg &cc, int eq, const void *store, const void *dir,
                                  const void *runs, const uint8_t *nruns, int32_t src_begin, int32_t src_end, const int32_t *src_list, int32_t src_base, uint32_t *rec_dst, unsigned long long *rec_val, uint64_t rec_cap,
                                  unsigned long long *counters, int n_cu, uint32_t *deg, unsigned long long *first, const ProbeBig *big,
                                  const unsigned long long *list_count /* device, may be null: list mode ends at min(src_end, *list_count) */,
                                  int sw /* 1 | 2: 64-bit words per offset mask (sources of up to 64 | 128 suffix windows) */, hipStream_t s,
                                  const uint32_t *skeys = nullptr, const uint32_t *sids = nullptr, int uniform_len = 0,
                                  const unsigned long long *pile_cnt = nullptr /* the pile path's sample: the build may have no entry array -- then the sorted (key, id) pairs stand in for it */);
void       launch_probe_stream(const NodesDev &nd, const PrefSufCfg &cfg, const ClusterCfg &cc, int eq, const void *store, const void *dir,
                               const void *runs, const uint8_t *nruns, int32_t src_begin, int32_t src_end, bool by_key /* the range is one of entry-array positions */,
                               unsigned long long *counters, int n_cu, uint32_t *deg, unsigned long long *first, unsigned long long *second, int32_t *defer_list,
                               uint32_t defer_cap, const unsigned long long *pile_cnt /* null, or the pile kernel's two counters: leave at once where that kernel works */,
                               hipStream_t s, uint32_t slot_stride = 0 /* != 0: second[] holds three slots per source, slot_stride entries apart */);
// the probe through piles (prefsuf_pile.hip)
void       launch_pile_deg(int32_t n, unsigned long long *first, uint32_t *deg, const unsigned long long *pile_cnt, hipStream_t s);
bool       pile_plan(const PrefSufCfg &cfg, const ClusterCfg &cc, int eq, int uniform_len, bool masks);
size_t     pile_record_bytes(uint64_t n);
size_t     pile_table_bytes(uint32_t n_buckets);
void       launch_pile_sample(const NodesDev &nd, const ClusterCfg &cc, int uniform_len, const uint32_t *skeys /* sorted keys */, const uint32_t *sids /* their node ids */, const void *dir,
                              unsigned long long *pile_cnt /* of the sample: [0] buckets (raised to entries / 8), [1] irregular buckets, [2] entries */, int no_sample, hipStream_t s);
constexpr int PILE_CNT_WORDS = 6;      // {sampled buckets, irregular ones, sampled entries, own-list ids, members checked, members whose lists differ}
size_t     pile_own_mask_bytes(uint64_t n);
void       launch_pile_build(const NodesDev &nd, const PrefSufCfg &cfg, const ClusterCfg &cc, int uniform_len, const uint32_t *skeys, const uint32_t *sids, const void *dir, void *rec,
                             void *rec2 /* run lists of the further groups, at the groups' slots */, void *tab, uint32_t epoch, void *side, const void *runs, int nwin, const unsigned long long *pile_cnt,
                             uint32_t *own_mask /* null: the piles' run lists from their outer members' own lists (round 4) */, hipStream_t s);
void       launch_pile_own_ids(const uint32_t *own_mask, const uint32_t *sids, uint64_t n_entries, int32_t *list, uint32_t cap, unsigned long long *pile_cnt, hipStream_t s);
void       launch_pile_check(const void *side, uint64_t n_entries, uint32_t n_buckets, const void *tab, const void *rec2, uint32_t epoch, const void *runs, int n_nodes, int nwin,
                             unsigned long long *pile_cnt, hipStream_t s);
// mixed form of a pile-path build: k_probe_stream over the sources on src_list (count on the device: counters[CNT_DEFERRED]), rejects to defer2, then the swap
void       launch_probe_stream_list(const NodesDev &nd, const PrefSufCfg &cfg, const ClusterCfg &cc, int eq, const void *store, const void *dir, const void *runs, const uint8_t *nruns,
                                    int32_t *src_list, uint32_t list_cap, unsigned long long *counters, int n_cu, uint32_t *deg, unsigned long long *first,
                                    unsigned long long *second, int32_t *defer2, const unsigned long long *pile_cnt, hipStream_t s, uint32_t slot_stride = 0,
                                    int32_t src_base = 0 /* the slot arrays count from this id (a rank's range) */);
void       launch_pile_probe(const NodesDev &nd, const PrefSufCfg &cfg, const ClusterCfg &cc, int uniform_len, const void *tab, uint32_t epoch, const void *rec, const void *rec2, const void *side,
                             const void *runs, unsigned long long *counters, uint32_t *deg, unsigned long long *first, unsigned long long *second, int32_t *defer_list,
                             uint32_t defer_cap, const unsigned long long *pile_cnt, int n_cu, hipStream_t s, int32_t src_begin, int32_t src_end,
                             void *side_range /* a range that is not all nodes: (src_end - src_begin + 64) * 16 B of scratch */, unsigned long long *cursor /* ... and a word */);

// radix_sort.hip: the engine's own stable LSD radix sort of (u32 key, u32 value) pairs on the key bits [begin_bit, 32)
size_t     rsort_u32_pairs_temp_bytes(uint64_t n);
void       rsort_set_variant(int v);                       // tuning only
hipError_t rsort_u32_pairs(void *temp, size_t temp_bytes, const uint32_t *keys_in, uint32_t *keys_out, const uint32_t *vals_in, uint32_t *vals_out, uint64_t n,
                           int begin_bit, hipStream_t s);
size_t     sort_u32_pairs_temp_bytes(uint64_t n);
hipError_t sort_u32_pairs(void *temp, size_t temp_bytes, const uint32_t *keys_in, uint32_t *keys_out, const uint32_t *vals_in, uint32_t *vals_out,
                          uint64_t n, int begin_bit /* the bits below it are not looked at */, hipStream_t s, bool own_sort = true /* radix_sort.hip; false: rocPRIM */);
size_t     rsort_u64_pairs_temp_bytes(uint64_t n);
// radix_sort.hip: stable sort of (u64 key, u64 value) records on the key bits [0, bits), bits <= 50 (the supplement's k-mer entries)
hipError_t rsort_u64_pairs(void *temp, size_t temp_bytes, const unsigned long long *keys_in, unsigned long long *keys_out, const unsigned long long *vals_in,
                           unsigned long long *vals_out, uint64_t n, int bits, hipStream_t s);
hipError_t rsort_u32_u64(void *temp, size_t temp_bytes, const uint32_t *keys_in, uint32_t *keys_out, const unsigned long long *vals_in, unsigned long long *vals_out,
                         uint64_t n, int begin_bit, int end_bit, hipStream_t s);          // temp: rsort_u64_pairs_temp_bytes(n)
size_t     sort_u64_pairs_temp_bytes(uint64_t n, int bits);
hipError_t sort_u64_pairs(void *temp, size_t temp_bytes, const unsigned long long *keys_in, unsigned long long *keys_out,
                          const unsigned long long *vals_in, unsigned long long *vals_out, uint64_t n, int bits, hipStream_t s);

void launch_make_keys(const uint32_t *rec_dst, uint64_t n_rec, int32_t dst_begin, int32_t dst_end, uint32_t *keys,
                      unsigned long long *n_valid, hipStream_t s);
size_t     sort_records_temp_bytes(uint64_t n, int bits);
hipError_t sort_records(void *temp, size_t temp_bytes, const uint32_t *keys_in, uint32_t *keys_out, const unsigned long long *vals_in,
                        unsigned long long *vals_out, uint64_t n, int bits, hipStream_t s);
void launch_rowptr_from_sorted(const uint32_t *keys, const unsigned long long *n_valid_ptr, uint64_t n_rec_max, int32_t n_owned,
                               uint32_t *rowptr, hipStream_t s);

size_t   scan_scratch_bytes(uint64_t n);
uint64_t scan_total_index(uint64_t n); // scratch[scan_total_index(n)] holds the 64-bit total after the scan
void launch_exclusive_scan(const uint32_t *in, uint64_t n, uint32_t *out, uint64_t *scratch, hipStream_t s);
// up to MAIL_SEGS pieces of device memory (32-bit words) -> a pinned host block through its device address (engine: h_counters), one kernel
constexpr int MAIL_SEGS = 6;
struct MailSeg { const uint32_t *src; uint32_t words, dst; };            // dst: word offset in the host block
struct MailArgs {
    MailSeg seg[MAIL_SEGS];
    int n = 0;
    void add(const void *src, uint32_t words, uint32_t dst_word) { seg[n].src = (const uint32_t *) src; seg[n].words = words; seg[n].dst = dst_word; n++; }
};
void launch_mail(const MailArgs &a, uint32_t *host_block_dev, hipStream_t s);

int  reduce_targets_per_block(uint64_t n_records, uint64_t n_targets);
void launch_gather_heads(const NodesDev &nd, const unsigned long long *seg_val, const unsigned long long *n_valid_ptr, uint64_t n_rec_max,
                         void *heads /* 16 B per record */, hipStream_t s);
void launch_reduce_targets(const NodesDev &nd, const PrefSufCfg &cfg, int32_t dst_begin, int32_t n_owned, int32_t targets_per_block,
                           const uint32_t *rowptr, unsigned long long *seg_val, const void *heads, uint32_t *out_cnt, uint32_t *outdeg,
                           unsigned long long *counters, hipStream_t s);
void launch_scatter_by_source(const PrefSufCfg &cfg, int32_t dst_begin, int32_t n_owned, const uint32_t *rowptr,
                              const unsigned long long *seg_val, const uint32_t *out_cnt, const uint32_t *out_rowptr, uint32_t *out_cursor,
                              alga_edge_dev *edges, hipStream_t s);
void launch_sort_rows(int32_t n, const uint32_t *out_rowptr, alga_edge_dev *edges, hipStream_t s);

void launch_edges_to_keys(const alga_edge_dev *e, uint64_t n, unsigned long long *keys, uint32_t *vals, hipStream_t s);
void launch_keys_to_edges(const unsigned long long *keys, const uint32_t *vals, uint64_t n, alga_edge_dev *e, hipStream_t s);
size_t     sort_edges_temp_bytes(uint64_t n);
hipError_t sort_edges(void *temp, size_t temp_bytes, const unsigned long long *keys_in, unsigned long long *keys_out, const uint32_t *vals_in,
                      uint32_t *vals_out, uint64_t n, int src_bits, hipStream_t s);

} // namespace alga
