// alga_amd/csrc/pkb_kernels.h -- launchers of the approximate-supplement kernels (pkb_kernels.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "prefsuf_common.h"
#include "prefsuf_kernels.h"

namespace alga {

constexpr int PKB_MAX_INTERVALS = 16;

struct PkbCfg {
    int32_t min_overlap_area;     // Params::MIN_OVERLAP_AREA
    int32_t max_offset_pct;       // Params::MAX_OFFSET_CONSIDERED_FOR_ALIGNMENT (% of |r1|)
    int32_t min_identity_pct;     // Params::MINIMAL_OVERLAP_FOR_LCS_LOW_ERROR
    int32_t same_ends;            // Params::ALIGNMENT_CONTROLLER_SAME_ENDS_LENGTH
    int32_t li_k, li_intervals;   // Params::LI_KMER_LENGTH / LI_KMER_INTERVALS
    int32_t kmer_length_bucket;   // Params::KMER_LENGTH_BUCKET (reads shorter than this give no k-mers)
};

void launch_can_align_batch(const NodesDev &nd, const PkbCfg &c, const int32_t *triples, uint64_t n, uint8_t *out, hipStream_t s);
void launch_li_kmers_slots(const NodesDev &nd, const PkbCfg &c, const int32_t prio[4], uint64_t *hash, int32_t *ind, int32_t *count, hipStream_t s);
void launch_pkb_edge_keys(const alga_edge_dev *e, uint64_t n, unsigned long long *keys, unsigned long long *counts /* [0] bad offsets, [1] unsorted */, hipStream_t s);
void launch_pkb_rowptr(const unsigned long long *keys, uint64_t E, int32_t n, uint32_t *rowptr, hipStream_t s);
// the additions in key order: src << shift as a 32-bit sort key; then the keys by sorted position with every run of one src put in order
void launch_pkb_src_keys(const unsigned long long *keys, uint64_t n, int shift, uint32_t *k32, hipStream_t s);
void launch_pkb_gather_sorted_runs(const unsigned long long *keys, const uint32_t *k32_sorted, const uint32_t *idx, uint64_t n, unsigned long long *out, hipStream_t s);
// first key per (src, dst) of a sorted key list: flags, then (after an exclusive scan of them into pos) the kept keys and the row pointers of the result
void launch_pkb_unique_flags(const unsigned long long *in, uint64_t n, uint32_t *flag, hipStream_t s);
void launch_pkb_unique_scatter(const unsigned long long *in, uint64_t n, const uint32_t *flag, const uint32_t *pos, int32_t n_nodes, unsigned long long *out,
                               uint32_t *rowptr, hipStream_t s);
void launch_pkb_keys_to_edges(const unsigned long long *keys, uint64_t E, alga_edge_dev *out, hipStream_t s);
void launch_pkb_masks(int32_t n, const uint32_t *rowptr, const unsigned long long *keys, uint64_t m, uint32_t *indeg, uint8_t *mask, hipStream_t s);
void launch_pkb_tip_flags(const NodesDev &nd, const PkbCfg &c, const uint8_t *mask, uint32_t *flag, hipStream_t s);
void launch_pkb_tip_list(const NodesDev &nd, const PkbCfg &c, const uint32_t *flag, const uint32_t *pos, uint32_t *tips, uint32_t *kcount,
                         unsigned long long *max_len, uint32_t *tipidx /* node -> place in tips[], ~0 if none */, hipStream_t s);
void launch_pkb_kmers(const NodesDev &nd, const PkbCfg &c, const int32_t prio[4], const uint32_t *tips, const uint32_t *koff, uint32_t n_tips, int sort_bits,
                      unsigned long long *keys, unsigned long long *vals, const void *tiprec, int wide /* 1: round 4's 128-bit rolling value */, hipStream_t s);
bool launch_pkb_kmers_all(const NodesDev &nd, const PkbCfg &c, const int32_t prio[4] /* of round 0 */, int first, int count, const uint32_t *tips, const uint32_t *koff, uint32_t n_tips,
                          int sort_bits, unsigned long long *keys, unsigned long long *vals, size_t round_stride, const void *tiprec, hipStream_t s);
// tip records (pkb_kernels.hip: PkbTipRec): row + id + first snapshot keys of every node that takes part, 128 bytes each
size_t pkb_tiprec_bytes(uint32_t n_tips);
void launch_pkb_tiprec_rows(const NodesDev &nd, const uint32_t *tips, uint32_t n_tips, void *tiprec, hipStream_t s);
void launch_pkb_tiprec_snap_srcs(const unsigned long long *sorted_adds, uint64_t n_adds, const uint32_t *tipidx, const uint32_t *rowptr, const unsigned long long *gkeys,
                                 void *tiprec, hipStream_t s);
void launch_pkb_tiprec_snap(const uint32_t *tips, uint32_t n_tips, const uint32_t *rowptr, const unsigned long long *gkeys, void *tiprec, hipStream_t s);
void launch_pkb_fix_runs(unsigned long long *keys, unsigned long long *vals, uint64_t n, int bits, uint32_t *list, uint32_t list_cap, unsigned long long *counter,
                         hipStream_t s);
void launch_pkb_fix_runs_loop(unsigned long long *keys, unsigned long long *vals, uint64_t n, int bits, hipStream_t s);
void launch_pkb_group_sizes(const unsigned long long *keys, uint64_t n, unsigned long long *big_words, unsigned long long *max_d, uint32_t *head_flag,
                            uint32_t *gsize, int rank, int n_ranks /* the groups of rank mix(key) mod n_ranks alone (1: all) */, hipStream_t s);
void launch_pkb_head_list(const uint32_t *head_flag, const uint32_t *pos, const uint32_t *gsize, uint64_t n, uint32_t *heads, uint32_t *hkey, hipStream_t s);
// group_sizes + scan + head_list in one pass (the list in no particular order; *n_heads: zeroed by the caller, the number of groups afterwards)
void launch_pkb_heads(const unsigned long long *keys, uint64_t n, unsigned long long *big_words, unsigned long long *max_d, unsigned long long *n_heads,
                      uint32_t *heads, uint32_t *hkey, int rank, int n_ranks, hipStream_t s);
void launch_pkb_class_bounds(const uint32_t *hkey_sorted, uint32_t n_heads, uint32_t *bound /* 257 entries */, hipStream_t s);
void launch_pkb_groups(const NodesDev &nd, const PkbCfg &c, const uint32_t *rowptr, const unsigned long long *gkeys, const unsigned long long *keys,
                       const uint32_t *heads, const uint32_t *hkey, const uint32_t *bound /* launch_pkb_class_bounds */, uint32_t n_heads, unsigned long long *vals, uint64_t n,
                       unsigned long long *marks, unsigned long long *big_marks, unsigned long long *big_cursor, unsigned long long *add_keys, uint64_t add_dense,
                       uint64_t add_cap, unsigned long long *add_overflow, unsigned long long *counters, uint32_t *n_add, uint32_t *left, int n_cu,
                       const uint32_t *tips, const void *tiprec, int legacy /* bit 0: groups of 8 .. 16 through the wave kernel (round 4's form); bit 3: their replay inside the pair kernel */, hipStream_t s);
void launch_pkb_gather_adds(const uint32_t *heads, const uint32_t *n_add, const uint32_t *pos, uint32_t n_heads, const unsigned long long *add_keys,
                            uint64_t add_dense, uint64_t n_dense_total, uint64_t n_ovf, unsigned long long *out, hipStream_t s);

// sort_records.hip
size_t     sort_u64_keys_temp_bytes(uint64_t n);
hipError_t sort_u64_keys(void *temp, size_t temp_bytes, const unsigned long long *keys_in, unsigned long long *keys_out, uint64_t n, hipStream_t s);
hipError_t sort_u64_keys_bits(void *temp, size_t temp_bytes, const unsigned long long *keys_in, unsigned long long *keys_out, uint64_t n, int bits, hipStream_t s);
hipError_t sort_u32_pairs_bits(void *temp, size_t temp_bytes, const uint32_t *keys_in, uint32_t *keys_out, const uint32_t *vals_in, uint32_t *vals_out,
                               uint64_t n, int bits, hipStream_t s);
size_t     merge_u64_temp_bytes(uint64_t na, uint64_t nb);
hipError_t merge_u64(void *temp, size_t temp_bytes, const unsigned long long *a, uint64_t na, const unsigned long long *b, uint64_t nb, unsigned long long *out,
                     hipStream_t s);
size_t     unique_edge_keys_temp_bytes(uint64_t n);
hipError_t unique_edge_keys(void *temp, size_t temp_bytes, const unsigned long long *in, unsigned long long *out, unsigned long long *d_count, uint64_t n,
                            hipStream_t s);

} // namespace alga
