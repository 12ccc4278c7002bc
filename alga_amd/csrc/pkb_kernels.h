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
void launch_pkb_masks(int32_t n, const uint32_t *rowptr, const alga_edge_dev *edges, uint64_t m, uint32_t *indeg, uint8_t *mask, hipStream_t s);
void launch_pkb_tip_flags(const NodesDev &nd, const PkbCfg &c, const uint8_t *mask, uint32_t *flag, hipStream_t s);
void launch_pkb_tip_list(int32_t n, const uint32_t *flag, const uint32_t *pos, uint32_t *tips, hipStream_t s);
void launch_pkb_kmers(const NodesDev &nd, const PkbCfg &c, const int32_t prio[4], const uint32_t *tips, uint32_t n_tips, unsigned long long *keys,
                      unsigned long long *vals, unsigned long long *counter, hipStream_t s);
void launch_pkb_group_sizes(const unsigned long long *keys, uint64_t n, unsigned long long *big_words, unsigned long long *stats, uint32_t *head_flag,
                            hipStream_t s);
void launch_pkb_head_list(const uint32_t *head_flag, const uint32_t *pos, uint64_t n, uint32_t *heads, hipStream_t s);
void launch_pkb_groups(const NodesDev &nd, const PkbCfg &c, const uint32_t *rowptr, const alga_edge_dev *edges, const unsigned long long *keys, const uint32_t *heads, uint32_t n_heads,
                       unsigned long long *vals, uint64_t n, unsigned long long *marks, unsigned long long *big_marks,
                       unsigned long long *big_cursor, alga_edge_dev *add_edges, uint64_t add_dense, uint64_t add_cap,
                       unsigned long long *add_overflow, unsigned long long *counters, hipStream_t s);
void launch_pkb_valid_flags(const alga_edge_dev *e, uint64_t n, uint32_t *flag, hipStream_t s);
void launch_pkb_edge_keys_dense(const alga_edge_dev *e, const uint32_t *flag, const uint32_t *pos, uint64_t n, unsigned long long *keys,
                                unsigned long long *bad, hipStream_t s);
void launch_pkb_unique_flags(const unsigned long long *keys, uint64_t n, uint32_t *flag, hipStream_t s);
void launch_pkb_compact(const unsigned long long *keys, const uint32_t *flag, const uint32_t *pos, uint64_t n, alga_edge_dev *out,
                        uint32_t *outdeg, hipStream_t s);

// sort_records.hip
size_t     sort_u64_keys_temp_bytes(uint64_t n);
hipError_t sort_u64_keys(void *temp, size_t temp_bytes, const unsigned long long *keys_in, unsigned long long *keys_out, uint64_t n, hipStream_t s);

} // namespace alga
