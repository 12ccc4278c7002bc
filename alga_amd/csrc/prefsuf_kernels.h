// alga_amd/csrc/prefsuf_kernels.h -- host-callable launchers of the kernels in prefsuf_kernels.hip
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include "prefsuf_common.h"

namespace alga {

struct alga_edge_dev { int32_t src, dst, offset; }; // layout == alga_edge of include/alga_amd.h

void launch_node_stats(const NodesDev &nd, unsigned long long *counters, int *max_len, hipStream_t s);
void launch_seed_build(const NodesDev &nd, const PrefSufCfg &cfg, unsigned long long *table, uint32_t mask, hipStream_t s);
void launch_probe(const NodesDev &nd, const PrefSufCfg &cfg, const unsigned long long *table, uint32_t mask,
                  int32_t src_begin, int32_t src_end, uint32_t *rec_dst, uint32_t *rec_src, uint32_t *rec_ol, uint64_t rec_cap,
                  uint32_t *indeg, int32_t dst_begin, int32_t dst_end, unsigned long long *counters, int n_cu, hipStream_t s);
uint64_t probe_record_slack(int n_cu, uint64_t n_src);
void launch_count_targets(const uint32_t *rec_dst, uint64_t n_rec, int32_t dst_begin, int32_t dst_end, uint32_t *indeg, hipStream_t s);

size_t   scan_scratch_bytes(uint64_t n);
uint64_t scan_total_index(uint64_t n); // scratch[scan_total_index(n)] holds the 64-bit total after the scan
void launch_exclusive_scan(const uint32_t *in, uint64_t n, uint32_t *out, uint64_t *scratch, hipStream_t s);

void launch_scatter_by_target(const uint32_t *rec_dst, const uint32_t *rec_src, const uint32_t *rec_ol,
                              const unsigned long long *n_rec_ptr, uint64_t n_rec_max, int32_t dst_begin, int32_t dst_end,
                              const uint32_t *rowptr, uint32_t *cursor, uint32_t *seg_src, uint32_t *seg_ol, hipStream_t s);
void launch_reduce_targets(const NodesDev &nd, const PrefSufCfg &cfg, int32_t dst_begin, int32_t n_owned, const uint32_t *rowptr,
                           uint32_t *seg_src, uint32_t *seg_ol, uint32_t *out_cnt, uint32_t *outdeg, unsigned long long *counters, hipStream_t s);
void launch_scatter_by_source(const PrefSufCfg &cfg, int32_t dst_begin, int32_t n_owned, const uint32_t *rowptr, const uint32_t *seg_src,
                              const uint32_t *seg_ol, const uint32_t *out_cnt, const uint32_t *out_rowptr, uint32_t *out_cursor,
                              alga_edge_dev *edges, hipStream_t s);
void launch_sort_rows(int32_t n, const uint32_t *out_rowptr, alga_edge_dev *edges, hipStream_t s);

} // namespace alga
