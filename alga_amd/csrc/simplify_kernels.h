// alga_amd/csrc/simplify_kernels.h -- launchers of simplify_kernels.hip
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "prefsuf_kernels.h"

namespace alga {
void launch_revcomp_rows(uint32_t *rows /* 2M rows, the first M filled */, int stride, int32_t *len /* 2M */, int32_t M, hipStream_t s);
void launch_trim_left(const alga_edge_dev *e, uint64_t m, const int32_t *len, int32_t M, int32_t *trim /* M, zeroed */, hipStream_t s);
void launch_edge_rowptr(const alga_edge_dev *e, uint64_t m, int32_t n, uint32_t *rowptr /* n + 1 */, hipStream_t s);
void launch_cut_triangles(const alga_edge_dev *e, const uint32_t *rowptr, int32_t n, int32_t max_offset_parallel_paths, alga_edge_dev *sorted /* m */,
                          alga_edge_dev *lst /* m */, uint32_t *out_cnt /* n */, unsigned long long *removed_total, hipStream_t s);
void launch_compact_rows(const alga_edge_dev *work, const uint32_t *rowptr, const uint32_t *out_cnt, const uint32_t *out_rowptr, int32_t n,
                         alga_edge_dev *out, hipStream_t s);
}
