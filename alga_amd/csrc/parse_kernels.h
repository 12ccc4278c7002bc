// alga_amd/csrc/parse_kernels.h -- launchers of parse_kernels.hip (input stage N2 on the GPU)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace alga {

struct ParseCfg { int32_t trim_left, trim_right, rna; };

uint64_t nl_tiles(uint64_t n_bytes);                       // workgroups (= counters) of the newline passes
void launch_nl_count(const uint8_t *b, uint64_t n, uint32_t *tile_cnt, hipStream_t s);
void launch_nl_write(const uint8_t *b, uint64_t n, const uint32_t *tile_off, unsigned long long *nl_pos, hipStream_t s);
void launch_line_stats(const unsigned long long *nl_pos, uint64_t n_nl, uint64_t n_bytes, uint64_t n_lines, int lines_per_record, uint64_t n_cand,
                       unsigned long long *out /* [0] longest sequence line (zeroed), [1] first empty record (all ones) */, hipStream_t s);
void launch_parse_records(const uint8_t *bytes, const unsigned long long *nl_pos, uint64_t n_nl, uint64_t n_bytes, uint64_t n_lines, int lines_per_record,
                          uint64_t n_rec, int file_index, int paired, const ParseCfg &pc, uint32_t *rows, int W, int32_t *len,
                          unsigned long long *tally /* [0] N, [1] STR, [2] sum len, [3] kept, [4] bad record << 8 | char (all ones) */, hipStream_t s);
void launch_len_stats(const int32_t *len, uint64_t n, unsigned long long *out /* [0] max, [1] live; zeroed */, hipStream_t s);

} // namespace alga
