// alga_amd/csrc/ingest_kernels.h -- launchers of ingest_kernels.hip (duplicate / prefix-read removal, id compaction)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace alga {

void launch_pp_iota(uint32_t *perm, uint64_t n, hipStream_t s);
void launch_pp_keys(const uint32_t *rows, int stride, int used_words, const int32_t *len, const uint32_t *perm, uint64_t n, int pass /* -1 = length */,
                    unsigned long long *keys, hipStream_t s);
void launch_pp_mark(const uint32_t *rows, int stride, const int32_t *len, const uint32_t *perm, uint64_t n_live, int mode, uint8_t *mark, hipStream_t s);
void launch_pp_apply(int32_t *len, const uint8_t *mark /* may be null */, uint64_t n_reads, uint32_t *keep,
                     unsigned long long *tally /* [0] removed, [1] twin errors, [2] max len, [3] emptied */, hipStream_t s);
void launch_pp_compact(const uint32_t *rows, int stride_in, const int32_t *len, const uint32_t *keep, const uint32_t *pos, uint64_t n_reads,
                       int min_keep_len, uint32_t *out_rows, int stride_out, int32_t *out_len, uint8_t *out_pair, unsigned long long *tally, hipStream_t s);

size_t     sort_u64_u32_temp_bytes(uint64_t n);
hipError_t sort_u64_u32(void *temp, size_t temp_bytes, const unsigned long long *keys_in, unsigned long long *keys_out, const uint32_t *vals_in,
                        uint32_t *vals_out, uint64_t n, int bits, hipStream_t s);

} // namespace alga
