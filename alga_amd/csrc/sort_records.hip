// alga_amd/csrc/sort_records.hip -- grouping steps of the engine as device radix sorts.
//
// Grouping overlap records by target with atomics (histogram + cursor scatter) runs at the chip's
// random-atomic rate (~20 G/s: 1.1 ms for 22 M records at BASELINE configs[1]); a radix sort moves
// the same records with streaming passes instead.  The sort itself is the vendor primitive
// (rocPRIM, header-only, part of ROCm) -- a plain library sort, like a plain library GEMM; the
// kernels that carry the algorithm of the reference are in prefsuf_kernels.hip.
#include <algorithm>
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "prefsuf_kernels.h"
#include "pkb_kernels.h"
#include "ingest_kernels.h"
#include "prefsuf_shard.h"

namespace alga {

size_t sort_records_temp_bytes(uint64_t n, int bits) {
    size_t bytes = 0;
    (void) rocprim::radix_sort_pairs(nullptr, bytes, (const uint32_t *) nullptr, (uint32_t *) nullptr,
                                     (const unsigned long long *) nullptr, (unsigned long long *) nullptr, (size_t) n, 0u,
                                     (unsigned) bits, (hipStream_t) 0);
    return bytes;
}

hipError_t sort_records(void *temp, size_t temp_bytes, const uint32_t *keys_in, uint32_t *keys_out, const unsigned long long *vals_in,
                        unsigned long long *vals_out, uint64_t n, int bits, hipStream_t s) {
    if (n == 0) return hipSuccess;
    return rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t) n, 0u, (unsigned) bits, s);
}

size_t sort_edges_temp_bytes(uint64_t n) {
    size_t bytes = 0;
    (void) rocprim::radix_sort_pairs(nullptr, bytes, (const unsigned long long *) nullptr, (unsigned long long *) nullptr,
                                     (const uint32_t *) nullptr, (uint32_t *) nullptr, (size_t) n, 0u, 64u, (hipStream_t) 0);
    return bytes;
}

// keys = (src << 32) | dst, values = offset
hipError_t sort_edges(void *temp, size_t temp_bytes, const unsigned long long *keys_in, unsigned long long *keys_out, const uint32_t *vals_in,
                      uint32_t *vals_out, uint64_t n, int src_bits, hipStream_t s) {
    if (n == 0) return hipSuccess;
    return rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t) n, 0u, (unsigned) (32 + src_bits), s);
}

size_t sort_u64_pairs_temp_bytes(uint64_t n, int bits) {
    size_t bytes = 0;
    (void) rocprim::radix_sort_pairs(nullptr, bytes, (const unsigned long long *) nullptr, (unsigned long long *) nullptr,
                                     (const unsigned long long *) nullptr, (unsigned long long *) nullptr, (size_t) n, 0u, (unsigned) bits,
                                     (hipStream_t) 0);
    return bytes;
}

hipError_t sort_u64_pairs(void *temp, size_t temp_bytes, const unsigned long long *keys_in, unsigned long long *keys_out,
                          const unsigned long long *vals_in, unsigned long long *vals_out, uint64_t n, int bits, hipStream_t s) {
    if (n == 0) return hipSuccess;
    return rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t) n, 0u, (unsigned) bits, s);
}

// stable sort of (u32 key, u32 value) on the key bits [begin_bit, 32): targets by minimizer key (prefsuf_cluster.hip).  rocPRIM's
// onesweep sort moves every pair once per pass; its gfx950 default takes 8 bits per pass (1024 threads x 16 items, match ranking).
// 25 .. 30 significant bits -- the north-star index build sorts 29 -- go in THREE passes of 10 bits instead of four of 8: a 10-bit pass
// costs 0.75 ms at 90.6 M pairs against 0.66 ms (tools/micro/sort_bits.hip: 2.26 against 2.65 ms), 9 bits per pass cost the same as
// 8, 11 bits (512 x 16: the counters of 1024 threads no longer fit the LDS) twice as much.
using Sort10 = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                          rocprim::radix_sort_onesweep_config<rocprim::kernel_config<1024, 16>, rocprim::kernel_config<1024, 16>, 10,
                                                                              rocprim::block_radix_rank_algorithm::match>>;
static bool sort_u32_three_wide_passes(int begin_bit) { const int bits = 32 - begin_bit; return bits > 24 && bits <= 30; }

size_t sort_u32_pairs_temp_bytes(uint64_t n) {
    size_t a = 0, b = 0;
    (void) rocprim::radix_sort_pairs(nullptr, a, (const uint32_t *) nullptr, (uint32_t *) nullptr, (const uint32_t *) nullptr,
                                     (uint32_t *) nullptr, (size_t) n, 0u, 32u, (hipStream_t) 0);
    (void) rocprim::radix_sort_pairs<Sort10>(nullptr, b, (const uint32_t *) nullptr, (uint32_t *) nullptr, (const uint32_t *) nullptr,
                                             (uint32_t *) nullptr, (size_t) n, 2u, 32u, (hipStream_t) 0);
    const size_t own = rsort_u32_pairs_temp_bytes(n);
    return std::max(own, a > b ? a : b);
}

hipError_t sort_u32_pairs(void *temp, size_t temp_bytes, const uint32_t *keys_in, uint32_t *keys_out, const uint32_t *vals_in, uint32_t *vals_out,
                          uint64_t n, int begin_bit, hipStream_t s, bool own_sort) {
    if (n == 0) return hipSuccess;
    if (begin_bit < 0 || begin_bit > 31) return hipErrorInvalidValue;
    // round 5: the engine's own radix sort (radix_sort.hip) unless the caller asks for the library's (engine option "own_sort" = 0: A/B and tests)
    if (own_sort) return rsort_u32_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, n, begin_bit, s);
    // Below rocPRIM's merge-sort limit (2^20 pairs by default) a sort on the bits [b, 32) of a 32-bit key goes through
    // radix_merge_compare, whose mask is built as (T(1) << 32) - 1 -- undefined, in practice 0: the comparison then looks at the bits
    // BELOW b only (rocPRIM 4.2.0, device/detail/device_radix_sort.hpp:685).  Small inputs are sorted on all 32 bits; the skipped
    // bits only ever matter for the pass count of the onesweep sort of large ones.
    if (n < (1ull << 22)) begin_bit = 0;
    if (sort_u32_three_wide_passes(begin_bit))
        return rocprim::radix_sort_pairs<Sort10>(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t) n, (unsigned) begin_bit, 32u, s);
    return rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t) n, (unsigned) begin_bit, 32u, s);
}

// run descriptors of the bucket-sharded N-GPU build (prefsuf_shard.hip): (u32 cluster key, u64 {source id, q | p0 | p1}) by bucket =
// the key bits [begin_bit, end_bit) (the bits above are the same for every bucket of one rank's range)
size_t sort_desc_temp_bytes(uint64_t n) { return std::max(sort_records_temp_bytes(n, 32), rsort_u64_pairs_temp_bytes(n)); }
hipError_t sort_desc(void *temp, size_t temp_bytes, const uint32_t *keys_in, uint32_t *keys_out, const unsigned long long *vals_in, unsigned long long *vals_out, uint64_t n,
                     int begin_bit, int end_bit, hipStream_t s, bool own_sort) {
    if (n == 0) return hipSuccess;
    if (begin_bit < 0 || end_bit > 32 || end_bit <= begin_bit) { begin_bit = 0; end_bit = 32; }
    // round 5: the engine's own radix sort (radix_sort.hip: 12-byte records through the 16-byte kernels); the library's on request (option own_sort = 0)
    if (own_sort) return rsort_u32_u64(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, n, begin_bit, end_bit, s);
    if (n < (1ull << 22)) { begin_bit = 0; end_bit = 32; }      // (small inputs: see sort_u32_pairs)
    return rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t) n, (unsigned) begin_bit, (unsigned) end_bit, s);
}

size_t sort_u64_keys_temp_bytes(uint64_t n) {
    size_t bytes = 0;
    (void) rocprim::radix_sort_keys(nullptr, bytes, (const unsigned long long *) nullptr, (unsigned long long *) nullptr, (size_t) n, 0u, 64u,
                                    (hipStream_t) 0);
    return bytes;
}

hipError_t sort_u64_keys(void *temp, size_t temp_bytes, const unsigned long long *keys_in, unsigned long long *keys_out, uint64_t n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    return rocprim::radix_sort_keys(temp, temp_bytes, keys_in, keys_out, (size_t) n, 0u, 64u, s);
}

// (u32 key, u32 value) on the low `bits` key bits: group heads by group size (pkb_kernels.hip)
hipError_t sort_u32_pairs_bits(void *temp, size_t temp_bytes, const uint32_t *keys_in, uint32_t *keys_out, const uint32_t *vals_in, uint32_t *vals_out,
                               uint64_t n, int bits, hipStream_t s) {
    if (n == 0) return hipSuccess;
    return rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t) n, 0u, (unsigned) bits, s);
}

hipError_t sort_u64_keys_bits(void *temp, size_t temp_bytes, const unsigned long long *keys_in, unsigned long long *keys_out, uint64_t n, int bits,
                              hipStream_t s) {
    if (n == 0) return hipSuccess;
    return rocprim::radix_sort_keys(temp, temp_bytes, keys_in, keys_out, (size_t) n, 0u, (unsigned) bits, s);
}

// the supplement's graph (sorted unique edge keys, pkb_kernels.hip): merge of two sorted key lists, and "first key of every
// (src, dst) run" = equality on key >> 9
struct EdgePairEq { __device__ bool operator()(unsigned long long a, unsigned long long b) const { return (a >> 9) == (b >> 9); } };

size_t merge_u64_temp_bytes(uint64_t na, uint64_t nb) {
    size_t bytes = 0;
    (void) rocprim::merge(nullptr, bytes, (const unsigned long long *) nullptr, (const unsigned long long *) nullptr, (unsigned long long *) nullptr,
                          (size_t) na, (size_t) nb, rocprim::less<unsigned long long>(), (hipStream_t) 0);
    return bytes;
}

hipError_t merge_u64(void *temp, size_t temp_bytes, const unsigned long long *a, uint64_t na, const unsigned long long *b, uint64_t nb,
                     unsigned long long *out, hipStream_t s) {
    if (na + nb == 0) return hipSuccess;
    return rocprim::merge(temp, temp_bytes, a, b, out, (size_t) na, (size_t) nb, rocprim::less<unsigned long long>(), s);
}

size_t unique_edge_keys_temp_bytes(uint64_t n) {
    size_t bytes = 0;
    (void) rocprim::unique(nullptr, bytes, (const unsigned long long *) nullptr, (unsigned long long *) nullptr, (unsigned long long *) nullptr, (size_t) n,
                           EdgePairEq(), (hipStream_t) 0);
    return bytes;
}

// *d_count (device) = number of keys kept
hipError_t unique_edge_keys(void *temp, size_t temp_bytes, const unsigned long long *in, unsigned long long *out, unsigned long long *d_count, uint64_t n,
                            hipStream_t s) {
    return rocprim::unique(temp, temp_bytes, in, out, d_count, (size_t) n, EdgePairEq(), s);
}

// stable sort of (u64 key, u32 value) on the low `bits` key bits: the LSD passes of ingest_kernels.hip
size_t sort_u64_u32_temp_bytes(uint64_t n) { return sort_edges_temp_bytes(n); }
hipError_t sort_u64_u32(void *temp, size_t temp_bytes, const unsigned long long *keys_in, unsigned long long *keys_out, const uint32_t *vals_in,
                        uint32_t *vals_out, uint64_t n, int bits, hipStream_t s) {
    if (n == 0) return hipSuccess;
    return rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t) n, 0u, (unsigned) bits, s);
}

} // namespace alga
