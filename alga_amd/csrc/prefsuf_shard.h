// alga_amd/csrc/prefsuf_shard.h -- host-callable launchers of prefsuf_shard.hip (the seed-bucket-sharded N-GPU build)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "prefsuf_common.h"
#include "prefsuf_kernels.h"

namespace alga {

// buckets per rank: rank g owns the buckets [g * bpr, min((g + 1) * bpr, n_buckets))
inline uint32_t shard_buckets_per_rank(uint32_t n_buckets, int n_ranks) { return (uint32_t) (((uint64_t) n_buckets + (uint64_t) n_ranks - 1) / (uint64_t) n_ranks); }

void launch_shard_select(const uint32_t *keys, uint32_t n, int shift, uint32_t b_lo, uint32_t b_hi, uint32_t *okeys, uint32_t *ovals, unsigned long long *cursor, hipStream_t s);
hipError_t launch_cluster_store_slice(const NodesDev &nd, const ClusterCfg &cc, int eq, uint64_t m, uint32_t bucket_base, uint32_t n_buckets_local, uint32_t *keys,
                                      uint32_t *vals, uint32_t *keys2, uint32_t *vals2, const uint32_t *meta, int uniform_len, void *sort_temp, size_t sort_temp_bytes,
                                      void *store, void *dir, unsigned long long *bad_flag, hipStream_t s, bool own_sort = true);
void launch_shard_export(bool count, const void *runs, int32_t node_begin, int32_t node_end, int shift, uint32_t bpr, uint32_t n_ranks, unsigned long long *counts,
                         int32_t *flagged_list, unsigned long long *flagged_count, uint32_t flagged_cap, const unsigned long long *seg_off, unsigned long long *cursor,
                         uint32_t *out, hipStream_t s);
void launch_shard_export_flagged(const NodesDev &nd, const PrefSufCfg &cfg, const ClusterCfg &cc, const int32_t *flagged_list, uint32_t n_flagged, uint32_t bpr, uint32_t n_ranks,
                                 const unsigned long long *seg_off, unsigned long long *cursor, uint32_t *out, hipStream_t s);
void launch_shard_desc_split(const uint32_t *in, uint64_t n, uint32_t key_base /* first key of the rank's bucket range */, uint32_t *dkey, unsigned long long *dval, hipStream_t s);
size_t     sort_desc_temp_bytes(uint64_t n);
hipError_t sort_desc(void *temp, size_t temp_bytes, const uint32_t *keys_in, uint32_t *keys_out, const unsigned long long *vals_in, unsigned long long *vals_out, uint64_t n,
                     int begin_bit, int end_bit, hipStream_t s, bool own_sort = true);
uint64_t shard_join_record_slack(int n_cu);
void launch_shard_groups(const uint32_t *dkey, uint64_t n_desc, int shift, uint32_t *flag, uint32_t *pos, uint32_t *gstart, uint64_t *scan_scratch, hipStream_t s);
void launch_shard_join(const NodesDev &nd, const PrefSufCfg &cfg, const ClusterCfg &cc, int eq, int uniform_len, const void *store, const void *dir, uint32_t bucket_base,
                       const uint32_t *dkey, const unsigned long long *dval, uint64_t n_desc, const uint32_t *gstart, uint32_t n_groups, uint32_t *rec_dst,
                       unsigned long long *rec_val, uint64_t rec_cap, unsigned long long *counters, unsigned long long *small_top, unsigned long long *declined,
                       int dmax /* descriptors of one bucket the join takes (<= 4096) */, int n_cu, hipStream_t s);
void launch_shard_pending_src(const uint32_t *rec_dst, const unsigned long long *rec_val, uint64_t n_rec, uint32_t *list, uint32_t cap, unsigned long long *count, hipStream_t s);
void launch_shard_bitmap_set(const uint32_t *ids, uint64_t n, uint32_t n_nodes, uint32_t *bitmap, hipStream_t s);
void launch_shard_small_emit(const unsigned long long *dval, const unsigned long long *small_top, uint64_t n_desc, const uint32_t *bitmap, uint32_t *out, uint32_t cap,
                             unsigned long long *count, hipStream_t s);
void launch_shard_small_split(const uint32_t *in, uint64_t n, uint32_t *ssrc, unsigned long long *skey, hipStream_t s);
void launch_shard_resolve(uint32_t *rec_dst, const unsigned long long *rec_val, uint64_t n_rec, const uint32_t *ssrc, const unsigned long long *skey, uint64_t n_small,
                          unsigned long long *dropped, hipStream_t s);
void launch_shard_edges_out(bool count, const uint32_t *rec_dst, const unsigned long long *rec_val, uint64_t n_rec, uint32_t chunk, uint32_t n_ranks,
                            unsigned long long *counts, const unsigned long long *seg_off, unsigned long long *cursor, alga_edge_dev *out, hipStream_t s);
void launch_shard_place_count(const alga_edge_dev *in, uint64_t n, int32_t src_base, int32_t n_src, uint32_t *deg, unsigned long long *bad, hipStream_t s);
void launch_shard_place_fill(const alga_edge_dev *in, uint64_t n, int32_t src_base, int32_t n_src, const uint32_t *rowptr, uint32_t *cursor, alga_edge_dev *out, hipStream_t s);

} // namespace alga
