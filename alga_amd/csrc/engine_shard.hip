// alga_amd/csrc/engine_shard.hip -- host side of the seed-bucket-sharded N-GPU build (C ABI: include/alga_amd.h, alga_shard_*).
//
// One rank's part of the protocol described in prefsuf_shard.hip / alga_amd.h; the exchanges between the phases are the caller's
// (engine_multi.hip: RCCL or peer copies inside one process; alga_amd/multigpu.py: torch.distributed).  There is no CPU fallback: a
// phase either completes on the device or reports why not; ALGA_ERR_UNSUPPORTED means "take the replicated form, all ranks together".
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <vector>

#include "engine_internal.h"
#include "prefsuf_shard.h"

using namespace alga;

namespace {

struct EvPair { hipEvent_t a = nullptr, b = nullptr; };

// device time between two points of the stream (events made on demand: five phases a build, not worth permanent handles)
struct PhaseTimer {
    hipStream_t s; hipEvent_t a = nullptr, b = nullptr;
    explicit PhaseTimer(hipStream_t st) : s(st) { if (hipEventCreate(&a) != hipSuccess) a = nullptr; if (hipEventCreate(&b) != hipSuccess) b = nullptr; if (a) (void) hipEventRecord(a, s); }
    double stop() { float ms = 0.f; if (a && b && hipEventRecord(b, s) == hipSuccess && hipEventSynchronize(b) == hipSuccess) (void) hipEventElapsedTime(&ms, a, b); return ms; }
    ~PhaseTimer() { if (a) (void) hipEventDestroy(a); if (b) (void) hipEventDestroy(b); }
};

// the phase's counters (sh_cnt, 256 entries) -> host
int shard_sync_counters(alga_engine *e, hipStream_t s, const unsigned long long *d, int n, std::vector<unsigned long long> &h) {
    h.assign(256, 0ull);
    HIP_TRY(e, hipStreamSynchronize(s));
    HIP_TRY(e, hipMemcpy(h.data(), d, (size_t) n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return ALGA_OK;
}

} // namespace

extern "C" {

int alga_shard_index_device(alga_engine *e, const alga_nodes *nodes, const alga_prefsuf_params *p, int32_t rank, int32_t n_ranks, void *hip_stream,
                            const uint32_t **d_desc, uint64_t *desc_counts, uint64_t *desc_offsets) {
    if (!e) return ALGA_ERR_INVALID_ARGUMENT;
    e->err.clear();
    if (!d_desc || !desc_counts || !desc_offsets) return alga_fail(e, ALGA_ERR_INVALID_ARGUMENT, "output pointers must not be NULL");
    *d_desc = nullptr;
    if (n_ranks < 1 || n_ranks > 64 || rank < 0 || rank >= n_ranks) return alga_fail(e, ALGA_ERR_INVALID_ARGUMENT, "bad rank / n_ranks (1 .. 64 ranks)");
    for (int q = 0; q < n_ranks; q++) { desc_counts[q] = 0; desc_offsets[q] = 0; }
    e->sh.phase = 0;
    HIP_TRY(e, hipSetDevice(e->device));
    hipStream_t s = hip_stream ? (hipStream_t) hip_stream : e->own_stream;
    // the key pass (alga_prefsuf_keys_device) left keys / runs of my node range; the caller's all-gather completed the key array
    const int32_t kb = e->keyed_begin, ke = e->keyed_end;
    if (!nodes || e->keyed_n != nodes->n || e->keyed_words != (const void *) nodes->words)
        return alga_fail(e, ALGA_ERR_INVALID_ARGUMENT, "alga_shard_index_device: alga_prefsuf_keys_device has not been called on this node set");
    alga_prefsuf_params p2 = *p;
    p2.keys_shared = 2;                                    // reuse the node statistics of the key pass (same node set, nothing in between)
    e->store_n = -1;
    AlgaPrepared pp;
    {
        // prepare() with keys_shared = 2 takes the cached statistics when they match; the key pass's own state must survive it
        const int32_t kn = e->keyed_n;
        int rc = alga_prepare(e, nodes, &p2, s, pp);
        if (rc) return rc;
        e->keyed_n = kn;
    }
    if (!pp.local_ok || pp.cluster_eq == 0 || pp.cluster_eq > 4 || pp.local_sw != 1 || pp.reduction == ALGA_REDUCTION_PER_TARGET)
        return alga_fail(e, ALGA_ERR_UNSUPPORTED, "the bucket-sharded form takes reads of up to 208 nt with max_len - min_overlap <= 63 (one-word offset masks) and the source-side preconditions");
    memset(&e->shard_stats, 0, sizeof(e->shard_stats));
    const ClusterCfg cc = pp.cluster;
    const uint32_t n = (uint32_t) nodes->n;
    const uint32_t bpr = shard_buckets_per_rank(cc.n_buckets, n_ranks);
    const uint32_t b_lo = std::min<uint64_t>((uint64_t) rank * bpr, cc.n_buckets), b_hi = std::min<uint64_t>((uint64_t) (rank + 1) * bpr, cc.n_buckets);
    int rc;
    if ((rc = alga_ensure(e, e->sh_cnt, 256 * sizeof(unsigned long long)))) return rc;
    unsigned long long *cnt = (unsigned long long *) e->sh_cnt.p;       // [0] selected targets, [1] flagged sources, [2] index-bad flag, [64..127] counts, [128..191] cursors, [192..255] segment offsets
    HIP_TRY(e, hipMemsetAsync(cnt, 0, 256 * sizeof(unsigned long long), s));
    PhaseTimer t_index(s);
    // ---- my targets: select, sort, entries, directory ----
    // an upper bound of my share without a first counting pass: all targets (4 B / node each for two key and two id arrays)
    for (int k = 0; k < 2; k++) {
        if ((rc = alga_ensure(e, e->sh_keys[k], ((size_t) n + 1) * sizeof(uint32_t)))) return rc;
        if ((rc = alga_ensure(e, e->sh_vals[k], ((size_t) n + 1) * sizeof(uint32_t)))) return rc;
    }
    launch_shard_select((const uint32_t *) e->cl_keys[0].p, n, cc.idx_shift, b_lo, b_hi, (uint32_t *) e->sh_keys[0].p, (uint32_t *) e->sh_vals[0].p, cnt, s);
    if ((rc = alga_check_launch(e, "k_shard_select"))) return rc;
    // ---- my sources' runs: count per owner (and list the sources whose runs have to be made by brute force) ----
    const uint32_t flagged_cap = 1u << 20;
    if ((rc = alga_ensure(e, e->sh_flagged, (size_t) flagged_cap * sizeof(int32_t)))) return rc;
    launch_shard_export(true, e->cl_runs.p, kb, ke, cc.idx_shift, bpr, (uint32_t) n_ranks, cnt + 64, (int32_t *) e->sh_flagged.p, cnt + 1, flagged_cap, nullptr, nullptr, nullptr, s);
    if ((rc = alga_check_launch(e, "k_shard_export (count)"))) return rc;
    std::vector<unsigned long long> hc;
    if ((rc = shard_sync_counters(e, s, cnt, 128, hc))) return rc;
    const uint64_t m = hc[0], n_flagged = hc[1];
    if (n_flagged > flagged_cap) return alga_fail(e, ALGA_ERR_UNSUPPORTED, "more than 2^20 sources of this rank need brute-force window minimizers");
    std::vector<unsigned long long> seg((size_t) n_ranks + 1, 0ull);
    for (int q = 0; q < n_ranks; q++) seg[(size_t) q + 1] = seg[(size_t) q] + hc[64 + (size_t) q] + 64ull * n_flagged;      // a flagged source has <= 64 runs
    const uint64_t desc_cap = seg[(size_t) n_ranks];
    if (desc_cap >= (1ull << 32) - 16) return alga_fail(e, ALGA_ERR_CAPACITY, "more than 2^32 run descriptors on one rank");
    if ((rc = alga_ensure(e, e->sh_desc_out, (desc_cap + 1) * 12))) return rc;
    HIP_TRY(e, hipMemcpyAsync(cnt + 192, seg.data(), (size_t) n_ranks * sizeof(unsigned long long), hipMemcpyHostToDevice, s));
    // entries + directory of my bucket range
    const uint32_t n_b_local = b_hi - b_lo;
    if ((rc = alga_ensure(e, e->sh_store, (m + 2) * 16 * (size_t) pp.cluster_eq))) return rc;
    if ((rc = alga_ensure(e, e->sh_dir, ((size_t) n_b_local + 2) * 16))) return rc;
    if ((rc = alga_ensure(e, e->sort_temp, std::max(cluster_sort_temp_bytes(m), sort_desc_temp_bytes(1))))) return rc;
    HIP_TRY(e, launch_cluster_store_slice(pp.nd, cc, pp.cluster_eq, m, b_lo, n_b_local, (uint32_t *) e->sh_keys[0].p, (uint32_t *) e->sh_vals[0].p, (uint32_t *) e->sh_keys[1].p,
                                          (uint32_t *) e->sh_vals[1].p, (const uint32_t *) e->cl_meta.p, pp.uniform_len, e->sort_temp.p, cluster_sort_temp_bytes(m),
                                          e->sh_store.p, e->sh_dir.p, cnt + 2, s, e->opt_own_sort != 0));
    e->shard_stats.ms_index = t_index.stop();
    PhaseTimer t_export(s);
    launch_shard_export(false, e->cl_runs.p, kb, ke, cc.idx_shift, bpr, (uint32_t) n_ranks, nullptr, nullptr, nullptr, 0u, cnt + 192, cnt + 128, (uint32_t *) e->sh_desc_out.p, s);
    if ((rc = alga_check_launch(e, "k_shard_export"))) return rc;
    launch_shard_export_flagged(pp.nd, pp.cfg, cc, (const int32_t *) e->sh_flagged.p, (uint32_t) n_flagged, bpr, (uint32_t) n_ranks, cnt + 192, cnt + 128,
                                (uint32_t *) e->sh_desc_out.p, s);
    if ((rc = alga_check_launch(e, "k_shard_export_flagged"))) return rc;
    e->shard_stats.ms_export = t_export.stop();
    if ((rc = shard_sync_counters(e, s, cnt, 192, hc))) return rc;
    if (hc[2] != 0) return alga_fail(e, ALGA_ERR_HIP, "bucket-sharded index: the sorted key array is not in order");
    uint64_t total = 0;
    for (int q = 0; q < n_ranks; q++) {
        desc_counts[q] = hc[128 + (size_t) q]; desc_offsets[q] = seg[(size_t) q];
        if (desc_counts[q] > seg[(size_t) q + 1] - seg[(size_t) q]) return alga_fail(e, ALGA_ERR_HIP, "bucket-sharded index: descriptor segment overflow");
        total += desc_counts[q];
    }
    e->shard_stats.targets_owned = m; e->shard_stats.descriptors_out = total; e->shard_stats.flagged_sources = n_flagged;
    e->sh.phase = 1; e->sh.rank = rank; e->sh.n_ranks = n_ranks; e->sh.eq = pp.cluster_eq; e->sh.uniform_len = pp.uniform_len; e->sh.n = nodes->n;
    e->sh.words = (const void *) nodes->words; e->sh.cfg = pp.cfg; e->sh.cc = cc; e->sh.bucket_base = b_lo; e->sh.bpr = bpr; e->sh.n_targets = m;
    *d_desc = (const uint32_t *) e->sh_desc_out.p;
    return ALGA_OK;
}

int alga_shard_join_device(alga_engine *e, const alga_nodes *nodes, const uint32_t *d_desc_in, uint64_t n_desc, void *hip_stream, const uint32_t **d_pending_src,
                           uint64_t *n_pending_src) {
    if (!e) return ALGA_ERR_INVALID_ARGUMENT;
    e->err.clear();
    if (!d_pending_src || !n_pending_src) return alga_fail(e, ALGA_ERR_INVALID_ARGUMENT, "output pointers must not be NULL");
    *d_pending_src = nullptr; *n_pending_src = 0;
    if (e->sh.phase != 1 || !nodes || nodes->n != e->sh.n || (const void *) nodes->words != e->sh.words)
        return alga_fail(e, ALGA_ERR_INVALID_ARGUMENT, "alga_shard_join_device follows alga_shard_index_device on the same node set");
    if (n_desc && !d_desc_in) return alga_fail(e, ALGA_ERR_INVALID_ARGUMENT, "descriptor array must not be NULL");
    if (n_desc >= (1ull << 32) - 16) return alga_fail(e, ALGA_ERR_CAPACITY, "more than 2^32 run descriptors on one rank");
    HIP_TRY(e, hipSetDevice(e->device));
    hipStream_t s = hip_stream ? (hipStream_t) hip_stream : e->own_stream;
    int rc;
    NodesDev nd;
    nd.words = nodes->words; nd.len = nodes->len; nd.from = nodes->align_from; nd.to = nodes->align_to; nd.n = nodes->n; nd.stride = nodes->stride_words;
    unsigned long long *cnt = (unsigned long long *) e->counters.p;
    unsigned long long *scnt = (unsigned long long *) e->sh_cnt.p;
    // ---- descriptors by bucket ----
    PhaseTimer t_sort(s);
    for (int k = 0; k < 2; k++) {
        if ((rc = alga_ensure(e, e->sh_dkey[k], (n_desc + 64) * sizeof(uint32_t)))) return rc;
        if ((rc = alga_ensure(e, e->sh_dval[k], (n_desc + 64) * sizeof(unsigned long long)))) return rc;
    }
    if ((rc = alga_ensure(e, e->sort_temp, sort_desc_temp_bytes(n_desc)))) return rc;
    if ((rc = alga_ensure(e, e->sh_small_top, (n_desc + 1) * 3 * sizeof(unsigned long long)))) return rc;
    const int shift = e->sh.cc.idx_shift;
    int range_bits = 1;                                    // bits of a bucket index relative to my range
    while (range_bits < 32 - shift && (1ull << range_bits) < (uint64_t) e->sh.bpr) range_bits++;
    launch_shard_desc_split(d_desc_in, n_desc, e->sh.bucket_base << shift, (uint32_t *) e->sh_dkey[0].p, (unsigned long long *) e->sh_dval[0].p, s);
    if ((rc = alga_check_launch(e, "k_shard_desc_split"))) return rc;
    HIP_TRY(e, sort_desc(e->sort_temp.p, sort_desc_temp_bytes(n_desc), (const uint32_t *) e->sh_dkey[0].p, (uint32_t *) e->sh_dkey[1].p, (const unsigned long long *) e->sh_dval[0].p,
                         (unsigned long long *) e->sh_dval[1].p, n_desc, shift, std::min(32, shift + range_bits), s, e->opt_own_sort != 0));
    // the groups (one per bucket) of the sorted array
    if ((rc = alga_ensure(e, e->sh_gflag, (n_desc + 2) * sizeof(uint32_t)))) return rc;
    if ((rc = alga_ensure(e, e->sh_gpos, (n_desc + 2) * sizeof(uint32_t)))) return rc;
    if ((rc = alga_ensure(e, e->sh_gstart, (n_desc + 2) * sizeof(uint32_t)))) return rc;
    if ((rc = alga_ensure(e, e->scan_scratch, scan_scratch_bytes(n_desc + 1)))) return rc;
    launch_shard_groups((const uint32_t *) e->sh_dkey[1].p, n_desc, e->sh.cc.idx_shift, (uint32_t *) e->sh_gflag.p, (uint32_t *) e->sh_gpos.p, (uint32_t *) e->sh_gstart.p,
                        (uint64_t *) e->scan_scratch.p, s);
    if ((rc = alga_check_launch(e, "k_shard_groups"))) return rc;
    uint32_t n_groups = 0;
    if (n_desc) {
        HIP_TRY(e, hipMemcpyAsync(e->h_counters, (const uint32_t *) e->sh_gpos.p + n_desc, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        HIP_TRY(e, hipStreamSynchronize(s));
        n_groups = *(const uint32_t *) e->h_counters;
    }
    e->shard_stats.ms_sort = t_sort.stop();
    e->sh.d_keys_sorted = (uint32_t *) e->sh_dkey[1].p; e->sh.d_vals_sorted = (unsigned long long *) e->sh_dval[1].p;
    // ---- join: records = surviving overlaps (small ones pending their source's cap) ----
    PhaseTimer t_join(s);
    const uint64_t slack = shard_join_record_slack(e->n_cu);
    uint64_t cap = std::max<uint64_t>(e->rec_cap_hint, 2 * e->sh.n_targets + n_desc / 4 + 4096) + slack;
    uint64_t n_rec = 0;
    bool done = false;
    unsigned long long join_passes[2] = {0, 0};
    for (int attempt = 0; attempt < 4 && !done; attempt++) {
        if (cap >= (1ull << 32) - 16) return alga_fail(e, ALGA_ERR_CAPACITY, "more than 2^32 overlap records; shard the input");
        if ((rc = alga_ensure(e, e->rec_dst, cap * sizeof(uint32_t)))) return rc;
        if ((rc = alga_ensure(e, e->rec_val, cap * sizeof(unsigned long long)))) return rc;
        HIP_TRY(e, hipMemsetAsync(cnt, 0, CNT_TOTAL * sizeof(unsigned long long), s));
        HIP_TRY(e, hipMemsetAsync(scnt, 0, 8 * sizeof(unsigned long long), s));
        launch_shard_join(nd, e->sh.cfg, e->sh.cc, e->sh.eq, e->sh.uniform_len, e->sh_store.p, e->sh_dir.p, e->sh.bucket_base, e->sh.d_keys_sorted, e->sh.d_vals_sorted, n_desc,
                          (const uint32_t *) e->sh_gstart.p, n_groups, (uint32_t *) e->rec_dst.p, (unsigned long long *) e->rec_val.p, cap, cnt, (unsigned long long *) e->sh_small_top.p, scnt + 3, e->opt_shard_dmax, e->n_cu, s);
        if ((rc = alga_check_launch(e, "k_shard_join"))) return rc;
        HIP_TRY(e, hipMemcpyAsync(e->h_counters, cnt, CNT_TOTAL * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
        HIP_TRY(e, hipMemcpyAsync(e->h_counters + CNT_TOTAL, scnt + 3, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
        HIP_TRY(e, hipMemcpyAsync(join_passes, scnt + 4, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
        HIP_TRY(e, hipStreamSynchronize(s));
        if (e->h_counters[CNT_TOTAL] != 0) return alga_fail(e, ALGA_ERR_UNSUPPORTED, "a bucket holds more run descriptors than the bucket-sharded join takes (repeat-rich input)");
        const uint64_t need = e->h_counters[CNT_RECORDS];
        if (need <= cap) { n_rec = need; e->rec_cap_hint = std::max<uint64_t>(e->rec_cap_hint, need + need / 16 + 4096); done = true; }
        else cap = need + need / 16 + 4096 + slack;
    }
    if (!done) return alga_fail(e, ALGA_ERR_HIP, "record buffer kept overflowing");
    e->shard_stats.ms_join = t_join.stop();
    e->shard_stats.records = e->h_counters[CNT_VALID_RECORDS];
    e->shard_stats.join_passes = join_passes[0]; e->shard_stats.join_passes_serial = join_passes[1];
    e->shard_stats.descriptors_in = n_desc;
    // ---- the sources of the pending (small) survivors ----
    const uint32_t pend_cap = (uint32_t) std::min<uint64_t>(e->shard_stats.records + 1, (1ull << 32) - 16);
    if ((rc = alga_ensure(e, e->sh_pending, (size_t) pend_cap * sizeof(uint32_t)))) return rc;
    HIP_TRY(e, hipMemsetAsync(scnt + 4, 0, sizeof(unsigned long long), s));
    launch_shard_pending_src((const uint32_t *) e->rec_dst.p, (const unsigned long long *) e->rec_val.p, n_rec, (uint32_t *) e->sh_pending.p, pend_cap, scnt + 4, s);
    if ((rc = alga_check_launch(e, "k_shard_pending_src"))) return rc;
    HIP_TRY(e, hipMemcpyAsync(e->h_counters, scnt + 4, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    HIP_TRY(e, hipStreamSynchronize(s));
    e->shard_stats.pending = e->h_counters[0];
    e->sh.n_desc = n_desc; e->sh.n_rec = n_rec; e->sh.phase = 2;
    *d_pending_src = (const uint32_t *) e->sh_pending.p;
    *n_pending_src = e->h_counters[0];
    return ALGA_OK;
}

int alga_shard_small_keys_device(alga_engine *e, const uint32_t *d_pending_src_all, uint64_t n_all, void *hip_stream, const uint32_t **d_small, uint64_t *n_small) {
    if (!e) return ALGA_ERR_INVALID_ARGUMENT;
    e->err.clear();
    if (!d_small || !n_small) return alga_fail(e, ALGA_ERR_INVALID_ARGUMENT, "output pointers must not be NULL");
    *d_small = nullptr; *n_small = 0;
    if (e->sh.phase != 2) return alga_fail(e, ALGA_ERR_INVALID_ARGUMENT, "alga_shard_small_keys_device follows alga_shard_join_device");
    if (n_all && !d_pending_src_all) return alga_fail(e, ALGA_ERR_INVALID_ARGUMENT, "source list must not be NULL");
    HIP_TRY(e, hipSetDevice(e->device));
    hipStream_t s = hip_stream ? (hipStream_t) hip_stream : e->own_stream;
    int rc;
    unsigned long long *scnt = (unsigned long long *) e->sh_cnt.p;
    e->shard_stats.pending_sources = n_all;
    e->sh.phase = 3;
    if (n_all == 0) return ALGA_OK;                        // nothing is pending anywhere
    PhaseTimer t(s);
    const size_t bm_words = ((size_t) e->sh.n + 31) / 32 + 1;
    if ((rc = alga_ensure(e, e->sh_bitmap, bm_words * sizeof(uint32_t)))) return rc;
    HIP_TRY(e, hipMemsetAsync(e->sh_bitmap.p, 0, bm_words * sizeof(uint32_t), s));
    launch_shard_bitmap_set(d_pending_src_all, n_all, (uint32_t) e->sh.n, (uint32_t *) e->sh_bitmap.p, s);
    if ((rc = alga_check_launch(e, "k_shard_bitmap_set"))) return rc;
    uint64_t cap = std::max<uint64_t>(3 * 8 * n_all + 4096, 1 << 16);     // a listed source has ~3 of its <= 8 runs here, three keys each; retried if short
    for (int attempt = 0; attempt < 3; attempt++) {
        if (cap >= (1ull << 32) - 16) return alga_fail(e, ALGA_ERR_CAPACITY, "more than 2^32 small keys");
        if ((rc = alga_ensure(e, e->sh_small_out, cap * 12))) return rc;
        HIP_TRY(e, hipMemsetAsync(scnt + 5, 0, sizeof(unsigned long long), s));
        launch_shard_small_emit(e->sh.d_vals_sorted, (const unsigned long long *) e->sh_small_top.p, e->sh.n_desc, (const uint32_t *) e->sh_bitmap.p, (uint32_t *) e->sh_small_out.p,
                                (uint32_t) cap, scnt + 5, s);
        if ((rc = alga_check_launch(e, "k_shard_small_emit"))) return rc;
        HIP_TRY(e, hipMemcpyAsync(e->h_counters, scnt + 5, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
        HIP_TRY(e, hipStreamSynchronize(s));
        if (e->h_counters[0] <= cap) break;
        cap = e->h_counters[0] + 4096;
        if (attempt == 2) return alga_fail(e, ALGA_ERR_HIP, "small-key list kept overflowing");
    }
    e->shard_stats.ms_cap += t.stop();
    e->shard_stats.small_keys_out = e->h_counters[0];
    *d_small = (const uint32_t *) e->sh_small_out.p;
    *n_small = e->h_counters[0];
    return ALGA_OK;
}

int alga_shard_resolve_device(alga_engine *e, const uint32_t *d_small_all, uint64_t n_small_all, void *hip_stream, const alga_edge **d_edges_out, uint64_t *edge_counts,
                              uint64_t *edge_offsets) {
    if (!e) return ALGA_ERR_INVALID_ARGUMENT;
    e->err.clear();
    if (!d_edges_out || !edge_counts || !edge_offsets) return alga_fail(e, ALGA_ERR_INVALID_ARGUMENT, "output pointers must not be NULL");
    *d_edges_out = nullptr;
    if (e->sh.phase != 3) return alga_fail(e, ALGA_ERR_INVALID_ARGUMENT, "alga_shard_resolve_device follows alga_shard_small_keys_device");
    if (n_small_all && !d_small_all) return alga_fail(e, ALGA_ERR_INVALID_ARGUMENT, "small-key list must not be NULL");
    if (n_small_all >= (1ull << 32) - 16) return alga_fail(e, ALGA_ERR_CAPACITY, "more than 2^32 small keys");
    HIP_TRY(e, hipSetDevice(e->device));
    hipStream_t s = hip_stream ? (hipStream_t) hip_stream : e->own_stream;
    int rc;
    const int N = e->sh.n_ranks;
    unsigned long long *scnt = (unsigned long long *) e->sh_cnt.p;
    HIP_TRY(e, hipMemsetAsync(scnt + 6, 0, sizeof(unsigned long long), s));
    HIP_TRY(e, hipMemsetAsync(scnt + 64, 0, 192 * sizeof(unsigned long long), s));
    PhaseTimer t_cap(s);
    if (e->shard_stats.pending > 0) {
        // every pending edge's source has its small keys in the gathered list (its own key among them): by source, then count the larger ones
        for (int k = 0; k < 2; k++) {
            if ((rc = alga_ensure(e, e->sh_ssrc[k], (n_small_all + 1) * sizeof(uint32_t)))) return rc;
            if ((rc = alga_ensure(e, e->sh_skey[k], (n_small_all + 1) * sizeof(unsigned long long)))) return rc;
        }
        if ((rc = alga_ensure(e, e->sort_temp, sort_records_temp_bytes(n_small_all, 32)))) return rc;
        launch_shard_small_split(d_small_all, n_small_all, (uint32_t *) e->sh_ssrc[0].p, (unsigned long long *) e->sh_skey[0].p, s);
        if ((rc = alga_check_launch(e, "k_shard_small_split"))) return rc;
        int src_bits = 1;
        while (src_bits < 32 && (1ll << src_bits) < (long long) e->sh.n) src_bits++;
        HIP_TRY(e, sort_records(e->sort_temp.p, sort_records_temp_bytes(n_small_all, 32), (const uint32_t *) e->sh_ssrc[0].p, (uint32_t *) e->sh_ssrc[1].p,
                                (const unsigned long long *) e->sh_skey[0].p, (unsigned long long *) e->sh_skey[1].p, n_small_all, src_bits, s));
        launch_shard_resolve((uint32_t *) e->rec_dst.p, (const unsigned long long *) e->rec_val.p, e->sh.n_rec, (const uint32_t *) e->sh_ssrc[1].p,
                             (const unsigned long long *) e->sh_skey[1].p, n_small_all, scnt + 6, s);
        if ((rc = alga_check_launch(e, "k_shard_resolve"))) return rc;
    }
    e->shard_stats.ms_cap += t_cap.stop();
    e->shard_stats.small_keys_in = n_small_all;
    // ---- final edges, grouped by the rank that owns the source id ----
    PhaseTimer t_out(s);
    const uint32_t chunk = (uint32_t) std::max<int64_t>(2, 2 * (((int64_t) e->sh.n + 2 * (int64_t) N - 1) / (2 * (int64_t) N)));      // alga_amd/multigpu.py: shard_chunk
    launch_shard_edges_out(true, (const uint32_t *) e->rec_dst.p, (const unsigned long long *) e->rec_val.p, e->sh.n_rec, chunk, (uint32_t) N, scnt + 64, nullptr, nullptr, nullptr, s);
    if ((rc = alga_check_launch(e, "k_shard_edges_out (count)"))) return rc;
    std::vector<unsigned long long> hc;
    if ((rc = shard_sync_counters(e, s, scnt, 128, hc))) return rc;
    e->shard_stats.dropped = hc[6];
    std::vector<unsigned long long> seg((size_t) N + 1, 0ull);
    for (int q = 0; q < N; q++) seg[(size_t) q + 1] = seg[(size_t) q] + hc[64 + (size_t) q];
    const uint64_t total = seg[(size_t) N];
    if (total >= (1ull << 32) - 16) return alga_fail(e, ALGA_ERR_CAPACITY, "more than 2^32 edges");
    if ((rc = alga_ensure(e, e->sh_edges_out, (total + 1) * sizeof(alga_edge_dev)))) return rc;
    HIP_TRY(e, hipMemcpyAsync(scnt + 192, seg.data(), (size_t) N * sizeof(unsigned long long), hipMemcpyHostToDevice, s));
    launch_shard_edges_out(false, (const uint32_t *) e->rec_dst.p, (const unsigned long long *) e->rec_val.p, e->sh.n_rec, chunk, (uint32_t) N, nullptr, scnt + 192, scnt + 128,
                           (alga_edge_dev *) e->sh_edges_out.p, s);
    if ((rc = alga_check_launch(e, "k_shard_edges_out"))) return rc;
    e->shard_stats.ms_edges_out = t_out.stop();
    for (int q = 0; q < N; q++) { edge_counts[q] = hc[64 + (size_t) q]; edge_offsets[q] = seg[(size_t) q]; }
    e->shard_stats.edges_out = total;
    e->sh.phase = 4;
    *d_edges_out = (const alga_edge *) e->sh_edges_out.p;
    return ALGA_OK;
}

int alga_shard_place_device(alga_engine *e, const alga_edge *d_edges_in, uint64_t n_in, int32_t src_begin, int32_t src_end, void *hip_stream, const alga_edge **d_edges,
                            uint64_t *n_edges) {
    if (!e) return ALGA_ERR_INVALID_ARGUMENT;
    e->err.clear();
    if (!d_edges || !n_edges) return alga_fail(e, ALGA_ERR_INVALID_ARGUMENT, "output pointers must not be NULL");
    *d_edges = nullptr; *n_edges = 0;
    if (n_in && !d_edges_in) return alga_fail(e, ALGA_ERR_INVALID_ARGUMENT, "edge array must not be NULL");
    if (src_begin < 0 || src_end < src_begin) return alga_fail(e, ALGA_ERR_INVALID_ARGUMENT, "bad source range");
    if (n_in >= (1ull << 32) - 16) return alga_fail(e, ALGA_ERR_CAPACITY, "more than 2^32 edges");
    HIP_TRY(e, hipSetDevice(e->device));
    hipStream_t s = hip_stream ? (hipStream_t) hip_stream : e->own_stream;
    int rc;
    const uint64_t n_src = (uint64_t) (src_end - src_begin);
    if ((rc = alga_ensure(e, e->sh_cnt, 256 * sizeof(unsigned long long)))) return rc;
    unsigned long long *scnt = (unsigned long long *) e->sh_cnt.p;
    PhaseTimer t(s);
    if ((rc = alga_ensure(e, e->sh_deg, (n_src + 1) * sizeof(uint32_t)))) return rc;
    if ((rc = alga_ensure(e, e->sh_cursor, (n_src + 1) * sizeof(uint32_t)))) return rc;
    if ((rc = alga_ensure(e, e->sh_rowptr, (n_src + 2) * sizeof(uint32_t)))) return rc;
    if ((rc = alga_ensure(e, e->scan_scratch, scan_scratch_bytes(n_src)))) return rc;
    if ((rc = alga_ensure(e, e->sh_edges, (n_in + 1) * sizeof(alga_edge_dev)))) return rc;
    HIP_TRY(e, hipMemsetAsync(e->sh_deg.p, 0, (n_src + 1) * sizeof(uint32_t), s));
    HIP_TRY(e, hipMemsetAsync(e->sh_cursor.p, 0, (n_src + 1) * sizeof(uint32_t), s));
    HIP_TRY(e, hipMemsetAsync(scnt + 7, 0, sizeof(unsigned long long), s));
    launch_shard_place_count((const alga_edge_dev *) d_edges_in, n_in, src_begin, (int32_t) n_src, (uint32_t *) e->sh_deg.p, scnt + 7, s);
    if ((rc = alga_check_launch(e, "k_shard_place_count"))) return rc;
    launch_exclusive_scan((const uint32_t *) e->sh_deg.p, n_src, (uint32_t *) e->sh_rowptr.p, (uint64_t *) e->scan_scratch.p, s);
    if ((rc = alga_check_launch(e, "scan(deg)"))) return rc;
    launch_shard_place_fill((const alga_edge_dev *) d_edges_in, n_in, src_begin, (int32_t) n_src, (const uint32_t *) e->sh_rowptr.p, (uint32_t *) e->sh_cursor.p,
                            (alga_edge_dev *) e->sh_edges.p, s);
    if ((rc = alga_check_launch(e, "k_shard_place_fill"))) return rc;
    launch_sort_rows((int32_t) n_src, (const uint32_t *) e->sh_rowptr.p, (alga_edge_dev *) e->sh_edges.p, s);
    if ((rc = alga_check_launch(e, "k_sort_rows"))) return rc;
    HIP_TRY(e, hipMemcpyAsync(e->h_counters, scnt + 7, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    HIP_TRY(e, hipStreamSynchronize(s));
    e->shard_stats.ms_place = t.stop();
    if (e->h_counters[0] != 0) return alga_fail(e, ALGA_ERR_INVALID_ARGUMENT, "alga_shard_place_device: an edge's source lies outside [src_begin, src_end)");
    e->shard_stats.edges_in = n_in; e->shard_stats.edges = n_in;
    e->sh.phase = 0;
    *d_edges = (const alga_edge *) e->sh_edges.p;
    *n_edges = n_in;
    return ALGA_OK;
}

int alga_shard_last_stats(const alga_engine *e, alga_shard_stats *out) {
    if (!e || !out) return ALGA_ERR_INVALID_ARGUMENT;
    *out = e->shard_stats;
    return ALGA_OK;
}

} // extern "C"
