// alga_amd/csrc/engine_simplify.hip -- C ABI of the first simplifier step (simplify_kernels.hip).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "engine_internal.h"
#include "simplify_kernels.h"

using namespace alga;

namespace {

int cut_impl(alga_engine *e, int32_t n, const alga_edge_dev *d_in, uint64_t m, int32_t mopp, hipStream_t s, const alga_edge **d_out, uint64_t *m_out,
             uint64_t *removed) {
    int rc;
    if ((rc = alga_ensure(e, e->counters, (CNT_TOTAL + 2) * sizeof(unsigned long long)))) return rc;
    if ((rc = alga_ensure(e, e->sp_rowptr, ((size_t) n + 2) * sizeof(uint32_t)))) return rc;
    if ((rc = alga_ensure(e, e->sp_sorted, (size_t) (m + 1) * sizeof(alga_edge_dev)))) return rc;
    if ((rc = alga_ensure(e, e->sp_list, (size_t) (m + 1) * sizeof(alga_edge_dev)))) return rc;
    if ((rc = alga_ensure(e, e->sp_cnt, ((size_t) n + 2) * sizeof(uint32_t)))) return rc;
    if ((rc = alga_ensure(e, e->sp_orow, ((size_t) n + 2) * sizeof(uint32_t)))) return rc;
    if ((rc = alga_ensure(e, e->sp_out, (size_t) (m + 1) * sizeof(alga_edge_dev)))) return rc;
    if ((rc = alga_ensure(e, e->scan_scratch, scan_scratch_bytes((uint64_t) n)))) return rc;
    unsigned long long *cnt = (unsigned long long *) e->counters.p;
    HIP_TRY(e, hipMemsetAsync(cnt, 0, sizeof(unsigned long long), s));
    launch_edge_rowptr(d_in, m, n, (uint32_t *) e->sp_rowptr.p, s);
    if ((rc = alga_check_launch(e, "k_edge_rowptr"))) return rc;
    launch_cut_triangles(d_in, (const uint32_t *) e->sp_rowptr.p, n, mopp, (alga_edge_dev *) e->sp_sorted.p, (alga_edge_dev *) e->sp_list.p,
                         (uint32_t *) e->sp_cnt.p, cnt, s);
    if ((rc = alga_check_launch(e, "k_cut_triangles"))) return rc;
    launch_exclusive_scan((const uint32_t *) e->sp_cnt.p, (uint64_t) n, (uint32_t *) e->sp_orow.p, (uint64_t *) e->scan_scratch.p, s);
    if ((rc = alga_check_launch(e, "scan(out_cnt)"))) return rc;
    launch_compact_rows((const alga_edge_dev *) e->sp_list.p, (const uint32_t *) e->sp_rowptr.p, (const uint32_t *) e->sp_cnt.p,
                        (const uint32_t *) e->sp_orow.p, n, (alga_edge_dev *) e->sp_out.p, s);
    if ((rc = alga_check_launch(e, "k_compact_rows"))) return rc;
    HIP_TRY(e, hipMemcpyAsync(e->h_counters, cnt, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    HIP_TRY(e, hipStreamSynchronize(s));
    *removed = e->h_counters[0];
    *m_out = m - *removed;
    *d_out = (const alga_edge *) e->sp_out.p;
    return ALGA_OK;
}

} // namespace

extern "C" {

int alga_cut_triangles_device(alga_engine *e, int32_t n_nodes, const alga_edge *d_edges, uint64_t n_edges, int32_t max_offset_parallel_paths,
                              void *hip_stream, const alga_edge **d_edges_out, uint64_t *n_edges_out, uint64_t *n_removed) {
    if (!e) return ALGA_ERR_INVALID_ARGUMENT;
    e->err.clear();
    if (!d_edges_out || !n_edges_out) return alga_fail(e, ALGA_ERR_INVALID_ARGUMENT, "output pointers must not be NULL");
    *d_edges_out = nullptr; *n_edges_out = 0;
    if (n_nodes < 0 || (n_edges && !d_edges)) return alga_fail(e, ALGA_ERR_INVALID_ARGUMENT, "bad graph");
    if (n_edges >= (1ull << 32) - 16) return alga_fail(e, ALGA_ERR_CAPACITY, "more than 2^32 edges");
    HIP_TRY(e, hipSetDevice(e->device));
    hipStream_t s = hip_stream ? (hipStream_t) hip_stream : e->own_stream;
    uint64_t removed = 0;
    int rc = cut_impl(e, n_nodes, (const alga_edge_dev *) d_edges, n_edges, max_offset_parallel_paths, s, d_edges_out, n_edges_out, &removed);
    if (rc == ALGA_OK && n_removed) *n_removed = removed;
    return rc;
}

int alga_cut_triangles_host(alga_engine *e, int32_t n_nodes, const alga_edge *edges, uint64_t n_edges, int32_t max_offset_parallel_paths,
                            alga_edge **edges_out, uint64_t *n_edges_out) {
    if (!e) return ALGA_ERR_INVALID_ARGUMENT;
    e->err.clear();
    if (!edges_out || !n_edges_out) return alga_fail(e, ALGA_ERR_INVALID_ARGUMENT, "output pointers must not be NULL");
    *edges_out = nullptr; *n_edges_out = 0;
    if (n_nodes < 0 || (n_edges && !edges)) return alga_fail(e, ALGA_ERR_INVALID_ARGUMENT, "bad graph");
    for (uint64_t k = 0; k < n_edges; k++) {
        if (edges[k].src < 0 || edges[k].src >= n_nodes || edges[k].dst < 0 || edges[k].dst >= n_nodes) return alga_fail(e, ALGA_ERR_INVALID_ARGUMENT, "edge endpoint out of range");
        if (k && (edges[k - 1].src > edges[k].src || (edges[k - 1].src == edges[k].src && edges[k - 1].dst > edges[k].dst)))
            return alga_fail(e, ALGA_ERR_INVALID_ARGUMENT, "edges must be sorted by (src, dst)");
    }
    HIP_TRY(e, hipSetDevice(e->device));
    int rc;
    if ((rc = alga_ensure(e, e->sp_in, (size_t) (n_edges + 1) * sizeof(alga_edge)))) return rc;
    if ((rc = alga_staged_h2d(e, e->sp_in.p, edges, (size_t) n_edges * sizeof(alga_edge)))) return rc;
    const alga_edge *d_out = nullptr;
    uint64_t m = 0, removed = 0;
    if ((rc = cut_impl(e, n_nodes, (const alga_edge_dev *) e->sp_in.p, n_edges, max_offset_parallel_paths, e->own_stream, &d_out, &m, &removed))) return rc;
    alga_edge *h = (alga_edge *) alga_host_list_take(e, (size_t) (m ? m : 1) * sizeof(alga_edge));
    if (!h) return alga_fail(e, ALGA_ERR_OUT_OF_MEMORY, "host edge buffer");
    if (m && (rc = alga_staged_d2h(e, h, d_out, (size_t) m * sizeof(alga_edge)))) { alga_host_list_give(e, h); return rc; }
    *edges_out = h; *n_edges_out = m;
    return ALGA_OK;
}

int alga_contig_trim_host(alga_engine *e, const uint32_t *words, int32_t stride_words, const int32_t *len, int32_t n_contigs, int32_t threshold,
                          int32_t *trim_left) {
    if (!e) return ALGA_ERR_INVALID_ARGUMENT;
    e->err.clear();
    if (n_contigs < 0 || (n_contigs && (!words || !len || !trim_left || stride_words <= 0))) return alga_fail(e, ALGA_ERR_INVALID_ARGUMENT, "bad contig arrays");
    if (threshold < 1 || threshold > 501) return alga_fail(e, ALGA_ERR_INVALID_ARGUMENT, "threshold must be in [1, 501]");
    if (n_contigs == 0) return ALGA_OK;
    if (n_contigs > 0x3FFFFFFF) return alga_fail(e, ALGA_ERR_CAPACITY, "too many contigs");
    const size_t M = (size_t) n_contigs;
    int32_t max_len = 0;
    for (size_t i = 0; i < M; i++) { if (len[i] < 0) return alga_fail(e, ALGA_ERR_INVALID_ARGUMENT, "negative contig length"); max_len = std::max(max_len, len[i]); }
    if ((int64_t) blocks_of(max_len) > (int64_t) stride_words) return alga_fail(e, ALGA_ERR_INVALID_ARGUMENT, "stride_words is smaller than the longest contig needs");
    const size_t row_bytes = (size_t) stride_words * sizeof(uint32_t);
    if (2 * M * row_bytes > (64ull << 30)) return alga_fail(e, ALGA_ERR_CAPACITY, "contig rows at one fixed stride would take more than 64 GB: trim on the host");
    HIP_TRY(e, hipSetDevice(e->device));
    hipStream_t s = e->own_stream;
    int rc;
    // nodes as src/main.cpp:636-645 numbers them: contigs 0 .. M-1, then their reverse complements M .. 2M-1
    alga_forget_node_set(e);
    if ((rc = alga_ensure(e, e->up_words, 2 * M * row_bytes))) return rc;
    if ((rc = alga_ensure(e, e->up_len, 2 * M * sizeof(int32_t)))) return rc;
    if ((rc = alga_ensure(e, e->sp_cnt, (M + 2) * sizeof(int32_t)))) return rc;
    HIP_TRY(e, hipStreamSynchronize(s));
    if ((rc = alga_staged_h2d(e, e->up_words.p, words, M * row_bytes))) return rc;
    if ((rc = alga_staged_h2d(e, e->up_len.p, len, M * sizeof(int32_t)))) return rc;
    launch_revcomp_rows((uint32_t *) e->up_words.p, stride_words, (int32_t *) e->up_len.p, n_contigs, s);
    if ((rc = alga_check_launch(e, "k_revcomp_rows"))) return rc;
    alga_nodes nd{(const uint32_t *) e->up_words.p, stride_words, (const int32_t *) e->up_len.p, 2 * n_contigs, nullptr, nullptr};
    alga_prefsuf_params p;
    alga_prefsuf_default_params(&p);
    p.min_overlap = threshold;                             // src/main.cpp:651-653
    p.rsoe_min_overlap = threshold;
    const alga_edge *d_edges = nullptr;
    uint64_t m = 0;
    if ((rc = alga_prefsuf_build_device(e, &nd, &p, (void *) s, &d_edges, &m))) return rc;
    // the reference does not call retainOnlySmallestOffset after this creator run; the largest overlap into a contig is the
    // smallest offset of its (source, contig) pair, which that call keeps: trimLeft is the same either way
    HIP_TRY(e, hipMemsetAsync(e->sp_cnt.p, 0, (M + 2) * sizeof(int32_t), s));
    launch_trim_left((const alga_edge_dev *) d_edges, m, (const int32_t *) e->up_len.p, n_contigs, (int32_t *) e->sp_cnt.p, s);
    if ((rc = alga_check_launch(e, "k_trim_left"))) return rc;
    HIP_TRY(e, hipMemcpyAsync(trim_left, e->sp_cnt.p, M * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(e, hipStreamSynchronize(s));
    return ALGA_OK;
}

} // extern "C"
