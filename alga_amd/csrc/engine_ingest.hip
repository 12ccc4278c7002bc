// alga_amd/csrc/engine_ingest.hip -- C ABI entry of the GPU input stage N1 (kernels: ingest_kernels.hip)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>

#include "engine_internal.h"
#include "ingest_kernels.h"

using namespace alga;

extern "C" int alga_preprocess_nodes(alga_engine *e, const alga_preprocess_input *in, alga_device_node_set *out) {
    if (!e) return ALGA_ERR_INVALID_ARGUMENT;
    e->err.clear();
    if (!in || !out) return alga_fail(e, ALGA_ERR_INVALID_ARGUMENT, "input/output must not be NULL");
    memset(out, 0, sizeof(*out));
    if (in->n_nodes < 0 || (in->n_nodes & 1)) return alga_fail(e, ALGA_ERR_INVALID_ARGUMENT, "n_nodes must be even and >= 0");
    if (in->n_nodes >= 0x7FFFFFFELL) return alga_fail(e, ALGA_ERR_CAPACITY, "too many nodes");
    if (in->n_nodes && (!in->rows || !in->len || in->stride_words <= 0)) return alga_fail(e, ALGA_ERR_INVALID_ARGUMENT, "bad node arrays");
    if (in->remove_pref_reads < 1 || in->remove_pref_reads > 3) return alga_fail(e, ALGA_ERR_INVALID_ARGUMENT, "remove_pref_reads must be 1, 2 or 3");
    HIP_TRY(e, hipSetDevice(e->device));
    hipStream_t s = e->own_stream;
    int rc;
    const uint64_t n = (uint64_t) in->n_nodes, R = n / 2;
    const int stride = in->stride_words;
    // lengths decide the pass count and must fit the rows: one host pass (the caller's arrays are pageable memory anyway)
    int max_len = 0;
    uint64_t live = 0;
    for (uint64_t i = 0; i < n; i++) { const int l = in->len[i]; if (l >= 0) { live++; max_len = std::max(max_len, l); } }
    if ((int64_t) blocks_of(max_len) > (int64_t) stride) return alga_fail(e, ALGA_ERR_INVALID_ARGUMENT, "stride_words is smaller than the longest read needs");
    const int used_words = std::max(1, blocks_of(max_len));
    const int stride_out = hbm_row_stride(used_words);
    if ((rc = alga_ensure(e, e->pp_rows, n * (size_t) stride * sizeof(uint32_t)))) return rc;
    if ((rc = alga_ensure(e, e->pp_len, (n + 2) * sizeof(int32_t)))) return rc;
    if ((rc = alga_ensure(e, e->pp_tally, 8 * sizeof(unsigned long long)))) return rc;
    if (n) {
        HIP_TRY(e, hipMemcpyAsync(e->pp_rows.p, in->rows, n * (size_t) stride * sizeof(uint32_t), hipMemcpyHostToDevice, s));
        HIP_TRY(e, hipMemcpyAsync(e->pp_len.p, in->len, n * sizeof(int32_t), hipMemcpyHostToDevice, s));
    }
    HIP_TRY(e, hipMemsetAsync(e->pp_tally.p, 0, 8 * sizeof(unsigned long long), s));
    HIP_TRY(e, hipEventRecord(e->ev[EV_START], s));
    const uint32_t *rows = (const uint32_t *) e->pp_rows.p;
    int32_t *len = (int32_t *) e->pp_len.p;
    unsigned long long *tally = (unsigned long long *) e->pp_tally.p;
    const uint8_t *mark = nullptr;
    if (in->remove_pref_reads != 3 && live > 1) {
        const size_t temp = sort_u64_u32_temp_bytes(n);
        for (int k = 0; k < 2; k++) {
            if ((rc = alga_ensure(e, e->pp_perm[k], (n + 1) * sizeof(uint32_t)))) return rc;
            if ((rc = alga_ensure(e, e->pp_keys[k], (n + 1) * sizeof(unsigned long long)))) return rc;
        }
        if ((rc = alga_ensure(e, e->sort_temp, temp))) return rc;
        if ((rc = alga_ensure(e, e->pp_mark, n + 16))) return rc;
        int cur = 0;
        launch_pp_iota((uint32_t *) e->pp_perm[cur].p, n, s);
        int len_bits = 1;
        while (len_bits < 31 && (1ll << len_bits) <= (long long) max_len) len_bits++;
        for (int pass = -1; pass < (used_words + 1) / 2; pass++) {
            // least significant digit first: the length, then the word pairs from the last to the first
            const int p = pass < 0 ? -1 : (used_words + 1) / 2 - 1 - pass;
            launch_pp_keys(rows, stride, used_words, len, (const uint32_t *) e->pp_perm[cur].p, n, p, (unsigned long long *) e->pp_keys[0].p, s);
            if ((rc = alga_check_launch(e, "k_pp_keys"))) return rc;
            // removed nodes carry the all-ones key: one more bit than the length needs keeps them behind every live node
            HIP_TRY(e, sort_u64_u32(e->sort_temp.p, temp, (const unsigned long long *) e->pp_keys[0].p, (unsigned long long *) e->pp_keys[1].p,
                                    (const uint32_t *) e->pp_perm[cur].p, (uint32_t *) e->pp_perm[cur ^ 1].p, n, p < 0 ? len_bits + 1 : 64, s));
            cur ^= 1;
        }
        HIP_TRY(e, hipMemsetAsync(e->pp_mark.p, 0, n + 16, s));
        launch_pp_mark(rows, stride, len, (const uint32_t *) e->pp_perm[cur].p, live, in->remove_pref_reads, (uint8_t *) e->pp_mark.p, s);
        if ((rc = alga_check_launch(e, "k_pp_mark"))) return rc;
        mark = (const uint8_t *) e->pp_mark.p;
    }
    if ((rc = alga_ensure(e, e->pp_keep, (R + 2) * sizeof(uint32_t)))) return rc;
    if ((rc = alga_ensure(e, e->pp_pos, (R + 2) * sizeof(uint32_t)))) return rc;
    if ((rc = alga_ensure(e, e->scan_scratch, scan_scratch_bytes(R)))) return rc;
    launch_pp_apply(len, mark, R, (uint32_t *) e->pp_keep.p, tally, s);
    if ((rc = alga_check_launch(e, "k_pp_apply"))) return rc;
    launch_exclusive_scan((const uint32_t *) e->pp_keep.p, R, (uint32_t *) e->pp_pos.p, (uint64_t *) e->scan_scratch.p, s);
    if ((rc = alga_check_launch(e, "scan(keep)"))) return rc;
    // the output is sized for every read surviving: no round trip for the count before the copy
    if ((rc = alga_ensure(e, e->pp_out_rows, (n + 2) * (size_t) stride_out * sizeof(uint32_t)))) return rc;
    if ((rc = alga_ensure(e, e->pp_out_len, (n + 2) * sizeof(int32_t)))) return rc;
    if ((rc = alga_ensure(e, e->pp_out_pair, n + 16))) return rc;
    launch_pp_compact(rows, stride, len, (const uint32_t *) e->pp_keep.p, (const uint32_t *) e->pp_pos.p, R, in->min_keep_len, (uint32_t *) e->pp_out_rows.p,
                      stride_out, (int32_t *) e->pp_out_len.p, (uint8_t *) e->pp_out_pair.p, tally, s);
    if ((rc = alga_check_launch(e, "k_pp_compact"))) return rc;
    HIP_TRY(e, hipEventRecord(e->ev[EV_EMIT], s));
    uint64_t *d_total = (uint64_t *) e->scan_scratch.p + scan_total_index(R);
    HIP_TRY(e, hipMemcpyAsync(&e->h_counters[8], d_total, sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(e, hipMemcpyAsync(e->h_counters, tally, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    HIP_TRY(e, hipStreamSynchronize(s));
    if (e->h_counters[1]) return alga_fail(e, ALGA_ERR_INVALID_ARGUMENT, "a read is kept but its reverse complement is removed (the reference asserts, src/main.cpp:171)");
    float ms = 0.f;
    (void) hipEventElapsedTime(&ms, e->ev[EV_START], e->ev[EV_EMIT]);
    out->d_words = (const uint32_t *) e->pp_out_rows.p;
    out->d_len = (const int32_t *) e->pp_out_len.p;
    out->d_pair_off = (const uint8_t *) e->pp_out_pair.p;
    out->n = (int32_t) (2 * e->h_counters[8]);
    out->stride_words = stride_out;
    out->removed_prefix = (int32_t) e->h_counters[0];
    out->removed_short = (int32_t) e->h_counters[3];
    out->max_len = (int32_t) e->h_counters[2];
    out->ms_device = ms;
    return ALGA_OK;
}
