// alga_amd/csrc/engine_ingest.hip -- C ABI entry of the GPU input stage N1 (kernels: ingest_kernels.hip)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>

#include "engine_internal.h"
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <string>

#include "ingest_kernels.h"
#include "parse_kernels.h"

using namespace alga;

namespace {

// N1 on device-resident rows (`rows`: n rows of `stride` words, `len`: n lengths, -1 = removed; both are modified / consumed):
// duplicate / prefix-read removal, id compaction, too-short reads emptied.  max_len / live: of the live input nodes.
int preprocess_impl(alga_engine *e, const uint32_t *rows, int stride, int32_t *len, uint64_t n, int max_len, uint64_t live, int remove_pref_reads,
                    int min_keep_len, hipStream_t s, alga_device_node_set *out) {
    int rc;
    const uint64_t R = n / 2;
    const int used_words = std::max(1, blocks_of(max_len));
    const int stride_out = hbm_row_stride(used_words);
    if ((rc = alga_ensure(e, e->pp_tally, 8 * sizeof(unsigned long long)))) return rc;
    HIP_TRY(e, hipMemsetAsync(e->pp_tally.p, 0, 8 * sizeof(unsigned long long), s));
    HIP_TRY(e, hipEventRecord(e->ev[EV_START], s));
    unsigned long long *tally = (unsigned long long *) e->pp_tally.p;
    const uint8_t *mark = nullptr;
    if (remove_pref_reads != 3 && live > 1) {
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
        launch_pp_mark(rows, stride, len, (const uint32_t *) e->pp_perm[cur].p, live, remove_pref_reads, (uint8_t *) e->pp_mark.p, s);
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
    alga_forget_node_set(e);
    if ((rc = alga_ensure(e, e->pp_out_rows, (n + 2) * (size_t) stride_out * sizeof(uint32_t)))) return rc;
    if ((rc = alga_ensure(e, e->pp_out_len, (n + 2) * sizeof(int32_t)))) return rc;
    if ((rc = alga_ensure(e, e->pp_out_pair, n + 16))) return rc;
    launch_pp_compact(rows, stride, len, (const uint32_t *) e->pp_keep.p, (const uint32_t *) e->pp_pos.p, R, min_keep_len, (uint32_t *) e->pp_out_rows.p,
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

struct MappedFile {
    const uint8_t *p = nullptr;
    size_t n = 0;
    int fd = -1;
    bool open_(const char *path) {
        fd = ::open(path, O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0) return false;
        n = (size_t) st.st_size;
        if (n == 0) return true;
        void *m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) return false;
        (void) madvise(m, n, MADV_SEQUENTIAL);
        p = (const uint8_t *) m;
        return true;
    }
    ~MappedFile() { if (p) munmap((void *) p, n); if (fd >= 0) ::close(fd); }
};

int lines_per_record_of(const char *path) {               // src/Params.cpp:315-333; 0 = a type this stage does not take
    std::string s(path);
    const size_t sl = s.rfind('/');
    const std::string base = sl == std::string::npos ? s : s.substr(sl + 1);
    const size_t dot = base.rfind('.');
    if (dot == std::string::npos) return 0;
    const std::string ext = base.substr(dot + 1);
    if (ext == "fasta") return 2;
    if (ext == "fastq" || ext == "fq") return 4;
    return 0;
}

} // namespace

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
    const uint64_t n = (uint64_t) in->n_nodes;
    const int stride = in->stride_words;
    // lengths decide the pass count and must fit the rows: one host pass (the caller's arrays are pageable memory anyway)
    int max_len = 0;
    uint64_t live = 0;
    for (uint64_t i = 0; i < n; i++) { const int l = in->len[i]; if (l >= 0) { live++; max_len = std::max(max_len, l); } }
    if ((int64_t) blocks_of(max_len) > (int64_t) stride) return alga_fail(e, ALGA_ERR_INVALID_ARGUMENT, "stride_words is smaller than the longest read needs");
    if ((rc = alga_ensure(e, e->pp_rows, n * (size_t) stride * sizeof(uint32_t)))) return rc;
    if ((rc = alga_ensure(e, e->pp_len, (n + 2) * sizeof(int32_t)))) return rc;
    if (n) {
        HIP_TRY(e, hipStreamSynchronize(s));
        if ((rc = alga_staged_h2d(e, e->pp_rows.p, in->rows, n * (size_t) stride * sizeof(uint32_t)))) return rc;
        if ((rc = alga_staged_h2d(e, e->pp_len.p, in->len, n * sizeof(int32_t)))) return rc;
    }
    return preprocess_impl(e, (const uint32_t *) e->pp_rows.p, stride, (int32_t *) e->pp_len.p, n, max_len, live, in->remove_pref_reads, in->min_keep_len, s, out);
}

// Files -> node set resident in HBM: the host only maps the files and moves their bytes; line splitting, trimming, the N / STR
// filters, packing, reverse complements (N2, parse_kernels.hip) and duplicate / prefix removal + compaction (N1) run on the GPU.
extern "C" int alga_ingest_device(alga_engine *e, const char *file1, const char *file2, const alga_ingest_params *p, alga_device_node_set *out,
                                  alga_ingest_info *info) {
    if (!e) return ALGA_ERR_INVALID_ARGUMENT;
    e->err.clear();
    if (!file1 || !p || !out || !info) return alga_fail(e, ALGA_ERR_INVALID_ARGUMENT, "arguments must not be NULL");
    memset(out, 0, sizeof(*out));
    memset(info, 0, sizeof(*info));
    const bool paired = file2 && file2[0];
    const int lpr = lines_per_record_of(file1);
    if (lpr == 0) return alga_fail(e, ALGA_ERR_UNSUPPORTED, "device ingest takes .fasta / .fastq / .fq files (others: alga_parse_files + alga_preprocess_nodes)");
    if (!p->remove_reads_with_n) return alga_fail(e, ALGA_ERR_UNSUPPORTED, "device ingest needs remove_reads_with_n = 1 (the random replacement of N draws from one generator in file order: alga_parse_files)");
    if (p->remove_pref_reads < 1 || p->remove_pref_reads > 3) return alga_fail(e, ALGA_ERR_INVALID_ARGUMENT, "remove_pref_reads must be 1, 2 or 3");
    using clk = std::chrono::steady_clock;
    const auto t0 = clk::now();
    HIP_TRY(e, hipSetDevice(e->device));
    hipStream_t s = e->own_stream;
    int rc;
    MappedFile mf[2];
    const int nf = paired ? 2 : 1;
    const char *paths[2] = {file1, file2};
    for (int f = 0; f < nf; f++) if (!mf[f].open_(paths[f])) return alga_fail(e, ALGA_ERR_IO, (std::string("cannot open ") + paths[f]).c_str());
    unsigned long long *hc = e->h_counters;                // pinned
    uint64_t n_nl[2] = {0, 0}, n_lines[2] = {0, 0}, n_rec[2] = {0, 0}, maxline = 0;
    HIP_TRY(e, hipStreamSynchronize(s));
    if ((rc = alga_ensure(e, e->counters, (CNT_TOTAL + 2) * sizeof(unsigned long long)))) return rc;
    unsigned long long *dc = (unsigned long long *) e->counters.p;
    double ms_upload = 0;
    for (int f = 0; f < nf; f++) {
        const uint64_t nb = mf[f].n;
        const auto tu = clk::now();
        if ((rc = alga_ensure(e, e->in_bytes[f], nb + 64))) return rc;
        if ((rc = alga_staged_h2d(e, e->in_bytes[f].p, mf[f].p, nb))) return rc;
        ms_upload += std::chrono::duration<double, std::milli>(clk::now() - tu).count();
        const uint64_t tiles = nl_tiles(nb);
        if ((rc = alga_ensure(e, e->in_tiles, (tiles + 2) * sizeof(uint32_t)))) return rc;
        if ((rc = alga_ensure(e, e->in_tile_off, (tiles + 2) * sizeof(uint32_t)))) return rc;
        if ((rc = alga_ensure(e, e->scan_scratch, scan_scratch_bytes(tiles)))) return rc;
        launch_nl_count((const uint8_t *) e->in_bytes[f].p, nb, (uint32_t *) e->in_tiles.p, s);
        if ((rc = alga_check_launch(e, "k_nl_count"))) return rc;
        launch_exclusive_scan((const uint32_t *) e->in_tiles.p, tiles, (uint32_t *) e->in_tile_off.p, (uint64_t *) e->scan_scratch.p, s);
        if ((rc = alga_check_launch(e, "scan(newlines)"))) return rc;
        HIP_TRY(e, hipMemcpyAsync(&hc[0], (uint64_t *) e->scan_scratch.p + scan_total_index(tiles), sizeof(uint64_t), hipMemcpyDeviceToHost, s));
        HIP_TRY(e, hipStreamSynchronize(s));
        n_nl[f] = nb ? hc[0] : 0;
        if (n_nl[f] >= (1ull << 32) - 16) return alga_fail(e, ALGA_ERR_CAPACITY, "more than 2^32 lines in one file");
        n_lines[f] = n_nl[f] + ((nb && mf[f].p[nb - 1] != '\n') ? 1 : 0);
        if ((rc = alga_ensure(e, e->in_nl[f], (n_nl[f] + 2) * sizeof(unsigned long long)))) return rc;
        launch_nl_write((const uint8_t *) e->in_bytes[f].p, nb, (const uint32_t *) e->in_tile_off.p, (unsigned long long *) e->in_nl[f].p, s);
        if ((rc = alga_check_launch(e, "k_nl_write"))) return rc;
        // records: the sequence line of record r is line r * lpr + 1; the input ends at the first empty one (InputReader.cpp:284)
        const uint64_t n_cand = n_lines[f] >= 2 ? (n_lines[f] - 2) / (uint64_t) lpr + 1 : 0;
        HIP_TRY(e, hipMemsetAsync(dc, 0, sizeof(unsigned long long), s));
        HIP_TRY(e, hipMemsetAsync(dc + 1, 0xFF, sizeof(unsigned long long), s));
        launch_line_stats((const unsigned long long *) e->in_nl[f].p, n_nl[f], nb, n_lines[f], lpr, n_cand, dc, s);
        if ((rc = alga_check_launch(e, "k_line_stats"))) return rc;
        HIP_TRY(e, hipMemcpyAsync(hc, dc, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
        HIP_TRY(e, hipStreamSynchronize(s));
        maxline = std::max<uint64_t>(maxline, hc[0]);
        n_rec[f] = std::min<uint64_t>(n_cand, hc[1]);
    }
    if (paired && n_rec[0] != n_rec[1]) return alga_fail(e, ALGA_ERR_IO, "paired files differ in record count");
    const uint64_t R = paired ? 2 * n_rec[0] : n_rec[0], n = 2 * R;
    if (n >= 0x7FFFFFFEull) return alga_fail(e, ALGA_ERR_CAPACITY, "too many nodes");
    if (maxline > (1u << 20)) return alga_fail(e, ALGA_ERR_CAPACITY, "a sequence line is longer than 1 Mi characters");
    const int W = std::max(1, blocks_of((int) maxline));
    if ((rc = alga_ensure(e, e->pp_rows, (n + 2) * (size_t) W * sizeof(uint32_t)))) return rc;
    if ((rc = alga_ensure(e, e->pp_len, (n + 2) * sizeof(int32_t)))) return rc;
    HIP_TRY(e, hipMemsetAsync(dc, 0, 4 * sizeof(unsigned long long), s));
    HIP_TRY(e, hipMemsetAsync(dc + 4, 0xFF, sizeof(unsigned long long), s));
    const ParseCfg pc{p->trim_left, p->trim_right, p->rna};
    for (int f = 0; f < nf; f++) {
        launch_parse_records((const uint8_t *) e->in_bytes[f].p, (const unsigned long long *) e->in_nl[f].p, n_nl[f], mf[f].n, n_lines[f], lpr, n_rec[f], f,
                             paired ? 1 : 0, pc, (uint32_t *) e->pp_rows.p, W, (int32_t *) e->pp_len.p, dc, s);
        if ((rc = alga_check_launch(e, "k_parse_records"))) return rc;
    }
    HIP_TRY(e, hipMemsetAsync(dc + 5, 0, 2 * sizeof(unsigned long long), s));
    launch_len_stats((const int32_t *) e->pp_len.p, n, dc + 5, s);
    if ((rc = alga_check_launch(e, "k_len_stats"))) return rc;
    HIP_TRY(e, hipMemcpyAsync(hc, dc, 7 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    HIP_TRY(e, hipStreamSynchronize(s));
    if (hc[4] != ~0ull) {
        char msg[160];
        snprintf(msg, sizeof msg, "s[i] = %c   but should be A,C,G,T,N or U (record %llu)", (char) (hc[4] & 0xFF), (unsigned long long) (hc[4] >> 8));
        return alga_fail(e, ALGA_ERR_IO, msg);             // the reference prints this and exits (InputReader.cpp:323-326)
    }
    const auto t1 = clk::now();
    // parameters of src/main.cpp:93-115 (mixed float / int arithmetic, truncating)
    const uint64_t kept = hc[3], sum_len = hc[2];
    info->records = (int64_t) R;
    info->removed_n = (int32_t) (2 * hc[0]);
    info->removed_str = (int32_t) (2 * hc[1]);
    info->avg_len = kept ? (double) (2 * sum_len) / (double) (2 * kept) : 0.0;
    info->LEN = (int32_t) (info->avg_len + p->trim_left + p->trim_right);
    int Lmin = p->min_overlap, rso = p->rsoemo, likl;
    if (Lmin == -1) {
        const int L = (int) ((float) info->LEN * p->scale);
        const int RSOEMO = (int) ((float) info->LEN * (p->scale + 1) / 2);
        likl = std::min(2 * L / 3, 60);
        Lmin = L;
        if (rso == -1) rso = RSOEMO;
    } else {
        likl = Lmin;
        if (rso == -1) rso = (Lmin + info->LEN) / 2;
    }
    info->min_overlap = Lmin; info->rsoemo = rso; info->li_kmer_length = likl;
    info->paired = paired ? 1 : 0;
    rc = preprocess_impl(e, (const uint32_t *) e->pp_rows.p, W, (int32_t *) e->pp_len.p, n, (int) hc[5], hc[6], p->remove_pref_reads, 3 + likl, s, out);
    if (rc) return rc;
    const auto t2 = clk::now();
    info->ms_upload = ms_upload;
    info->ms_parse = std::chrono::duration<double, std::milli>(t1 - t0).count();
    info->ms_preprocess = std::chrono::duration<double, std::milli>(t2 - t1).count();
    return ALGA_OK;
}
