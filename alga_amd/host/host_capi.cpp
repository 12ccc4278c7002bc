// alga_amd/host/host_capi.cpp -- C ABI of the host-side input stages (include/alga_amd.h, "input stages" section)
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/alga_amd.h"
#include "ingest.hpp"

extern "C" {

void alga_ingest_default_params(alga_ingest_params *p) {
    if (!p) return;
    alga_host::IngestParams d;
    p->trim_left = d.trim_left; p->trim_right = d.trim_right; p->remove_reads_with_n = d.remove_reads_with_n; p->rna = d.rna;
    p->scale = d.scale; p->min_overlap = d.min_overlap; p->rsoemo = d.rsoemo; p->remove_pref_reads = d.remove_pref_reads; p->threads = d.threads;
}

int alga_ingest_files(const char *file1, const char *file2, const alga_ingest_params *p, alga_node_set *out, char *errbuf, size_t errlen) {
    if (!file1 || !p || !out) return ALGA_ERR_INVALID_ARGUMENT;
    memset(out, 0, sizeof(*out));
    alga_host::IngestParams ip;
    ip.trim_left = p->trim_left; ip.trim_right = p->trim_right; ip.remove_reads_with_n = p->remove_reads_with_n; ip.rna = p->rna;
    ip.scale = p->scale; ip.min_overlap = p->min_overlap; ip.rsoemo = p->rsoemo; ip.remove_pref_reads = p->remove_pref_reads;
    ip.threads = p->threads < 1 ? 1 : p->threads;
    alga_host::NodeSet ns;
    std::string err = alga_host::ingest(file1, file2 ? file2 : "", ip, ns);
    if (!err.empty()) {
        if (errbuf && errlen) snprintf(errbuf, errlen, "%s", err.c_str());
        return ALGA_ERR_IO;
    }
    const size_t n = (size_t) ns.n;
    out->n = ns.n; out->stride_words = ns.stride;
    out->words = (uint32_t *) malloc(sizeof(uint32_t) * (n * (size_t) ns.stride + 1));
    out->len = (int32_t *) malloc(sizeof(int32_t) * (n + 1));
    out->pair_off = (uint8_t *) malloc(n + 1);
    if (!out->words || !out->len || !out->pair_off) { alga_free_node_set(out); return ALGA_ERR_OUT_OF_MEMORY; }
    if (n) {
        memcpy(out->words, ns.words.data(), sizeof(uint32_t) * n * (size_t) ns.stride);
        memcpy(out->len, ns.len.data(), sizeof(int32_t) * n);
        memcpy(out->pair_off, ns.pair_off.data(), n);
    }
    out->LEN = ns.LEN; out->min_overlap = ns.min_overlap; out->rsoemo = ns.rsoemo; out->li_kmer_length = ns.li_kmer_length;
    out->records = ns.records; out->removed_n = ns.removed_n; out->removed_str = ns.removed_str;
    out->removed_prefix = ns.removed_prefix; out->removed_short = ns.removed_short; out->avg_len = ns.avg_len;
    return ALGA_OK;
}

static alga_host::IngestParams to_host_params(const alga_ingest_params *p) {
    alga_host::IngestParams ip;
    ip.trim_left = p->trim_left; ip.trim_right = p->trim_right; ip.remove_reads_with_n = p->remove_reads_with_n; ip.rna = p->rna;
    ip.scale = p->scale; ip.min_overlap = p->min_overlap; ip.rsoemo = p->rsoemo; ip.remove_pref_reads = p->remove_pref_reads;
    ip.threads = p->threads < 1 ? 1 : p->threads;
    return ip;
}

int alga_parse_files(const char *file1, const char *file2, const alga_ingest_params *p, alga_parsed_reads *out, char *errbuf, size_t errlen) {
    if (!file1 || !p || !out) return ALGA_ERR_INVALID_ARGUMENT;
    memset(out, 0, sizeof(*out));
    alga_host::Parsed *P = new alga_host::Parsed();
    std::string err = alga_host::parse(file1, file2 ? file2 : "", to_host_params(p), *P);
    if (!err.empty()) {
        if (errbuf && errlen) snprintf(errbuf, errlen, "%s", err.c_str());
        delete P;
        return ALGA_ERR_IO;
    }
    out->n_nodes = (int64_t) (2 * P->R); out->stride_words = P->W; out->rows = P->rows.data(); out->len = P->len.data();
    out->paired = P->paired ? 1 : 0; out->records = P->records; out->removed_n = P->removed_n; out->removed_str = P->removed_str;
    out->LEN = P->LEN; out->min_overlap = P->min_overlap; out->rsoemo = P->rsoemo; out->li_kmer_length = P->li_kmer_length;
    out->avg_len = P->avg_len; out->owner = P;
    return ALGA_OK;
}

void alga_free_parsed_reads(alga_parsed_reads *pr) {
    if (!pr) return;
    delete (alga_host::Parsed *) pr->owner;
    memset(pr, 0, sizeof(*pr));
}

void alga_free_node_set(alga_node_set *ns) {
    if (!ns) return;
    free(ns->words); free(ns->len); free(ns->pair_off);
    memset(ns, 0, sizeof(*ns));
}

} // extern "C"
