/*
 * oracle/alga_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see alga_oracle.h).
 *
 * Single-threaded CPU restatement of swacisko/ALGA's overlap-graph construction, in the
 * --threads=1 processing order (the canonical, run-to-run deterministic order of the reference).
 * Pinned by reference dumps under tests/golden/ (tools/make_golden.py).
 */
#define _GNU_SOURCE
#include "alga_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

/* ------------------------------------------------------------------------------------------ */
/* small helpers                                                                               */
/* ------------------------------------------------------------------------------------------ */

static inline int blocks_of(int len_nt) { /* Bitset::blocks(), include/DataStructures/Bitset.h:206 */
    int nbits = len_nt << 1;
    return nbits == 0 ? 0 : ((nbits - 1) >> 5) + 1;
}

static inline int nt_at(const uint32_t *w, int pos) { /* Read::operator[], src/DataStructures/Read.cpp:127-129 */
    return (int) ((w[pos >> 4] >> ((pos & 15) << 1)) & 3u);
}

/* 32 bits of the bit string `w` (nw valid words, zero beyond) starting at bit position `bit` */
static inline uint32_t bits32_at(const uint32_t *w, int nw, int bit) {
    int q = bit >> 5, r = bit & 31;
    uint32_t lo = q < nw ? w[q] : 0u;
    if (r == 0) return lo;
    uint32_t hi = (q + 1) < nw ? w[q + 1] : 0u;
    return (lo >> r) | (hi << (32 - r));
}

/* include/Utils/MyUtils.h:160-170 : minimal period by the KMP prefix function */
int oracle_min_period(const char *s, int n) {
    if (n <= 0) return 0;
    int *pre = (int *) malloc(sizeof(int) * (size_t) (n + 1));
    int k = 0;
    pre[0] = 0;
    pre[1] = 0;
    for (int q = 1; q < n; q++) {
        while (k > 0 && s[k] != s[q]) k = pre[k];
        if (s[k] == s[q]) k++;
        pre[q + 1] = k;
    }
    int res = n - pre[n];
    free(pre);
    return res;
}

/* src/DataStructures/Read.cpp:40-68 : C sets bit 2i, G sets bit 2i+1, T both, anything else 00 */
void oracle_pack(const char *s, int n, uint32_t *words, int W) {
    memset(words, 0, sizeof(uint32_t) * (size_t) W);
    for (int i = 0; i < n; i++) {
        uint32_t v;
        switch (s[i]) {
            case 'C': v = 1; break;
            case 'G': v = 2; break;
            case 'T': v = 3; break;
            default:  v = 0; break;
        }
        words[i >> 4] |= v << ((i & 15) << 1);
    }
}

void oracle_default_ingest_params(oracle_ingest_params *p) {
    p->trim_left = 3;            /* src/Params.cpp:729 */
    p->trim_right = 3;           /* src/Params.cpp:730 */
    p->remove_reads_with_n = 1;  /* src/Params.cpp:740 */
    p->rna = 0;
    p->scale = 0.55f;            /* src/Params.cpp:678 */
    p->min_overlap = -1;
    p->rsoemo = -1;
    p->remove_pref_reads = 2;    /* PREF_READS_ALL_PREFIX_READS, src/Params.cpp:758 */
}

/* ------------------------------------------------------------------------------------------ */
/* input: src/IO/InputReader.cpp                                                               */
/* ------------------------------------------------------------------------------------------ */

enum { FT_MY_INPUT = 0, FT_FASTA, FT_PFASTA, FT_FASTQ };

static int file_type_of(const char *path) { /* src/Params.cpp:315-333 */
    const char *base = strrchr(path, '/');
    base = base ? base + 1 : path;
    const char *dot = strrchr(base, '.');
    if (!dot) return FT_MY_INPUT;
    dot++;
    if (!strcmp(dot, "fasta")) return FT_FASTA;
    if (!strcmp(dot, "pfasta")) return FT_PFASTA;
    if (!strcmp(dot, "fastq") || !strcmp(dot, "fq")) return FT_FASTQ;
    return FT_MY_INPUT;
}

typedef struct { char *buf; size_t size, pos; } textbuf;

static int slurp(const char *path, textbuf *t) {
    FILE *f = fopen(path, "rb");
    if (!f) return -1;
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    t->buf = (char *) malloc((size_t) sz + 1);
    t->size = fread(t->buf, 1, (size_t) sz, f);
    t->buf[t->size] = 0;
    t->pos = 0;
    fclose(f);
    return 0;
}

/* std::getline: returns the line without '\n'; at EOF yields an empty string */
static void next_line(textbuf *t, const char **s, int *n) {
    if (t->pos >= t->size) { *s = t->buf + t->size; *n = 0; return; }
    const char *b = t->buf + t->pos;
    const char *e = (const char *) memchr(b, '\n', t->size - t->pos);
    if (!e) { *n = (int) (t->size - t->pos); t->pos = t->size; }
    else    { *n = (int) (e - b); t->pos += (size_t) (*n) + 1; }
    *s = b;
}

static void next_token(textbuf *t, const char **s, int *n) { /* istream >> string */
    while (t->pos < t->size && (t->buf[t->pos] == ' ' || t->buf[t->pos] == '\n' || t->buf[t->pos] == '\t' ||
                                t->buf[t->pos] == '\r' || t->buf[t->pos] == '\v' || t->buf[t->pos] == '\f')) t->pos++;
    size_t b = t->pos;
    while (t->pos < t->size && !(t->buf[t->pos] == ' ' || t->buf[t->pos] == '\n' || t->buf[t->pos] == '\t' ||
                                 t->buf[t->pos] == '\r' || t->buf[t->pos] == '\v' || t->buf[t->pos] == '\f')) t->pos++;
    *s = t->buf + b;
    *n = (int) (t->pos - b);
}

/* InputReader::readOneRead1, src/IO/InputReader.cpp:142-180 */
static void read_one(textbuf *t, int type, const char **s, int *n) {
    const char *e; int en;
    switch (type) {
        case FT_MY_INPUT: next_token(t, s, n); break;
        case FT_FASTA:
        case FT_PFASTA:   next_line(t, &e, &en); next_line(t, s, n); break; /* ADD_PAIRED_READS==1 */
        default:          next_line(t, &e, &en); next_line(t, s, n); next_line(t, &e, &en); next_line(t, &e, &en); break;
    }
}

typedef struct {   /* growing list of reads as read from the files; len<0 = nullptr */
    int32_t *len; uint32_t **w; int64_t n, cap;
} readlist;

static void rl_push(readlist *rl, const char *s, int n, int is_null) {
    if (rl->n == rl->cap) {
        rl->cap = rl->cap ? rl->cap * 2 : 1024;
        rl->len = (int32_t *) realloc(rl->len, sizeof(int32_t) * (size_t) rl->cap);
        rl->w = (uint32_t **) realloc(rl->w, sizeof(uint32_t *) * (size_t) rl->cap);
    }
    if (is_null) { rl->len[rl->n] = -1; rl->w[rl->n] = NULL; }
    else {
        int W = blocks_of(n); if (W == 0) W = 1;
        rl->w[rl->n] = (uint32_t *) malloc(sizeof(uint32_t) * (size_t) W);
        oracle_pack(s, n, rl->w[rl->n], W);
        rl->len[rl->n] = n;
    }
    rl->n++;
}

/* InputReader::readParallelJob with THREADS=1, thread_id=0 (src/IO/InputReader.cpp:272-391) */
static int read_file(const char *path, int type, const oracle_ingest_params *p, readlist *rl,
                     int64_t *records, int *n_removed, int *str_removed) {
    textbuf t;
    if (slurp(path, &t)) { fprintf(stderr, "oracle: cannot open %s\n", path); return -1; }
    uint32_t rng = 1; /* std::minstd_rand0(0): a zero seed is replaced by 1 */
    char *s = NULL; int cap = 0;
    for (;;) {
        const char *ls; int ln;
        read_one(&t, type, &ls, &ln);
        if (ln == 0) break;                                              /* :284 */
        if (ln + 1 > cap) { cap = ln + 64; s = (char *) realloc(s, (size_t) cap); }
        int b = 0; while (b < ln && ls[b] == ' ') b++;                   /* :286-288 */
        int e = b; while (e < ln && ls[e] != ' ') e++;                   /* :289-291 */
        int n = e - b;
        memcpy(s, ls + b, (size_t) n);
        if (!(n < p->trim_left + p->trim_right + 10)) {                  /* :298-303 */
            int l = p->trim_left < n ? p->trim_left : n;
            memmove(s, s + l, (size_t) (n - l)); n -= l;
            int r = p->trim_right < n ? p->trim_right : n;
            n -= r;
        }
        s[n] = 0;
        int containsN = 0;
        for (int i = 0; i < n; i++) {                                    /* :321-336 */
            char c = s[i];
            if (c != 'A' && c != 'C' && c != 'G' && c != 'T' && c != 'N' && c != 'U') {
                fprintf(stderr, "oracle: s[i] = %c but should be A,C,G,T,N or U\n", c);
                free(s); free(t.buf); return -2;
            }
            if (c == 'N' && p->remove_reads_with_n) containsN = 1;
            else if (c == 'N') { rng = (uint32_t) (((uint64_t) rng * 16807u) % 2147483647u); s[i] = "ACGT"[rng & 3]; }
            else if (p->rna && c == 'U') s[i] = 'T';
        }
        (*records)++;
        int is_null;
        if (p->remove_reads_with_n && containsN) { is_null = 1; (*n_removed)++; }   /* :344-346 */
        else if (oracle_min_period(s, n) <= 20) { is_null = 1; (*str_removed)++; }  /* :348-353 */
        else is_null = 0;
        rl_push(rl, s, n, is_null);
        /* reverse complement twin, :363-377 (getComplimentaryString :23-33: only A,C,G,T are mapped) */
        for (int i = 0, j = n - 1; i < j; i++, j--) { char c = s[i]; s[i] = s[j]; s[j] = c; }
        for (int i = 0; i < n; i++) {
            switch (s[i]) { case 'A': s[i] = 'T'; break; case 'C': s[i] = 'G'; break;
                            case 'G': s[i] = 'C'; break; case 'T': s[i] = 'A'; break; default: break; }
        }
        if (p->remove_reads_with_n && containsN) is_null = 1;
        else is_null = oracle_min_period(s, n) <= 20;
        rl_push(rl, s, n, is_null);
    }
    free(s); free(t.buf);
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* duplicate / prefix read removal: src/IO/ReadPreprocess.cpp:13-152                            */
/* ------------------------------------------------------------------------------------------ */

typedef struct { const readlist *rl; } sortctx;

/* comparator of getSortedReads (:115-132). The bucket pre-pass (:86-108) orders by the first
 * 15 bits read LSB-first, which is a prefix of this same order, so one sort reproduces it. */
static int cmp_reads(const void *pa, const void *pb, void *vctx) {
    const readlist *rl = ((sortctx *) vctx)->rl;
    int64_t a = *(const int64_t *) pa, b = *(const int64_t *) pb;
    const uint32_t *wa = rl->w[a], *wb = rl->w[b];
    int ba = blocks_of(rl->len[a]), bb = blocks_of(rl->len[b]);
    int m = ba < bb ? ba : bb;
    for (int p = 0; p < m; p++) {
        if (wa[p] != wb[p]) {
            int ind = __builtin_ctz(wa[p] ^ wb[p]);
            int bita = (wa[p] >> ind) & 1, bitb = (wb[p] >> ind) & 1;
            return bita < bitb ? -1 : 1;
        }
    }
    if (rl->len[a] != rl->len[b]) return rl->len[a] < rl->len[b] ? -1 : 1;
    return a < b ? -1 : (a > b ? 1 : 0);
}

/* Bitset::mismatch >> 1 (src/DataStructures/Bitset.cpp:858-877) */
static int lcp_nt(const readlist *rl, int64_t a, int64_t b) {
    int ba = blocks_of(rl->len[a]), bb = blocks_of(rl->len[b]);
    int m = ba < bb ? ba : bb;
    int64_t ind = 1000000000;
    for (int i = 0; i < m; i++) {
        if (rl->w[a][i] != rl->w[b][i]) { ind = (int64_t) i * 32 + __builtin_ctz(rl->w[a][i] ^ rl->w[b][i]); break; }
    }
    int64_t ms = 2 * (int64_t) (rl->len[a] < rl->len[b] ? rl->len[a] : rl->len[b]);
    if (ind < ms) return (int) (ind >> 1);
    return (int) (ms >> 1);
}

static int remove_prefix_reads(readlist *rl, int mode) {
    int64_t nv = 0;
    for (int64_t i = 0; i < rl->n; i++) if (rl->len[i] >= 0) nv++;
    int64_t *ord = (int64_t *) malloc(sizeof(int64_t) * (size_t) (nv ? nv : 1));
    nv = 0;
    for (int64_t i = 0; i < rl->n; i++) if (rl->len[i] >= 0) ord[nv++] = i;
    sortctx ctx = { rl };
    qsort_r(ord, (size_t) nv, sizeof(int64_t), cmp_reads, &ctx);
    uint8_t *mark = (uint8_t *) calloc((size_t) (rl->n ? rl->n : 1), 1);
    for (int64_t i = 0; i + 1 < nv; i++) {                               /* lcpFun :28-51 */
        int64_t a = ord[i], b = ord[i + 1];
        int l = lcp_nt(rl, a, b);
        if (mode == 1) { if (l == rl->len[a] && rl->len[a] == rl->len[b]) mark[a] = 1; }
        else if (mode == 2 && l == rl->len[a]) {
            mark[a] = 1;
            if (rl->len[a] < rl->len[b]) mark[a ^ 1] = 1;               /* getIdOfCompRevRead */
        }
    }
    int cnt = 0;
    for (int64_t i = 0; i < rl->n; i++) {                                /* src/main.cpp:136-141 */
        if (mark[i]) {
            cnt++;
            if (rl->len[i] >= 0) { free(rl->w[i]); rl->w[i] = NULL; rl->len[i] = -1; }
        }
    }
    free(mark); free(ord);
    return cnt;
}

/* ------------------------------------------------------------------------------------------ */
/* oracle_ingest                                                                               */
/* ------------------------------------------------------------------------------------------ */

int oracle_ingest(const char *file1, const char *file2, const oracle_ingest_params *p, oracle_nodes *out) {
    memset(out, 0, sizeof(*out));
    readlist rl = {0};
    int type = file_type_of(file1);
    int64_t records = 0; int nrem = 0, strrem = 0;
    if (read_file(file1, type, p, &rl, &records, &nrem, &strrem)) return -1;
    if (file2 && file2[0] && type != FT_PFASTA) {                        /* :53-76 */
        int64_t n1 = rl.n;
        if (read_file(file2, type, p, &rl, &records, &nrem, &strrem)) return -1;
        int64_t N = rl.n;
        if (N != 2 * n1) { fprintf(stderr, "oracle: paired files differ in record count\n"); return -3; }
        int32_t *len2 = (int32_t *) malloc(sizeof(int32_t) * (size_t) N);
        uint32_t **w2 = (uint32_t **) malloc(sizeof(uint32_t *) * (size_t) N);
        for (int64_t i = 0; 4 * i < N; i++) {
            int64_t src[4] = { 2 * i, 2 * i + 1, N / 2 + 2 * i, N / 2 + 2 * i + 1 };
            for (int k = 0; k < 4; k++) { len2[4 * i + k] = rl.len[src[k]]; w2[4 * i + k] = rl.w[src[k]]; }
        }
        free(rl.len); free(rl.w); rl.len = len2; rl.w = w2; rl.cap = N;
    }
    for (int64_t i = 0; i + 1 < rl.n; i += 2) {                          /* :78-80 : [r, rc] -> [rc, r] */
        int32_t tl = rl.len[i]; rl.len[i] = rl.len[i + 1]; rl.len[i + 1] = tl;
        uint32_t *tw = rl.w[i]; rl.w[i] = rl.w[i + 1]; rl.w[i + 1] = tw;
    }
    /* src/main.cpp:93-115, include/Global.h:133-145 */
    double sum = 0; int64_t cnt = 0;
    for (int64_t i = 0; i < rl.n; i++) if (rl.len[i] >= 0) { sum += rl.len[i]; cnt++; }
    double avg = cnt ? sum / (double) cnt : 0.0;
    int LEN = (int) (avg + p->trim_left + p->trim_right);
    int Lmin = p->min_overlap, rso = p->rsoemo, likl;
    if (Lmin == -1) {
        int L = (int) ((float) LEN * p->scale);
        int RSOEMO = (int) ((float) LEN * (p->scale + 1) / 2);
        likl = (2 * L / 3) < 60 ? (2 * L / 3) : 60;
        Lmin = L;
        if (rso == -1) rso = RSOEMO;
    } else {
        likl = Lmin;                                                     /* src/Params.cpp:447-457 (mfup / -l) */
        if (rso == -1) rso = (Lmin + LEN) / 2;
    }
    int removed_prefix = 0;
    if (p->remove_pref_reads != 3) removed_prefix = remove_prefix_reads(&rl, p->remove_pref_reads);

    /* compaction, src/main.cpp:150-232 (in-place copy of surviving [rc, r] pairs in order) */
    int64_t nn = 0;
    for (int64_t i = 0; i + 1 < rl.n; i += 2) if (rl.len[i] >= 0) nn += 2;
    int maxlen = 0;
    for (int64_t i = 0; i < rl.n; i++) if (rl.len[i] > maxlen) maxlen = rl.len[i];
    int W = blocks_of(maxlen); if (W == 0) W = 1;
    out->n = (int32_t) nn; out->W = W;
    out->words = (uint32_t *) calloc((size_t) (nn ? nn : 1) * (size_t) W, sizeof(uint32_t));
    out->len = (int32_t *) calloc((size_t) (nn ? nn : 1), sizeof(int32_t));
    out->pair_off = (uint8_t *) calloc((size_t) (nn ? nn : 1), 1);
    int64_t bi = 0;
#define COPY_PAIR(i_, po_) do {                                                                      \
        for (int k_ = 0; k_ < 2; k_++) {                                                             \
            int64_t s_ = (i_) + k_;                                                                  \
            if (rl.len[s_] < 0) { fprintf(stderr, "oracle: read %lld kept but its twin is removed "  \
                                  "(reference asserts here, src/main.cpp:171)\n", (long long) s_);   \
                                  return -4; }                                                       \
            out->len[bi] = rl.len[s_];                                                               \
            memcpy(out->words + (size_t) bi * W, rl.w[s_], sizeof(uint32_t) * (size_t) blocks_of(rl.len[s_])); \
            out->pair_off[bi] = (uint8_t) (po_);                                                     \
            bi++;                                                                                    \
        } } while (0)
    for (int64_t i = 0; i + 1 < rl.n; i += 2) {
        if (rl.len[i] < 0) continue;
        if ((i & 3) == 0) {
            if (i + 2 < rl.n && rl.len[i + 2] >= 0) { COPY_PAIR(i, 1); COPY_PAIR(i + 2, 2); }
            else COPY_PAIR(i, 0);
        } else if (rl.len[i - 2] < 0) COPY_PAIR(i, 0);
    }
#undef COPY_PAIR
    if (bi != nn) { fprintf(stderr, "oracle: compaction mismatch %lld vs %lld\n", (long long) bi, (long long) nn); return -5; }

    /* src/main.cpp:253-266 : reads shorter than LI_KMER_INTERVALS(3) + LI_KMER_LENGTH are deleted
     * (their node slots stay) */
    for (int64_t i = 0; i < nn; i++) {
        if (out->len[i] < 3 + likl) { out->len[i] = 0; memset(out->words + (size_t) i * W, 0, sizeof(uint32_t) * (size_t) W); }
    }
    out->LEN = LEN; out->min_overlap = Lmin; out->rsoemo = rso; out->li_kmer_length = likl;
    out->reads_in_file = records; out->removed_n = 2 * nrem; out->removed_str = 2 * strrem;
    out->removed_prefix = removed_prefix; out->avg_len = avg;
    for (int64_t i = 0; i < rl.n; i++) free(rl.w[i]);
    free(rl.w); free(rl.len);
    return 0;
}

void oracle_free_nodes(oracle_nodes *nd) {
    free(nd->words); free(nd->len); free(nd->pair_off);
    memset(nd, 0, sizeof(*nd));
}

/* ------------------------------------------------------------------------------------------ */
/* Graph container subset: src/DataStructures/Graph.cpp                                        */
/* ------------------------------------------------------------------------------------------ */

typedef struct { int32_t nbr, off; } pii;
typedef struct { pii *a; int32_t size, cap; } adj;

static inline void adj_push(adj *v, int32_t nbr, int32_t off) { /* pushDirectedEdge :73-75 */
    if (v->size == v->cap) { v->cap = v->cap ? v->cap * 2 : 4; v->a = (pii *) realloc(v->a, sizeof(pii) * (size_t) v->cap); }
    v->a[v->size].nbr = nbr; v->a[v->size].off = off; v->size++;
}

static int cmp_pii(const void *x, const void *y) {
    const pii *a = (const pii *) x, *b = (const pii *) y;
    if (a->nbr != b->nbr) return a->nbr < b->nbr ? -1 : 1;
    if (a->off != b->off) return a->off < b->off ? -1 : 1;
    return 0;
}

static void retain_only_smallest_offset(adj *G, int n) { /* :348-387 */
    for (int i = 0; i < n; i++) {
        adj *v = &G[i];
        if (v->size > 1) qsort(v->a, (size_t) v->size, sizeof(pii), cmp_pii);
        int q = 0, p = 0;
        while (p < v->size) {
            v->a[q++] = v->a[p++];
            while (p < v->size && v->a[p - 1].nbr == v->a[p].nbr) p++;
        }
        v->size = q;
    }
}

static void reverse_graph_in_place(adj *G, int n) { /* :926-971 */
    int32_t *degs = (int32_t *) malloc(sizeof(int32_t) * (size_t) (n ? n : 1));
    for (int i = 0; i < n; i++) degs[i] = G[i].size;
    for (int j = 0; j < n; j++) {
        int d = degs[j];
        if (d == 0) continue;
        pii *neigh = (pii *) malloc(sizeof(pii) * (size_t) d);
        memcpy(neigh, G[j].a, sizeof(pii) * (size_t) d);
        for (int i = 0; i < d; i++) adj_push(&G[neigh[i].nbr], j, neigh[i].off);
        memmove(G[j].a, G[j].a + d, sizeof(pii) * (size_t) (G[j].size - d));
        G[j].size -= d;
        free(neigh);
    }
    free(degs);
}

static int64_t count_edges(const adj *G, int n) { /* :654-681 */
    int64_t r = 0;
    for (int i = 0; i < n; i++) r += G[i].size;
    return r;
}

/* ------------------------------------------------------------------------------------------ */
/* GraphCreatorPrefSuf: src/GraphCreators/GraphCreatorPrefSuf.cpp                               */
/* ------------------------------------------------------------------------------------------ */

#define MAX_HASH 1000000000000000003ULL /* Params::MAX_HASH_CONSIDERED, src/Params.cpp:721 */
#define MAX_ADD  1000000007u            /* MAX_ADDITIONAL_HASH, GraphCreatorPrefSuf.h:42      */
#define SOES     3                      /* GraphCreatorPrefSuf.h:62                           */
#define MAX_BLOCKS 10                   /* GraphCreatorPrefSuf.cpp:361                        */

typedef struct {
    const uint32_t *words; const int32_t *len; int n, W;
    uint8_t *from, *to;
    uint64_t *pk, *sk; uint32_t *pka, *ska;
    int buckets; int32_t *bstart, *bitems;
    adj *G; uint8_t *to_remove; int32_t *toR; int toR_n, toR_cap;
    int Lmin, rsoemo, cur, maxlen;
    uint64_t fac; uint32_t faca;
    oracle_graph *st;
} ps;

static int upd_prefix(ps *s, int id, int L, uint64_t fac, uint32_t faca) { /* :213-223 */
    if (L > s->len[id]) return 0;
    int x = nt_at(s->words + (size_t) id * s->W, L - 1);
    s->pk[id] += (uint64_t) x * fac;
    if (s->pk[id] >= MAX_HASH) s->pk[id] %= MAX_HASH;
    s->pka[id] += (uint32_t) x * faca;
    if (s->pka[id] >= MAX_ADD) s->pka[id] %= MAX_ADD;
    return 1;
}

static int upd_suffix(ps *s, int id, int L) { /* :225-236, MIN_OFFSET_FOR_ALIGNMENT = 0 */
    if (L > s->len[id]) return 0;
    int x = nt_at(s->words + (size_t) id * s->W, s->len[id] - L);
    s->sk[id] <<= 2; s->sk[id] += (uint64_t) x;
    if (s->sk[id] >= MAX_HASH) s->sk[id] %= MAX_HASH;
    s->ska[id] <<= 2; s->ska[id] += (uint32_t) x;
    if (s->ska[id] >= MAX_ADD) s->ska[id] %= MAX_ADD;
    return 1;
}

static void create_initial_state(ps *s) { /* :129-211 */
    for (int i = 0; i < s->n; i++) {
        if (s->len[i] > 0) { s->pk[i] = s->sk[i] = 0; s->pka[i] = s->ska[i] = 0; }
        else { s->pk[i] = s->sk[i] = ~0ULL; s->pka[i] = s->ska[i] = ~0u; }
    }
    for (int i = 0; i < s->n; i++) {
        if (!(s->from[i] || s->to[i])) continue;
        uint64_t fac = 1; uint32_t faca = 1; int cur = 0;
        for (int l = 0; l < s->Lmin - 1; l++) {
            cur++;
            int pu = s->to[i] ? upd_prefix(s, i, cur, fac, faca) : 0;
            if (!pu) s->to[i] = 0;
            int su = s->from[i] ? upd_suffix(s, i, cur) : 0;
            if (!su) s->from[i] = 0;
            fac <<= 2; if (fac >= MAX_HASH) fac %= MAX_HASH;
            uint64_t fa = (uint64_t) faca << 2; if (fa >= MAX_ADD) fa %= MAX_ADD; faca = (uint32_t) fa;
        }
    }
    s->cur = 0; s->fac = 1; s->faca = 1;
    for (int l = 0; l < s->Lmin - 1; l++) {
        s->cur++;
        s->fac <<= 2; if (s->fac >= MAX_HASH) s->fac %= MAX_HASH;
        uint64_t fa = (uint64_t) s->faca << 2; if (fa >= MAX_ADD) fa %= MAX_ADD; s->faca = (uint32_t) fa;
    }
}

/* "no mismatch among the first `pos` bits" of B versus (A >> shift_bits)
 * == !Bitset::mismatchBounded (src/DataStructures/Bitset.cpp:879-903) on the operands built at
 * GraphCreatorPrefSuf.cpp:434-451 (setBlock/getBlock/operator<<= : Bitset.cpp:116-163,905-909) */
static int equal_first_bits(const uint32_t *B, int nwB, const uint32_t *A, int nwA, int shift_bits, int pos) {
    for (int k = 0; k * 32 < pos; k++) {
        uint32_t a = bits32_at(A, nwA, shift_bits + 32 * k);
        uint32_t b = k < nwB ? B[k] : 0u;
        uint32_t x = a ^ b;
        int rem = pos - 32 * k;
        if (rem < 32) x &= (1u << rem) - 1u;
        if (x) return 0;
    }
    return 1;
}

static void add_edges_for_suffix(ps *s, int suffId) { /* nextPrefSufIterationJobAddEdges :369-487 */
    int L = s->cur;
    int upd = 0;
    if (s->from[suffId]) upd = upd_suffix(s, suffId, L);
    if (!upd) { s->from[suffId] = 0; return; }
    const uint32_t *rB = s->words + (size_t) suffId * s->W;
    int lenB = s->len[suffId];
    int b = (int) (s->sk[suffId] & (uint64_t) (s->buckets - 1));
    int offset = lenB - L;
    uint64_t sh = s->sk[suffId]; uint32_t sha = s->ska[suffId];
    s->st->bucket_entries_scanned += s->bstart[b + 1] - s->bstart[b];
    for (int q = s->bstart[b]; q < s->bstart[b + 1]; q++) {
        int prefId = s->bitems[q];
        if (prefId == suffId || s->pk[prefId] != sh || s->pka[prefId] != sha) continue;
        s->st->hash_equal_pairs++;
        /* Read::calculateReadOverlap, include/DataStructures/Read.h:84-86 */
        int lenC = s->len[prefId];
        int ov = (lenB < lenC + offset ? lenB : lenC + offset) - offset;
        if (ov < L) continue;
        if (L < s->rsoemo) {                                             /* :397-401 */
            adj *v = &s->G[suffId];
            if (v->size == SOES) { memmove(v->a, v->a + 1, sizeof(pii) * (size_t) (v->size - 1)); v->size--; }
            adj_push(v, prefId, offset);
        } else {                                                         /* :403-483 */
            adj *lst = &s->G[prefId];
            if (offset > 0) {
                int cnt = lst->size;
                s->st->transitive_checks += cnt;
                for (int t = 0; t < cnt; t++) {
                    int A = lst->a[t].nbr;
                    int offAC = lst->a[t].off;
                    int offsetDiff = offAC - offset;
                    if (offsetDiff < 0) continue;
                    if (A == suffId) continue;
                    int lenA = s->len[A];
                    if (lenB + offsetDiff - lenA < 0) continue;          /* Read::getRightOffset, Read.h:91 */
                    const uint32_t *rA = s->words + (size_t) A * s->W;
                    unsigned begBlock = (unsigned) (offsetDiff << 1) >> 5;
                    unsigned endBlock = (unsigned) (offAC << 1) >> 5;
                    int pos = (endBlock - begBlock + 1 < MAX_BLOCKS) ? (offset << 1) : ((lenA - offsetDiff) << 1);
                    int eq = equal_first_bits(rB, blocks_of(lenB), rA, blocks_of(lenA), offsetDiff << 1, pos);
                    if (eq) {
                        if (!s->to_remove[A]) {
                            s->to_remove[A] = 1;
                        }
                        if (s->toR_n == s->toR_cap) { s->toR_cap *= 2; s->toR = (int32_t *) realloc(s->toR, sizeof(int32_t) * (size_t) s->toR_cap); }
                        s->toR[s->toR_n++] = A;
                        s->st->transitive_removed++;
                    }
                }
            }
            if (s->toR_n == s->toR_cap) { s->toR_cap *= 2; s->toR = (int32_t *) realloc(s->toR, sizeof(int32_t) * (size_t) s->toR_cap); }
            s->toR[s->toR_n++] = suffId;
            s->to_remove[suffId] = 1;
            for (int j = lst->size - 1; j >= 0; j--) {                   /* :466-472 */
                if (s->to_remove[lst->a[j].nbr]) { lst->a[j] = lst->a[lst->size - 1]; lst->size--; }
            }
            for (int t = 0; t < s->toR_n; t++) s->to_remove[s->toR[t]] = 0;
            s->toR_n = 0;
            adj_push(lst, suffId, offset);                               /* :477 */
        }
    }
}

static void next_iteration(ps *s) { /* nextPrefSufIteration :238-315 */
    s->cur++;
    for (int i = 0; i < s->n; i++) {                                      /* updatePrexihHashJob :347-354 */
        if (s->to[i]) { if (!upd_prefix(s, i, s->cur, s->fac, s->faca)) s->to[i] = 0; }
    }
    /* buckets :317-332 ; ids land in ascending order inside a bucket with one thread */
    memset(s->bstart, 0, sizeof(int32_t) * (size_t) (s->buckets + 1));
    for (int i = 0; i < s->n; i++) if (s->to[i]) s->bstart[(s->pk[i] & (uint64_t) (s->buckets - 1)) + 1]++;
    for (int b = 0; b < s->buckets; b++) s->bstart[b + 1] += s->bstart[b];
    {
        int32_t *fill = (int32_t *) malloc(sizeof(int32_t) * (size_t) s->buckets);
        memcpy(fill, s->bstart, sizeof(int32_t) * (size_t) s->buckets);
        for (int i = 0; i < s->n; i++) if (s->to[i]) s->bitems[fill[s->pk[i] & (uint64_t) (s->buckets - 1)]++] = i;
        free(fill);
    }
    if (s->cur == s->rsoemo) {                                            /* :288-296 */
        reverse_graph_in_place(s->G, s->n);
        retain_only_smallest_offset(s->G, s->n);
    }
    for (int i = 0; i < s->n; i++) add_edges_for_suffix(s, i);            /* :299-306 */
    s->fac <<= 2; if (s->fac >= MAX_HASH) s->fac %= MAX_HASH;              /* :309-313 */
    uint64_t fa = (uint64_t) s->faca << 2; if (fa >= MAX_ADD) fa %= MAX_ADD; s->faca = (uint32_t) fa;
}

static int cmp_edge(const void *x, const void *y) {
    const oracle_edge *a = (const oracle_edge *) x, *b = (const oracle_edge *) y;
    if (a->src != b->src) return a->src < b->src ? -1 : 1;
    if (a->dst != b->dst) return a->dst < b->dst ? -1 : 1;
    if (a->offset != b->offset) return a->offset < b->offset ? -1 : 1;
    return 0;
}

int oracle_prefsuf(const uint32_t *words, const int32_t *len, int32_t n, int32_t W,
                   const uint8_t *align_from, const uint8_t *align_to,
                   int32_t min_overlap, int32_t rsoemo, oracle_graph *out) {
    memset(out, 0, sizeof(*out));
    ps s; memset(&s, 0, sizeof(s));
    s.words = words; s.len = len; s.n = n; s.W = W; s.Lmin = min_overlap; s.rsoemo = rsoemo; s.st = out;
    size_t nn = (size_t) (n ? n : 1);
    s.from = (uint8_t *) malloc(nn); s.to = (uint8_t *) malloc(nn);
    for (int i = 0; i < n; i++) {
        s.from[i] = (uint8_t) (len[i] > 0 && (!align_from || align_from[i]));
        s.to[i]   = (uint8_t) (len[i] > 0 && (!align_to || align_to[i]));
    }
    s.pk = (uint64_t *) malloc(sizeof(uint64_t) * nn); s.sk = (uint64_t *) malloc(sizeof(uint64_t) * nn);
    s.pka = (uint32_t *) malloc(sizeof(uint32_t) * nn); s.ska = (uint32_t *) malloc(sizeof(uint32_t) * nn);
    /* ctor :41-45 */
    {
        int lg = n >= 2 ? (int) log2((double) n / 2) : 0;
        long long bk = 1ll << lg;
        if (bk * 3ll > n) bk >>= 1;
        if (bk < 3) bk = 3;
        s.buckets = (int) bk;
    }
    s.bstart = (int32_t *) malloc(sizeof(int32_t) * (size_t) (s.buckets + 1));
    s.bitems = (int32_t *) malloc(sizeof(int32_t) * nn);
    s.G = (adj *) calloc(nn, sizeof(adj));
    s.to_remove = (uint8_t *) calloc(nn, 1);
    s.toR_cap = 1024; s.toR = (int32_t *) malloc(sizeof(int32_t) * (size_t) s.toR_cap);
    s.maxlen = 0;
    for (int i = 0; i < n; i++) if (len[i] > s.maxlen) s.maxlen = len[i]; /* :52-56 */

    create_initial_state(&s);
    if (s.maxlen > 500) s.maxlen = 500;                                   /* :92 */
    int iters = s.maxlen - s.cur + 1; if (iters < 0) iters = 0;
    out->edges_after_iter = (int64_t *) calloc((size_t) (iters ? iters : 1), sizeof(int64_t));
    out->n_iters = 0;
    while (s.cur <= s.maxlen) {                                           /* :94-100 */
        next_iteration(&s);
        out->edges_after_iter[out->n_iters++] = count_edges(s.G, n);
    }
    reverse_graph_in_place(s.G, n);                                       /* :107 */
    retain_only_smallest_offset(s.G, n);                                  /* src/main.cpp:291 */

    int64_t m = count_edges(s.G, n);
    out->edges = (oracle_edge *) malloc(sizeof(oracle_edge) * (size_t) (m ? m : 1));
    out->n_edges = m;
    int64_t k = 0;
    for (int i = 0; i < n; i++)
        for (int j = 0; j < s.G[i].size; j++) { out->edges[k].src = i; out->edges[k].dst = s.G[i].a[j].nbr; out->edges[k].offset = s.G[i].a[j].off; k++; }
    qsort(out->edges, (size_t) m, sizeof(oracle_edge), cmp_edge);

    for (int i = 0; i < n; i++) free(s.G[i].a);
    free(s.G); free(s.to_remove); free(s.toR); free(s.bstart); free(s.bitems);
    free(s.pk); free(s.sk); free(s.pka); free(s.ska); free(s.from); free(s.to);
    return 0;
}

void oracle_free_graph(oracle_graph *g) {
    free(g->edges); free(g->edges_after_iter);
    memset(g, 0, sizeof(*g));
}

/* Graph::serializeGraph, src/DataStructures/Graph.cpp:269-297:
 *   u32 n; n x { i32 id; i32 deg; deg x { i32 neighbour; i32 offset } }   (native endian) */
/* ---- first simplifier step: edges sorted by increasing offset, non- and weakly-metric triangles cut -------------------- */
static int cmp_off_dst(const void *a, const void *b) {          /* Graph.cpp:606-609 */
    const oracle_edge *x = (const oracle_edge *) a, *y = (const oracle_edge *) b;
    if (x->offset != y->offset) return x->offset < y->offset ? -1 : 1;
    return x->dst < y->dst ? -1 : (x->dst > y->dst ? 1 : 0);
}

int oracle_cut_triangles(int32_t n, const oracle_edge *edges_in, int64_t m_in, int32_t mopp, oracle_edge **edges_out, int64_t *m_out) {
    int64_t *row = (int64_t *) calloc((size_t) n + 2, sizeof(int64_t));
    oracle_edge *e = (oracle_edge *) malloc((size_t) (m_in ? m_in : 1) * sizeof(oracle_edge));
    oracle_edge *out = (oracle_edge *) malloc((size_t) (m_in ? m_in : 1) * sizeof(oracle_edge));
    if (!row || !e || !out) { free(row); free(e); free(out); return -1; }
    for (int64_t k = 0; k < m_in; k++) row[edges_in[k].src + 1]++;
    for (int32_t i = 0; i < n; i++) row[i + 1] += row[i];
    {   /* adjacency lists, then sortEdgesByIncreasingOffset */
        int64_t *cur = (int64_t *) malloc((size_t) (n + 1) * sizeof(int64_t));
        if (!cur) { free(row); free(e); free(out); return -1; }
        memcpy(cur, row, (size_t) (n + 1) * sizeof(int64_t));
        for (int64_t k = 0; k < m_in; k++) e[cur[edges_in[k].src]++] = edges_in[k];
        free(cur);
        for (int32_t i = 0; i < n; i++) qsort(e + row[i], (size_t) (row[i + 1] - row[i]), sizeof(oracle_edge), cmp_off_dst);
    }
    int64_t m = 0;
    for (int32_t i = 0; i < n; i++) {
        const int64_t b0 = row[i], deg = row[i + 1] - row[i];
        oracle_edge *L = out + m;                                /* the list of i as the reference will leave it */
        memcpy(L, e + b0, (size_t) deg * sizeof(oracle_edge));
        int64_t size = deg;
        for (int64_t k = 0; k < deg; k++) {                      /* GraphSimplifier.cpp:297-318, decisions on the unchanged graph */
            const int32_t b = e[b0 + k].dst, w = e[b0 + k].offset;
            if (w > mopp) continue;                              /* :301-303 */
            int have = 0; int32_t best = 0;                      /* dst[b]: shortest two-edge path i -> a -> b (:283-295) */
            for (int64_t k2 = 0; k2 < deg; k2++) {
                const int32_t a = e[b0 + k2].dst, w1 = e[b0 + k2].offset;
                for (int64_t j = row[a]; j < row[a + 1]; j++)
                    if (e[j].dst == b) { const int32_t d = w1 + e[j].offset; if (!have || d < best) { best = d; have = 1; } }
            }
            if (have && best == w) {                             /* :310: equal distances only */
                /* removeDirectedEdge(i, b), Graph.cpp:104-114: every entry with that neighbour, from the back, swapped with the last */
                int64_t p = size - 1;
                for (int64_t x = size - 1; x >= 0; x--)
                    if (L[x].dst == b) { oracle_edge t = L[x]; L[x] = L[p]; L[p] = t; size--; p--; }
            }
        }
        m += size;
    }
    free(row); free(e);
    *edges_out = out; *m_out = m;
    return 0;
}

int oracle_write_graph(const char *path, int32_t n, const oracle_edge *edges, int64_t n_edges) {
    FILE *f = fopen(path, "wb");
    if (!f) return -1;
    uint32_t s = (uint32_t) n;
    fwrite(&s, sizeof(s), 1, f);
    int64_t k = 0;
    for (int32_t i = 0; i < n; i++) {
        int64_t e = k;
        while (e < n_edges && edges[e].src == i) e++;
        int32_t t = (int32_t) (e - k);
        fwrite(&i, sizeof(i), 1, f);
        fwrite(&t, sizeof(t), 1, f);
        for (; k < e; k++) { fwrite(&edges[k].dst, 4, 1, f); fwrite(&edges[k].offset, 4, 1, f); }
    }
    fclose(f);
    return 0;
}
