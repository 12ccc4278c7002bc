/*
 * oracle/alga_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, single thread) of the overlap-graph hot path of swacisko/ALGA and of
 * the input stages that feed it, written from the reference's behaviour; every function cites
 * the reference file:line it follows (paths relative to the reference root).
 *
 * Pinning: the reference has no tests for this path, so the oracle is pinned by
 * outputs of the reference itself, built from its own sources by oracle/Makefile into
 * oracle/_ref/ALGA and run with --threads=1 --serialize=1 (tools/make_golden.py); the dumps are
 * committed under tests/golden/ and tests/test_oracle_golden.py requires byte identity.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#ifndef ALGA_ORACLE_H
#define ALGA_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Node set handed to the graph creator: 2-bit packed reads, LSB-first inside uint32 blocks,
 * nucleotide i in bits (2i, 2i+1), A0 C1 G2 T3, tail bits zero
 * (include/DataStructures/Bitset.h:41-50, src/DataStructures/Read.cpp:40-68). */
typedef struct {
    int32_t   n;         /* nodes (reads incl. reverse complements), even                    */
    int32_t   W;         /* uint32 words per node in `words` (row stride)                     */
    uint32_t *words;     /* n*W                                                                */
    int32_t  *len;       /* length in nt, 0 = removed node (READS[i]==nullptr)                 */
    uint8_t  *pair_off;  /* Global::pairedReadOffset (src/main.cpp:150-232)                    */
    /* parameters derived by src/main.cpp:93-115 */
    int32_t   LEN;       /* int(avg_len + trimL + trimR)                                       */
    int32_t   min_overlap;   /* MIN_OVERLAP_PREF_SUF                                           */
    int32_t   rsoemo;        /* REMOVE_SMALL_OVERLAP_EDGES_MIN_OVERLAP                         */
    int32_t   li_kmer_length;/* min(2L/3, 60)                                                  */
    /* bookkeeping the reference prints */
    int64_t   reads_in_file; /* records parsed (file1+file2)                                   */
    int32_t   removed_n, removed_str, removed_prefix;
    double    avg_len;
} oracle_nodes;

typedef struct {
    int32_t trim_left, trim_right;   /* READ_END_TRIM_* default 3/3 (src/Params.cpp:729-730) */
    int32_t remove_reads_with_n;     /* default 1                                             */
    int32_t rna;                     /* default 0                                             */
    float   scale;                   /* Params::SCALE default 0.55f                           */
    int32_t min_overlap;             /* -l / mfup; -1 = derive                                */
    int32_t rsoemo;                  /* --rsoemo; -1 = derive                                 */
    int32_t remove_pref_reads;       /* 1 duplicates, 2 all prefix reads (default), 3 none    */
} oracle_ingest_params;

void oracle_default_ingest_params(oracle_ingest_params *p);

/* FASTA/FASTQ/plain -> compacted node set; mirrors --threads=1 of
 * src/IO/InputReader.cpp:44-139,272-391 + src/main.cpp:93-232,253-266.
 * Returns 0 on success. file2 may be NULL/"" for single-end input. */
int  oracle_ingest(const char *file1, const char *file2, const oracle_ingest_params *p, oracle_nodes *out);
void oracle_free_nodes(oracle_nodes *nd);

/* Directed overlap edge a -> b : "b starts at position offset of a". */
typedef struct { int32_t src, dst, offset; } oracle_edge;

typedef struct {
    oracle_edge *edges;      /* sorted by (src, dst, offset)                                   */
    int64_t      n_edges;
    int64_t     *edges_after_iter;  /* G->countEdges() after the iteration for overlap length L,
                                       index L - min_overlap (GraphCreatorPrefSuf.cpp:96-99)   */
    int32_t      n_iters;
    /* work counters == GATHER_STATISTICS (include/GraphCreators/GraphCreatorPrefSuf.h:112-118) */
    int64_t      bucket_entries_scanned, hash_equal_pairs, transitive_checks, transitive_removed;
} oracle_graph;

/* GraphCreatorPrefSuf::startAlignmentGraphCreation + the caller's retainOnlySmallestOffset
 * (src/GraphCreators/GraphCreatorPrefSuf.cpp:73-488, src/main.cpp:282-291), --threads=1 order.
 * align_from/align_to: n bytes each, may be NULL (= all true for live nodes). */
int  oracle_prefsuf(const uint32_t *words, const int32_t *len, int32_t n, int32_t W,
                    const uint8_t *align_from, const uint8_t *align_to,
                    int32_t min_overlap, int32_t rsoemo, oracle_graph *out);
void oracle_free_graph(oracle_graph *g);

/* ---- approximate supplement (error_rate > 0.01): oracle/alga_oracle_pkb.cpp ---------------------------------- */
typedef struct {
    int32_t min_overlap_area;    /* Params::MIN_OVERLAP_AREA = int((1+SCALE)*avg/2)          (src/main.cpp:333) */
    int32_t max_offset_pct;      /* Params::MAX_OFFSET_CONSIDERED_FOR_ALIGNMENT, % of |r1|   (src/main.cpp:335) */
    int32_t min_identity_pct;    /* Params::MINIMAL_OVERLAP_FOR_LCS_LOW_ERROR = 99 - ERROR_RATE (:336)         */
    int32_t same_ends;           /* Params::ALIGNMENT_CONTROLLER_SAME_ENDS_LENGTH = 3                          */
    int32_t li_k, li_intervals;  /* 35, 6 (src/main.cpp:339-340)                                               */
    int32_t rounds;              /* min(4, LI_PRIORITIES_TO_CONSIDER) = 4                                      */
} oracle_pkb_params;

void oracle_pkb_derive_params(double avg_len, float scale, int error_rate_percent, oracle_pkb_params *p);
int  oracle_can_align(const uint32_t *words, const int32_t *len, int32_t W, int32_t r1, int32_t r2, int32_t offset,
                      const oracle_pkb_params *p);
int  oracle_li_kmers(const uint32_t *row, int32_t len, int32_t k, int32_t intervals, const int32_t *prio,
                     uint64_t *hash_out, int32_t *ind_out);
#define ORACLE_PKB_TIES_BY_ID 1   /* equal k-mers ordered by read id instead of std::sort's tie order            */
#define ORACLE_PKB_SNAPSHOT    2   /* groups of a round see the round-start graph: the GPU engine's semantics     */
int  oracle_supplement(const uint32_t *words, const int32_t *len, int32_t n, int32_t W, const oracle_edge *edges_in, int64_t m_in,
                       const oracle_pkb_params *p, int32_t kmer_length_bucket, int32_t flags, oracle_edge **edges_out, int64_t *m_out,
                       int64_t *can_align_calls);

/* First step of the simplifier on the PrefSuf path (src/GraphSimplifiers/GraphSimplifier.cpp:90-125 simplifyGraphOld):
 * Graph::sortEdgesByIncreasingOffset (src/DataStructures/Graph.cpp:584-614) and GraphSimplifier::cutNonAndWeaklyMetricTriangles
 * (src/GraphSimplifiers/GraphSimplifier.cpp:228-348).  edges_in: any order, grouped or not; *edges_out: grouped by src, every
 * adjacency list in the order the reference leaves it (sorted by (offset, dst), then Graph::removeDirectedEdge's swap-with-last
 * removals in list order, src/DataStructures/Graph.cpp:96-119).  Pinned by tests/golden/ after-cut dumps made through oracle/ref_driver.cpp. */
int  oracle_cut_triangles(int32_t n, const oracle_edge *edges_in, int64_t m_in, int32_t max_offset_parallel_paths,
                          oracle_edge **edges_out, int64_t *m_out);

/* Graph::serializeGraph wire format (src/DataStructures/Graph.cpp:269-297). */
int  oracle_write_graph(const char *path, int32_t n, const oracle_edge *edges, int64_t n_edges);

/* helpers exported for function-level tests */
int  oracle_min_period(const char *s, int n);                 /* include/Utils/MyUtils.h:160-170 */
void oracle_pack(const char *s, int n, uint32_t *words, int W);/* src/DataStructures/Read.cpp:40-68 */

#ifdef __cplusplus
}
#endif
#endif
