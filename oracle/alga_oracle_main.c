/*
 * oracle/alga_oracle_main.c -- TEST INFRASTRUCTURE: command-line front end of the CPU oracle.
 *
 *   alga_oracle --file1=a.fasta [--file2=b.fasta] [--graph=out.graph] [-l N] [--rsoemo=N] [--scale=F]
 *
 * Writes the graph in the reference's dump format and prints the per-iteration edge counts in
 * the wording of src/GraphCreators/GraphCreatorPrefSuf.cpp:96-99 so they can be diffed against
 * the reference's stderr.
 */
#include "alga_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static const char *optval(const char *arg, const char *name) {
    size_t n = strlen(name);
    if (strncmp(arg, name, n) == 0 && arg[n] == '=') return arg + n + 1;
    return NULL;
}

int main(int argc, char **argv) {
    const char *f1 = NULL, *f2 = NULL, *graph = NULL;
    oracle_ingest_params p;
    oracle_default_ingest_params(&p);
    for (int i = 1; i < argc; i++) {
        const char *v;
        if ((v = optval(argv[i], "--file1"))) f1 = v;
        else if ((v = optval(argv[i], "--file2"))) f2 = v;
        else if ((v = optval(argv[i], "--graph"))) graph = v;
        else if ((v = optval(argv[i], "--rsoemo"))) p.rsoemo = atoi(v);
        else if ((v = optval(argv[i], "--scale"))) p.scale = strtof(v, NULL);
        else if ((v = optval(argv[i], "--retl"))) p.trim_left = atoi(v);
        else if ((v = optval(argv[i], "--retr"))) p.trim_right = atoi(v);
        else if ((v = optval(argv[i], "--remove_reads_with_n"))) p.remove_reads_with_n = atoi(v);
        else if ((v = optval(argv[i], "--rna"))) p.rna = atoi(v);
        else if (!strcmp(argv[i], "-l") && i + 1 < argc) p.min_overlap = atoi(argv[++i]);
        else { fprintf(stderr, "unknown option %s\n", argv[i]); return 2; }
    }
    if (!f1) { fprintf(stderr, "usage: alga_oracle --file1=reads.fasta [--file2=...] [--graph=out.graph]\n"); return 2; }
    oracle_nodes nd;
    int rc = oracle_ingest(f1, f2, &p, &nd);
    if (rc) return 1;
    fprintf(stderr, "nodes %d  W %d  LEN %d  MIN_OVERLAP_PREF_SUF %d  RSOEMO %d  removed: N %d STR %d prefix %d\n",
            nd.n, nd.W, nd.LEN, nd.min_overlap, nd.rsoemo, nd.removed_n, nd.removed_str, nd.removed_prefix);
    oracle_graph g;
    clock_t t0 = clock();
    rc = oracle_prefsuf(nd.words, nd.len, nd.n, nd.W, NULL, NULL, nd.min_overlap, nd.rsoemo, &g);
    double dt = (double) (clock() - t0) / CLOCKS_PER_SEC;
    if (rc) return 1;
    for (int i = 0; i < g.n_iters; i++)
        printf("After Iteration %d.  There are already %lld edges in the graph\n", nd.min_overlap + i,
               (long long) g.edges_after_iter[i]);
    fprintf(stderr, "edges %lld  scanned %lld  hash_equal %lld  transitive_checks %lld  removed %lld  creator %.3f s\n",
            (long long) g.n_edges, (long long) g.bucket_entries_scanned, (long long) g.hash_equal_pairs,
            (long long) g.transitive_checks, (long long) g.transitive_removed, dt);
    if (graph && oracle_write_graph(graph, nd.n, g.edges, g.n_edges)) { fprintf(stderr, "cannot write %s\n", graph); return 1; }
    oracle_free_graph(&g);
    oracle_free_nodes(&nd);
    return 0;
}
