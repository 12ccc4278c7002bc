// oracle/ref_adapter.cpp -- TEST INFRASTRUCTURE.  Proves the in-process drop-in of INTEGRATION.md section 2: a driver of my own
// that LINKS AGAINST THE REFERENCE'S OBJECT FILES (everything under /root/reference/src except main.cpp, compiled in place by
// oracle/Makefile) and walks the reference's creator call site with the creator swapped for the adapter
// (alga_amd/host/adapter/GraphCreatorPrefSufHIP.h, GraphCreatorLIHIP.h).  Nothing of the reference is copied: this file includes
// its headers at build time and calls its classes.
//
//   ref_adapter <hip|cpu> <nodes.bin> <graph_out> <min_overlap> <rsoemo> <li_kmer_length> [<error_rate_percent> <kmer_length_bucket> <graph_out2>]
//   ref_adapter <hip|cpu> <nodes.bin> <graph_out> <min_overlap> <rsoemo> <li_kmer_length> <max_offset_parallel_paths>
//       (8 arguments: the exact graph, then the simplifier's first step -- sortEdgesByIncreasingOffset + cutNonAndWeaklyMetricTriangles,
//        src/GraphSimplifiers/GraphSimplifier.cpp:113-117 -- on the CPU or through alga_adapter::first_simplifier_step; <graph_out> is
//        the graph after that step)
//
// The call sequence is the one of src/main.cpp:239-296 (exact graph) and, with the last three arguments, :300-347 (supplement):
// Graph(READS.size()); new <creator>; masks for short / removed reads; startAlignmentGraphCreation(); delete;
// retainOnlySmallestOffset(); serializeGraph().  `cpu` takes the reference's own GraphCreatorPrefSuf / GraphCreatorLI,
// `hip` the adapters: the two dumps must be identical (tests/test_adapter.py).
// node file: i32 n, i32 W, i32 len[n], u32 words[n*W]   (reference bit layout; len 0 = nullptr)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

#include <GraphCreators/GraphCreatorLI.h>
#include <GraphCreators/GraphCreatorPrefSuf.h>
#include <GraphSimplifiers/GraphSimplifier.h>
#include <Global.h>
#include <Params.h>

#include "../alga_amd/host/adapter/GraphCreatorLIHIP.h"
#include "../alga_amd/host/adapter/GraphCreatorPrefSufHIP.h"

static void die(const char *m) { fprintf(stderr, "ref_adapter: %s\n", m); exit(2); }

static void load_nodes(const char *path) {
    FILE *f = fopen(path, "rb");
    if (!f) die("cannot open node file");
    int32_t n, W;
    if (fread(&n, 4, 1, f) != 1 || fread(&W, 4, 1, f) != 1) die("short node file");
    std::vector<int32_t> len((size_t) n);
    std::vector<uint32_t> words((size_t) n * W);
    if (n && (fread(len.data(), 4, (size_t) n, f) != (size_t) n || fread(words.data(), 4, (size_t) n * W, f) != (size_t) n * W)) die("short node file");
    fclose(f);
    Global::READS.clear();
    for (int i = 0; i < n; i++) {
        if (len[(size_t) i] == 0) { Global::READS.push_back(nullptr); continue; }
        std::string s((size_t) len[(size_t) i], 'A');
        for (int k = 0; k < len[(size_t) i]; k++) s[(size_t) k] = "ACGT"[(words[(size_t) i * W + (k >> 4)] >> ((k & 15) << 1)) & 3];
        Global::READS.push_back(new Read(i, s));
    }
}

int main(int argc, char **argv) {
    if (argc != 7 && argc != 8 && argc != 10) die("usage: ref_adapter <hip|cpu> <nodes.bin> <graph_out> <min_overlap> <rsoemo> <li_kmer_length> [<error_rate_percent> <kmer_length_bucket> <graph_out2>]");
    const bool hip = !strcmp(argv[1], "hip");
    Read::priorities = VI(4);
    std::iota(Read::priorities.begin(), Read::priorities.end(), 0);
    Bitset::initializeStaticBlock();
    Params::THREADS = 1;                                   // the canonical, run-to-run deterministic order of the reference
    load_nodes(argv[2]);
    Params::MIN_OVERLAP_PREF_SUF = Params::MIN_OVERLAP_AREA = Params::MOST_FREQUENTLY_USED_PARAMETER = atoi(argv[4]);
    Params::REMOVE_SMALL_OVERLAP_EDGES_MIN_OVERLAP = atoi(argv[5]);
    Params::LI_KMER_LENGTH = atoi(argv[6]);
    std::vector<Read *> *READS = &Global::READS;

    // ---- the creator call site (src/main.cpp:239-296) ----
    Global::GRAPH = Graph((int) Global::READS.size());
    Graph *G = &Global::GRAPH;
    GraphCreator *graphCreator = hip ? (GraphCreator *) new GraphCreatorPrefSufHIP(READS, G) : (GraphCreator *) new GraphCreatorPrefSuf(READS, G, false);
    for (int i = 0; i < (int) READS->size(); i++)
        if ((*READS)[i] != nullptr && (*READS)[i]->size() < Params::LI_KMER_INTERVALS + Params::LI_KMER_LENGTH) {
            graphCreator->setAlignFrom(i, false);
            graphCreator->setAlignTo(i, false);
        }
    for (int i = 0; i < (int) READS->size(); i++)
        if ((*READS)[i] != nullptr && !graphCreator->getAlignFrom(i) && !graphCreator->getAlignTo(i)) Global::removeRead(i);
    for (int i = 0; i < G->size(); i++)
        if ((*READS)[i] == nullptr) { graphCreator->setAlignFrom(i, false); graphCreator->setAlignTo(i, false); }
    if (hip) Params::THREADS = std::max(1u, std::thread::hardware_concurrency());   // only the adapter's marshalling and Graph::V fill use it
    const auto t_creator = std::chrono::steady_clock::now();
    graphCreator->startAlignmentGraphCreation();
    graphCreator->clear();
    delete graphCreator;
    G->retainOnlySmallestOffset();
    fprintf(stdout, "creator_region_ms %.1f\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_creator).count());
    fprintf(stdout, "edges %lld\n", (long long) G->countEdges());
    if (argc == 8) {
        // ---- the simplifier's first step (src/GraphSimplifiers/GraphSimplifier.cpp:113-117) ----
        Params::MAX_OFFSET_PARALLEL_PATHS = atoi(argv[7]);
        if (hip) alga_adapter::first_simplifier_step(G);
        else {
            GraphSimplifier simplifier(Global::GRAPH, Global::READS);
            G->sortEdgesByIncreasingOffset();
            simplifier.cutNonAndWeaklyMetricTriangles();
        }
        fprintf(stdout, "edges_after_cut %lld\n", (long long) G->countEdges());
    }
    G->serializeGraph(argv[3]);
    if (argc <= 8) return 0;

    // ---- the supplement call site (src/main.cpp:300-347) ----
    Params::ERROR_RATE = atoi(argv[7]);
    Params::KMER_LENGTH_BUCKET = atoi(argv[8]);
    GraphCreator *gc = hip ? (GraphCreator *) new GraphCreatorLIHIP(&Global::READS, G) : (GraphCreator *) new GraphCreatorLI(&Global::READS, G);
    VI *inDeg = G->getInDegrees();
    for (int i = 0; i < G->size(); i++) {
        gc->setAlignFrom(i, false);
        gc->setAlignTo(i, false);
        if ((*inDeg)[i] == 0 && (*G)[i].size() > 0) gc->setAlignTo(i, true);
        if ((*inDeg)[i] > 0 && (*G)[i].size() == 0) gc->setAlignFrom(i, true);
    }
    delete inDeg;
    const double avg = Global::calculateAvgReadLength();
    Params::MIN_OVERLAP_AREA = (1.f + Params::SCALE) * avg / 2;
    Params::MAX_OFFSET_CONSIDERED_FOR_ALIGNMENT = (1.f - Params::SCALE) * avg / 2;
    Params::MINIMAL_OVERLAP_FOR_LCS_LOW_ERROR = 99 - Params::ERROR_RATE;
    Params::LI_KMER_INTERVALS = 6;
    Params::LI_KMER_LENGTH = 35;
    gc->startAlignmentGraphCreation();
    G->retainOnlySmallestOffset();
    delete gc;
    fprintf(stdout, "edges_after_supplement %lld\n", (long long) G->countEdges());
    G->serializeGraph(argv[9]);
    return 0;
}
