// alga_amd/host/adapter/GraphCreatorPrefSufHIP.h -- REFERENCE-SIDE binding of the MI355X overlap engine.
//
// A `GraphCreator` subclass (reference include/GraphCreators/GraphCreator.h:12-62) a maintainer drops into the reference
// tree: replace `new GraphCreatorPrefSuf(READS, G, false)` at src/main.cpp:249 by `new GraphCreatorPrefSufHIP(READS, G)` and
// link -lalga_amd.  It compiles against the REFERENCE's headers (Read, Bitset, Graph, Params) and against include/alga_amd.h;
// oracle/Makefile builds it (target `adapter`) together with oracle/ref_adapter.cpp into oracle/_ref/ref_adapter, which
// tests/test_adapter.py runs against the reference's own creator and the golden dumps.
//
// Same life cycle as GraphCreatorPrefSuf: construct -> setAlignFrom/To -> startAlignmentGraphCreation() -> delete; the
// caller's G->retainOnlySmallestOffset() (src/main.cpp:291) finds the lists already deduplicated and sorted.
#ifndef ALGA_AMD_GRAPHCREATORPREFSUFHIP_H
#define ALGA_AMD_GRAPHCREATORPREFSUFHIP_H

#include <algorithm>
#include <thread>
#include <vector>

#include <GraphCreators/GraphCreator.h>
#include <Params.h>

#include <alga_amd.h>

namespace alga_adapter {

// vector<Read*> -> the engine's node arrays: 2-bit rows as the Bitset holds them (Bitset.h:41-50,175), 0 = READS[i] == nullptr
struct NodeArrays {
    std::vector<uint32_t> words;
    std::vector<int32_t> len;
    int stride = 4;
    explicit NodeArrays(std::vector<Read *> &reads) {
        const size_t n = reads.size();
        int max_len = 0;
        for (Read *r : reads) if (r != nullptr) max_len = std::max(max_len, r->size());
        stride = std::max(4, (((2 * max_len + 31) / 32) + 3) & ~3);          // 16-byte aligned rows
        words.assign(n * (size_t) stride, 0u);
        len.assign(n, 0);
        const int T = std::max(1, Params::THREADS);
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++)
            th.emplace_back([&, t]() {
                for (size_t i = n * t / T; i < n * (t + 1) / T; i++) {
                    Read *r = reads[i];
                    if (r == nullptr) continue;
                    len[i] = r->size();
                    Bitset &b = r->getSequence();
                    const int nb = (int) b.countBlocks();
                    for (int k = 0; k < nb; k++) words[i * (size_t) stride + k] = b.getBlock(k);
                }
            });
        for (std::thread &x : th) x.join();
    }
};

// edge triples grouped by src, lists sorted by (dst, offset) -> Graph::V, one exact-size allocation per node and no
// per-edge pushDirectedEdge (Graph.cpp:73-75 grows a vector edge by edge): rows are independent, so threads split the nodes
inline void fill_graph(Graph *G, const alga_edge *e, uint64_t m) {
    const int n = G->size();
    std::vector<uint64_t> first((size_t) n + 1, m);
    for (uint64_t k = m; k-- > 0;) first[(size_t) e[k].src] = k;
    for (int a = n - 1; a >= 0; a--) if (first[(size_t) a] == m) first[(size_t) a] = first[(size_t) a + 1];   // empty rows
    const int T = std::max(1, Params::THREADS);
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++)
        th.emplace_back([&, t]() {
            for (int a = (int) ((long long) n * t / T); a < (int) ((long long) n * (t + 1) / T); a++) {
                VPII &row = (*G)[a];
                const uint64_t k0 = first[(size_t) a], k1 = first[(size_t) a + 1];
                row.clear();
                row.reserve((size_t) (k1 - k0));
                for (uint64_t k = k0; k < k1; k++) row.emplace_back(e[k].dst, e[k].offset);
            }
        });
    for (std::thread &x : th) x.join();
}

[[noreturn]] inline void die(alga_engine *e, const char *what, int rc);

// The simplifier's first step on the GPU: replaces `G->sortEdgesByIncreasingOffset(); cutNonAndWeaklyMetricTriangles();` of
// GraphSimplifier::simplifyGraphOld (src/GraphSimplifiers/GraphSimplifier.cpp:113-117) -- G comes back with every list in the
// order those two calls leave it in.
inline void first_simplifier_step(Graph *G, int hip_device = 0) {
    const int n = G->size();
    std::vector<alga_edge> in;
    for (int a = 0; a < n; a++) {
        VPII row = (*G)[a];
        std::sort(row.begin(), row.end());                                  // (neighbour, offset): the order the engine's lists have
        for (const PII &x : row) in.push_back(alga_edge{a, x.first, x.second});
    }
    alga_engine *e = nullptr;
    int rc = alga_engine_create(hip_device, &e);
    if (rc != ALGA_OK) die(nullptr, "no usable HIP device", rc);
    alga_edge *out = nullptr;
    uint64_t m = 0;
    rc = alga_cut_triangles_host(e, n, in.data(), (uint64_t) in.size(), Params::MAX_OFFSET_PARALLEL_PATHS, &out, &m);
    if (rc != ALGA_OK) die(e, "triangle cut", rc);
    fill_graph(G, out, m);
    alga_free_edges(e, out);
    alga_engine_destroy(e);
}

// The contig-trimming block on the GPU: replaces src/main.cpp:636-697 (contigs + reverse complements through one more
// GraphCreatorPrefSuf run at threshold 25, trimLeft from its edges); the caller keeps the string surgery of :700-712.
inline std::vector<int> contig_trim_left(std::vector<Read *> &contigs, int threshold = 25, int hip_device = 0) {
    NodeArrays nodes(contigs);
    std::vector<int32_t> trim(contigs.size(), 0);
    alga_engine *e = nullptr;
    int rc = alga_engine_create(hip_device, &e);
    if (rc != ALGA_OK) die(nullptr, "no usable HIP device", rc);
    rc = alga_contig_trim_host(e, nodes.words.data(), nodes.stride, nodes.len.data(), (int32_t) contigs.size(), threshold, trim.data());
    if (rc != ALGA_OK) die(e, "contig trimming", rc);
    alga_engine_destroy(e);
    return std::vector<int>(trim.begin(), trim.end());
}

[[noreturn]] inline void die(alga_engine *e, const char *what, int rc) {   // the reference's convention: cerr + exit(1)
    std::cerr << "alga_amd: " << what << ": " << (e ? alga_last_error(e) : "no engine") << " (status " << rc << ")" << std::endl;
    exit(1);
}

} // namespace alga_adapter

class GraphCreatorPrefSufHIP : public GraphCreator {
public:
    GraphCreatorPrefSufHIP(std::vector<Read *> *reads, Graph *G, int hip_device = 0) : GraphCreator(reads, G), device(hip_device) {}

    void startAlignmentGraphCreation() override {
        const int n = G->size();
        alga_adapter::NodeArrays nodes(*reads);
        std::vector<uint8_t> from((size_t) n), to((size_t) n);
        for (int i = 0; i < n; i++) { from[(size_t) i] = alignFrom[i]; to[(size_t) i] = alignTo[i]; }
        alga_engine *e = nullptr;
        int rc = alga_engine_create(device, &e);
        if (rc != ALGA_OK) alga_adapter::die(nullptr, "no usable HIP device", rc);
        alga_prefsuf_params p;
        alga_prefsuf_default_params(&p);
        p.min_overlap = Params::MIN_OVERLAP_PREF_SUF;                           // read at GraphCreatorPrefSuf.cpp:76,167
        p.rsoe_min_overlap = Params::REMOVE_SMALL_OVERLAP_EDGES_MIN_OVERLAP;    // :288,397
        // p.reduction stays ALGA_REDUCTION_AUTO: same graph either way, the engine picks the faster exact form
        alga_nodes nd = {nodes.words.data(), nodes.stride, nodes.len.data(), n, from.data(), to.data()};
        alga_edge *edges = nullptr;
        uint64_t m = 0;
        rc = alga_prefsuf_build_host(e, &nd, &p, &edges, &m);
        if (rc != ALGA_OK) alga_adapter::die(e, "overlap graph", rc);
        alga_adapter::fill_graph(G, edges, m);
        alga_free_edges(e, edges);
        alga_engine_destroy(e);
    }

private:
    int device;
};

#endif
