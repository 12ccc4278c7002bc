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
#include <atomic>
#include <thread>
#include <vector>

#include <GraphCreators/GraphCreator.h>
#include <Params.h>

#include <alga_amd.h>

namespace alga_adapter {

// vector<Read*> -> the engine's node arrays: 2-bit rows as the Bitset holds them (Bitset.h:41-50,175), 0 = READS[i] == nullptr.
// twins: ALGA's read set comes in pairs -- READS[2k] is the reverse complement of READS[2k + 1] (src/IO/InputReader.cpp:78-80,363-377; the
// duplicate removal deletes twins together, src/main.cpp:150-232).  Where that holds only the ODD reads' rows are packed and uploaded --
// half of the PCIe traffic -- and the engine rebuilds the even rows on the device (alga_upload_twin_nodes).  It is CHECKED, not assumed: the
// lengths of every pair first (cheap: any other vector, e.g. the contigs of the trimming stage -- contigs first, reverse complements
// behind -- fails here), then, inside the packing loop, EVERY even read's blocks against the reverse complement of its odd twin's
// (word-parallel; the blocks of the even read are read instead of being packed, so the check costs what packing it would have).  One pair
// that is not an exact reverse complement and the whole set is packed again and travels whole.
struct NodeArrays {
    std::vector<uint32_t> words;
    std::vector<int32_t> len;
    int stride = 1;
    bool twins = false;
    static bool twin_lengths(std::vector<Read *> &reads) {
        const size_t n = reads.size();
        if (n == 0 || (n & 1)) return false;
        for (size_t k = 0; k + 1 < n; k += 2) {
            if (reads[k] == nullptr) continue;
            if (reads[k + 1] == nullptr || reads[k]->size() != reads[k + 1]->size()) return false;
        }
        return true;
    }
    static inline uint32_t rev_groups(uint32_t x) {              // the sixteen 2-bit groups of a block in reverse order
        x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
        x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
        return __builtin_bswap32(x);
    }
    // blocks of `even` == reverse complement of the m nucleotides in `odd` (nb blocks, tail bits zero as the Bitset keeps them)
    static bool is_revcomp(Bitset &even, const uint32_t *odd, int m, int nb) {
        const int pad = nb * 32 - 2 * m;                         // bits of the last block the read does not use
        for (int k = 0; k < nb; k++) {
            const uint32_t lo = rev_groups(~odd[nb - 1 - k]);
            const uint32_t hi = k + 1 < nb ? rev_groups(~odd[nb - 2 - k]) : 0u;
            uint32_t w = pad ? (lo >> pad) | (hi << (32 - pad)) : lo;
            if (k == nb - 1 && pad) w &= 0xFFFFFFFFu >> pad;
            if (w != (uint32_t) even.getBlock(k)) return false;
        }
        return true;
    }
    explicit NodeArrays(std::vector<Read *> &reads, bool allow_twins = true) {
        int max_len = 0;
        for (Read *r : reads) if (r != nullptr) max_len = std::max(max_len, r->size());
        // rows as tight as the Bitset itself (9 words for a 150-bp read): what crosses PCIe is this array, and the engine re-strides it to
        // its own HBM layout on the device (alga_upload_nodes); 16-byte aligned rows (12 words) cost a third more upload for nothing
        stride = std::max(1, (2 * max_len + 31) / 32);
        twins = allow_twins && twin_lengths(reads);
        if (!pack(reads) && twins) {                             // a pair that is no reverse-complement pair: every row travels
            twins = false;
            pack(reads);
        }
    }

private:
    bool pack(std::vector<Read *> &reads) {
        const size_t n = reads.size();
        const size_t rows = twins ? n / 2 : n;
        words.assign(rows * (size_t) stride, 0u);
        len.assign(n, 0);
        const int T = std::max(1, Params::THREADS);
        std::atomic<bool> ok{true};
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++)
            th.emplace_back([&, t]() {
                // whole pairs per thread (the odd row is packed before its even twin is compared with it)
                const size_t p0 = (n / 2) * t / T, p1 = (n / 2) * (t + 1) / T;
                const size_t i0 = twins ? 2 * p0 : n * t / T, i1 = twins ? 2 * p1 : n * (t + 1) / T;
                for (size_t j = i0; j < i1 && ok.load(std::memory_order_relaxed); j++) {
                    const size_t i = twins ? (j ^ 1) : j;        // twins: odd read first, then its even twin
                    Read *r = reads[i];
                    if (r == nullptr) continue;
                    len[i] = r->size();
                    Bitset &b = r->getSequence();
                    const int nb = (int) b.countBlocks();
                    if (twins && (i & 1) == 0) {                 // an even read: rebuilt on the device from its odd twin -- which it must equal reversed and complemented
                        if (!is_revcomp(b, words.data() + (i / 2) * (size_t) stride, r->size(), nb)) ok.store(false, std::memory_order_relaxed);
                        continue;
                    }
                    uint32_t *row = words.data() + (twins ? i / 2 : i) * (size_t) stride;
                    for (int k = 0; k < nb; k++) row[k] = b.getBlock(k);
                }
            });
        for (std::thread &x : th) x.join();
        return ok.load();
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

// the same from the COMPACT form (alga_compact_edges: a degree byte per node, 5 bytes per edge -- what alga_download_edges_compact brings down in
// half the PCIe time of the triples): the lists' starts are a prefix sum of the degree bytes
inline void fill_graph_compact(Graph *G, const alga_compact_edges &c) {
    const int n = G->size();
    std::vector<uint64_t> first((size_t) n + 1, 0);
    for (int a = 0; a < n; a++) first[(size_t) a + 1] = first[(size_t) a] + (a < c.n_nodes ? c.degree[a] : 0);
    const int T = std::max(1, Params::THREADS);
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++)
        th.emplace_back([&, t]() {
            for (int a = (int) ((long long) n * t / T); a < (int) ((long long) n * (t + 1) / T); a++) {
                VPII &row = (*G)[a];
                const uint64_t k0 = first[(size_t) a], k1 = first[(size_t) a + 1];
                row.clear();
                row.reserve((size_t) (k1 - k0));
                for (uint64_t k = k0; k < k1; k++) row.emplace_back((int) c.dst[k], (int) c.offset[k]);
            }
        });
    for (std::thread &x : th) x.join();
}

[[noreturn]] inline void die(alga_engine *e, const char *what, int rc);

// ONE engine and ONE resident node set per process.  The stages of an ALGA run that go through this library -- exact graph
// (src/main.cpp:246-291), approximate supplement (:300-347), first simplifier step, contig trimming (:633-725) -- share them: the node
// set crosses PCIe once (6 GB at 90 M nodes) and the engine's device buffers are allocated once, instead of once per stage.
// The resident copy is matched on the vector's identity, its size and a sampled fingerprint of the reads (4096 evenly spaced reads:
// length and first block); a caller that edits reads in place between two stages calls Session::get(dev).forget().
class Session {
public:
    static Session &get(int hip_device) {
        static Session s;
        if (s.e_ == nullptr || s.device_ != hip_device) {
            s.close();
            int rc = alga_engine_create(hip_device, &s.e_);
            if (rc != ALGA_OK) die(nullptr, "no usable HIP device", rc);
            s.device_ = hip_device;
        }
        return s;
    }
    alga_engine *engine() { return e_; }
    void forget() { resident_ = false; }
    // the node set of `reads`, resident in HBM (uploaded now unless it already is)
    alga_nodes nodes_of(std::vector<Read *> &reads) {
        const uint64_t fp = fingerprint(reads);
        if (!(resident_ && reads_id_ == (const void *) &reads && n_ == reads.size() && fp_ == fp)) {
            NodeArrays host(reads);
            alga_nodes hn = {host.words.data(), host.stride, host.len.data(), (int32_t) reads.size(), nullptr, nullptr};
            // every device buffer of the builds to come, while nothing else is in flight (alga_engine_reserve, include/alga_amd.h)
            int max_len = 0;
            for (int32_t l : host.len) max_len = std::max(max_len, (int) l);
            if (max_len > 0) (void) alga_engine_reserve(e_, hn.n, max_len, std::max(1, Params::MIN_OVERLAP_PREF_SUF), 0);
            int rc = host.twins ? alga_upload_twin_nodes(e_, &hn, &dev_) : alga_upload_nodes(e_, &hn, &dev_);
            if (rc != ALGA_OK) die(e_, "upload of the reads", rc);
            resident_ = true; reads_id_ = (const void *) &reads; n_ = reads.size(); fp_ = fp;
        }
        alga_nodes nd = dev_;
        nd.align_from = nullptr; nd.align_to = nullptr;
        return nd;
    }
    // n bytes -> a session-owned device buffer (slot 0 / 1: the two masks; slot 2: an edge list)
    const void *to_device(int slot, const void *host, size_t bytes) {
        if (cap_[slot] < bytes) {
            if (buf_[slot]) alga_device_free(e_, buf_[slot]);
            buf_[slot] = nullptr; cap_[slot] = 0;
            int rc = alga_device_alloc(e_, bytes, &buf_[slot]);
            if (rc != ALGA_OK) die(e_, "device buffer", rc);
            cap_[slot] = bytes;
        }
        if (bytes) { int rc = alga_copy_to_device(e_, buf_[slot], host, bytes); if (rc != ALGA_OK) die(e_, "copy to the device", rc); }
        return buf_[slot];
    }
    // (no destructor work: at static-destruction time the HIP runtime may already be gone; a caller that wants the HBM back before the
    // process ends calls shutdown())
    static void shutdown(int hip_device) { get(hip_device).close(); }

private:
    Session() = default;
    void close() {
        if (!e_) return;
        for (int k = 0; k < 3; k++) { if (buf_[k]) alga_device_free(e_, buf_[k]); buf_[k] = nullptr; cap_[k] = 0; }
        alga_engine_destroy(e_);
        e_ = nullptr; resident_ = false;
    }
    static uint64_t fingerprint(std::vector<Read *> &reads) {
        uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t) reads.size();
        const size_t n = reads.size(), step = std::max<size_t>(1, n / 4096);
        for (size_t i = 0; i < n; i += step) {
            Read *r = reads[i];
            uint64_t x = r == nullptr ? ~0ull : (((uint64_t) r->size() << 32) | (r->size() > 0 ? r->getSequence().getBlock(0) : 0u));
            h = (h ^ x) * 0x100000001B3ull;
        }
        return h;
    }
    alga_engine *e_ = nullptr;
    int device_ = -1;
    bool resident_ = false;
    const void *reads_id_ = nullptr;
    size_t n_ = 0;
    uint64_t fp_ = 0;
    alga_nodes dev_{};
    void *buf_[3] = {nullptr, nullptr, nullptr};
    size_t cap_[3] = {0, 0, 0};
};

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
    alga_engine *e = Session::get(hip_device).engine();
    alga_edge *out = nullptr;
    uint64_t m = 0;
    int rc = alga_cut_triangles_host(e, n, in.data(), (uint64_t) in.size(), Params::MAX_OFFSET_PARALLEL_PATHS, &out, &m);
    if (rc != ALGA_OK) die(e, "triangle cut", rc);
    fill_graph(G, out, m);
    alga_free_edges(e, out);
}

// The contig-trimming block on the GPU: replaces src/main.cpp:636-697 (contigs + reverse complements through one more
// GraphCreatorPrefSuf run at threshold 25, trimLeft from its edges); the caller keeps the string surgery of :700-712.
inline std::vector<int> contig_trim_left(std::vector<Read *> &contigs, int threshold = 25, int hip_device = 0) {
    NodeArrays nodes(contigs, false);
    std::vector<int32_t> trim(contigs.size(), 0);
    Session &ses = Session::get(hip_device);
    alga_engine *e = ses.engine();
    ses.forget();                                          // the contigs take the engine's upload buffers: the reads are no longer resident
    int rc = alga_contig_trim_host(e, nodes.words.data(), nodes.stride, nodes.len.data(), (int32_t) contigs.size(), threshold, trim.data());
    if (rc != ALGA_OK) die(e, "contig trimming", rc);
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
        alga_adapter::Session &ses = alga_adapter::Session::get(device);
        alga_engine *e = ses.engine();
        alga_nodes nd = ses.nodes_of(*reads);                                   // uploaded once per process, shared with the later stages
        std::vector<uint8_t> from((size_t) n), to((size_t) n);
        bool masked = false;
        for (int i = 0; i < n; i++) { from[(size_t) i] = alignFrom[i]; to[(size_t) i] = alignTo[i]; masked = masked || !alignFrom[i] || !alignTo[i]; }
        if (masked) {                                                           // (all-true masks need not travel: NULL means all 1)
            nd.align_from = (const uint8_t *) ses.to_device(0, from.data(), from.size());
            nd.align_to = (const uint8_t *) ses.to_device(1, to.data(), to.size());
        }
        alga_prefsuf_params p;
        alga_prefsuf_default_params(&p);
        p.min_overlap = Params::MIN_OVERLAP_PREF_SUF;                           // read at GraphCreatorPrefSuf.cpp:76,167
        p.rsoe_min_overlap = Params::REMOVE_SMALL_OVERLAP_EDGES_MIN_OVERLAP;    // :288,397
        // p.reduction stays ALGA_REDUCTION_AUTO: same graph either way, the engine picks the faster exact form
        const alga_edge *d_edges = nullptr;
        alga_edge *edges = nullptr;
        uint64_t m = 0;
        int rc = alga_prefsuf_build_device(e, &nd, &p, nullptr, &d_edges, &m);
        if (rc != ALGA_OK) alga_adapter::die(e, "overlap graph", rc);
        // the graph comes down in compact form (a degree byte per node, 5 bytes per edge) where it fits -- every short-read set --, as triples else
        alga_compact_edges ce;
        rc = alga_download_edges_compact(e, n, d_edges, m, &ce);
        if (rc == ALGA_OK) {
            alga_adapter::fill_graph_compact(G, ce);
            alga_free_compact_edges(e, &ce);
            return;
        }
        if (rc != ALGA_ERR_UNSUPPORTED) alga_adapter::die(e, "overlap graph (edges to the host)", rc);
        rc = alga_download_edges(e, d_edges, m, &edges);
        if (rc != ALGA_OK) alga_adapter::die(e, "overlap graph (edges to the host)", rc);
        alga_adapter::fill_graph(G, edges, m);
        alga_free_edges(e, edges);
    }

private:
    int device;
};

#endif
