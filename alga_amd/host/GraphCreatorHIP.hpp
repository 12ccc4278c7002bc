// alga_amd/host/GraphCreatorHIP.hpp -- C++ mirror of the reference's GraphCreator interface over the C ABI.
//
// Same names, argument meaning and error behaviour as include/GraphCreators/GraphCreator.h:12-62 and
// include/GraphCreators/GraphCreatorPrefSuf.h:20-30 of the reference, on plain buffers instead of
// vector<Read*> / Graph (those types live in the reference; the adapter that converts them is in INTEGRATION.md).
#pragma once
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/alga_amd.h"

namespace alga_host {

class GraphCreatorPrefSufHIP {
public:
    // `words` / `len` are borrowed, like GraphCreator borrows `reads` (src/GraphCreators/GraphCreator.cpp:10-18)
    GraphCreatorPrefSufHIP(const uint32_t *words, int stride_words, const int32_t *len, int n, int min_overlap, int rsoe_min_overlap,
                           int hip_device = 0)
        : words_(words), stride_(stride_words), len_(len), n_(n), alignFrom((size_t) n, 1), alignTo((size_t) n, 1) {
        alga_prefsuf_default_params(&params_);
        params_.min_overlap = min_overlap;
        params_.rsoe_min_overlap = rsoe_min_overlap;
        int rc = alga_engine_create(hip_device, &engine_);
        if (rc != ALGA_OK) die("cannot create the HIP engine (no usable MI355X / HIP device)", rc);
    }
    // Node set already resident in HBM (e.g. left there by alga_preprocess_nodes on `engine`): the engine is borrowed too.
    // alignFrom / alignTo start all-true (src/main.cpp:253-278 clears them for removed reads only, which carry len 0 here); a
    // mask the caller clears is uploaded next to the node set when the graph is built.
    GraphCreatorPrefSufHIP(alga_engine *engine, const uint32_t *d_words, int stride_words, const int32_t *d_len, int n, int min_overlap,
                           int rsoe_min_overlap)
        : words_(d_words), stride_(stride_words), len_(d_len), n_(n), device_resident_(true), owns_engine_(false),
          alignFrom((size_t) n, 1), alignTo((size_t) n, 1), engine_(engine) {
        alga_prefsuf_default_params(&params_);
        params_.min_overlap = min_overlap;
        params_.rsoe_min_overlap = rsoe_min_overlap;
    }
    ~GraphCreatorPrefSufHIP() {
        clear();
        if (d_from_) alga_device_free(engine_, d_from_);
        if (d_to_) alga_device_free(engine_, d_to_);
        if (engine_ && owns_engine_) alga_engine_destroy(engine_);
    }

    void setAlignTo(int id, bool val) { check(id); alignTo[(size_t) id] = val; masks_touched_ = true; }
    void setAlignFrom(int id, bool val) { check(id); alignFrom[(size_t) id] = val; masks_touched_ = true; }
    bool getAlignTo(int id) const { check(id); return alignTo[(size_t) id] != 0; }
    bool getAlignFrom(int id) const { check(id); return alignFrom[(size_t) id] != 0; }

    // == startAlignmentGraphCreation() followed by the caller's G->retainOnlySmallestOffset() (src/main.cpp:282-291)
    void startAlignmentGraphCreation() {
        clear();
        if (device_resident_) {
            alga_nodes nd{words_, stride_, len_, n_, nullptr, nullptr};
            if (masks_touched_) {                                   // masks travel only when the caller cleared an entry
                upload(d_from_, alignFrom); upload(d_to_, alignTo);
                nd.align_from = d_from_; nd.align_to = d_to_;
            }
            int rc = alga_prefsuf_build_device(engine_, &nd, &params_, nullptr, &d_edges_, &n_edges_);
            if (rc != ALGA_OK) die(alga_last_error(engine_), rc);
            return;
        }
        alga_nodes nd{words_, stride_, len_, n_, alignFrom.data(), alignTo.data()};
        int rc = alga_prefsuf_build_host(engine_, &nd, &params_, &edges_, &n_edges_);
        if (rc != ALGA_OK) die(alga_last_error(engine_), rc);
    }
    const alga_edge *deviceEdges() const { return d_edges_; }    // device-resident form: engine-owned, valid until the next build

    const alga_edge *edges() const { return edges_; }
    uint64_t countEdges() const { return n_edges_; }             // Graph::countEdges
    void clear() { if (edges_) { alga_free_edges(engine_, edges_); edges_ = nullptr; n_edges_ = 0; } }
    alga_prefsuf_stats stats() const { alga_prefsuf_stats s; alga_prefsuf_last_stats(engine_, &s); return s; }
    void collectStats(bool on) { params_.collect_stats = on ? 1 : 0; }

private:
    [[noreturn]] static void die(const char *msg, int rc) {      // the reference's convention: cerr + exit(1)
        std::fprintf(stderr, "alga_amd: %s (status %d)\n", msg, rc);
        std::exit(1);
    }
    void check(int id) const { if (id < 0 || id >= n_) die("node id out of range in setAlign/getAlign", ALGA_ERR_INVALID_ARGUMENT); }
    void upload(uint8_t *&d, const std::vector<uint8_t> &h) {
        if (!d) { int rc = alga_device_alloc(engine_, h.size(), (void **) &d); if (rc != ALGA_OK) die(alga_last_error(engine_), rc); }
        int rc = alga_copy_to_device(engine_, d, h.data(), h.size());
        if (rc != ALGA_OK) die(alga_last_error(engine_), rc);
    }
    uint8_t *d_from_ = nullptr, *d_to_ = nullptr;
    bool masks_touched_ = false;
    const uint32_t *words_; int stride_; const int32_t *len_; int n_;
    bool device_resident_ = false, owns_engine_ = true;
    const alga_edge *d_edges_ = nullptr;
    std::vector<uint8_t> alignFrom, alignTo;
    alga_prefsuf_params params_;
    alga_engine *engine_ = nullptr;
    alga_edge *edges_ = nullptr;
    uint64_t n_edges_ = 0;
};

} // namespace alga_host
