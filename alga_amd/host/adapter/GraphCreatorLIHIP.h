// alga_amd/host/adapter/GraphCreatorLIHIP.h -- REFERENCE-SIDE binding of the approximate supplement (error_rate > 0.01).
//
// Replaces `new GraphCreatorLI(&Global::READS, &Global::GRAPH)` at src/main.cpp:306: the caller sets the tip masks
// (:308-322) and the Params of :332-340 exactly as before, then startAlignmentGraphCreation() runs the four LI rounds on the
// GPU (alga_pkb_supplement_host) on top of the graph G holds and refills G; the caller's retainOnlySmallestOffset (:347) is
// a no-op on the result.  Built and tested like GraphCreatorPrefSufHIP.h.
#ifndef ALGA_AMD_GRAPHCREATORLIHIP_H
#define ALGA_AMD_GRAPHCREATORLIHIP_H

#include "GraphCreatorPrefSufHIP.h"

class GraphCreatorLIHIP : public GraphCreator {
public:
    GraphCreatorLIHIP(std::vector<Read *> *reads, Graph *G, int hip_device = 0) : GraphCreator(reads, G), device(hip_device) {}

    void startAlignmentGraphCreation() override {
        const int n = G->size();
        alga_adapter::Session &ses = alga_adapter::Session::get(device);
        alga_engine *e = ses.engine();
        alga_nodes nd = ses.nodes_of(*reads);                                   // resident since the exact graph was built: no second upload
        std::vector<alga_edge> in;                                             // the exact graph, lists sorted by (dst, offset)
        for (int a = 0; a < n; a++) {
            VPII row = (*G)[a];
            std::sort(row.begin(), row.end());
            for (const PII &x : row) in.push_back(alga_edge{a, x.first, x.second});
        }
        // The engine derives the tip masks from the degrees of the incoming graph itself (src/main.cpp:308-322) and takes the
        // parameters the caller has just stored in Params (src/main.cpp:332-340).
        alga_pkb_params p;
        p.min_overlap_area = Params::MIN_OVERLAP_AREA;
        p.max_offset_pct = Params::MAX_OFFSET_CONSIDERED_FOR_ALIGNMENT;
        p.min_identity_pct = Params::MINIMAL_OVERLAP_FOR_LCS_LOW_ERROR;
        p.same_ends = Params::ALIGNMENT_CONTROLLER_SAME_ENDS_LENGTH;
        p.li_k = Params::LI_KMER_LENGTH;
        p.li_intervals = Params::LI_KMER_INTERVALS;
        p.rounds = 4;
        p.kmer_length_bucket = Params::KMER_LENGTH_BUCKET;
        const alga_edge *d_in = (const alga_edge *) ses.to_device(2, in.data(), in.size() * sizeof(alga_edge));
        const alga_edge *d_out = nullptr;
        alga_edge *out = nullptr;
        uint64_t m = 0;
        int rc = alga_pkb_supplement_device(e, &nd, &p, d_in, (uint64_t) in.size(), nullptr, &d_out, &m);
        if (rc != ALGA_OK) alga_adapter::die(e, "approximate supplement", rc);
        rc = alga_download_edges(e, d_out, m, &out);
        if (rc != ALGA_OK) alga_adapter::die(e, "approximate supplement (edges to the host)", rc);
        alga_adapter::fill_graph(G, out, m);
        alga_free_edges(e, out);
    }

private:
    int device;
};

#endif
