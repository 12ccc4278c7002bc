// oracle/ref_driver.cpp -- TEST INFRASTRUCTURE.  A small driver of my own that LINKS AGAINST THE REFERENCE'S OWN
// OBJECT FILES (everything under /root/reference/src except main.cpp, compiled in place by oracle/Makefile into
// oracle/_ref/obj) and calls its classes directly, to produce function-level golden vectors the stock binary cannot
// dump.  Nothing of the reference is copied: this file only includes its headers at build time.
//
//   ref_driver supplement <nodes.bin> <graph_in> <graph_out> <error_rate_percent> <kmer_length_bucket>
//       rebuilds Global::READS from a node file, loads the pre-supplement graph dump, runs the reference's
//       approximate supplement exactly as src/main.cpp:300-347 drives it (GraphCreatorLI, 4 priority rounds,
//       --threads=1) and writes the resulting graph in the reference's dump format.
//   ref_driver canalign <nodes.bin> <triples.bin> <out.bin> <min_overlap_area> <max_offset_pct> <min_identity_pct>
//       AlignmentControllerHybrid::canAlign on (r1, r2, offset) int32 triples -> one byte each.
//   ref_driver likmers <nodes.bin> <out.bin> <k> <intervals>
//       Read::getKmers for every live node under the 4 rotations of Read::priorities (src/GraphCreators/GraphCreatorLI.cpp:18-28)
//       -> per node and rotation: count, then (hash u64, indInRead i32) pairs, in the order the reference returns them.
//
//   ref_driver triangles <graph_in> <graph_out> <max_offset_parallel_paths>
//       the first step of the simplifier on a graph dump (src/GraphSimplifiers/GraphSimplifier.cpp:90-125: the PrefSuf path of
//       simplifyGraphOld): Graph::sortEdgesByIncreasingOffset, GraphSimplifier::cutNonAndWeaklyMetricTriangles; writes the
//       resulting graph in the reference's dump format (in-list order as the reference leaves it).
//
//   ref_driver trim <contigs.bin> <out.txt> <avg_read_length>
//       the contig-trimming block of src/main.cpp:633-725 on the contigs of a node file: contigs + their reverse complements as
//       "reads", the reference's GraphCreatorPrefSuf with MIN_OVERLAP_PREF_SUF = REMOVE_SMALL_OVERLAP_EDGES_MIN_OVERLAP = 25,
//       trimLeft[d] = longest overlap of an edge between two forward contigs that ends in d, sequences cut accordingly.
//       Writes one line per contig: trimLeft and the trimmed sequence.
//
// node file: i32 n, i32 W, i32 len[n], u32 words[n*W]   (reference bit layout; len 0 = nullptr)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

#include <AlignmentControllers/AlignmentControllerHybrid.h>
#include <GraphCreators/GraphCreatorLI.h>
#include <GraphCreators/GraphCreatorPrefSuf.h>
#include <GraphSimplifiers/GraphSimplifier.h>
#include <Utils/MyUtils.h>
#include <Global.h>
#include <Params.h>

static void die(const char *m) { fprintf(stderr, "ref_driver: %s\n", m); exit(2); }

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

static void init_static() {
    Read::priorities = VI(4);
    std::iota(Read::priorities.begin(), Read::priorities.end(), 0);
    Bitset::initializeStaticBlock();
    Params::THREADS = 1;
}

int main(int argc, char **argv) {
    if (argc < 2) die("usage");
    init_static();
    std::string mode = argv[1];
    if (mode == "supplement") {
        if (argc != 7) die("supplement <nodes> <graph_in> <graph_out> <error_rate_percent> <kmer_length_bucket>");
        load_nodes(argv[2]);
        Global::GRAPH = Graph((int) Global::READS.size());
        Graph *G = &Global::GRAPH;
        if (!G->deserializeGraph(argv[3])) die("cannot load graph");
        Params::ERROR_RATE = atoi(argv[5]);
        Params::KMER_LENGTH_BUCKET = atoi(argv[6]);
        // what the caller of the supplement does before running it
        GraphCreator *gc = new GraphCreatorLI(&Global::READS, G);
        VI *inDeg = G->getInDegrees();
        for (int i = 0; i < G->size(); i++) {
            gc->setAlignFrom(i, false);
            gc->setAlignTo(i, false);
            if ((*inDeg)[i] == 0 && (*G)[i].size() > 0) gc->setAlignTo(i, true);
            if ((*inDeg)[i] > 0 && (*G)[i].size() == 0) gc->setAlignFrom(i, true);
        }
        delete inDeg;
        double avg = Global::calculateAvgReadLength();
        Params::MIN_OVERLAP_AREA = (1.f + Params::SCALE) * avg / 2;
        Params::MAX_OFFSET_CONSIDERED_FOR_ALIGNMENT = (1.f - Params::SCALE) * avg / 2;
        Params::MINIMAL_OVERLAP_FOR_LCS_LOW_ERROR = 99 - Params::ERROR_RATE;
        Params::LI_KMER_INTERVALS = 6;
        Params::LI_KMER_LENGTH = 35;
        fprintf(stdout, "MIN_OVERLAP_AREA %d MAX_OFFSET %d MIN_IDENTITY %d avg %.6f edges_before %lld\n", Params::MIN_OVERLAP_AREA,
                Params::MAX_OFFSET_CONSIDERED_FOR_ALIGNMENT, Params::MINIMAL_OVERLAP_FOR_LCS_LOW_ERROR, avg, (long long) G->countEdges());
        gc->startAlignmentGraphCreation();
        G->retainOnlySmallestOffset();
        delete gc;
        fprintf(stdout, "edges_after %lld\n", (long long) G->countEdges());
        G->serializeGraph(argv[4]);
        return 0;
    }
    if (mode == "triangles") {
        if (argc != 5) die("triangles <graph_in> <graph_out> <max_offset_parallel_paths>");
        FILE *f = fopen(argv[2], "rb");
        if (!f) die("cannot open graph");
        uint32_t n = 0;
        if (fread(&n, 4, 1, f) != 1) die("short graph file");
        fclose(f);
        Global::READS.assign((size_t) n, nullptr);
        Global::GRAPH = Graph((int) n);
        Graph *G = &Global::GRAPH;
        if (!G->deserializeGraph(argv[2])) die("cannot load graph");
        Params::MAX_OFFSET_PARALLEL_PATHS = atoi(argv[4]);
        const long long before = G->countEdges();
        {
            GraphSimplifier simplifier(Global::GRAPH, Global::READS);
            G->sortEdgesByIncreasingOffset();
            simplifier.cutNonAndWeaklyMetricTriangles();
        }
        fprintf(stdout, "edges_before %lld edges_after %lld\n", before, (long long) G->countEdges());
        G->serializeGraph(argv[3]);
        return 0;
    }
    if (mode == "trim") {
        if (argc != 5) die("trim <contigs.bin> <out.txt> <avg_read_length>");
        load_nodes(argv[2]);                               // the contigs, as Reads
        std::vector<Read *> contigs = Global::READS;
        // src/main.cpp:636-656
        std::vector<Read *> newReads;
        for (auto t : contigs) newReads.push_back(t);
        int cnt = 0;
        for (auto t : contigs) newReads.push_back(new Read(cnt++, MyUtils::getComplimentaryString(MyUtils::getReverse(t->getSequenceAsString()))));
        cnt = 0;
        for (auto t : newReads) t->setId(cnt++);
        Graph *newGraph = new Graph((int) newReads.size());
        GraphCreatorPrefSuf gcps(&newReads, newGraph);
        const int THRESHOLD = 25;
        Params::MIN_OVERLAP_PREF_SUF = THRESHOLD;
        Params::REMOVE_SMALL_OVERLAP_EDGES_MIN_OVERLAP = THRESHOLD;
        gcps.startAlignmentGraphCreation();
        // :676-697
        const int M = (int) newReads.size() / 2;
        std::vector<int> trimLeft((size_t) M, 0), trimRight((size_t) M, 0);
        for (int i = 0; i < (int) newReads.size(); i++)
            for (PII neigh : (*newGraph)[i]) {
                const int d = neigh.first, offset = neigh.second, overlap = newReads[i]->size() - offset;
                if (i < M && d < M) trimLeft[d] = std::max(trimLeft[d], overlap);
            }
        FILE *o = fopen(argv[3], "w");
        for (int i = 0; i < M; i++) {                      // :700-712
            std::string sq = newReads[i]->getSequenceAsString();
            if (trimLeft[i] + trimRight[i] + 10 < (int) sq.size()) sq = sq.substr(trimLeft[i], std::max(1, (int) sq.size() - trimLeft[i] - trimRight[i]));
            else sq = "CCCC";
            fprintf(o, "%d %s\n", trimLeft[i], sq.c_str());
        }
        fclose(o);
        fprintf(stdout, "contigs %d edges %lld\n", M, (long long) newGraph->countEdges());
        return 0;
    }
    if (mode == "canalign") {
        if (argc != 8) die("canalign <nodes> <triples> <out> <min_overlap_area> <max_offset_pct> <min_identity_pct>");
        load_nodes(argv[2]);
        Params::MIN_OVERLAP_AREA = atoi(argv[5]);
        Params::MAX_OFFSET_CONSIDERED_FOR_ALIGNMENT = atoi(argv[6]);
        Params::MINIMAL_OVERLAP_FOR_LCS_LOW_ERROR = atoi(argv[7]);
        FILE *f = fopen(argv[3], "rb");
        if (!f) die("cannot open triples");
        std::vector<int32_t> t;
        int32_t buf[3];
        while (fread(buf, 4, 3, f) == 3) { t.push_back(buf[0]); t.push_back(buf[1]); t.push_back(buf[2]); }
        fclose(f);
        AlignmentControllerHybrid ach;
        FILE *o = fopen(argv[4], "wb");
        for (size_t i = 0; i + 2 < t.size(); i += 3) {
            unsigned char r = ach.canAlign(Global::READS[(size_t) t[i]], Global::READS[(size_t) t[i + 1]], t[i + 2]) ? 1 : 0;
            fwrite(&r, 1, 1, o);
        }
        fclose(o);
        return 0;
    }
    if (mode == "likmers") {
        if (argc != 6) die("likmers <nodes> <out> <k> <intervals>");
        load_nodes(argv[2]);
        Params::LI_KMER_LENGTH = atoi(argv[4]);
        Params::LI_KMER_INTERVALS = atoi(argv[5]);
        FILE *o = fopen(argv[3], "wb");
        for (int rot = 0; rot < 4; rot++) {
            for (Read *r : Global::READS) {
                if (r == nullptr || r->size() < Params::LI_KMER_LENGTH) continue;
                vector<Kmer> km = r->getKmers(Params::LI_KMER_LENGTH);
                int32_t c = (int32_t) km.size();
                fwrite(&c, 4, 1, o);
                for (Kmer &k : km) { unsigned long long h = k.hash; int32_t ind = k.indInRead; fwrite(&h, 8, 1, o); fwrite(&ind, 4, 1, o); }
            }
            std::rotate(Read::priorities.begin(), Read::priorities.begin() + 1, Read::priorities.end());
        }
        fclose(o);
        return 0;
    }
    die("unknown mode");
    return 2;
}
