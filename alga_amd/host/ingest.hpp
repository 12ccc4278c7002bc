// alga_amd/host/ingest.hpp -- host side (C++, multithreaded) of the drop-in: reads -> node set.
//
// Reproduces what the reference does between its command line and the GraphCreator constructor
// (paths relative to the reference root):
//   src/IO/InputReader.cpp:44-139,272-391   record parsing, end trimming, N / STR filters, 2-bit packing,
//                                            reverse-complement twins, pair interleave, [rc, r] order
//   src/main.cpp:93-115                      parameter derivation (mixed float/int, truncating)
//   src/IO/ReadPreprocess.cpp:13-152         duplicate / prefix read removal
//   src/main.cpp:150-232,253-266             id compaction, pairedReadOffset, removal of too-short reads
// in the --threads=1 order of the reference (file order), whatever number of threads is used here.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace alga_host {

struct IngestParams {
    int   trim_left = 3, trim_right = 3;     // Params::READ_END_TRIM_* (src/Params.cpp:729-730)
    int   remove_reads_with_n = 1;           // src/Params.cpp:740
    int   rna = 0;
    float scale = 0.55f;                     // Params::SCALE
    int   min_overlap = -1;                  // -l / mfup ; -1 = derive
    int   rsoemo = -1;                       // --rsoemo  ; -1 = derive
    int   remove_pref_reads = 2;             // 1 duplicates, 2 all prefix reads (default), 3 none
    int   threads = 1;
};

struct NodeSet {
    int32_t n = 0;
    int32_t stride = 0;                      // uint32 words per row, multiple of 4
    std::vector<uint32_t> words;             // n * stride, reference bit layout
    std::vector<int32_t>  len;               // 0 = removed node
    std::vector<uint8_t>  pair_off;          // Global::pairedReadOffset
    int LEN = 0, min_overlap = 0, rsoemo = 0, li_kmer_length = 0;
    int64_t records = 0;
    int removed_n = 0, removed_str = 0, removed_prefix = 0, removed_short = 0;
    double avg_len = 0;
};

// Stage 1 result: every record's two nodes in the reference's node order, before the removals that depend on other reads.
struct Parsed {
    size_t R = 0;                            // reads (= records; both files)
    bool   paired = false;
    int    W = 0;                            // uint32 words per row
    std::vector<uint32_t> rows;              // 2R rows: node 2k = reverse complement, 2k+1 = forward of read k
    std::vector<int32_t>  len;               // 2R; -1 = removed (N / STR)
    int64_t records = 0, live = 0;
    int removed_n = 0, removed_str = 0;
    int LEN = 0, min_overlap = 0, rsoemo = 0, li_kmer_length = 0;
    double avg_len = 0;
    uint32_t *row(size_t i) { return rows.data() + i * (size_t) W; }
    const uint32_t *row(size_t i) const { return rows.data() + i * (size_t) W; }
};

// Each returns "" on success, else an error message (the reference would print it and exit(1)).
std::string parse(const std::string &file1, const std::string &file2, const IngestParams &p, Parsed &out);
std::string preprocess_host(Parsed &P, const IngestParams &p, NodeSet &out);      // consumes P.len
std::string ingest(const std::string &file1, const std::string &file2, const IngestParams &p, NodeSet &out);

// "ALGA_<basename of file1 without extension>_scale<100*scale>_<noN|randN>" (src/Params.cpp:343,554-557)
std::string test_name(const std::string &file1, float scale, int remove_reads_with_n);

} // namespace alga_host
