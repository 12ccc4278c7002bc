// tests/sanitize/ingest_asan.cpp -- the host-side input stages (alga_amd/host/ingest.cpp behind include/alga_amd.h's
// alga_ingest_files / alga_parse_files) under AddressSanitizer + UndefinedBehaviorSanitizer, on the CPU: the golden fixtures' inputs and
// a few hundred generated messy FASTA / FASTQ files (N, lower case, short reads, STRs, duplicates, blank lines, CRLF, missing final
// newline, truncated records, empty files).  TEST INFRASTRUCTURE: built and run by tests/test_host_sanitize_cpu.py.
//   g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=undefined -pthread ingest_asan.cpp ../../alga_amd/host/ingest.cpp ../../alga_amd/host/host_capi.cpp
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../../include/alga_amd.h"

static int run_one(const char *f1, const char *f2, int threads, bool expect_ok, long *nodes) {
    alga_ingest_params p;
    alga_ingest_default_params(&p);
    p.threads = threads;
    char err[512] = {0};
    alga_node_set ns;
    int rc = alga_ingest_files(f1, f2, &p, &ns, err, sizeof err);
    if (rc == ALGA_OK) {
        uint64_t sum = 0;                                  // touch every byte the call handed out
        for (int64_t i = 0; i < (int64_t) ns.n * ns.stride_words; i++) sum += ns.words[i];
        for (int64_t i = 0; i < ns.n; i++) sum += (uint64_t) ns.len[i] + ns.pair_off[i];
        if (nodes) *nodes += ns.n + (long) (sum & 1);
        alga_free_node_set(&ns);
    }
    alga_parsed_reads pr;
    int rc2 = alga_parse_files(f1, f2, &p, &pr, err, sizeof err);
    if (rc2 == ALGA_OK) {
        uint64_t sum = 0;
        for (int64_t i = 0; i < pr.n_nodes / 2 * pr.stride_words; i++) sum += pr.rows[i];
        if (nodes) *nodes += (long) (sum & 1);
        alga_free_parsed_reads(&pr);
    }
    if (expect_ok && (rc != ALGA_OK || rc2 != ALGA_OK)) { fprintf(stderr, "unexpected failure on %s: %d %d %s\n", f1, rc, rc2, err); return 1; }
    return 0;
}

static std::string messy(std::mt19937_64 &g, bool fastq, bool broken) {
    std::uniform_int_distribution<int> len(0, 260), coin(0, 99), base(0, 3);
    const char *ACGT = "ACGT", *acgt = "acgt";
    std::string s;
    const int n = 1 + (int) (g() % 400);
    std::string genome;
    for (int i = 0; i < 3000; i++) genome += ACGT[base(g)];
    for (int r = 0; r < n; r++) {
        const char *eol = coin(g) < 10 ? "\r\n" : "\n";
        s += fastq ? "@r" : ">r"; s += std::to_string(r); if (coin(g) < 20) s += " some text/1"; s += eol;
        int L = coin(g) < 70 ? 150 : len(g);
        std::string seq;
        if (coin(g) < 8) { const int per = 1 + (int) (g() % 12); for (int i = 0; i < L; i++) seq += ACGT[(i % per) % 4]; }   // STR
        else { const size_t st = g() % (genome.size() - 261); seq = genome.substr(st, (size_t) L); }
        for (auto &c : seq) { const int x = coin(g); if (x < 1) c = 'N'; else if (x < 3) c = acgt[base(g)]; else if (x < 4 && broken) c = "XU*-"[g() % 4]; }
        if (!fastq && coin(g) < 15 && seq.size() > 80) { seq.insert(seq.size() / 2, eol); }       // multi-line FASTA record
        s += seq; s += eol;
        if (fastq) { s += "+"; s += eol; s += std::string(broken && coin(g) < 5 ? (size_t) L / 2 : seq.size(), 'I'); s += eol; }
        if (coin(g) < 3) s += eol;                         // blank line
    }
    if (coin(g) < 30 && !s.empty()) s.pop_back();          // no final newline
    if (broken && coin(g) < 30) s.resize(s.size() * (size_t) (40 + coin(g) % 60) / 100);           // truncated file
    return s;
}

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: ingest_asan <scratch dir> [fixture file1[:file2]] ...\n"); return 2; }
    const std::string dir = argv[1];
    long nodes = 0;
    int bad = 0;
    for (int a = 2; a < argc; a++) {                       // the golden inputs: must parse
        std::string f1 = argv[a], f2;
        const size_t c = f1.find(':');
        if (c != std::string::npos) { f2 = f1.substr(c + 1); f1.resize(c); }
        for (int t : {1, 4}) bad += run_one(f1.c_str(), f2.empty() ? nullptr : f2.c_str(), t, true, &nodes);
    }
    std::mt19937_64 g(12345);
    for (int k = 0; k < 240; k++) {                        // generated: may be rejected, must not misbehave
        const bool fastq = k % 2, broken = k % 3 == 0, paired = k % 5 == 0;
        const std::string p1 = dir + "/m1." + (fastq ? "fastq" : "fasta"), p2 = dir + "/m2." + (fastq ? "fastq" : "fasta");
        for (const std::string &p : {p1, p2}) {
            FILE *f = fopen(p.c_str(), "wb");
            if (!f) { perror("fopen"); return 2; }
            const std::string body = (k == 7) ? std::string() : messy(g, fastq, broken);
            fwrite(body.data(), 1, body.size(), f);
            fclose(f);
        }
        bad += run_one(p1.c_str(), paired ? p2.c_str() : nullptr, 1 + k % 4, false, &nodes);
    }
    bad += run_one((dir + "/does_not_exist.fasta").c_str(), nullptr, 1, false, &nodes);
    printf("ingest_asan: %d unexpected failures, %ld nodes seen\n", bad, nodes);
    return bad ? 1 : 0;
}
