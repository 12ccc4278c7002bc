// alga_amd/host/ingest.cpp -- see ingest.hpp
#include "ingest.hpp"

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <numeric>
#include <thread>
#include <chrono>

namespace alga_host {

namespace {

enum FileType { MY_INPUT, FASTA, PFASTA, FASTQ };

FileType file_type_of(const std::string &path) {             // src/Params.cpp:315-333
    size_t sl = path.rfind('/');
    std::string base = sl == std::string::npos ? path : path.substr(sl + 1);
    size_t dot = base.rfind('.');
    if (dot == std::string::npos) return MY_INPUT;
    std::string ext = base.substr(dot + 1);
    if (ext == "fasta") return FASTA;
    if (ext == "pfasta") return PFASTA;
    if (ext == "fastq" || ext == "fq") return FASTQ;
    return MY_INPUT;
}

inline int blocks_of(int len) { return len <= 0 ? 0 : ((2 * len - 1) >> 5) + 1; }

int min_period(const char *s, int n, std::vector<int> &pre) {   // include/Utils/MyUtils.h:160-170
    if (n <= 0) return 0;
    pre.assign((size_t) n + 1, 0);
    int k = 0;
    for (int q = 1; q < n; q++) {
        while (k > 0 && s[k] != s[q]) k = pre[k];
        if (s[k] == s[q]) k++;
        pre[q + 1] = k;
    }
    return n - pre[n];
}

struct Span { const char *p; int n; };

// split the file into sequence lines (std::getline semantics: '\n' only)
bool load_sequences(const std::string &path, FileType type, std::string &buf, std::vector<Span> &seqs, std::string &err) {
    std::ifstream f(path, std::ios::binary);
    if (!f) { err = "cannot open " + path; return false; }
    f.seekg(0, std::ios::end);
    std::streamoff sz = f.tellg();
    f.seekg(0);
    buf.resize((size_t) sz);
    if (sz) f.read(&buf[0], sz);
    const char *b = buf.data(), *e = b + buf.size();
    auto next_line = [&](const char *&p, Span &out) {
        if (p >= e) { out = {e, 0}; return; }
        const char *nl = (const char *) memchr(p, '\n', (size_t) (e - p));
        if (!nl) { out = {p, (int) (e - p)}; p = e; } else { out = {p, (int) (nl - p)}; p = nl + 1; }
    };
    const char *p = b;
    Span s, skip;
    if (type == MY_INPUT) {
        while (p < e) {
            while (p < e && isspace((unsigned char) *p)) p++;
            const char *q = p;
            while (q < e && !isspace((unsigned char) *q)) q++;
            if (q == p) break;
            seqs.push_back({p, (int) (q - p)});
            p = q;
        }
        return true;
    }
    for (;;) {                                                  // InputReader::readOneRead1, :142-180
        next_line(p, skip);
        next_line(p, s);
        if (type == FASTQ) { next_line(p, skip); next_line(p, skip); }
        if (s.n == 0) break;                                    // an empty sequence line ends the input (:284)
        seqs.push_back(s);
    }
    return true;
}

// is the minimal period of s <= 20 ?  (STR_THRESHOLD, InputReader.cpp:341-353; MyUtils.h:160-170)
//   any period p <= 20 bounds the minimal one, so for reads longer than 20 nt twenty early-exit compares decide it
bool is_str(const char *s, int n, std::vector<int> &pre) {
    if (n <= 20) return min_period(s, n, pre) <= 20;
    for (int p = 1; p <= 20; p++) if (memcmp(s, s + p, (size_t) (n - p)) == 0) return true;
    return false;
}

// character classes of a sequence line: 0..3 = A C G T (the 2-bit codes), 4 = N, 5 = U, 255 = anything else
struct CharTable {
    uint8_t t[256];
    CharTable() { memset(t, 255, sizeof(t)); t[(unsigned char) 'A'] = 0; t[(unsigned char) 'C'] = 1; t[(unsigned char) 'G'] = 2; t[(unsigned char) 'T'] = 3;
                  t[(unsigned char) 'N'] = 4; t[(unsigned char) 'U'] = 5; }
};
const CharTable g_chars;

void pack_codes(const uint8_t *c, int n, uint32_t *w) {        // src/DataStructures/Read.cpp:40-68
    int i = 0, q = 0;
    for (; i + 16 <= n; i += 16, q++) {
        uint32_t v = 0;
        for (int k = 0; k < 16; k++) v |= (uint32_t) c[i + k] << (2 * k);
        w[q] = v;
    }
    if (i < n) {
        uint32_t v = 0;
        for (int k = 0; i + k < n; k++) v |= (uint32_t) c[i + k] << (2 * k);
        w[q] = v;
    }
}

// InputReader::readParallelJob for one record (:286-377): trimmed length or -1 (removed); rows are written when kept.
// `code` / `rcode` are per-thread scratch.  Any other letter than A C G T counts as A in the packed read, like the
// reference's Read::createSequence; U becomes T only with --rna.
bool parse_record(Span in, const IngestParams &p, uint32_t *fw, uint32_t *rc, int &len_out, bool &had_n, bool &was_str,
                  std::vector<uint8_t> &code, std::vector<uint8_t> &rcode, std::vector<int> &pre, uint32_t *rng, std::string &err) {
    const char *s = in.p;
    int n = in.n, b = 0;
    while (b < n && s[b] == ' ') b++;
    int e = b;
    while (e < n && s[e] != ' ') e++;
    s += b; n = e - b;
    if (!(n < p.trim_left + p.trim_right + 10)) {
        int l = std::min(p.trim_left, n);
        s += l; n -= l;
        n -= std::min(p.trim_right, n);
    }
    code.resize((size_t) n + 1); rcode.resize((size_t) n + 1);
    bool containsN = false;
    for (int i = 0; i < n; i++) {
        uint8_t c = g_chars.t[(unsigned char) s[i]];
        if (c == 255) { err = std::string("s[i] = ") + s[i] + "   but should be A,C,G,T,N or U"; return false; }
        if (c == 4) {
            if (p.remove_reads_with_n) { containsN = true; c = 0; }
            else { *rng = (uint32_t) (((uint64_t) *rng * 16807u) % 2147483647u); c = (uint8_t) (*rng & 3); }
        } else if (c == 5) c = p.rna ? 3 : 6;                 // 6: a 'U' that stays a 'U' (packs as A, complements to itself)
        code[(size_t) i] = c;
    }
    had_n = p.remove_reads_with_n && containsN;
    was_str = false;
    len_out = -1;
    if (had_n) return true;
    if (is_str((const char *) code.data(), n, pre)) { was_str = true; return true; }
    len_out = n;
    for (int i = 0; i < n; i++) { const uint8_t c = code[(size_t) (n - 1 - i)]; rcode[(size_t) i] = c < 4 ? (uint8_t) (3 - c) : 0; }   // :23-33
    for (int i = 0; i < n; i++) if (code[(size_t) i] > 3) code[(size_t) i] = 0;
    pack_codes(code.data(), n, fw);
    pack_codes(rcode.data(), n, rc);
    return true;
}

// comparator of ReadPreprocess::getSortedReads (:115-132): bit string with bit 0 most significant, then size, then id
inline bool less_reads(const Parsed &P, int64_t a, int64_t b) {
    const uint32_t *wa = P.row((size_t) a), *wb = P.row((size_t) b);
    int m = std::min(blocks_of(P.len[(size_t) a]), blocks_of(P.len[(size_t) b]));
    for (int q = 0; q < m; q++) {
        if (wa[q] != wb[q]) {
            int ind = __builtin_ctz(wa[q] ^ wb[q]);
            return ((wa[q] >> ind) & 1u) < ((wb[q] >> ind) & 1u);
        }
    }
    if (P.len[(size_t) a] != P.len[(size_t) b]) return P.len[(size_t) a] < P.len[(size_t) b];
    return a < b;
}

inline int lcp_nt(const Parsed &P, int64_t a, int64_t b) {     // Bitset::mismatch >> 1 (Bitset.cpp:858-877)
    const uint32_t *wa = P.row((size_t) a), *wb = P.row((size_t) b);
    int m = std::min(blocks_of(P.len[(size_t) a]), blocks_of(P.len[(size_t) b]));
    int64_t ind = 1000000000;
    for (int i = 0; i < m; i++)
        if (wa[i] != wb[i]) { ind = (int64_t) i * 32 + __builtin_ctz(wa[i] ^ wb[i]); break; }
    int64_t ms = 2 * (int64_t) std::min(P.len[(size_t) a], P.len[(size_t) b]);
    return (int) ((ind < ms ? ind : ms) >> 1);
}

template <class It, class Cmp>
void parallel_sort(It b, It e, Cmp cmp, int threads) {
    size_t n = (size_t) (e - b);
    if (threads <= 1 || n < 50000) { std::sort(b, e, cmp); return; }
    int T = threads;
    std::vector<size_t> cut((size_t) T + 1);
    for (int t = 0; t <= T; t++) cut[(size_t) t] = n * (size_t) t / (size_t) T;
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++) th.emplace_back([&, t] { std::sort(b + (std::ptrdiff_t) cut[(size_t) t], b + (std::ptrdiff_t) cut[(size_t) t + 1], cmp); });
    for (auto &x : th) x.join();
    for (int step = 1; step < T; step *= 2) {
        std::vector<std::thread> mt;
        for (int t = 0; t + step < T; t += 2 * step) {
            size_t lo = cut[(size_t) t], mid = cut[(size_t) (t + step)], hi = cut[(size_t) std::min(T, t + 2 * step)];
            mt.emplace_back([=] { std::inplace_merge(b + (std::ptrdiff_t) lo, b + (std::ptrdiff_t) mid, b + (std::ptrdiff_t) hi, cmp); });
        }
        for (auto &x : mt) x.join();
    }
}

struct Lap {                                               // stage timings on stderr: compile with -DALGA_INGEST_TIMING
#ifdef ALGA_INGEST_TIMING
    bool on = true;
#else
    bool on = false;
#endif
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void operator()(const char *what) {
        if (!on) return;
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "ingest: %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
        t = now;
    }
};

} // namespace

std::string test_name(const std::string &file1, float scale, int remove_reads_with_n) {
    size_t sl = file1.rfind('/');
    std::string base = sl == std::string::npos ? file1 : file1.substr(sl + 1);
    size_t dot = base.rfind('.');
    std::string stem = dot == std::string::npos ? base : base.substr(0, dot);
    return "ALGA_" + stem + "_scale" + std::to_string((int) (100 * scale)) + (remove_reads_with_n ? "_noN" : "_randN");
}

// Stage 1: files -> rows of all nodes in the reference's node order, before any removal by content.
//   node order: per record [rc, r] (InputReader.cpp:78-80); with two files groups of four [rc_i, r_i, rc(p_i), p_i] (:53-76)
std::string parse(const std::string &file1, const std::string &file2, const IngestParams &p, Parsed &P) {
    P = Parsed();
    Lap lap;
    std::string err;
    const FileType type = file_type_of(file1);
    const bool paired = !file2.empty() && type != PFASTA;
    std::string buf[2];
    std::vector<Span> seqs[2];
    if (!load_sequences(file1, type, buf[0], seqs[0], err)) return err;
    if (paired) {
        if (!load_sequences(file2, type, buf[1], seqs[1], err)) return err;
        if (seqs[1].size() != seqs[0].size()) return "paired files differ in record count";
    }
    const size_t n1 = seqs[0].size(), R = paired ? 2 * n1 : n1;
    int maxline = 0;
    for (int f = 0; f < 2; f++) for (const Span &s : seqs[f]) maxline = std::max(maxline, s.n);
    P.R = R; P.paired = paired; P.records = (int64_t) R;
    P.W = std::max(1, blocks_of(maxline));
    P.rows.assign(2 * R * (size_t) P.W, 0u);
    P.len.assign(2 * R, -1);
    lap("read files, split lines");
    // the random replacement of N (remove_reads_with_n = 0) draws from one generator in file order: serial then
    const int T = p.remove_reads_with_n ? std::max(1, p.threads) : 1;
    for (int f = 0; f < (paired ? 2 : 1); f++) {
        std::vector<std::thread> th;
        std::vector<int> rn((size_t) T, 0), rs((size_t) T, 0);
        std::vector<std::string> errs((size_t) T);
        std::atomic<size_t> next{0};
        const size_t CH = 4096;
        const std::vector<Span> &sq = seqs[f];
        auto work = [&](int t) {
            std::vector<uint8_t> code, rcode;
            std::vector<int> pre;
            int my_n = 0, my_str = 0;                           // thread-local tallies (no shared cache line in the loop)
            uint32_t rng = 1;                                   // std::minstd_rand0(0)
            for (;;) {
                size_t s0 = next.fetch_add(CH);
                if (s0 >= sq.size()) break;
                size_t s1 = std::min(sq.size(), s0 + CH);
                for (size_t i = s0; i < s1; i++) {
                    const size_t k = paired ? 2 * i + (size_t) f : i;       // read slot in node order
                    bool hn, st;
                    int len;
                    if (!parse_record(sq[i], p, P.row(2 * k + 1), P.row(2 * k), len, hn, st, code, rcode, pre, &rng, errs[(size_t) t])) return;
                    P.len[2 * k] = P.len[2 * k + 1] = len;
                    my_n += hn; my_str += st;
                }
            }
            rn[(size_t) t] = my_n; rs[(size_t) t] = my_str;
        };
        if (T == 1) work(0);
        else { for (int t = 0; t < T; t++) th.emplace_back(work, t); for (auto &x : th) x.join(); }
        for (int t = 0; t < T; t++) {
            if (!errs[(size_t) t].empty()) return errs[(size_t) t];
            P.removed_n += 2 * rn[(size_t) t]; P.removed_str += 2 * rs[(size_t) t];
        }
    }
    lap("parse + pack");
    // src/main.cpp:93-115
    double sum = 0; int64_t cnt = 0;
    for (size_t i = 0; i < 2 * R; i++) if (P.len[i] >= 0) { sum += P.len[i]; cnt++; }
    P.live = cnt;
    P.avg_len = cnt ? sum / (double) cnt : 0.0;
    P.LEN = (int) (P.avg_len + p.trim_left + p.trim_right);
    int Lmin = p.min_overlap, rso = p.rsoemo, likl;
    if (Lmin == -1) {
        int L = (int) ((float) P.LEN * p.scale);
        int RSOEMO = (int) ((float) P.LEN * (p.scale + 1) / 2);
        likl = std::min(2 * L / 3, 60);
        Lmin = L;
        if (rso == -1) rso = RSOEMO;
    } else {
        likl = Lmin;
        if (rso == -1) rso = (Lmin + P.LEN) / 2;
    }
    P.min_overlap = Lmin; P.rsoemo = rso; P.li_kmer_length = likl;
    lap("parameters");
    return "";
}

// Stage 2 on the host: duplicate / prefix removal, compaction, short-read removal.  (The same stage on the GPU:
// alga_preprocess_nodes_device, alga_amd/csrc/ingest_kernels.hip.)
std::string preprocess_host(Parsed &P, const IngestParams &p, NodeSet &out) {
    out = NodeSet();
    Lap lap;
    const size_t R = P.R;
    // src/IO/ReadPreprocess.cpp:13-77
    int removed_prefix = 0;
    if (p.remove_pref_reads != 3) {
        std::vector<int64_t> ord;
        ord.reserve((size_t) P.live);
        for (size_t i = 0; i < 2 * R; i++) if (P.len[i] >= 0) ord.push_back((int64_t) i);
        parallel_sort(ord.begin(), ord.end(), [&](int64_t a, int64_t b) { return less_reads(P, a, b); }, p.threads);
        lap("sort");
        std::vector<uint8_t> mark(2 * R, 0);
        for (size_t i = 0; i + 1 < ord.size(); i++) {
            int64_t a = ord[i], b = ord[i + 1];
            int l = lcp_nt(P, a, b);
            if (p.remove_pref_reads == 1) { if (l == P.len[(size_t) a] && P.len[(size_t) a] == P.len[(size_t) b]) mark[(size_t) a] = 1; }
            else if (l == P.len[(size_t) a]) {
                mark[(size_t) a] = 1;
                if (P.len[(size_t) a] < P.len[(size_t) b]) mark[(size_t) (a ^ 1)] = 1;
            }
        }
        for (size_t i = 0; i < 2 * R; i++) if (mark[i]) { removed_prefix++; P.len[i] = -1; }
        lap("mark duplicates / prefixes");
    }
    // compaction (src/main.cpp:150-232)
    int maxlen = 0;
    size_t nn = 0;
    for (size_t i = 0; i + 1 < 2 * R; i += 2) if (P.len[i] >= 0) nn += 2;
    for (size_t i = 0; i < 2 * R; i++) maxlen = std::max(maxlen, P.len[i]);
    if (nn > 0x7FFFFFFEull) return "too many nodes";
    int stride = (std::max(1, blocks_of(maxlen)) + 3) & ~3;
    out.n = (int32_t) nn; out.stride = stride;
    out.words.assign(nn * (size_t) stride, 0u);
    out.len.assign(nn, 0);
    out.pair_off.assign(nn, 0);
    size_t bi = 0;
    std::string cerr_;
    auto copy_pair = [&](size_t i, int po) {
        for (int k = 0; k < 2; k++) {
            size_t s = i + (size_t) k;
            if (P.len[s] < 0) { cerr_ = "a read is kept but its reverse complement is removed (the reference asserts, src/main.cpp:171)"; return; }
            out.len[bi] = P.len[s];
            std::copy(P.row(s), P.row(s) + blocks_of(P.len[s]), out.words.begin() + (std::ptrdiff_t) (bi * (size_t) stride));
            out.pair_off[bi] = (uint8_t) po;
            bi++;
        }
    };
    for (size_t i = 0; i + 1 < 2 * R && cerr_.empty(); i += 2) {
        if (P.len[i] < 0) continue;
        if ((i & 3) == 0) {
            if (i + 2 < 2 * R && P.len[i + 2] >= 0) { copy_pair(i, 1); if (cerr_.empty()) copy_pair(i + 2, 2); }
            else copy_pair(i, 0);
        } else if (P.len[i - 2] < 0) copy_pair(i, 0);
    }
    if (!cerr_.empty()) return cerr_;
    if (bi != nn) return "compaction mismatch";
    // src/main.cpp:253-266
    for (size_t i = 0; i < nn; i++) {
        if (out.len[i] < 3 + P.li_kmer_length) {
            out.len[i] = 0; out.removed_short++;
            std::fill(out.words.begin() + (std::ptrdiff_t) (i * (size_t) stride), out.words.begin() + (std::ptrdiff_t) ((i + 1) * (size_t) stride), 0u);
        }
    }
    lap("compaction");
    out.records = P.records;
    out.LEN = P.LEN; out.min_overlap = P.min_overlap; out.rsoemo = P.rsoemo; out.li_kmer_length = P.li_kmer_length;
    out.removed_n = P.removed_n; out.removed_str = P.removed_str; out.removed_prefix = removed_prefix; out.avg_len = P.avg_len;
    return "";
}

std::string ingest(const std::string &file1, const std::string &file2, const IngestParams &p, NodeSet &out) {
    Parsed P;
    std::string err = parse(file1, file2, p, P);
    if (!err.empty()) return err;
    return preprocess_host(P, p, out);
}

} // namespace alga_host
