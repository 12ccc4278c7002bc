// alga_amd/host/ingest.cpp -- see ingest.hpp
#include "ingest.hpp"

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <numeric>
#include <thread>

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

struct RawRead {               // one parsed record = two nodes (forward, reverse complement)
    std::vector<uint32_t> fw, rc;
    int len = -1;              // -1: removed (N / STR)
};

void pack(const char *s, int n, std::vector<uint32_t> &w) {    // src/DataStructures/Read.cpp:40-68
    w.assign((size_t) std::max(1, blocks_of(n)), 0u);
    for (int i = 0; i < n; i++) {
        uint32_t v = s[i] == 'C' ? 1u : s[i] == 'G' ? 2u : s[i] == 'T' ? 3u : 0u;
        w[(size_t) i >> 4] |= v << ((i & 15) << 1);
    }
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

// InputReader::readParallelJob for one record (:286-377)
bool parse_record(Span in, const IngestParams &p, RawRead &out, bool &had_n, bool &was_str, std::string &tmp, std::vector<int> &pre,
                  uint32_t *rng, std::string &err) {
    const char *s = in.p;
    int n = in.n, b = 0;
    while (b < n && s[b] == ' ') b++;
    int e = b;
    while (e < n && s[e] != ' ') e++;
    tmp.assign(s + b, (size_t) (e - b));
    n = (int) tmp.size();
    if (!(n < p.trim_left + p.trim_right + 10)) {
        int l = std::min(p.trim_left, n);
        tmp.erase(0, (size_t) l);
        n -= l;
        int r = std::min(p.trim_right, n);
        tmp.erase((size_t) (n - r));
        n -= r;
    }
    bool containsN = false;
    for (int i = 0; i < n; i++) {
        char c = tmp[(size_t) i];
        if (c != 'A' && c != 'C' && c != 'G' && c != 'T' && c != 'N' && c != 'U') {
            err = std::string("s[i] = ") + c + "   but should be A,C,G,T,N or U";
            return false;
        }
        if (c == 'N' && p.remove_reads_with_n) containsN = true;
        else if (c == 'N') { *rng = (uint32_t) (((uint64_t) *rng * 16807u) % 2147483647u); tmp[(size_t) i] = "ACGT"[*rng & 3]; }
        else if (p.rna && c == 'U') tmp[(size_t) i] = 'T';
    }
    had_n = p.remove_reads_with_n && containsN;
    was_str = false;
    out.len = -1;
    if (had_n) return true;
    if (min_period(tmp.data(), n, pre) <= 20) { was_str = true; return true; }     // STR_THRESHOLD, :341-353
    out.len = n;
    pack(tmp.data(), n, out.fw);
    std::reverse(tmp.begin(), tmp.end());
    for (char &c : tmp) c = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : c;   // :23-33
    pack(tmp.data(), n, out.rc);
    return true;
}

bool read_file(const std::string &path, FileType type, const IngestParams &p, std::vector<RawRead> &reads, int &n_removed, int &str_removed,
               std::string &err) {
    std::string buf;
    std::vector<Span> seqs;
    if (!load_sequences(path, type, buf, seqs, err)) return false;
    const size_t base = reads.size();
    reads.resize(base + seqs.size());
    // the random replacement of N (remove_reads_with_n = 0) draws from one generator in file order: serial then
    int T = p.remove_reads_with_n ? std::max(1, p.threads) : 1;
    std::vector<std::thread> th;
    std::vector<int> rn((size_t) T, 0), rs((size_t) T, 0);
    std::vector<std::string> errs((size_t) T);
    std::atomic<size_t> next{0};
    const size_t CH = 4096;
    auto work = [&](int t) {
        std::string tmp;
        std::vector<int> pre;
        uint32_t rng = 1;                                       // std::minstd_rand0(0)
        for (;;) {
            size_t s0 = next.fetch_add(CH);
            if (s0 >= seqs.size()) break;
            size_t s1 = std::min(seqs.size(), s0 + CH);
            for (size_t i = s0; i < s1; i++) {
                bool hn, st;
                if (!parse_record(seqs[i], p, reads[base + i], hn, st, tmp, pre, &rng, errs[(size_t) t])) return;
                rn[(size_t) t] += hn; rs[(size_t) t] += st;
            }
        }
    };
    if (T == 1) work(0);
    else { for (int t = 0; t < T; t++) th.emplace_back(work, t); for (auto &x : th) x.join(); }
    for (int t = 0; t < T; t++) {
        if (!errs[(size_t) t].empty()) { err = errs[(size_t) t]; return false; }
        n_removed += rn[(size_t) t]; str_removed += rs[(size_t) t];
    }
    return true;
}

// node view during preprocessing: index 2i = twin A, 2i+1 = twin B of record i (after the [rc, r] swap: even = rc)
struct Pre {
    std::vector<const std::vector<uint32_t> *> w;
    std::vector<int> len;                                       // -1 = nullptr
};

// comparator of ReadPreprocess::getSortedReads (:115-132): bit string with bit 0 most significant, then size, then id
inline bool less_reads(const Pre &P, int64_t a, int64_t b) {
    const std::vector<uint32_t> &wa = *P.w[(size_t) a], &wb = *P.w[(size_t) b];
    int m = std::min(blocks_of(P.len[(size_t) a]), blocks_of(P.len[(size_t) b]));
    for (int q = 0; q < m; q++) {
        if (wa[(size_t) q] != wb[(size_t) q]) {
            int ind = __builtin_ctz(wa[(size_t) q] ^ wb[(size_t) q]);
            return ((wa[(size_t) q] >> ind) & 1u) < ((wb[(size_t) q] >> ind) & 1u);
        }
    }
    if (P.len[(size_t) a] != P.len[(size_t) b]) return P.len[(size_t) a] < P.len[(size_t) b];
    return a < b;
}

inline int lcp_nt(const Pre &P, int64_t a, int64_t b) {        // Bitset::mismatch >> 1 (Bitset.cpp:858-877)
    const std::vector<uint32_t> &wa = *P.w[(size_t) a], &wb = *P.w[(size_t) b];
    int m = std::min(blocks_of(P.len[(size_t) a]), blocks_of(P.len[(size_t) b]));
    int64_t ind = 1000000000;
    for (int i = 0; i < m; i++)
        if (wa[(size_t) i] != wb[(size_t) i]) { ind = (int64_t) i * 32 + __builtin_ctz(wa[(size_t) i] ^ wb[(size_t) i]); break; }
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

} // namespace

std::string test_name(const std::string &file1, float scale, int remove_reads_with_n) {
    size_t sl = file1.rfind('/');
    std::string base = sl == std::string::npos ? file1 : file1.substr(sl + 1);
    size_t dot = base.rfind('.');
    std::string stem = dot == std::string::npos ? base : base.substr(0, dot);
    return "ALGA_" + stem + "_scale" + std::to_string((int) (100 * scale)) + (remove_reads_with_n ? "_noN" : "_randN");
}

std::string ingest(const std::string &file1, const std::string &file2, const IngestParams &p, NodeSet &out) {
    out = NodeSet();
    std::string err;
    FileType type = file_type_of(file1);
    std::vector<RawRead> reads;
    int nrem = 0, strrem = 0;
    if (!read_file(file1, type, p, reads, nrem, strrem, err)) return err;
    size_t n1 = reads.size();
    bool paired = !file2.empty() && type != PFASTA;
    if (paired) {
        if (!read_file(file2, type, p, reads, nrem, strrem, err)) return err;
        if (reads.size() != 2 * n1) return "paired files differ in record count";
    }
    out.records = (int64_t) reads.size();
    // node order: per record [rc, r] (InputReader.cpp:78-80); with two files groups of four
    // [rc_i, r_i, rc(p_i), p_i] (:53-76)
    const size_t R = reads.size();
    std::vector<size_t> order(R);
    if (paired) { for (size_t i = 0; i < n1; i++) { order[2 * i] = i; order[2 * i + 1] = n1 + i; } }
    else std::iota(order.begin(), order.end(), (size_t) 0);
    Pre P;
    P.w.resize(2 * R); P.len.resize(2 * R);
    for (size_t k = 0; k < R; k++) {
        const RawRead &r = reads[order[k]];
        P.w[2 * k] = &r.rc; P.w[2 * k + 1] = &r.fw;
        P.len[2 * k] = P.len[2 * k + 1] = r.len;
    }
    // src/main.cpp:93-115
    double sum = 0; int64_t cnt = 0;
    for (size_t i = 0; i < 2 * R; i++) if (P.len[i] >= 0) { sum += P.len[i]; cnt++; }
    double avg = cnt ? sum / (double) cnt : 0.0;
    int LEN = (int) (avg + p.trim_left + p.trim_right);
    int Lmin = p.min_overlap, rso = p.rsoemo, likl;
    if (Lmin == -1) {
        int L = (int) ((float) LEN * p.scale);
        int RSOEMO = (int) ((float) LEN * (p.scale + 1) / 2);
        likl = std::min(2 * L / 3, 60);
        Lmin = L;
        if (rso == -1) rso = RSOEMO;
    } else {
        likl = Lmin;
        if (rso == -1) rso = (Lmin + LEN) / 2;
    }
    // src/IO/ReadPreprocess.cpp:13-77
    int removed_prefix = 0;
    if (p.remove_pref_reads != 3) {
        std::vector<int64_t> ord;
        ord.reserve((size_t) cnt);
        for (size_t i = 0; i < 2 * R; i++) if (P.len[i] >= 0) ord.push_back((int64_t) i);
        parallel_sort(ord.begin(), ord.end(), [&](int64_t a, int64_t b) { return less_reads(P, a, b); }, p.threads);
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
            std::copy(P.w[s]->begin(), P.w[s]->begin() + blocks_of(P.len[s]), out.words.begin() + (std::ptrdiff_t) (bi * (size_t) stride));
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
        if (out.len[i] < 3 + likl) {
            out.len[i] = 0; out.removed_short++;
            std::fill(out.words.begin() + (std::ptrdiff_t) (i * (size_t) stride), out.words.begin() + (std::ptrdiff_t) ((i + 1) * (size_t) stride), 0u);
        }
    }
    out.LEN = LEN; out.min_overlap = Lmin; out.rsoemo = rso; out.li_kmer_length = likl;
    out.removed_n = 2 * nrem; out.removed_str = 2 * strrem; out.removed_prefix = removed_prefix; out.avg_len = avg;
    return "";
}

} // namespace alga_host
