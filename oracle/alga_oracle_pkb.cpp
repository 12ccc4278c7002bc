/*
 * oracle/alga_oracle_pkb.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see alga_oracle.h).
 *
 * CPU restatement of the reference's approximate supplement (error_rate > 0.01)
 * in its --threads=1 order.  C++ only because the processing order of k-mer groups leaks the tie behaviour of
 * libstdc++'s std::sort (src/GraphCreators/GraphCreatorKmerBased.cpp:94-106): using the same std::sort with an
 * equivalent comparator on the same initial order reproduces it.  Pinned by tests/golden/f7_pkb.* (vectors made by
 * the reference's own code through oracle/ref_driver.cpp).
 */
#include "alga_oracle.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

inline int nt_at(const uint32_t *w, int pos) { return (int) ((w[pos >> 4] >> ((pos & 15) << 1)) & 3u); }
inline int blocks_of(int len) { return len <= 0 ? 0 : ((2 * len - 1) >> 5) + 1; }

inline uint32_t bits32_at(const uint32_t *w, int nw, int bit) {
    int q = bit >> 5, r = bit & 31;
    uint32_t lo = q < nw ? w[q] : 0u;
    if (r == 0) return lo;
    uint32_t hi = (q + 1) < nw ? w[q + 1] : 0u;
    return (lo >> r) | (hi << (32 - r));
}

/* popcount of bits [a, b] (inclusive, Bitset::count(a,b), src/DataStructures/Bitset.cpp:438-477) of X = (r1 >> 2*off) ^ r2,
 * where the xor only covers min(blocks) blocks (operator^=, :395-399) */
int xor_count(const uint32_t *r1, int n1, const uint32_t *r2, int n2, int off, int a, int b) {
    int m = n1 < n2 ? n1 : n2;
    int cnt = 0;
    for (int blk = a >> 5; blk <= (b >> 5); blk++) {
        uint32_t x = bits32_at(r1, n1, 2 * off + 32 * blk);
        if (blk >= n1) x = 0;                 /* operator<<= zero-fills past the end, Bitset.cpp:116-163 */
        if (blk < m) x ^= r2[blk];
        int lo = blk == (a >> 5) ? (a & 31) : 0;
        int hi = blk == (b >> 5) ? (b & 31) : 31;
        uint32_t mask = (hi == 31 ? 0xFFFFFFFFu : ((1u << (hi + 1)) - 1u)) & ~((1u << lo) - 1u);
        cnt += __builtin_popcount(x & mask);
    }
    return cnt;
}

} // namespace

extern "C" {

/* AlignmentControllerHybrid::canAlign (src/AlignmentControllers/AlignmentControllerHybrid.cpp:46-83) with the reference's
 * defaults USE_LCS_LOW_ERROR_FILTER = 1, USE_ACLER_INSTEAD_OF_ACLCS = 1 (src/Params.cpp:702-703), i.e.
 * AlignmentControllerLowErrorRate::canAlign (src/AlignmentControllers/AlignmentControllerLowErrorRate.cpp:15-49). */
int oracle_can_align(const uint32_t *words, const int32_t *len, int32_t W, int32_t r1, int32_t r2, int32_t offset,
                     const oracle_pkb_params *p) {
    const int l1 = len[r1], l2 = len[r2];
    if (100 * offset > p->max_offset_pct * l1) return 0;                       /* Hybrid :50-52 */
    if (offset < 0) return 0;                                                  /* MIN_OFFSET_FOR_ALIGNMENT = 0, :54 */
    const int ov = (l1 < l2 + offset ? l1 : l2 + offset) - offset;             /* Read::calculateReadOverlap */
    if (ov < p->min_overlap_area) return 0;                                    /* :57 */
    if (l2 + offset - l1 < 0) return 0;                                        /* Read::getRightOffset, :59 */
    const uint32_t *a = words + (size_t) r1 * W, *b = words + (size_t) r2 * W;
    const int n1 = blocks_of(l1), n2 = blocks_of(l2);
    int seq = (ov << 1) - xor_count(a, n1, b, n2, offset, 0, (ov << 1) - 1);   /* LowErrorRate :36-38 */
    seq >>= 1;
    const int se = p->same_ends;
    if (xor_count(a, n1, b, n2, offset, 0, se << 1) != 0) return 0;            /* :43 (inclusive: 2*se+1 bits) */
    if (xor_count(a, n1, b, n2, offset, (ov - se) << 1, (ov << 1) - 1) != 0) return 0;   /* :44 */
    return 100 * seq >= p->min_identity_pct * ov ? 1 : 0;                      /* :47 */
}

/* Read::getLIKmers (src/DataStructures/Read.cpp:145-226): per interval of start positions the k-mer that is
 * lexicographically smallest under the alphabet permutation `prio`; hash = value mod 10^18+3.  Returns the count;
 * order as the reference returns them (empty intervals are removed by swap-with-last). */
int oracle_li_kmers(const uint32_t *row, int32_t len, int32_t k, int32_t intervals, const int32_t *prio,
                    uint64_t *hash_out, int32_t *ind_out) {
    typedef unsigned __int128 u128;
    if (k > len || intervals <= 0) return 0;
    int p = 0, q = 0;
    u128 h = 0;
    while (q < k) { h <<= 2; h += (u128) prio[nt_at(row, q)]; q++; }
    u128 factor = 1;
    for (int i = 0; i < k - 1; i++) factor <<= 2;
    std::vector<u128> best((size_t) intervals, factor << 2);
    std::vector<int> have((size_t) intervals, 0);
    std::vector<uint64_t> bh((size_t) intervals, 0);
    std::vector<int32_t> bi((size_t) intervals, 0);
    const u128 M = (u128) 1000000000000000003ULL;
    best[0] = h; have[0] = 1; bh[0] = (uint64_t) (h % M); bi[0] = 0;
    int il = (int) std::ceil(((double) len - k + 1) / intervals);
    if (il <= 0) return -1;
    int interv = 0;
    while (q < len) {
        h -= factor * (u128) prio[nt_at(row, p)];
        h <<= 2;
        h += (u128) prio[nt_at(row, q)];
        p++; q++;
        interv = p / il;
        if (interv < 0 || interv >= intervals) return -1;
        if (h < best[(size_t) interv]) { best[(size_t) interv] = h; have[(size_t) interv] = 1; bh[(size_t) interv] = (uint64_t) (h % M); bi[(size_t) interv] = p; }
    }
    /* :212-222 : drop trailing intervals, then remove empty ones by swapping with the last */
    int cnt = interv + 1;
    std::vector<int> slot((size_t) cnt);
    for (int j = 0; j < cnt; j++) slot[(size_t) j] = j;
    for (int j = cnt - 1; j >= 0; j--) {
        if (!have[(size_t) slot[(size_t) j]]) { std::swap(slot[(size_t) j], slot.back()); slot.pop_back(); }
    }
    for (size_t j = 0; j < slot.size(); j++) { hash_out[j] = bh[(size_t) slot[j]]; ind_out[j] = bi[(size_t) slot[j]]; }
    return (int) slot.size();
}

/* src/main.cpp:332-336 + src/Params.cpp:357 */
void oracle_pkb_derive_params(double avg_len, float scale, int error_rate_percent, oracle_pkb_params *p) {
    p->min_overlap_area = (int) ((1.f + scale) * avg_len / 2);
    p->max_offset_pct = (int) ((1.f - scale) * avg_len / 2);
    p->min_identity_pct = 99 - error_rate_percent;
    p->same_ends = 3;
    p->li_k = 35;
    p->li_intervals = 6;
    p->rounds = 4;
}

} // extern "C"

namespace {

struct KmerRec { int32_t read; uint64_t hash; int32_t ind; int32_t rsize; };

struct KmerLess {   /* Kmer::operator< (src/DataStructures/Kmer.cpp:58-64) */
    bool by_id;     /* false: the reference's comparator (ties left to std::sort); true: ties broken by read id */
    bool operator()(const KmerRec &a, const KmerRec &b) const {
        if (a.hash != b.hash) return a.hash < b.hash;
        if (a.ind != b.ind) return a.ind > b.ind;
        if (a.rsize != b.rsize) return a.rsize < b.rsize;
        return by_id ? a.read < b.read : false;
    }
};

typedef std::vector<std::pair<int, int>> AdjList;

void add_directed_edge(std::vector<AdjList> &V, int a, int b, int off) {   /* Graph::addDirectedEdge, Graph.cpp:53-71 */
    if (a == b) return;
    for (auto &e : V[(size_t) a]) if (e.first == b) { if (off < e.second) e.second = off; return; }
    V[(size_t) a].push_back({b, off});
}

void retain_only_smallest_offset(std::vector<AdjList> &V) {                /* Graph.cpp:348-387 */
    for (auto &v : V) {
        std::sort(v.begin(), v.end());
        AdjList nv;
        size_t p = 0;
        while (p < v.size()) { nv.push_back(v[p]); p++; while (p < v.size() && v[p - 1].first == v[p].first) p++; }
        v.swap(nv);
    }
}

} // namespace

extern "C" {

/* The supplement as src/main.cpp:300-347 drives it: masks from the degrees of the incoming graph, GraphCreatorLI's four
 * priority rounds (src/GraphCreators/GraphCreatorLI.cpp:18-28), per round GraphCreatorKmerBased::startAlignmentGraphCreation
 * (src/GraphCreators/GraphCreatorKmerBased.cpp:28-92) with GraphCreatorPairwiseKmerBranch::createAlignmentsForKmers
 * (src/GraphCreators/GraphCreatorPairwiseKmerBranch.cpp:16-97) on every group of equal k-mer hash. */
int oracle_supplement(const uint32_t *words, const int32_t *len, int32_t n, int32_t W, const oracle_edge *edges_in, int64_t m_in,
                      const oracle_pkb_params *p, int32_t kmer_length_bucket, int32_t flags, oracle_edge **edges_out, int64_t *m_out,
                      int64_t *can_align_calls) {
    /* flags: 0 = the reference with --threads=1.
     *        ORACLE_PKB_TIES_BY_ID  : equal k-mers (same hash, position, read length) ordered by read id instead of by
     *                                 std::sort's unspecified tie order
     *        ORACLE_PKB_SNAPSHOT    : every group of a round sees the graph as it was when the round started (plus what
     *                                 the group itself added) -- the order-independent semantics of the GPU engine; the
     *                                 reference processes groups one after the other (and races when --threads > 1) */
    const bool by_id = (flags & ORACLE_PKB_TIES_BY_ID) != 0, snapshot = (flags & ORACLE_PKB_SNAPSHOT) != 0;
    std::vector<AdjList> V((size_t) n);
    for (int64_t i = 0; i < m_in; i++) V[(size_t) edges_in[i].src].push_back({edges_in[i].dst, edges_in[i].offset});
    /* src/main.cpp:308-322 */
    std::vector<int> indeg((size_t) n, 0);
    for (auto &v : V) for (auto &e : v) indeg[(size_t) e.first]++;
    std::vector<char> from((size_t) n, 0), to((size_t) n, 0);
    for (int i = 0; i < n; i++) {
        if (indeg[(size_t) i] == 0 && !V[(size_t) i].empty()) to[(size_t) i] = 1;
        if (indeg[(size_t) i] > 0 && V[(size_t) i].empty()) from[(size_t) i] = 1;
    }
    const long long BUCKETS_SORT = 1048576ll;
    const uint64_t MAXH = 1000000000000000003ULL;                           /* A = 0, B = MAX_HASH (KmerBased.cpp:205-209) */
    int32_t prio[4] = {0, 1, 2, 3};
    std::vector<int> neighbors((size_t) n, 1000000001);                       /* Params::INF */
    int64_t calls = 0;
    std::vector<uint64_t> hb((size_t) p->li_intervals);
    std::vector<int32_t> ib((size_t) p->li_intervals);
    for (int round = 0; round < p->rounds; round++) {
        std::vector<std::vector<KmerRec>> buckets((size_t) BUCKETS_SORT);
        for (int i = 0; i < n; i++) {                                         /* getKmersForBucketJob :202-259 */
            if (len[i] <= 0) continue;
            if (!(from[(size_t) i] || to[(size_t) i])) continue;
            if (kmer_length_bucket > len[i]) continue;                        /* Read::getKmers: length > size() -> none (Read.cpp:71) */
            int c = oracle_li_kmers(words + (size_t) i * W, len[i], p->li_k, p->li_intervals, prio, hb.data(), ib.data());
            if (c < 0) return -1;
            for (int j = 0; j < c; j++) {
                int ind = (int) ((BUCKETS_SORT - 1) * ((double) (hb[(size_t) j] - 0) / (double) (MAXH - 0)));
                buckets[(size_t) ind].push_back({i, hb[(size_t) j], ib[(size_t) j], len[i]});
            }
        }
        for (auto &b : buckets) if (!b.empty()) std::sort(b.begin(), b.end(), KmerLess{by_id});   /* sortBucketsJob :94-106 */
        std::vector<std::vector<char>> marks;
        std::vector<AdjList> Vsnap;
        std::vector<oracle_edge> added_round;
        if (snapshot) Vsnap = V;
        std::vector<std::pair<int, int>> added_local;             /* (id2, offset) additions of the current `i` inside a group */
        std::vector<oracle_edge> added_group;
        for (auto &km : buckets) {                                            /* createAlignmentForKmersJobNewGC :108-136 (clone: masks all true) */
            size_t P = 0, Q = 0;
            while (P < km.size()) {
                while (Q < km.size() && km[Q].hash == km[P].hash) Q++;
                const int pp = (int) P, qq = (int) Q - 1, D = qq - pp + 1;
                marks.assign((size_t) D, std::vector<char>((size_t) D, 0));   /* branchMarkers */
                added_group.clear();
                for (int i = qq - 1; i >= pp; i--) {                          /* PairwiseKmerBranch.cpp:34-94 */
                    const int id1 = km[(size_t) i].read, ind1 = km[(size_t) i].ind;
                    const AdjList &cur = snapshot ? Vsnap[(size_t) id1] : V[(size_t) id1];
                    for (auto &x : cur) neighbors[(size_t) x.first] = x.second;
                    added_local.clear();
                    if (snapshot) for (auto &g : added_group) if (g.src == id1) {
                        if (g.offset < neighbors[(size_t) g.dst]) neighbors[(size_t) g.dst] = g.offset;
                        added_local.push_back({g.dst, 0});
                    }
                    for (int j = i + 1; j <= qq; j++) {
                        const int id2 = km[(size_t) j].read;
                        if (id1 == id2) continue;
                        const int offset = ind1 - km[(size_t) j].ind;
                        if (offset < 0) continue;
                        if (100 * offset > p->max_offset_pct * len[id1]) break;
                        const int ov = (len[id1] < len[id2] + offset ? len[id1] : len[id2] + offset) - offset;
                        if (ov < p->min_overlap_area) continue;
                        if (len[id2] + offset - len[id1] < 0) continue;
                        if (!marks[(size_t) (i - pp)][(size_t) (j - pp)]) {
                            if (neighbors[(size_t) id2] > offset) {
                                calls++;
                                if (oracle_can_align(words, len, W, id1, id2, offset, p)) {
                                    if (snapshot) { added_round.push_back({id1, id2, offset}); added_group.push_back({id1, id2, offset}); added_local.push_back({id2, 0}); }
                                    else add_directed_edge(V, id1, id2, offset);
                                    neighbors[(size_t) id2] = offset;
                                }
                            }
                            if (neighbors[(size_t) id2] != 1000000001) {
                                marks[(size_t) (i - pp)][(size_t) (j - pp)] = 1;
                                for (int t = 0; t < D; t++) marks[(size_t) (i - pp)][(size_t) t] |= marks[(size_t) (j - pp)][(size_t) t];
                            }
                        }
                    }
                    for (auto &x : cur) neighbors[(size_t) x.first] = 1000000001;
                    for (auto &x : added_local) neighbors[(size_t) x.first] = 1000000001;
                }
                P = Q;
            }
        }
        if (snapshot) for (auto &g : added_round) add_directed_edge(V, g.src, g.dst, g.offset);
        retain_only_smallest_offset(V);                                       /* KmerBased.cpp:87 */
        std::rotate(prio, prio + 1, prio + 4);                                /* GraphCreatorLI.cpp:26 */
    }
    retain_only_smallest_offset(V);                                           /* src/main.cpp:347 */
    int64_t m = 0;
    for (auto &v : V) m += (int64_t) v.size();
    oracle_edge *out = (oracle_edge *) malloc(sizeof(oracle_edge) * (size_t) (m ? m : 1));
    int64_t k = 0;
    for (int i = 0; i < n; i++) for (auto &e : V[(size_t) i]) { out[k].src = i; out[k].dst = e.first; out[k].offset = e.second; k++; }
    *edges_out = out; *m_out = m;
    if (can_align_calls) *can_align_calls = calls;
    return 0;
}

} // extern "C"
