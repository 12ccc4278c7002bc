// alga_amd/csrc/engine_pkb.hip -- host side of the approximate supplement (C ABI: include/alga_amd.h, "approximate
// supplement" section).  Orchestrates pkb_kernels.hip; no CPU fallback.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "engine_internal.h"
#include "pkb_kernels.h"

using namespace alga;

namespace {

PkbCfg make_cfg(const alga_pkb_params *p) {
    PkbCfg c;
    c.min_overlap_area = p->min_overlap_area; c.max_offset_pct = p->max_offset_pct; c.min_identity_pct = p->min_identity_pct;
    c.same_ends = p->same_ends; c.li_k = p->li_k; c.li_intervals = p->li_intervals; c.kmer_length_bucket = p->kmer_length_bucket;
    return c;
}

int check_params(alga_engine *e, const alga_nodes *nodes, const alga_pkb_params *p) {
    if (!nodes || !p) return alga_fail(e, ALGA_ERR_INVALID_ARGUMENT, "nodes/params must not be NULL");
    if (nodes->n < 0 || (nodes->n > 0 && (!nodes->words || !nodes->len || nodes->stride_words <= 0)))
        return alga_fail(e, ALGA_ERR_INVALID_ARGUMENT, "bad node set");
    if (p->li_k < 1 || p->li_k > 63) return alga_fail(e, ALGA_ERR_INVALID_ARGUMENT, "li_k must be in [1, 63]");
    if (p->li_intervals < 1 || p->li_intervals > PKB_MAX_INTERVALS) return alga_fail(e, ALGA_ERR_INVALID_ARGUMENT, "li_intervals must be in [1, 16]");
    if (p->same_ends < 0 || p->same_ends > 15) return alga_fail(e, ALGA_ERR_INVALID_ARGUMENT, "same_ends must be in [0, 15]");
    if (p->rounds < 0 || p->rounds > 4) return alga_fail(e, ALGA_ERR_INVALID_ARGUMENT, "rounds must be in [0, 4]");
    if (p->min_overlap_area < p->same_ends) return alga_fail(e, ALGA_ERR_INVALID_ARGUMENT, "min_overlap_area must be >= same_ends");
    if (nodes->n >= (1 << 27)) return alga_fail(e, ALGA_ERR_CAPACITY, "more than 2^27 nodes in the supplement; shard the input");
    return ALGA_OK;
}

// uploads a host node set into the engine's staging buffers (rows re-strided to a multiple of 4 words)
int upload_nodes(alga_engine *e, const alga_nodes *nodes, hipStream_t s, alga_nodes *dn) {
    int rc;
    const size_t n = (size_t) nodes->n;
    int32_t max_len = 0;
    for (size_t i = 0; i < n; i++) max_len = std::max(max_len, nodes->len[i]);
    if ((int64_t) blocks_of(max_len) > (int64_t) nodes->stride_words)
        return alga_fail(e, ALGA_ERR_INVALID_ARGUMENT, "stride_words is smaller than the longest read needs");
    const int stride_up = alga::hbm_row_stride(nodes->stride_words);
    const size_t wbytes = n * (size_t) stride_up * sizeof(uint32_t);
    alga_forget_node_set(e);
    if ((rc = alga_ensure(e, e->up_words, wbytes))) return rc;
    if ((rc = alga_ensure(e, e->up_len, n * sizeof(int32_t)))) return rc;
    *dn = *nodes;
    if (n) {
        if (stride_up != nodes->stride_words) HIP_TRY(e, hipMemsetAsync(e->up_words.p, 0, wbytes, s));
        HIP_TRY(e, hipMemcpy2DAsync(e->up_words.p, (size_t) stride_up * 4, nodes->words, (size_t) nodes->stride_words * 4,
                                    (size_t) nodes->stride_words * 4, n, hipMemcpyHostToDevice, s));
        HIP_TRY(e, hipMemcpyAsync(e->up_len.p, nodes->len, n * sizeof(int32_t), hipMemcpyHostToDevice, s));
    }
    dn->words = (const uint32_t *) e->up_words.p; dn->len = (const int32_t *) e->up_len.p; dn->stride_words = stride_up;
    dn->align_from = nullptr; dn->align_to = nullptr;
    return ALGA_OK;
}

NodesDev nodes_dev(const alga_nodes *dn) {
    NodesDev nd;
    nd.words = dn->words; nd.len = dn->len; nd.from = nullptr; nd.to = nullptr; nd.n = dn->n; nd.stride = dn->stride_words;
    return nd;
}

int node_bits(int32_t n) { int b = 1; while (b < 28 && (1ll << b) < (long long) n) b++; return b; }

// The incoming edge list -> the supplement's graph form: sorted unique keys (pkb_kernels.hip) in `out` + row pointers.  The C ABI
// promises a list sorted by (src, dst) with one edge per pair; that is checked on the device, and a list that is not is sorted and
// reduced to the smallest offset per pair (Graph::addDirectedEdge / retainOnlySmallestOffset, src/DataStructures/Graph.cpp:53-71,348-387).
int graph_from_edges(alga_engine *e, int32_t n_nodes, const alga_edge_dev *in, uint64_t n_in, DevBuf &out, uint64_t *n_out, hipStream_t s) {
    int rc;
    if (n_in >= (1ull << 32) - 16) return alga_fail(e, ALGA_ERR_CAPACITY, "more than 2^32 edges");
    if ((rc = alga_ensure(e, out, (n_in + 1) * sizeof(unsigned long long)))) return rc;
    if ((rc = alga_ensure(e, e->pk_rowptr, (size_t) (n_nodes + 2) * sizeof(uint32_t)))) return rc;
    if ((rc = alga_ensure(e, e->pk_cnt, 16 * sizeof(unsigned long long)))) return rc;
    unsigned long long *cnt = (unsigned long long *) e->pk_cnt.p;
    HIP_TRY(e, hipMemsetAsync(cnt + 13, 0, 3 * sizeof(unsigned long long), s));
    launch_pkb_edge_keys(in, n_in, (unsigned long long *) out.p, cnt + 14, s);
    if ((rc = alga_check_launch(e, "k_pkb_edge_keys"))) return rc;
    HIP_TRY(e, hipMemcpyAsync(&e->h_counters[CNT_TOTAL], cnt + 14, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    HIP_TRY(e, hipStreamSynchronize(s));
    if (e->h_counters[CNT_TOTAL]) return alga_fail(e, ALGA_ERR_CAPACITY, "the supplement keeps edge offsets in 9 bits: an edge has an offset above 511 (reads longer than 512 nt?)");
    uint64_t E = n_in;
    if (e->h_counters[CNT_TOTAL + 1]) {
        const size_t temp = std::max(sort_u64_keys_temp_bytes(n_in), unique_edge_keys_temp_bytes(n_in));
        if ((rc = alga_ensure(e, e->pk_merged, (n_in + 1) * sizeof(unsigned long long)))) return rc;
        if ((rc = alga_ensure(e, e->sort_temp, temp))) return rc;
        HIP_TRY(e, sort_u64_keys(e->sort_temp.p, temp, (const unsigned long long *) out.p, (unsigned long long *) e->pk_merged.p, n_in, s));
        HIP_TRY(e, unique_edge_keys(e->sort_temp.p, temp, (const unsigned long long *) e->pk_merged.p, (unsigned long long *) out.p, cnt + 13, n_in, s));
        HIP_TRY(e, hipMemcpyAsync(&e->h_counters[CNT_TOTAL], cnt + 13, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
        HIP_TRY(e, hipStreamSynchronize(s));
        E = e->h_counters[CNT_TOTAL];
    }
    launch_pkb_rowptr((const unsigned long long *) out.p, E, n_nodes, (uint32_t *) e->pk_rowptr.p, s);
    if ((rc = alga_check_launch(e, "k_pkb_rowptr"))) return rc;
    *n_out = E;
    return ALGA_OK;
}

constexpr uint32_t PKB_FIX_LIST_CAP = 1u << 20;     // places where two hashes share their sorted low bits (k_pkb_fix_flag)

// The supplement in PHASES (state in e->pkb between them): begin -- graph form, masks, the tips and where their k-mers go; per round: this
// rank's additions (the pairwise join of the k-mer groups it owns: rank = mix(k-mer key) mod n_ranks; one rank: all of them), then the merge of
// the additions of ALL ranks into the graph; end -- the graph as an edge list.  One GPU runs them back to back (supplement_device_impl); N ranks
// exchange the additions between `round` and `merge` (alga_pkb_shard_*, include/alga_amd.h).  The engine's semantics make that exact: every
// group of a round sees the graph as it was when the round started, and the merge orders by key -- the result does not depend on N.
int pkb_drop_presort(alga_engine *e);

int pkb_begin(alga_engine *e, const alga_nodes *dn, const alga_pkb_params *p, const alga_edge *d_edges_in, uint64_t m_in, int rank, int n_ranks, hipStream_t s) {
    int rc;
    auto &st = e->pkb;
    st.phase = 0;
    if ((rc = pkb_drop_presort(e))) return rc;
    st.cfg = make_cfg(p); st.dn = *dn; st.rounds = p->rounds; st.rank = rank; st.n_ranks = n_ranks; st.round = 0; st.no_look_ahead = false;
    const NodesDev nd = nodes_dev(dn);
    const int32_t n = dn->n;
    const PkbCfg &c = st.cfg;
    memset(&e->pkb_stats, 0, sizeof(e->pkb_stats));
    HIP_TRY(e, hipEventRecord(e->ev[EV_START], s));
    if ((rc = alga_ensure(e, e->pk_cnt, 16 * sizeof(unsigned long long)))) return rc;
    unsigned long long *cnt = (unsigned long long *) e->pk_cnt.p;
    st.cur = 0; st.E = 0;
    if ((rc = graph_from_edges(e, n, (const alga_edge_dev *) d_edges_in, m_in, e->pk_g[st.cur], &st.E, s))) return rc;
    // masks from the degrees of the incoming graph, once (src/main.cpp:308-322)
    if ((rc = alga_ensure(e, e->pk_mask, (size_t) n + 16))) return rc;
    if ((rc = alga_ensure(e, e->outdeg, (size_t) (n + 1) * sizeof(uint32_t)))) return rc;
    launch_pkb_masks(n, (const uint32_t *) e->pk_rowptr.p, (const unsigned long long *) e->pk_g[st.cur].p, st.E, (uint32_t *) e->outdeg.p, (uint8_t *) e->pk_mask.p, s);
    if ((rc = alga_check_launch(e, "k_pkb_masks"))) return rc;
    // the nodes that take part, as a dense list (the masks are fixed for all rounds), and where their k-mers go
    if ((rc = alga_ensure(e, e->pk_tips, (size_t) (n + 1) * sizeof(uint32_t)))) return rc;
    if ((rc = alga_ensure(e, e->pk_gsz, (size_t) (n + 1) * sizeof(uint32_t)))) return rc;       // here: k-mers per tip
    if ((rc = alga_ensure(e, e->pk_koff, (size_t) (n + 2) * sizeof(uint32_t)))) return rc;
    if ((rc = alga_ensure(e, e->pk_flag, (size_t) (n + 2) * sizeof(uint32_t)))) return rc;
    if ((rc = alga_ensure(e, e->pk_pos, (size_t) (n + 2) * sizeof(uint32_t)))) return rc;
    if ((rc = alga_ensure(e, e->scan_scratch, scan_scratch_bytes((uint64_t) n)))) return rc;
    HIP_TRY(e, hipMemsetAsync(cnt + 12, 0, sizeof(unsigned long long), s));
    launch_pkb_tip_flags(nd, c, (const uint8_t *) e->pk_mask.p, (uint32_t *) e->pk_flag.p, s);
    if ((rc = alga_check_launch(e, "k_pkb_tip_flags"))) return rc;
    launch_exclusive_scan((const uint32_t *) e->pk_flag.p, (uint64_t) n, (uint32_t *) e->pk_pos.p, (uint64_t *) e->scan_scratch.p, s);
    if ((rc = alga_ensure(e, e->pk_tipidx, (size_t) (n + 2) * sizeof(uint32_t)))) return rc;
    launch_pkb_tip_list(nd, c, (const uint32_t *) e->pk_flag.p, (const uint32_t *) e->pk_pos.p, (uint32_t *) e->pk_tips.p, (uint32_t *) e->pk_gsz.p, cnt + 12,
                        (uint32_t *) e->pk_tipidx.p, s);
    if ((rc = alga_check_launch(e, "k_pkb_tip_list"))) return rc;
    {
        MailArgs m;
        m.add((uint64_t *) e->scan_scratch.p + scan_total_index((uint64_t) n), 2, 0);
        m.add(cnt + 12, 2, 2);
        launch_mail(m, e->h_counters_dev, s);
    }
    HIP_TRY(e, hipStreamSynchronize(s));
    st.n_tips = n > 0 ? (uint32_t) e->h_counters[0] : 0u;
    if (e->h_counters[1] >= 4096) return alga_fail(e, ALGA_ERR_CAPACITY, "the supplement keeps read lengths in 12 bits: a participating read has 4096 nt or more");
    st.nk = 0;
    if (st.n_tips) {
        launch_exclusive_scan((const uint32_t *) e->pk_gsz.p, (uint64_t) st.n_tips, (uint32_t *) e->pk_koff.p, (uint64_t *) e->scan_scratch.p, s);
        HIP_TRY(e, hipMemcpyAsync(e->h_counters, (uint64_t *) e->scan_scratch.p + scan_total_index((uint64_t) st.n_tips), sizeof(uint64_t), hipMemcpyDeviceToHost, s));
        HIP_TRY(e, hipStreamSynchronize(s));
        st.nk = e->h_counters[0];
    }
    if (st.nk >= (1ull << 31)) return alga_fail(e, ALGA_ERR_CAPACITY, "more than 2^31 k-mers in the supplement; shard the input");
    // one 128-byte record per node that takes part: its row and id now, its snapshot keys at the start of every round
    if (st.n_tips) {
        if ((rc = alga_ensure(e, e->pk_tiprec, pkb_tiprec_bytes(st.n_tips)))) return rc;
        launch_pkb_tiprec_rows(nd, (const uint32_t *) e->pk_tips.p, st.n_tips, e->pk_tiprec.p, s);
        if ((rc = alga_check_launch(e, "k_pkb_tiprec_rows"))) return rc;
    }
    st.prio[0] = 0; st.prio[1] = 1; st.prio[2] = 2; st.prio[3] = 3;
    st.key_bits = 36 + node_bits(n);
    st.phase = 1;
    return ALGA_OK;
}

// The sorted k-mer entries of a round and its unsorted group heads exist twice: while the groups and the merge of round r work on set r's, the side stream
// fills the other set for round r + 1 (pkb_presort).  Set 0 is what the engine had before; the head SORT's outputs (pk_heads2, pk_hsz2) exist once.
struct PkbSet { DevBuf &keys2, &vals2, &heads, &hsz; };
PkbSet pkb_set(alga_engine *e, int i) {
    return i ? PkbSet{e->pk_keys2b, e->pk_vals2b, e->pk_headsb, e->pk_hszb} : PkbSet{e->pk_keys2, e->pk_vals2, e->pk_heads, e->pk_hsz};
}

// Look-ahead: sort -> repair -> heads of round `round`'s k-mer entries (they were all made in round 0: pk_keys_all) into set `set`, on the engine's SIDE
// stream with scratch of its own, their counts through k_mail into h_counters + H_PRE, ev_side behind them.  Nothing here depends on the graph, so it runs
// beside the groups and the merge of the round before: those are bound by isolated 128-byte reads, the sort by streaming and LDS -- and the GPU has work
// while the host sits in the round's waits.
int pkb_presort(alga_engine *e, int round, int set, int sort_bits) {
    int rc;
    auto &st = e->pkb;
    const uint64_t nk = st.nk;
    hipStream_t q = e->side_stream;
    PkbSet B = pkb_set(e, set);
    const size_t temp = rsort_u64_pairs_temp_bytes(nk);
    if (e->opt_test_presort_oom) return alga_fail(e, ALGA_ERR_OUT_OF_MEMORY, "pkb_presort (option test_presort_oom)");
    for (DevBuf *b : {&B.keys2, &B.vals2})
        if ((rc = alga_ensure(e, *b, (nk + 1) * sizeof(unsigned long long)))) return rc;
    for (DevBuf *b : {&B.heads, &B.hsz})
        if ((rc = alga_ensure(e, *b, (nk + 2) * sizeof(uint32_t)))) return rc;
    if ((rc = alga_ensure(e, e->sort_temp2, temp))) return rc;
    if ((rc = alga_ensure(e, e->pk_fixlist2, (size_t) PKB_FIX_LIST_CAP * sizeof(uint32_t)))) return rc;
    if ((rc = alga_ensure(e, e->pk_cnt2, 16 * sizeof(unsigned long long)))) return rc;
    unsigned long long *c2 = (unsigned long long *) e->pk_cnt2.p;
    const unsigned long long *kin = (const unsigned long long *) e->pk_keys_all.p + (size_t) round * st.kmers_stride;
    const unsigned long long *vin = (const unsigned long long *) e->pk_vals_all.p + (size_t) round * st.kmers_stride;
    HIP_TRY(e, rsort_u64_pairs(e->sort_temp2.p, temp, kin, (unsigned long long *) B.keys2.p, vin, (unsigned long long *) B.vals2.p, nk, sort_bits, q));
    HIP_TRY(e, hipMemsetAsync(c2, 0, 12 * sizeof(unsigned long long), q));
    launch_pkb_fix_runs((unsigned long long *) B.keys2.p, (unsigned long long *) B.vals2.p, nk, sort_bits, (uint32_t *) e->pk_fixlist2.p, PKB_FIX_LIST_CAP, c2 + 9, q);
    if ((rc = alga_check_launch(e, "k_pkb_fix_runs (look-ahead)"))) return rc;
    launch_pkb_heads((const unsigned long long *) B.keys2.p, nk, c2 + 1, c2 + 3, c2 + 10, (uint32_t *) B.heads.p, (uint32_t *) B.hsz.p, st.rank, st.n_ranks, q);
    if ((rc = alga_check_launch(e, "k_pkb_heads (look-ahead)"))) return rc;
    MailArgs m;
    m.add(c2, 24, 2 * alga_engine::H_PRE);
    m.add(c2 + 10, 2, 2 * alga_engine::H_PRE + 24);
    launch_mail(m, e->h_counters_dev, q);
    HIP_TRY(e, hipEventRecord(e->ev_side, q));
    st.pre_round = round; st.pre_set = set;
    return ALGA_OK;
}

// a look-ahead nobody will take (a sequence given up, or begun anew): wait it out before its buffers are touched
int pkb_drop_presort(alga_engine *e) {
    if (e->pkb.pre_round >= 0) { e->pkb.pre_round = -1; HIP_TRY(e, hipStreamSynchronize(e->side_stream)); }
    return ALGA_OK;
}

// the additions of this rank's groups in the round at hand: *d_add (device, unsorted edge keys), *n_add
int pkb_round(alga_engine *e, hipStream_t s, const unsigned long long **d_add, uint64_t *n_add) {
    int rc;
    auto &st = e->pkb;
    *d_add = nullptr; *n_add = 0;
    if (st.phase != 1 || st.round >= st.rounds) return alga_fail(e, ALGA_ERR_INVALID_ARGUMENT, "supplement round: begin (and the merge of the round before) come first");
    const NodesDev nd = nodes_dev(&st.dn);
    const PkbCfg &c = st.cfg;
    unsigned long long *cnt = (unsigned long long *) e->pk_cnt.p;
    const uint64_t nk = st.nk;
    const int round = st.round, cur = st.cur;
    const uint32_t n_tips = st.n_tips;
    e->pkb_stats.kmers[round] = nk;
    st.phase = 2;
    if (nk < 2) return ALGA_OK;
    // k-mer entries are radix-sorted on this many low bits of their key (= top bits of the mixed hash) (expected places to repair: nk^2 / 2^(bits + 1) <= 2^18)
    // the engine's own sort (radix_sort.hip, round 5) takes 10 bits per pass: 30 bits while that leaves <= 2^18 places to repair (23.7 M k-mers), else 40, 50;
    // the library's (option pkb_legacy bit 1) 8 bits per pass: 32, 40, 48
    const bool own_sort = !(e->opt_pkb_legacy & 2);
    if (st.pre_round >= 0 && st.pre_round != round && (rc = pkb_drop_presort(e))) return rc;
    const bool have_pre = st.pre_round == round;
    const int set = have_pre ? st.pre_set : 0;
    st.cur_set = set;
    PkbSet B = pkb_set(e, set);
    const int sort_bits = own_sort ? (nk <= 23700000ull ? 30 : (nk <= (1ull << 29) ? 40 : 50)) : (nk <= (1ull << 25) ? 32 : (nk <= (1ull << 29) ? 40 : 48));
    const size_t temp = std::max(std::max(sort_u64_pairs_temp_bytes(nk, sort_bits), rsort_u64_pairs_temp_bytes(nk)), sort_u32_pairs_temp_bytes(nk));
    if ((rc = alga_ensure(e, e->pk_fixlist, (size_t) PKB_FIX_LIST_CAP * sizeof(uint32_t)))) return rc;
    for (DevBuf *b : {&e->pk_keys, &e->pk_vals, &B.keys2, &B.vals2, &e->pk_marks})
        if ((rc = alga_ensure(e, *b, (nk + 1) * sizeof(unsigned long long)))) return rc;
    for (DevBuf *b : {&e->pk_flag, &e->pk_pos, &B.heads, &e->pk_heads2, &B.hsz, &e->pk_hsz2, &e->pk_gsz, &e->pk_nadd})
        if ((rc = alga_ensure(e, *b, (nk + 2) * sizeof(uint32_t)))) return rc;
    if ((rc = alga_ensure(e, e->sort_temp, temp))) return rc;
    if ((rc = alga_ensure(e, e->scan_scratch, scan_scratch_bytes(nk)))) return rc;
    // the records' snapshot halves: all of them before the first round (or every round, option pkb_legacy bit 6); later the merge refreshes the rows it changed
    if (round == 0 || (e->opt_pkb_legacy & 64)) {
        launch_pkb_tiprec_snap((const uint32_t *) e->pk_tips.p, n_tips, (const uint32_t *) e->pk_rowptr.p, (const unsigned long long *) e->pk_g[cur].p, e->pk_tiprec.p, s);
        if ((rc = alga_check_launch(e, "k_pkb_tiprec_snap"))) return rc;
    }
    const bool mail = !(e->opt_pkb_legacy & 256);            // what the host waits for, through one kernel into the pinned block (bit 8: a copy per piece)
    // look-ahead (pkb_presort): with the engine's sort and heads kernel (option pkb_legacy bit 9: off), when every round's entries are made in round 0
    const bool ahead_opts = own_sort && mail && !(e->opt_pkb_legacy & (4 | 512)) && e->side_stream;
    // The k-mers of EVERY round in the first round's walk (the tips, their rows and the interval borders are the same; the priority rotates by one
    // per round): one kernel for what were four.  Option pkb_legacy bit 7, or a shape that kernel does not take: a walk per round.
    if (round == 0) st.kmers_all = false;
    if (round == 0 && !(e->opt_pkb_legacy & (128 | 32))) {
        const size_t stride = (size_t) nk + 1;
        if ((rc = alga_ensure(e, e->pk_keys_all, (size_t) st.rounds * stride * sizeof(unsigned long long)))) return rc;
        if ((rc = alga_ensure(e, e->pk_vals_all, (size_t) st.rounds * stride * sizeof(unsigned long long)))) return rc;
        if (launch_pkb_kmers_all(nd, c, st.prio, 0, st.rounds, (const uint32_t *) e->pk_tips.p, (const uint32_t *) e->pk_koff.p, n_tips, sort_bits,
                                 (unsigned long long *) e->pk_keys_all.p, (unsigned long long *) e->pk_vals_all.p, stride, e->pk_tiprec.p, s)) {
            if ((rc = alga_check_launch(e, "k_pkb_kmers_all"))) return rc;
            st.kmers_all = true; st.kmers_stride = stride; st.kmers_sort_bits = sort_bits;
        }
    }
    const bool look_ahead = ahead_opts && st.kmers_all && st.kmers_sort_bits == sort_bits && st.kmers_stride == (size_t) nk + 1;
    uint32_t n_heads = 0;
    uint64_t big_words = 0;
    int first_pass = 0;
    bool counted = false;
    if (have_pre) {
        // sorted, repaired and its heads listed while the round before ran; the stream at hand goes on behind that work
        st.pre_round = -1;
        HIP_TRY(e, hipStreamWaitEvent(s, e->ev_side, 0));
        HIP_TRY(e, hipEventSynchronize(e->ev_side));
        memcpy(e->h_counters, e->h_counters + alga_engine::H_PRE, 13 * sizeof(unsigned long long));
        if (e->h_counters[9] > PKB_FIX_LIST_CAP) first_pass = 1;                  // its repair list overflowed: the loop form, here
        else counted = true;
    } else {
        const unsigned long long *kin = (const unsigned long long *) e->pk_keys.p, *vin = (const unsigned long long *) e->pk_vals.p;
        if (st.kmers_all && st.kmers_sort_bits == sort_bits && st.kmers_stride == (size_t) nk + 1) {
            kin = (const unsigned long long *) e->pk_keys_all.p + (size_t) round * st.kmers_stride;
            vin = (const unsigned long long *) e->pk_vals_all.p + (size_t) round * st.kmers_stride;
        } else {
            launch_pkb_kmers(nd, c, st.prio, (const uint32_t *) e->pk_tips.p, (const uint32_t *) e->pk_koff.p, n_tips, sort_bits, (unsigned long long *) e->pk_keys.p,
                             (unsigned long long *) e->pk_vals.p, e->pk_tiprec.p, (e->opt_pkb_legacy & 32) != 0, s);
            if ((rc = alga_check_launch(e, "k_pkb_kmers"))) return rc;
        }
        // equal hashes become contiguous; inside a group the group kernel orders the entries itself
        if (own_sort) HIP_TRY(e, rsort_u64_pairs(e->sort_temp.p, temp, kin, (unsigned long long *) B.keys2.p, vin, (unsigned long long *) B.vals2.p, nk, sort_bits, s));
        else HIP_TRY(e, sort_u64_pairs(e->sort_temp.p, temp, kin, (unsigned long long *) B.keys2.p, vin, (unsigned long long *) B.vals2.p, nk, sort_bits, s));
    }
    for (int pass = first_pass; pass < 2 && !counted; pass++) {
        HIP_TRY(e, hipMemsetAsync(cnt, 0, 12 * sizeof(unsigned long long), s));
        if (pass == 0) launch_pkb_fix_runs((unsigned long long *) B.keys2.p, (unsigned long long *) B.vals2.p, nk, sort_bits, (uint32_t *) e->pk_fixlist.p,
                                           PKB_FIX_LIST_CAP, cnt + 9, s);
        else launch_pkb_fix_runs_loop((unsigned long long *) B.keys2.p, (unsigned long long *) B.vals2.p, nk, sort_bits, s);   // the list overflowed
        if ((rc = alga_check_launch(e, "k_pkb_fix_runs"))) return rc;
        if (e->opt_pkb_legacy & 4) {
            launch_pkb_group_sizes((const unsigned long long *) B.keys2.p, nk, cnt + 1, cnt + 3, (uint32_t *) e->pk_flag.p, (uint32_t *) e->pk_gsz.p, st.rank, st.n_ranks, s);
            if ((rc = alga_check_launch(e, "k_pkb_group_sizes"))) return rc;
            launch_exclusive_scan((const uint32_t *) e->pk_flag.p, nk, (uint32_t *) e->pk_pos.p, (uint64_t *) e->scan_scratch.p, s);
            launch_pkb_head_list((const uint32_t *) e->pk_flag.p, (const uint32_t *) e->pk_pos.p, (const uint32_t *) e->pk_gsz.p, nk, (uint32_t *) B.heads.p,
                                 (uint32_t *) B.hsz.p, s);
            if ((rc = alga_check_launch(e, "k_pkb_head_list"))) return rc;
            HIP_TRY(e, hipMemcpyAsync(e->h_counters + 12, (uint64_t *) e->scan_scratch.p + scan_total_index(nk), sizeof(uint64_t), hipMemcpyDeviceToHost, s));
        } else {
            launch_pkb_heads((const unsigned long long *) B.keys2.p, nk, cnt + 1, cnt + 3, cnt + 10, (uint32_t *) B.heads.p, (uint32_t *) B.hsz.p,
                             st.rank, st.n_ranks, s);
            if ((rc = alga_check_launch(e, "k_pkb_heads"))) return rc;
            if (mail) {
                MailArgs m;
                m.add(cnt, 24, 0);
                m.add(cnt + 10, 2, 24);
                launch_mail(m, e->h_counters_dev, s);
            } else HIP_TRY(e, hipMemcpyAsync(e->h_counters + 12, cnt + 10, sizeof(uint64_t), hipMemcpyDeviceToHost, s));
        }
        if (!mail || (e->opt_pkb_legacy & 4)) HIP_TRY(e, hipMemcpyAsync(e->h_counters, cnt, 12 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
        HIP_TRY(e, hipStreamSynchronize(s));
        if (pass == 0 && e->h_counters[9] > PKB_FIX_LIST_CAP) continue;
        counted = true;
    }
    big_words = e->h_counters[1];
    n_heads = (uint32_t) e->h_counters[12];
    if (look_ahead && !st.no_look_ahead && round + 1 < st.rounds && (rc = pkb_presort(e, round + 1, set ^ 1, sort_bits))) {
        // The look-ahead wants a second set of sorted entries and sort scratch (~50 B per k-mer entry): an input that fitted without it must not fail
        // because of it.  Its allocations come before its first launch: nothing is in flight; give the second set back, go on round by round.
        if (rc != ALGA_ERR_OUT_OF_MEMORY) return rc;
        (void) hipGetLastError();
        if ((set ^ 1) == 1)                                                    // (the second set is not the one this round works on)
            for (DevBuf *b : {&e->pk_keys2b, &e->pk_vals2b, &e->pk_headsb, &e->pk_hszb}) alga_release(*b);
        for (DevBuf *b : {&e->sort_temp2, &e->pk_fixlist2, &e->pk_cnt2}) alga_release(*b);
        e->err.clear();
        st.no_look_ahead = true; st.pre_round = -1;
        rc = ALGA_OK;
    }
    e->pkb_stats.groups[round] = n_heads;
    e->pkb_stats.max_group = std::max<uint64_t>(e->pkb_stats.max_group, e->h_counters[3]);             // exact for groups of more than 64
    if (!n_heads) return ALGA_OK;
    // groups in order of their size: the lanes of a wave replay groups of the same size
    HIP_TRY(e, sort_u32_pairs_bits(e->sort_temp.p, temp, (const uint32_t *) B.hsz.p, (uint32_t *) e->pk_hsz2.p, (const uint32_t *) B.heads.p,
                                   (uint32_t *) e->pk_heads2.p, n_heads, 8, s));
    if ((rc = alga_ensure(e, e->pk_bounds, 260 * sizeof(uint32_t)))) return rc;
    launch_pkb_class_bounds((const uint32_t *) e->pk_hsz2.p, n_heads, (uint32_t *) e->pk_bounds.p, s);
    if ((rc = alga_check_launch(e, "k_pkb_class_bounds"))) return rc;
    const uint64_t add_dense = 2 * nk;
    uint64_t add_ovf_cap = std::max<uint64_t>(1024, nk / 4);
    uint64_t n_dense = 0, n_ovf = 0;
    for (int attempt = 0; attempt < 3; attempt++) {
        const uint64_t add_cap = add_dense + add_ovf_cap;
        if ((rc = alga_ensure(e, e->pk_big, (big_words + 1) * sizeof(unsigned long long)))) return rc;
        if ((rc = alga_ensure(e, e->pk_add, (add_cap + 1) * sizeof(unsigned long long)))) return rc;
        HIP_TRY(e, hipMemsetAsync(cnt + 4, 0, 4 * sizeof(unsigned long long), s));                    // big cursor, overflow, calls
        launch_pkb_groups(nd, c, (const uint32_t *) e->pk_rowptr.p, (const unsigned long long *) e->pk_g[cur].p, (const unsigned long long *) B.keys2.p,
                          (const uint32_t *) e->pk_heads2.p, (const uint32_t *) e->pk_hsz2.p, (const uint32_t *) e->pk_bounds.p, n_heads, (unsigned long long *) B.vals2.p, nk,
                          (unsigned long long *) e->pk_marks.p, (unsigned long long *) e->pk_big.p, cnt + 4, (unsigned long long *) e->pk_add.p, add_dense,
                          add_cap, cnt + 5, cnt + 6, (uint32_t *) e->pk_nadd.p, (uint32_t *) e->pk_gsz.p, e->n_cu, (const uint32_t *) e->pk_tips.p, e->pk_tiprec.p,
                          e->opt_pkb_legacy, s);
        if ((rc = alga_check_launch(e, "k_pkb_groups"))) return rc;
        launch_exclusive_scan((const uint32_t *) e->pk_nadd.p, (uint64_t) n_heads, (uint32_t *) e->pk_pos.p, (uint64_t *) e->scan_scratch.p, s);
        if (mail) {
            MailArgs m;
            m.add(cnt, 16, 0);
            m.add((uint64_t *) e->scan_scratch.p + scan_total_index((uint64_t) n_heads), 2, 16);
            m.add(e->pk_bounds.p, 257, 2 * (CNT_TOTAL + 2));
            m.add(e->pk_hsz2.p, 1, 2 * (CNT_TOTAL + 2) + 258);                                          // the longest group's sort key
            launch_mail(m, e->h_counters_dev, s);
        } else {
            HIP_TRY(e, hipMemcpyAsync(e->h_counters, cnt, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
            HIP_TRY(e, hipMemcpyAsync(e->h_counters + 8, (uint64_t *) e->scan_scratch.p + scan_total_index((uint64_t) n_heads), sizeof(uint64_t),
                                      hipMemcpyDeviceToHost, s));
            HIP_TRY(e, hipMemcpyAsync(&e->h_first_hkey, e->pk_hsz2.p, sizeof(uint32_t), hipMemcpyDeviceToHost, s));    // the longest group's sort key
            HIP_TRY(e, hipMemcpyAsync(e->h_counters + CNT_TOTAL + 2, e->pk_bounds.p, 257 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        }
        HIP_TRY(e, hipStreamSynchronize(s));
        if (mail) e->h_first_hkey = ((const uint32_t *) (e->h_counters + CNT_TOTAL + 2))[258];
        if (e->h_counters[5] <= add_ovf_cap) { n_dense = e->h_counters[8]; n_ovf = e->h_counters[5]; break; }
        if (attempt == 2) return alga_fail(e, ALGA_ERR_HIP, "supplement: addition buffer kept overflowing");
        add_ovf_cap = e->h_counters[5] + 1024;
    }
    e->pkb_stats.can_align_calls[round] = e->h_counters[6];
    {
        const uint32_t *hb = (const uint32_t *) (e->h_counters + CNT_TOTAL + 2);                         // groups of exactly D members: [hb[255 - D], hb[256 - D])
        auto upto = [&](int d_lo, int d_hi) { return (uint64_t) (hb[256 - d_lo] - hb[255 - d_hi]); };        // sizes d_lo .. d_hi
        uint64_t *h = e->pkb_stats.group_hist[round];
        h[0] = upto(2, 2); h[1] = upto(3, 3); h[2] = upto(4, 4); h[3] = upto(5, 7); h[4] = upto(8, 15); h[5] = upto(16, 31); h[6] = upto(32, 64); h[7] = upto(65, 255);
    }
    e->pkb_stats.max_group = std::max<uint64_t>(e->pkb_stats.max_group, 255u - std::min<uint32_t>(255u, e->h_first_hkey));
    const uint64_t A = n_dense + n_ovf;
    if (A) {
        if ((rc = alga_ensure(e, e->pk_addk, (A + 1) * sizeof(unsigned long long)))) return rc;
        launch_pkb_gather_adds((const uint32_t *) e->pk_heads2.p, (const uint32_t *) e->pk_nadd.p, (const uint32_t *) e->pk_pos.p, n_heads,
                               (const unsigned long long *) e->pk_add.p, add_dense, n_dense, n_ovf, (unsigned long long *) e->pk_addk.p, s);
        if ((rc = alga_check_launch(e, "k_pkb_gather_adds"))) return rc;
        *d_add = (const unsigned long long *) e->pk_addk.p; *n_add = A;
    }
    return ALGA_OK;
}

// addDirectedEdge + retainOnlySmallestOffset for the additions of ALL ranks (A keys on this device, any order): sorted, merged into the graph,
// the first key per (src, dst) kept; closes the round
int pkb_merge(alga_engine *e, const unsigned long long *d_all, uint64_t A, hipStream_t s) {
    int rc;
    auto &st = e->pkb;
    if (st.phase != 2) return alga_fail(e, ALGA_ERR_INVALID_ARGUMENT, "supplement merge: the round's own additions come first");
    unsigned long long *cnt = (unsigned long long *) e->pk_cnt.p;
    if (A) {
        const uint64_t E = st.E;
        const int cur = st.cur, nxt = cur ^ 1;
        const size_t t2 = std::max(std::max(sort_u64_keys_temp_bytes(A), merge_u64_temp_bytes(E, A)), unique_edge_keys_temp_bytes(E + A));
        if ((rc = alga_ensure(e, e->pk_addk2, (A + 1) * sizeof(unsigned long long)))) return rc;
        if ((rc = alga_ensure(e, e->pk_merged, (E + A + 1) * sizeof(unsigned long long)))) return rc;
        if ((rc = alga_ensure(e, e->pk_g[nxt], (E + A + 1) * sizeof(unsigned long long)))) return rc;
        if ((rc = alga_ensure(e, e->sort_temp, t2))) return rc;
        if (A >= 1024 && !(e->opt_pkb_legacy & 16)) {     // (below: one library kernel does it)
            const int nb = st.key_bits - 36;                                          // bits of a node id
            const size_t t3 = rsort_u32_pairs_temp_bytes(A);
            PkbSet B = pkb_set(e, st.cur_set);                                          // (scratch: the round's unsorted heads are spent; the OTHER set's may be in the making)
            for (DevBuf *b : {&B.heads, &e->pk_heads2, &B.hsz})
                if ((rc = alga_ensure(e, *b, (A + 16) * sizeof(uint32_t)))) return rc;
            if ((rc = alga_ensure(e, e->sort_temp, std::max(t2, t3)))) return rc;
            launch_pkb_src_keys(d_all, A, 32 - nb, (uint32_t *) B.heads.p, s);
            HIP_TRY(e, rsort_u32_pairs(e->sort_temp.p, std::max(t2, t3), (const uint32_t *) B.heads.p, (uint32_t *) e->pk_heads2.p, nullptr, (uint32_t *) B.hsz.p, A,
                                       32 - nb, s));
            launch_pkb_gather_sorted_runs(d_all, (const uint32_t *) e->pk_heads2.p, (const uint32_t *) B.hsz.p, A, (unsigned long long *) e->pk_addk2.p, s);
            if ((rc = alga_check_launch(e, "k_pkb_gather_sorted_runs"))) return rc;
        } else HIP_TRY(e, sort_u64_keys_bits(e->sort_temp.p, t2, d_all, (unsigned long long *) e->pk_addk2.p, A, st.key_bits, s));
        HIP_TRY(e, merge_u64(e->sort_temp.p, t2, (const unsigned long long *) e->pk_g[cur].p, E, (const unsigned long long *) e->pk_addk2.p, A,
                             (unsigned long long *) e->pk_merged.p, s));
        if (e->opt_pkb_legacy & 16) {
            HIP_TRY(e, unique_edge_keys(e->sort_temp.p, t2, (const unsigned long long *) e->pk_merged.p, (unsigned long long *) e->pk_g[nxt].p, cnt + 13, E + A, s));
            HIP_TRY(e, hipMemcpyAsync(e->h_counters, cnt + 13, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
            HIP_TRY(e, hipStreamSynchronize(s));
            st.E = e->h_counters[0];
            st.cur = nxt;
            launch_pkb_rowptr((const unsigned long long *) e->pk_g[st.cur].p, st.E, st.dn.n, (uint32_t *) e->pk_rowptr.p, s);
            if ((rc = alga_check_launch(e, "k_pkb_rowptr"))) return rc;
        } else {
            // first key of every (src, dst) run: flags -> scan -> scatter; the scatter writes the new row pointers as well
            const uint64_t M = E + A;
            if ((rc = alga_ensure(e, e->pk_flag, (M + 2) * sizeof(uint32_t)))) return rc;
            if ((rc = alga_ensure(e, e->pk_pos, (M + 2) * sizeof(uint32_t)))) return rc;
            if ((rc = alga_ensure(e, e->scan_scratch, scan_scratch_bytes(M)))) return rc;
            launch_pkb_unique_flags((const unsigned long long *) e->pk_merged.p, M, (uint32_t *) e->pk_flag.p, s);
            launch_exclusive_scan((const uint32_t *) e->pk_flag.p, M, (uint32_t *) e->pk_pos.p, (uint64_t *) e->scan_scratch.p, s);
            launch_pkb_unique_scatter((const unsigned long long *) e->pk_merged.p, M, (const uint32_t *) e->pk_flag.p, (const uint32_t *) e->pk_pos.p, st.dn.n,
                                      (unsigned long long *) e->pk_g[nxt].p, (uint32_t *) e->pk_rowptr.p, s);
            if ((rc = alga_check_launch(e, "k_pkb_unique_scatter"))) return rc;
            HIP_TRY(e, hipMemcpyAsync(e->h_counters, (uint64_t *) e->scan_scratch.p + scan_total_index(M), sizeof(uint64_t), hipMemcpyDeviceToHost, s));
            HIP_TRY(e, hipStreamSynchronize(s));
            st.E = e->h_counters[0];
            st.cur = nxt;
        }
    }
    if (A && st.n_tips && !(e->opt_pkb_legacy & 64)) {
        // the tip records of the sources that got an edge (or a smaller offset): their snapshot halves from the new graph
        launch_pkb_tiprec_snap_srcs((const unsigned long long *) e->pk_addk2.p, A, (const uint32_t *) e->pk_tipidx.p, (const uint32_t *) e->pk_rowptr.p,
                                    (const unsigned long long *) e->pk_g[st.cur].p, e->pk_tiprec.p, s);
        if ((rc = alga_check_launch(e, "k_pkb_tiprec_snap_srcs"))) return rc;
    }
    e->pkb_stats.edges_after[st.round] = st.E;
    std::rotate(st.prio, st.prio + 1, st.prio + 4);                          // GraphCreatorLI.cpp:26
    st.round++;
    st.phase = 1;
    return ALGA_OK;
}

int pkb_end(alga_engine *e, hipStream_t s, const alga_edge **d_out, uint64_t *m_out) {
    int rc;
    auto &st = e->pkb;
    if (st.phase != 1) return alga_fail(e, ALGA_ERR_INVALID_ARGUMENT, "supplement end: a round is open");
    if ((rc = pkb_drop_presort(e))) return rc;
    if ((rc = alga_ensure(e, e->pk_edges[0], (st.E + 1) * sizeof(alga_edge_dev)))) return rc;
    launch_pkb_keys_to_edges((const unsigned long long *) e->pk_g[st.cur].p, st.E, (alga_edge_dev *) e->pk_edges[0].p, s);
    if ((rc = alga_check_launch(e, "k_pkb_keys_to_edges"))) return rc;
    HIP_TRY(e, hipEventRecord(e->ev[EV_EMIT], s));
    HIP_TRY(e, hipStreamSynchronize(s));
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e->ev[EV_START], e->ev[EV_EMIT]) == hipSuccess) e->pkb_stats.ms_total = ms;
    *d_out = (const alga_edge *) e->pk_edges[0].p;
    *m_out = st.E;
    st.phase = 0;
    return ALGA_OK;
}

int supplement_device_impl(alga_engine *e, const alga_nodes *dn, const alga_pkb_params *p, const alga_edge *d_edges_in, uint64_t m_in,
                           hipStream_t s, const alga_edge **d_out, uint64_t *m_out) {
    int rc = pkb_begin(e, dn, p, d_edges_in, m_in, 0, 1, s);
    if (rc) return rc;
    for (int round = 0; round < p->rounds; round++) {
        const unsigned long long *d_add = nullptr;
        uint64_t A = 0;
        if ((rc = pkb_round(e, s, &d_add, &A))) return rc;
        if ((rc = pkb_merge(e, d_add, A, s))) return rc;
    }
    return pkb_end(e, s, d_out, m_out);
}

} // namespace

extern "C" {

void alga_pkb_derive_params(double avg_len, float scale, double error_rate, int32_t kmer_length_bucket, alga_pkb_params *p) {
    if (!p) return;
    const int er = (int) (100 * error_rate);                                 // Params::ERROR_RATE = 100 * rate (src/Params.cpp:357)
    p->min_overlap_area = (int32_t) ((1.f + scale) * avg_len / 2);           // src/main.cpp:333
    p->max_offset_pct = (int32_t) ((1.f - scale) * avg_len / 2);             // :335
    p->min_identity_pct = 99 - er;                                           // :336
    p->same_ends = 3; p->li_k = 35; p->li_intervals = 6; p->rounds = 4;
    p->kmer_length_bucket = kmer_length_bucket;
}

int alga_can_align_batch_host(alga_engine *e, const alga_nodes *nodes, const alga_pkb_params *p, const int32_t *triples, uint64_t n, uint8_t *out) {
    if (!e) return ALGA_ERR_INVALID_ARGUMENT;
    e->err.clear();
    int rc = check_params(e, nodes, p);
    if (rc) return rc;
    if (n && (!triples || !out)) return alga_fail(e, ALGA_ERR_INVALID_ARGUMENT, "triples/out must not be NULL");
    HIP_TRY(e, hipSetDevice(e->device));
    hipStream_t s = e->own_stream;
    alga_nodes dn;
    if ((rc = upload_nodes(e, nodes, s, &dn))) return rc;
    if ((rc = alga_ensure(e, e->pk_io, (n + 1) * 3 * sizeof(int32_t)))) return rc;
    if ((rc = alga_ensure(e, e->pk_io2, n + 16))) return rc;
    if (n) HIP_TRY(e, hipMemcpyAsync(e->pk_io.p, triples, n * 3 * sizeof(int32_t), hipMemcpyHostToDevice, s));
    launch_can_align_batch(nodes_dev(&dn), make_cfg(p), (const int32_t *) e->pk_io.p, n, (uint8_t *) e->pk_io2.p, s);
    if ((rc = alga_check_launch(e, "k_can_align_batch"))) return rc;
    if (n) HIP_TRY(e, hipMemcpyAsync(out, e->pk_io2.p, n, hipMemcpyDeviceToHost, s));
    HIP_TRY(e, hipStreamSynchronize(s));
    return ALGA_OK;
}

int alga_li_kmers_host(alga_engine *e, const alga_nodes *nodes, const alga_pkb_params *p, const int32_t prio[4], uint64_t *hash, int32_t *ind,
                       int32_t *count) {
    if (!e) return ALGA_ERR_INVALID_ARGUMENT;
    e->err.clear();
    int rc = check_params(e, nodes, p);
    if (rc) return rc;
    if (!prio || (nodes->n && (!hash || !ind || !count))) return alga_fail(e, ALGA_ERR_INVALID_ARGUMENT, "output pointers must not be NULL");
    HIP_TRY(e, hipSetDevice(e->device));
    hipStream_t s = e->own_stream;
    alga_nodes dn;
    if ((rc = upload_nodes(e, nodes, s, &dn))) return rc;
    const size_t slots = (size_t) nodes->n * (size_t) p->li_intervals;
    if ((rc = alga_ensure(e, e->pk_keys, (slots + 1) * sizeof(uint64_t)))) return rc;
    if ((rc = alga_ensure(e, e->pk_io, (slots + 1) * sizeof(int32_t)))) return rc;
    if ((rc = alga_ensure(e, e->pk_io2, ((size_t) nodes->n + 1) * sizeof(int32_t)))) return rc;
    launch_li_kmers_slots(nodes_dev(&dn), make_cfg(p), prio, (uint64_t *) e->pk_keys.p, (int32_t *) e->pk_io.p, (int32_t *) e->pk_io2.p, s);
    if ((rc = alga_check_launch(e, "k_li_kmers_slots"))) return rc;
    if (nodes->n) {
        HIP_TRY(e, hipMemcpyAsync(hash, e->pk_keys.p, slots * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
        HIP_TRY(e, hipMemcpyAsync(ind, e->pk_io.p, slots * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        HIP_TRY(e, hipMemcpyAsync(count, e->pk_io2.p, (size_t) nodes->n * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(e, hipStreamSynchronize(s));
    return ALGA_OK;
}

int alga_pkb_supplement_device(alga_engine *e, const alga_nodes *nodes, const alga_pkb_params *p, const alga_edge *d_edges_in, uint64_t n_edges_in,
                               void *hip_stream, const alga_edge **d_edges_out, uint64_t *n_edges_out) {
    if (!e) return ALGA_ERR_INVALID_ARGUMENT;
    e->err.clear();
    if (!d_edges_out || !n_edges_out) return alga_fail(e, ALGA_ERR_INVALID_ARGUMENT, "output pointers must not be NULL");
    *d_edges_out = nullptr; *n_edges_out = 0;
    int rc = check_params(e, nodes, p);
    if (rc) return rc;
    if (n_edges_in && !d_edges_in) return alga_fail(e, ALGA_ERR_INVALID_ARGUMENT, "edges_in must not be NULL");
    HIP_TRY(e, hipSetDevice(e->device));
    hipStream_t s = hip_stream ? (hipStream_t) hip_stream : e->own_stream;
    return supplement_device_impl(e, nodes, p, d_edges_in, n_edges_in, s, d_edges_out, n_edges_out);
}

int alga_pkb_supplement_host(alga_engine *e, const alga_nodes *nodes, const alga_pkb_params *p, const alga_edge *edges_in, uint64_t n_edges_in,
                             alga_edge **edges_out, uint64_t *n_edges_out) {
    if (!e) return ALGA_ERR_INVALID_ARGUMENT;
    e->err.clear();
    if (!edges_out || !n_edges_out) return alga_fail(e, ALGA_ERR_INVALID_ARGUMENT, "output pointers must not be NULL");
    *edges_out = nullptr; *n_edges_out = 0;
    int rc = check_params(e, nodes, p);
    if (rc) return rc;
    if (n_edges_in && !edges_in) return alga_fail(e, ALGA_ERR_INVALID_ARGUMENT, "edges_in must not be NULL");
    for (uint64_t i = 0; i < n_edges_in; i++)
        if (edges_in[i].src < 0 || edges_in[i].src >= nodes->n || edges_in[i].dst < 0 || edges_in[i].dst >= nodes->n || edges_in[i].offset < 0 || edges_in[i].offset > 511)
            return alga_fail(e, ALGA_ERR_INVALID_ARGUMENT, "edge out of range");
    HIP_TRY(e, hipSetDevice(e->device));
    hipStream_t s = e->own_stream;
    alga_nodes dn;
    if ((rc = upload_nodes(e, nodes, s, &dn))) return rc;
    if ((rc = alga_ensure(e, e->pk_io, (n_edges_in + 1) * sizeof(alga_edge)))) return rc;
    if (n_edges_in) HIP_TRY(e, hipMemcpyAsync(e->pk_io.p, edges_in, n_edges_in * sizeof(alga_edge), hipMemcpyHostToDevice, s));
    const alga_edge *d_out = nullptr;
    uint64_t m = 0;
    if ((rc = supplement_device_impl(e, &dn, p, (const alga_edge *) e->pk_io.p, n_edges_in, s, &d_out, &m))) return rc;
    alga_edge *h = (alga_edge *) alga_host_list_take(e, (size_t) (m ? m : 1) * sizeof(alga_edge));
    if (!h) return alga_fail(e, ALGA_ERR_OUT_OF_MEMORY, "host edge buffer");
    if (m) {
        hipError_t err = hipMemcpy(h, d_out, (size_t) m * sizeof(alga_edge), hipMemcpyDeviceToHost);
        if (err != hipSuccess) { alga_host_list_give(e, h); return alga_fail(e, ALGA_ERR_HIP, "copy edges to host", err); }
    }
    *edges_out = h; *n_edges_out = m;
    return ALGA_OK;
}

// ---- the supplement on N ranks: the same phases, the exchange of a round's additions in the caller's hands (include/alga_amd.h) ----
int alga_pkb_shard_begin(alga_engine *e, const alga_nodes *nodes, const alga_pkb_params *p, const alga_edge *d_edges_in, uint64_t n_edges_in, int32_t rank, int32_t n_ranks,
                         void *hip_stream) {
    if (!e) return ALGA_ERR_INVALID_ARGUMENT;
    e->err.clear();
    int rc = check_params(e, nodes, p);
    if (rc) return rc;
    if (n_edges_in && !d_edges_in) return alga_fail(e, ALGA_ERR_INVALID_ARGUMENT, "edges_in must not be NULL");
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return alga_fail(e, ALGA_ERR_INVALID_ARGUMENT, "bad rank / n_ranks");
    HIP_TRY(e, hipSetDevice(e->device));
    return pkb_begin(e, nodes, p, d_edges_in, n_edges_in, rank, n_ranks, hip_stream ? (hipStream_t) hip_stream : e->own_stream);
}

int alga_pkb_shard_round(alga_engine *e, void *hip_stream, const uint64_t **d_additions, uint64_t *n_additions) {
    if (!e) return ALGA_ERR_INVALID_ARGUMENT;
    e->err.clear();
    if (!d_additions || !n_additions) return alga_fail(e, ALGA_ERR_INVALID_ARGUMENT, "output pointers must not be NULL");
    HIP_TRY(e, hipSetDevice(e->device));
    const unsigned long long *d = nullptr;
    uint64_t a = 0;
    const int rc = pkb_round(e, hip_stream ? (hipStream_t) hip_stream : e->own_stream, &d, &a);
    *d_additions = (const uint64_t *) d; *n_additions = a;
    return rc;
}

int alga_pkb_shard_merge(alga_engine *e, const uint64_t *d_all_additions, uint64_t n_all, void *hip_stream) {
    if (!e) return ALGA_ERR_INVALID_ARGUMENT;
    e->err.clear();
    if (n_all && !d_all_additions) return alga_fail(e, ALGA_ERR_INVALID_ARGUMENT, "additions must not be NULL");
    if (n_all >= (1ull << 32) - 16) return alga_fail(e, ALGA_ERR_CAPACITY, "more than 2^32 additions");
    HIP_TRY(e, hipSetDevice(e->device));
    return pkb_merge(e, (const unsigned long long *) d_all_additions, n_all, hip_stream ? (hipStream_t) hip_stream : e->own_stream);
}

int alga_pkb_shard_end(alga_engine *e, void *hip_stream, const alga_edge **d_edges_out, uint64_t *n_edges_out) {
    if (!e) return ALGA_ERR_INVALID_ARGUMENT;
    e->err.clear();
    if (!d_edges_out || !n_edges_out) return alga_fail(e, ALGA_ERR_INVALID_ARGUMENT, "output pointers must not be NULL");
    *d_edges_out = nullptr; *n_edges_out = 0;
    HIP_TRY(e, hipSetDevice(e->device));
    return pkb_end(e, hip_stream ? (hipStream_t) hip_stream : e->own_stream, d_edges_out, n_edges_out);
}

int alga_pkb_last_stats(const alga_engine *e, alga_pkb_stats *out) {
    if (!e || !out) return ALGA_ERR_INVALID_ARGUMENT;
    *out = e->pkb_stats;
    return ALGA_OK;
}

} // extern "C"
