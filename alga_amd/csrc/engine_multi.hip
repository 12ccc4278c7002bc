// alga_amd/csrc/engine_multi.hip -- the overlap graph on the N GPUs of one node, behind the C ABI (include/alga_amd.h: alga_multi_*).
//
// What the reference offers for parallelism is `--threads` (src/Params.cpp:237-294: worker threads of one process inside
// GraphCreatorPrefSuf, src/GraphCreators/GraphCreatorPrefSuf.cpp:150-161,299-306).  The counterpart here is ONE process with one
// host thread and one engine per GPU -- the shape a C++ caller (alga_hip --gpus N, the GraphCreator adapters) can use without a
// launcher:
//
//   1. every rank holds the node set (the caller uploads or ingests it per GPU: each GPU has its own PCIe link);
//   2. keys      rank r computes the minimizer keys and probe runs of ITS nodes (alga_prefsuf_keys_device), ids [b_r, b_r+1);
//   3. share     the per-node key array (4 B/node; the meta array too unless every read has the same length) is all-gathered
//                IN PLACE into every rank's engine array -- ncclAllGather over xGMI, or peer copies;
//   4. build     every rank sorts the gathered keys into its own copy of the bucket-ordered entry array and probes its own sources:
//                the final edges of the sources [b_r, b_r+1) (alga_prefsuf_build_range_device, keys_shared = 1); nothing a rank
//                computes here depends on another rank;
//   5. gather    the edge lists go to rank 0's GPU with their exact lengths, landed at their offsets of one list: grouped
//                ncclSend / ncclRecv (every peer over its own xGMI link), or peer copies.  The ranges are ascending and every list
//                is (src, dst)-ordered, so the concatenation IS the single-GPU byte order.
// A rank whose input the source-side form does not take (alga_status UNSUPPORTED: long reads, asymmetric masks, a repeat-rich
// source beyond the capacity) makes rank 0 build the whole graph alone -- the result never depends on N.
//
// Transports.  RCCL is loaded with dlopen (librccl.so.1: no link-time dependency, a box without it still gets the copy transport)
// and needs one GPU per rank.  The copy transport (hipMemcpyPeerAsync between the ranks' buffers, host barriers in between) also
// accepts several ranks on ONE device: that is how the driver is tested on a one-GPU box (tests/test_gpu_multi.py), with every
// step but the library calls themselves identical.
// NOT MEASURED ON MORE THAN ONE GPU: no multi-GPU node was available to this repository's builder.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "engine_internal.h"

namespace {

// ---- RCCL through dlopen: the handful of entry points the driver uses (rccl.h: ncclResult_t / ncclDataType_t are plain ints) ----
typedef void *nccl_comm_t;
enum { NCCL_INT32 = 2, NCCL_UINT32 = 3 };                  // ncclInt32, ncclUint32 (rccl.h, enum ncclDataType_t)
struct Rccl {
    void *lib = nullptr;
    int (*CommInitAll)(nccl_comm_t *, int, const int *) = nullptr;
    int (*CommDestroy)(nccl_comm_t) = nullptr;
    int (*CommAbort)(nccl_comm_t) = nullptr;               // optional: absent in a build without it, the driver then cannot unblock peers after a failed post
    const char *(*GetErrorString)(int) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*Send)(const void *, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    bool load(std::string &err) {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) { err = "RCCL (librccl.so.1) cannot be loaded"; return false; }
#define RCCL_SYM(field, sym) do { *(void **) (&field) = dlsym(lib, sym); if (!field) { err = std::string("RCCL lacks ") + sym; return false; } } while (0)
        RCCL_SYM(CommInitAll, "ncclCommInitAll"); RCCL_SYM(CommDestroy, "ncclCommDestroy"); RCCL_SYM(GetErrorString, "ncclGetErrorString");
        RCCL_SYM(AllGather, "ncclAllGather"); RCCL_SYM(Send, "ncclSend"); RCCL_SYM(Recv, "ncclRecv");
        RCCL_SYM(GroupStart, "ncclGroupStart"); RCCL_SYM(GroupEnd, "ncclGroupEnd");
#undef RCCL_SYM
        *(void **) (&CommAbort) = dlsym(lib, "ncclCommAbort");
        return true;
    }
};

// What every rank must see alike after a rendezvous: did any rank fail, did any rank decline the source-side form.  The LAST arriver of
// a barrier takes the snapshot under the barrier's mutex and every rank leaves with that one copy: a rank that runs ahead and fails (or
// resets its slot) in the next phase can no longer make two ranks read different answers and take different branches -- which would
// leave them in barriers of different phases for ever.
struct Agreed { bool failed = false, declined = false, shard_declined = false; };

// all ranks arrive, all leave; reusable.  n is fixed before the first rank thread runs (run_build's start gate).
struct Barrier {
    std::mutex mu; std::condition_variable cv; int n = 1, waiting = 0; unsigned long phase = 0;
    Agreed snap;
    template <class F> Agreed wait(F &&take_snapshot) {
        std::unique_lock<std::mutex> lk(mu);
        const unsigned long ph = phase;
        if (++waiting == n) { waiting = 0; snap = take_snapshot(); phase++; cv.notify_all(); }
        else cv.wait(lk, [&] { return phase != ph; });
        return snap;                                       // read under the mutex: the next phase's snapshot needs all n ranks back in wait()
    }
    void wait() { (void) wait([this] { return snap; }); }
};

inline double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

} // namespace

struct alga_multi {
    int n = 0;
    int transport = ALGA_TRANSPORT_COPY;
    std::vector<int> dev;
    std::vector<alga_engine *> eng;
    std::vector<hipStream_t> stream;
    Rccl rccl;
    std::vector<nccl_comm_t> comm;
    std::mutex comm_mu;
    bool comm_broken = false;                              // a post inside an RCCL group failed and the communicators were aborted: every later RCCL collective of this handle fails fast
    std::string err;
    Barrier bar;
    // per call, shared between the rank threads
    std::vector<int> rc;
    std::vector<std::string> rank_err;
    std::vector<alga_node_keys> keys;
    std::vector<const alga_edge *> d_edges;
    std::vector<uint64_t> counts;
    std::vector<char> declined;                            // rank r's build answered ALGA_ERR_UNSUPPORTED; written by r before the barrier, read by the barrier's snapshot only
    std::vector<char> shard_declined;                      // the same for a phase of the bucket-sharded form (all ranks then continue in the replicated form)
    int form = ALGA_MULTI_FORM_AUTO;
    // variable-length exchanges of the bucket-sharded form: what rank r offers (device pointer, per-destination counts and offsets in units), N x N
    std::vector<const void *> x_send;
    std::vector<uint64_t> x_cnt, x_off;
    std::vector<DevBuf> rx_desc, rx_pending, rx_small, rx_edges;   // per rank, on the rank's device: the receive buffers
    std::vector<alga_prefsuf_stats> stats;
    DevBuf gathered;                                       // rank 0's device: the complete edge list
    alga_multi_stats mstats{};
    uint64_t mstats_upload_bytes_per_rank = 0;             // host entry point: bytes of rows one rank brought up over its own PCIe link
};

namespace {

int mfail(alga_multi *m, int code, const std::string &what) { m->err = what; return code; }

// ids per rank: equal, even (a read and its reverse complement, ids 2i and 2i + 1, stay together); alga_amd/multigpu.py: shard_chunk
int64_t shard_chunk(int64_t n, int ranks) { return 2 * ((n + 2 * (int64_t) ranks - 1) / (2 * (int64_t) ranks)); }

// the snapshot the last arriver of a barrier takes (under the barrier's mutex; every rank has written its slots before arriving)
Agreed snapshot(const alga_multi *m) {
    Agreed a;
    for (int x : m->rc) a.failed = a.failed || x != ALGA_OK;
    for (char d : m->declined) a.declined = a.declined || d != 0;
    for (char d : m->shard_declined) a.shard_declined = a.shard_declined || d != 0;
    return a;
}
Agreed rendezvous(alga_multi *m) { return m->bar.wait([m] { return snapshot(m); }); }

// The collectives of the driver behind one small interface (m->transport): everything else of a rank's work is transport-agnostic.
//   all_gather_u32: every rank's slice [r * chunk, (r + 1) * chunk) of its own array `mine` -> the same slice of every rank's array
//   gather_edges  : rank q's list (counts[q] edges at d_edges[q]) -> rank 0's `out` at offset off[q]
// Both return with the data in place and every buffer free for reuse (stream synchronised, host barrier passed).
struct Collectives {
    alga_multi *m; int r;
    int fail(int code, const std::string &w) { m->rc[(size_t) r] = code; m->rank_err[(size_t) r] = w; return code; }
    int hip(hipError_t e, const char *what) { return e == hipSuccess ? ALGA_OK : fail(ALGA_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e)); }
    int nccl(int e, const char *what) { return e == 0 ? ALGA_OK : fail(ALGA_ERR_HIP, std::string(what) + ": " + m->rccl.GetErrorString(e)); }
    // A post inside an RCCL group failed on this rank: the peers' matching sends and receives will never complete, and their stream syncs
    // would wait for ever in front of the closing rendezvous.  All communicators live in this process (ncclCommInitAll), so
    // the failing rank aborts every one of them -- ncclCommAbort may be called while operations are in flight, that is what it is for --
    // the peers' syncs return, the rendezvous agrees on `failed`, and the handle refuses further RCCL collectives (create a new one).
    void abort_comms() {
        std::lock_guard<std::mutex> lk(m->comm_mu);
        if (m->comm_broken) return;
        m->comm_broken = true;
        if (m->rccl.CommAbort)
            for (nccl_comm_t &c : m->comm) if (c) { (void) m->rccl.CommAbort(c); c = nullptr; }
    }
    bool broken() { std::lock_guard<std::mutex> lk(m->comm_mu); return m->comm_broken; }
    int refuse() { return fail(ALGA_ERR_HIP, "RCCL communicators of this handle were aborted after a failed exchange: destroy it and create a new one"); }

    // (the collectives end in a rendezvous and hand back its snapshot: what every rank agrees on at that point)
    Agreed all_gather_u32(uint32_t *mine, uint32_t *const *all /* all[q] = rank q's array */, size_t chunk) {
        hipStream_t s = m->stream[(size_t) r];
        int rc = ALGA_OK;
        if (m->transport == ALGA_TRANSPORT_RCCL) {
            if (broken()) { refuse(); return rendezvous(m); }
            rc = nccl(m->rccl.AllGather(mine + (size_t) r * chunk, mine, chunk, NCCL_UINT32, m->comm[(size_t) r], s), "ncclAllGather(keys)");
            if (rc != ALGA_OK) abort_comms();              // (a peer already inside the collective would wait for this rank for ever)
            else rc = hip(hipStreamSynchronize(s), "all-gather of the keys");
            return rendezvous(m);
        }
        rc = hip(hipStreamSynchronize(s), "key pass");     // my slice is complete before a peer reads it
        m->bar.wait();
        for (int q = 0; q < m->n && rc == ALGA_OK; q++) {
            if (q == r) continue;
            rc = hip(hipMemcpyPeerAsync(mine + (size_t) q * chunk, m->dev[(size_t) r], all[q] + (size_t) q * chunk, m->dev[(size_t) q], chunk * sizeof(uint32_t), s),
                     "peer copy of a key slice");
        }
        if (rc == ALGA_OK) rc = hip(hipStreamSynchronize(s), "peer copies of the key slices");
        return rendezvous(m);                              // nobody's build (which sorts the key array in place) starts while a peer still reads it
    }

    // Variable-length exchange in units of `unit` bytes (a multiple of 4): rank r offers cnt[q] units at send + off[q] * unit to every rank
    // q; afterwards recv holds, rank by rank, what the ranks offered to r (*recv_total units).  An all-gather is the same thing with
    // one segment offered to everybody.  Ends in a rendezvous: the senders keep their buffers until everybody has what it wanted.
    Agreed exchange_v(const void *send, const uint64_t *cnt, const uint64_t *off, size_t unit, DevBuf &recv, uint64_t *recv_total, uint64_t *sent_bytes) {
        const int N = m->n;
        hipStream_t s = m->stream[(size_t) r];
        m->x_send[(size_t) r] = send;
        for (int q = 0; q < N; q++) { m->x_cnt[(size_t) r * N + q] = cnt[q]; m->x_off[(size_t) r * N + q] = off[q]; }
        int rc = hip(hipStreamSynchronize(s), "exchange: my segments");      // complete before a peer reads them
        Agreed ag = rendezvous(m);
        uint64_t total = 0, out_bytes = 0;
        for (int q = 0; q < N; q++) { total += m->x_cnt[(size_t) q * N + r]; if (q != r) out_bytes += cnt[q] * unit; }
        *recv_total = total;
        if (sent_bytes) *sent_bytes = out_bytes;
        if (rc == ALGA_OK && !ag.failed && recv.cap < (total + 1) * unit) {
            if (recv.p) (void) hipFree(recv.p);
            recv.p = nullptr; recv.cap = 0;
            if (hipMalloc(&recv.p, (total + 1) * unit + (total + 1) * unit / 8) != hipSuccess) { recv.p = nullptr; rc = fail(ALGA_ERR_OUT_OF_MEMORY, "receive buffer of an exchange"); }
            else recv.cap = (total + 1) * unit + (total + 1) * unit / 8;
        }
        if (m->transport == ALGA_TRANSPORT_RCCL) {
            // a rank that could not allocate still takes part (with nothing to receive into it fails the build at the rendezvous below; its
            // peers' sends to it complete into a scratch of the same size class only if it posts receives -- so it posts none and the
            // group is skipped by ALL ranks): agree first
            if (broken()) refuse();
            ag = rendezvous(m);
            if (!ag.failed) {
                rc = nccl(m->rccl.GroupStart(), "ncclGroupStart");
                uint64_t at = 0;
                for (int q = 0; q < N && rc == ALGA_OK; q++) {
                    const uint64_t cq = m->x_cnt[(size_t) q * N + r];
                    if (q == r) {
                        if (cq) rc = hip(hipMemcpyAsync((char *) recv.p + at * unit, (const char *) send + off[r] * unit, cq * unit, hipMemcpyDeviceToDevice, s), "exchange: own segment");
                    } else {
                        if (cq) rc = nccl(m->rccl.Recv((char *) recv.p + at * unit, cq * unit / 4, NCCL_UINT32, q, m->comm[(size_t) r], s), "ncclRecv(exchange)");
                        if (rc == ALGA_OK && cnt[q]) rc = nccl(m->rccl.Send((const char *) send + off[q] * unit, cnt[q] * unit / 4, NCCL_UINT32, q, m->comm[(size_t) r], s), "ncclSend(exchange)");
                    }
                    at += cq;
                }
                { const int rc2 = nccl(m->rccl.GroupEnd(), "ncclGroupEnd"); if (rc == ALGA_OK) rc = rc2; }
                if (rc != ALGA_OK) abort_comms();          // my posts are incomplete: the peers' matching halves must not wait for them
                else rc = hip(hipStreamSynchronize(s), "exchange");
            }
            return rendezvous(m);
        }
        if (rc == ALGA_OK && !ag.failed) {                   // pull: every rank copies what the others offered to it
            uint64_t at = 0;
            for (int q = 0; q < N && rc == ALGA_OK; q++) {
                const uint64_t cq = m->x_cnt[(size_t) q * N + r];
                if (cq) rc = hip(hipMemcpyPeerAsync((char *) recv.p + at * unit, m->dev[(size_t) r], (const char *) m->x_send[(size_t) q] + m->x_off[(size_t) q * N + r] * unit,
                                                    m->dev[(size_t) q], cq * unit, s), "peer copy of an exchange segment");
                at += cq;
            }
            if (rc == ALGA_OK) rc = hip(hipStreamSynchronize(s), "peer copies of an exchange");
        }
        return rendezvous(m);
    }
    Agreed all_gather_v(const void *send, uint64_t count, size_t unit, DevBuf &recv, uint64_t *recv_total, uint64_t *sent_bytes) {
        std::vector<uint64_t> cnt((size_t) m->n, count), off((size_t) m->n, 0);
        return exchange_v(send, cnt.data(), off.data(), unit, recv, recv_total, sent_bytes);
    }

    Agreed gather_edges(alga_edge *out /* rank 0 */, const std::vector<uint64_t> &off) {
        hipStream_t s = m->stream[(size_t) r];
        int rc = ALGA_OK;
        if (m->transport == ALGA_TRANSPORT_RCCL) {
            // every rank's transfers inside one group (rank 0: its receives; a peer: its one send), so that no call blocks on its partner
            if (broken()) { refuse(); return rendezvous(m); }
            rc = nccl(m->rccl.GroupStart(), "ncclGroupStart");
            if (r == 0) {
                for (int q = 1; q < m->n && rc == ALGA_OK; q++)
                    if (m->counts[(size_t) q]) rc = nccl(m->rccl.Recv(out + off[(size_t) q], m->counts[(size_t) q] * 3, NCCL_INT32, q, m->comm[0], s), "ncclRecv(edges)");
            } else if (m->counts[(size_t) r]) {
                rc = nccl(m->rccl.Send(m->d_edges[(size_t) r], m->counts[(size_t) r] * 3, NCCL_INT32, 0, m->comm[(size_t) r], s), "ncclSend(edges)");
            }
            { const int rc2 = nccl(m->rccl.GroupEnd(), "ncclGroupEnd"); if (rc == ALGA_OK) rc = rc2; }
            if (rc != ALGA_OK) abort_comms();
            if (r == 0 && rc == ALGA_OK && m->counts[0])
                rc = hip(hipMemcpyAsync(out, m->d_edges[0], m->counts[0] * sizeof(alga_edge), hipMemcpyDeviceToDevice, s), "own edges");
            if (rc == ALGA_OK) rc = hip(hipStreamSynchronize(s), "gather of the edge lists");
            return rendezvous(m);
        }
        if (r == 0) {                                      // every build has ended in a host sync and the barrier before this call: rank 0 pulls
            for (int q = 0; q < m->n && rc == ALGA_OK; q++)
                if (m->counts[(size_t) q])
                    rc = hip(hipMemcpyPeerAsync(out + off[(size_t) q], m->dev[0], m->d_edges[(size_t) q], m->dev[(size_t) q], m->counts[(size_t) q] * sizeof(alga_edge), s),
                             "peer copy of an edge list");
            if (rc == ALGA_OK) rc = hip(hipStreamSynchronize(s), "peer copies of the edge lists");
        }
        return rendezvous(m);                              // the peers keep their lists until rank 0 has them
    }
};

// The build step of the bucket-sharded form (alga_shard_*, include/alga_amd.h) for rank r, keys already all-gathered: on success the
// rank's (src, dst)-ordered edge list of its own source range is in d_edges[r] / counts[r], exactly what the replicated build leaves.
// A phase that answers ALGA_ERR_UNSUPPORTED on ANY rank ends the attempt for all of them (snapshot.shard_declined): the caller goes
// on in the replicated form with the keys still in place.
Agreed shard_build(alga_multi *m, int r, Collectives &co, const alga_nodes *nodes_r, const alga_prefsuf_params *p, int32_t b0, int32_t b1, double *t_shard /* 5 */) {
    alga_engine *e = m->eng[(size_t) r];
    hipStream_t s = m->stream[(size_t) r];
    const int N = m->n;
    std::vector<uint64_t> cnt((size_t) N, 0), off((size_t) N, 0);
    auto note = [&](int rc) { if (rc == ALGA_ERR_UNSUPPORTED) m->shard_declined[(size_t) r] = 1; else if (rc != ALGA_OK) co.fail(rc, alga_last_error(e)); };
    double t0 = now_ms(), tx = 0;
    // ---- my slice of the index; my sources' run descriptors by owner ----
    const uint32_t *d_desc = nullptr;
    note(alga_shard_index_device(e, nodes_r, p, r, N, (void *) s, &d_desc, cnt.data(), off.data()));
    Agreed ag = rendezvous(m);
    double t1 = now_ms();
    if (ag.failed || ag.shard_declined) return ag;
    uint64_t n_desc = 0, xb = 0;
    ag = co.exchange_v(d_desc, cnt.data(), off.data(), 12, m->rx_desc[(size_t) r], &n_desc, &xb);
    if (r == 0) m->mstats.xbytes_descriptors = xb;
    double t2 = now_ms(); tx += t2 - t1;
    if (ag.failed) return ag;
    // ---- join in my buckets ----
    const uint32_t *d_pend = nullptr;
    uint64_t n_pend = 0;
    note(alga_shard_join_device(e, nodes_r, (const uint32_t *) m->rx_desc[(size_t) r].p, n_desc, (void *) s, &d_pend, &n_pend));
    ag = rendezvous(m);
    double t3 = now_ms();
    if (ag.failed || ag.shard_declined) return ag;
    // ---- the per-source cap on the pending small survivors ----
    uint64_t n_pend_all = 0, n_small = 0, n_small_all = 0;
    ag = co.all_gather_v(d_pend, n_pend, 4, m->rx_pending[(size_t) r], &n_pend_all, &xb);
    if (r == 0) m->mstats.xbytes_pending = xb;
    if (ag.failed) return ag;
    const uint32_t *d_small = nullptr;
    note(alga_shard_small_keys_device(e, (const uint32_t *) m->rx_pending[(size_t) r].p, n_pend_all, (void *) s, &d_small, &n_small));
    ag = co.all_gather_v(d_small, n_small, 12, m->rx_small[(size_t) r], &n_small_all, &xb);
    if (r == 0) m->mstats.xbytes_small_keys = xb;
    if (ag.failed) return ag;
    const alga_edge *d_out = nullptr;
    note(alga_shard_resolve_device(e, (const uint32_t *) m->rx_small[(size_t) r].p, n_small_all, (void *) s, &d_out, cnt.data(), off.data()));
    ag = rendezvous(m);
    double t4 = now_ms();
    if (ag.failed) return ag;
    // ---- edges to the owner of the source id, adjacency lists there ----
    uint64_t n_in = 0;
    ag = co.exchange_v(d_out, cnt.data(), off.data(), sizeof(alga_edge), m->rx_edges[(size_t) r], &n_in, &xb);
    if (r == 0) m->mstats.xbytes_edges = xb;
    double t5 = now_ms(); tx += t5 - t4;
    if (ag.failed) return ag;
    const alga_edge *d = nullptr;
    uint64_t k = 0;
    note(alga_shard_place_device(e, (const alga_edge *) m->rx_edges[(size_t) r].p, n_in, b0, b1, (void *) s, &d, &k));
    m->d_edges[(size_t) r] = d; m->counts[(size_t) r] = k;
    memset(&m->stats[(size_t) r], 0, sizeof(alga_prefsuf_stats));
    m->stats[(size_t) r].edges = k; m->stats[(size_t) r].probe_used = ALGA_PROBE_CLUSTER; m->stats[(size_t) r].reduction_used = ALGA_REDUCTION_PER_TARGET;
    ag = rendezvous(m);
    double t6 = now_ms();
    if (r == 0) { t_shard[0] = t1 - t0; t_shard[1] = tx; t_shard[2] = t3 - t2; t_shard[3] = t4 - t3; t_shard[4] = t6 - t5; }
    return ag;
}

// one rank's part of a build; nodes_r: the node set on THIS rank's device.  Every branch that contains a rendezvous is taken on an
// `Agreed` snapshot (the same copy on every rank), never on a flag another rank may be rewriting.
void rank_main(alga_multi *m, int r, const alga_nodes *nodes_r, const alga_prefsuf_params *p, std::vector<uint64_t> *off, double *t_phase /* 5 */,
               alga_edge **host_edges /* null: the complete list is gathered on rank 0's GPU; else: every rank downloads its own range over its own PCIe link into ONE host list */) {
    Collectives co{m, r};
    alga_engine *e = m->eng[(size_t) r];
    hipStream_t s = m->stream[(size_t) r];
    (void) hipSetDevice(m->dev[(size_t) r]);
    const int N = m->n;
    const int64_t n = nodes_r->n, chunk = shard_chunk(n, N);
    const int32_t b0 = (int32_t) std::min<int64_t>(n, (int64_t) r * chunk), b1 = (int32_t) std::min<int64_t>(n, (int64_t) (r + 1) * chunk);
    double t0 = now_ms();
    // ---- 2. keys of my nodes ----
    // (up to four ranks: no shared key pass -- every rank's build computes the target keys of all nodes itself and goes through the pile path
    // for its id range: 16.0 / 13.9 ms of compute per rank at two / four ranks against 23.4 / 14.8 through the pairwise kernels, and no key
    // all-gather; from five ranks on the shared key pass and the pairwise kernels: tools/emulate_rank.py)
    const bool keys_local = N <= 4 && m->form != ALGA_MULTI_FORM_BUCKET_SHARDED;
    alga_node_keys k{};
    int rc = ALGA_OK;
    if (!keys_local) {
        rc = alga_prefsuf_keys_device(e, nodes_r, p, b0, b1, (void *) s, &k);
        if (rc != ALGA_OK) co.fail(rc, alga_last_error(e));
    }
    m->keys[(size_t) r] = k;
    Agreed ag = rendezvous(m);
    bool shared = !keys_local && !ag.failed && (int64_t) N * chunk <= n + ALGA_KEY_ARRAY_SLACK;
    for (int q = 0; q < N && shared; q++) shared = m->keys[(size_t) q].eligible != 0;      // (written before the rendezvous, not touched again in this build)
    double t1 = now_ms();
    // ---- 3. share ----
    if (shared) {
        std::vector<uint32_t *> all((size_t) N);
        for (int q = 0; q < N; q++) all[(size_t) q] = m->keys[(size_t) q].d_keys;
        ag = co.all_gather_u32(k.d_keys, all.data(), (size_t) chunk);
        bool meta = false;
        for (int q = 0; q < N; q++) meta = meta || m->keys[(size_t) q].meta_needed != 0;
        if (meta) {                                        // (taken by all ranks or none; a rank that failed above still joins the rendezvous inside)
            for (int q = 0; q < N; q++) all[(size_t) q] = m->keys[(size_t) q].d_meta;
            ag = co.all_gather_u32(k.d_meta, all.data(), (size_t) chunk);
        }
    }
    double t2 = now_ms();
    // ---- 4. build: the final edges of my sources ----
    // (AUTO = REPLICATED: by the one-GPU emulation of both forms at the north-star size the sharded one costs a rank MORE device time than
    // the replicated one up to eight ranks -- tools/emulate_shard.py against tools/emulate_rank.py, DESIGN.md section 7 -- so it is opt-in)
    const bool try_sharded = shared && N > 1 && m->form == ALGA_MULTI_FORM_BUCKET_SHARDED;
    bool have = false;
    if (!ag.failed && try_sharded) {
        double t_shard[5] = {0, 0, 0, 0, 0};
        ag = shard_build(m, r, co, nodes_r, p, b0, b1, t_shard);
        have = !ag.failed && !ag.shard_declined;
        if (r == 0 && have) {
            m->mstats.form = ALGA_MULTI_FORM_BUCKET_SHARDED;
            m->mstats.ms_shard_index = t_shard[0]; m->mstats.ms_shard_exchange = t_shard[1]; m->mstats.ms_shard_join = t_shard[2]; m->mstats.ms_shard_cap = t_shard[3];
            m->mstats.ms_shard_place = t_shard[4];
        }
    }
    if (!ag.failed && !have) {
        if (r == 0) m->mstats.form = ALGA_MULTI_FORM_REPLICATED;
        alga_prefsuf_params p2 = *p;
        p2.keys_shared = shared ? 1 : 0;
        const alga_edge *d = nullptr;
        uint64_t cnt = 0;
        rc = alga_prefsuf_build_range_device(e, nodes_r, &p2, b0, b1, (void *) s, &d, &cnt);
        if (rc == ALGA_ERR_UNSUPPORTED) { m->declined[(size_t) r] = 1; cnt = 0; d = nullptr; }
        else if (rc != ALGA_OK) co.fail(rc, alga_last_error(e));
        m->d_edges[(size_t) r] = d; m->counts[(size_t) r] = cnt;
        (void) alga_prefsuf_last_stats(e, &m->stats[(size_t) r]);
    }
    ag = rendezvous(m);
    double t3 = now_ms();
    if (r == 0 && shared) m->mstats.xbytes_keys = (uint64_t) (N - 1) * (uint64_t) chunk * 4u;
    // ---- 5. gather (or, a rank having declined the source-side form: rank 0 builds the whole graph, the general way) ----
    if (!ag.failed && ag.declined) {
        if (r == 0) {
            const alga_edge *d = nullptr;
            uint64_t cnt = 0;
            alga_prefsuf_params p2 = *p;
            p2.keys_shared = 0;
            rc = alga_prefsuf_build_device(e, nodes_r, &p2, (void *) s, &d, &cnt);
            if (rc != ALGA_OK) co.fail(rc, alga_last_error(e));
            m->d_edges[0] = d; m->counts[0] = cnt;
            (void) alga_prefsuf_last_stats(e, &m->stats[0]);
            m->mstats.fell_back_to_one_gpu = 1;
            if (host_edges && rc == ALGA_OK) {
                rc = alga_download_edges(e, d, cnt, host_edges);
                if (rc != ALGA_OK) co.fail(rc, alga_last_error(e));
            }
        }
        (void) rendezvous(m);                              // (counts[r != 0] are not read after a fallback: run_build takes rank 0's list alone)
    } else if (!ag.failed) {
        if (r == 0) {
            uint64_t total = 0;
            for (int q = 0; q < N; q++) { (*off)[(size_t) q] = total; total += m->counts[(size_t) q]; }
            (*off)[(size_t) N] = total;
            if (total >= (1ull << 32) - 16) co.fail(ALGA_ERR_CAPACITY, "more than 2^32 edges");
            else if (host_edges) {
                *host_edges = (alga_edge *) alga_host_list_take(m->eng[0], (size_t) (total ? total : 1) * sizeof(alga_edge));
                if (!*host_edges) co.fail(ALGA_ERR_OUT_OF_MEMORY, "host edge list of the whole graph");
            } else if (N > 1) {
                hipError_t he = hipSuccess;
                if (m->gathered.cap < (total + 1) * sizeof(alga_edge)) {
                    if (m->gathered.p) (void) hipFree(m->gathered.p);
                    m->gathered.p = nullptr; m->gathered.cap = 0;
                    he = hipMalloc(&m->gathered.p, (total + 1) * sizeof(alga_edge));
                    if (he == hipSuccess) m->gathered.cap = (total + 1) * sizeof(alga_edge);
                }
                if (he != hipSuccess) co.fail(ALGA_ERR_OUT_OF_MEMORY, "edge list of the whole graph on rank 0");
            }
        }
        ag = rendezvous(m);
        if (!ag.failed && host_edges) {
            // the consumer is the HOST: every GPU has its own PCIe link, no list crosses xGMI first (ranges ascending, lists (src, dst)-ordered:
            // the host list is the single-GPU byte order)
            if (m->counts[(size_t) r]) {
                rc = alga_staged_d2h(e, *host_edges + (*off)[(size_t) r], m->d_edges[(size_t) r], (size_t) m->counts[(size_t) r] * sizeof(alga_edge));
                if (rc != ALGA_OK) co.fail(rc, alga_last_error(e));
            }
            (void) rendezvous(m);
        } else if (!ag.failed && N > 1) (void) co.gather_edges((alga_edge *) m->gathered.p, *off);
        if (r == 0) m->mstats.xbytes_gather = 0;           // rank 0 only receives
    }
    double t4 = now_ms();
    if (r == 0) { t_phase[0] = t1 - t0; t_phase[1] = t2 - t1; t_phase[2] = t3 - t2; t_phase[3] = t4 - t3; t_phase[4] = t4 - t0; }
}

int run_build(alga_multi *m, const alga_nodes *per_rank, const alga_prefsuf_params *p, const alga_edge **d_edges, uint64_t *n_edges, alga_edge **host_edges = nullptr) {
    const int N = m->n;
    for (int r = 1; r < N; r++)
        if (per_rank[r].n != per_rank[0].n || per_rank[r].stride_words != per_rank[0].stride_words) return mfail(m, ALGA_ERR_INVALID_ARGUMENT, "the ranks hold different node sets");
    m->rc.assign((size_t) N, ALGA_OK); m->rank_err.assign((size_t) N, "");
    m->keys.assign((size_t) N, alga_node_keys{}); m->d_edges.assign((size_t) N, nullptr); m->counts.assign((size_t) N, 0);
    m->declined.assign((size_t) N, 0); m->shard_declined.assign((size_t) N, 0);
    m->x_send.assign((size_t) N, nullptr); m->x_cnt.assign((size_t) N * N, 0); m->x_off.assign((size_t) N * N, 0);
    m->stats.assign((size_t) N, alga_prefsuf_stats{});
    memset(&m->mstats, 0, sizeof(m->mstats));
    std::vector<uint64_t> off((size_t) N + 1, 0);
    double t_phase[5] = {0, 0, 0, 0, 0};
    // Start gate: no rank thread touches the barrier before ALL of them exist.  A thread that cannot be started (std::system_error)
    // is noticed while the others still wait at the gate, and they leave without ever entering a rendezvous -- the barrier's count is
    // never rewritten under a waiting thread.
    struct Gate { std::mutex mu; std::condition_variable cv; int state = 0; /* 0 wait, 1 go, 2 abort */ } gate;
    auto gated = [&](int r) {
        { std::unique_lock<std::mutex> lk(gate.mu); gate.cv.wait(lk, [&] { return gate.state != 0; }); if (gate.state == 2) return; }
        rank_main(m, r, &per_rank[r], p, &off, t_phase, host_edges);
    };
    std::vector<std::thread> th;
    bool started = true;
    try {
        for (int r = 1; r < N; r++) th.emplace_back(gated, r);
    } catch (...) { started = false; }
    { std::lock_guard<std::mutex> lk(gate.mu); gate.state = started ? 1 : 2; }
    gate.cv.notify_all();
    if (host_edges) *host_edges = nullptr;
    if (started) rank_main(m, 0, &per_rank[0], p, &off, t_phase, host_edges);
    for (std::thread &x : th) x.join();
    if (!started) return mfail(m, ALGA_ERR_OUT_OF_MEMORY, "cannot start a host thread per GPU");
    for (int r = 0; r < N; r++)
        if (m->rc[(size_t) r] != ALGA_OK) {
            if (host_edges && *host_edges) { alga_free_edges(m->eng[0], *host_edges); *host_edges = nullptr; }
            return mfail(m, m->rc[(size_t) r], "rank " + std::to_string(r) + ": " + m->rank_err[(size_t) r]);
        }
    const bool one = N == 1 || m->mstats.fell_back_to_one_gpu;
    *d_edges = one ? m->d_edges[0] : (const alga_edge *) m->gathered.p;
    *n_edges = one ? m->counts[0] : off[(size_t) N];
    m->mstats.ms_keys = t_phase[0]; m->mstats.ms_share = t_phase[1]; m->mstats.ms_build = t_phase[2]; m->mstats.ms_gather = t_phase[3]; m->mstats.ms_total = t_phase[4];
    m->mstats.n_ranks = N; m->mstats.transport = m->transport; m->mstats.edges = *n_edges;
    return ALGA_OK;
}

// fn(rank) on one host thread per rank, behind a start gate (no rank touches the barrier before all of them exist); false: a thread could not be started
template <class F>
bool run_rank_threads(alga_multi *m, F &&fn) {
    const int N = m->n;
    struct Gate { std::mutex mu; std::condition_variable cv; int state = 0; } gate;
    auto gated = [&](int r) {
        { std::unique_lock<std::mutex> lk(gate.mu); gate.cv.wait(lk, [&] { return gate.state != 0; }); if (gate.state == 2) return; }
        fn(r);
    };
    std::vector<std::thread> th;
    bool started = true;
    try { for (int r = 1; r < N; r++) th.emplace_back(gated, r); } catch (...) { started = false; }
    { std::lock_guard<std::mutex> lk(gate.mu); gate.state = started ? 1 : 2; }
    gate.cv.notify_all();
    if (started) fn(0);
    for (std::thread &x : th) x.join();
    return started;
}

// The approximate supplement on the N ranks of the handle (alga_pkb_shard_*): the exact graph -- complete on rank 0 -- goes to every rank (an
// all-gather in which rank 0 offers the whole list and the others nothing), then per round every rank joins the k-mer groups it owns, the
// additions of all ranks are all-gathered and merged by everybody.  All ranks end with the same graph; rank 0's is handed back.
int run_supplement(alga_multi *m, const alga_nodes *per_rank, const alga_pkb_params *p, const alga_edge *d_edges0, uint64_t n_edges0, const alga_edge **d_out, uint64_t *n_out) {
    const int N = m->n;
    m->rc.assign((size_t) N, ALGA_OK);
    m->rank_err.assign((size_t) N, std::string());
    m->declined.assign((size_t) N, 0);
    m->shard_declined.assign((size_t) N, 0);
    m->x_send.assign((size_t) N, nullptr);
    m->x_cnt.assign((size_t) N * N, 0); m->x_off.assign((size_t) N * N, 0);
    std::vector<const alga_edge *> outs((size_t) N, nullptr);
    std::vector<uint64_t> outn((size_t) N, 0);
    auto body = [&](int r) {
        Collectives co{m, r};
        alga_engine *e = m->eng[(size_t) r];
        hipStream_t s = m->stream[(size_t) r];
        (void) hipSetDevice(m->dev[(size_t) r]);
        auto note = [&](int rc) { if (rc != ALGA_OK) co.fail(rc, alga_last_error(e)); };
        uint64_t n_all = 0, xb = 0;
        // the exact graph to everybody (rx_edges[r]: the receive buffer of the edge exchange of the build, free by now)
        Agreed ag = co.all_gather_v(r == 0 ? (const void *) d_edges0 : nullptr, r == 0 ? n_edges0 : 0, sizeof(alga_edge), m->rx_edges[(size_t) r], &n_all, &xb);
        if (ag.failed) return;
        note(alga_pkb_shard_begin(e, &per_rank[r], p, (const alga_edge *) m->rx_edges[(size_t) r].p, n_all, r, N, (void *) s));
        ag = rendezvous(m);
        if (ag.failed) return;
        for (int round = 0; round < p->rounds; round++) {
            const uint64_t *d_add = nullptr;
            uint64_t a = 0, a_all = 0;
            note(alga_pkb_shard_round(e, (void *) s, &d_add, &a));
            ag = co.all_gather_v(d_add, a, sizeof(uint64_t), m->rx_small[(size_t) r], &a_all, &xb);
            if (ag.failed) return;
            note(alga_pkb_shard_merge(e, (const uint64_t *) m->rx_small[(size_t) r].p, a_all, (void *) s));
            ag = rendezvous(m);
            if (ag.failed) return;
        }
        note(alga_pkb_shard_end(e, (void *) s, &outs[(size_t) r], &outn[(size_t) r]));
        (void) rendezvous(m);
    };
    if (!run_rank_threads(m, body)) return mfail(m, ALGA_ERR_OUT_OF_MEMORY, "cannot start a host thread per GPU");
    for (int r = 0; r < N; r++)
        if (m->rc[(size_t) r] != ALGA_OK) return mfail(m, m->rc[(size_t) r], "rank " + std::to_string(r) + ": " + m->rank_err[(size_t) r]);
    for (int r = 1; r < N; r++)
        if (outn[(size_t) r] != outn[0]) return mfail(m, ALGA_ERR_HIP, "supplement: the ranks ended with different graphs");
    *d_out = outs[0]; *n_out = outn[0];
    return ALGA_OK;
}

} // namespace

extern "C" {

int alga_multi_create(const int32_t *hip_devices, int32_t n_ranks, int32_t transport, alga_multi **out) {
    if (!out) return ALGA_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (!hip_devices || n_ranks < 1 || n_ranks > 64 || transport < ALGA_TRANSPORT_AUTO || transport > ALGA_TRANSPORT_COPY) return ALGA_ERR_INVALID_ARGUMENT;
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    alga_multi *m = new (std::nothrow) alga_multi();
    if (!m) return ALGA_ERR_OUT_OF_MEMORY;
    m->n = n_ranks; m->bar.n = n_ranks;
    m->rx_desc.assign((size_t) n_ranks, DevBuf{}); m->rx_pending.assign((size_t) n_ranks, DevBuf{}); m->rx_small.assign((size_t) n_ranks, DevBuf{});
    m->rx_edges.assign((size_t) n_ranks, DevBuf{});
    bool distinct = true;
    for (int r = 0; r < n_ranks; r++) for (int q = 0; q < r; q++) distinct = distinct && hip_devices[r] != hip_devices[q];
    if (transport == ALGA_TRANSPORT_AUTO) transport = (distinct && n_ranks > 1) ? ALGA_TRANSPORT_RCCL : ALGA_TRANSPORT_COPY;
    int rc = ALGA_OK;
    if (transport == ALGA_TRANSPORT_RCCL && !distinct) rc = ALGA_ERR_INVALID_ARGUMENT;           // RCCL wants one GPU per rank
    for (int r = 0; r < n_ranks && rc == ALGA_OK; r++) {
        alga_engine *e = nullptr;
        rc = alga_engine_create(hip_devices[r], &e);
        if (rc != ALGA_OK) break;
        m->eng.push_back(e); m->dev.push_back(hip_devices[r]);
        hipStream_t s = nullptr;
        if (hipSetDevice(hip_devices[r]) != hipSuccess || hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { rc = ALGA_ERR_HIP; break; }
        m->stream.push_back(s);
    }
    if (rc == ALGA_OK && transport == ALGA_TRANSPORT_COPY && distinct) {
        for (int r = 0; r < n_ranks; r++)                  // peer access where the hardware offers it (xGMI); copies work without it, through the host
            for (int q = 0; q < n_ranks; q++) {
                int can = 0;
                if (q != r && hipSetDevice(hip_devices[r]) == hipSuccess && hipDeviceCanAccessPeer(&can, hip_devices[r], hip_devices[q]) == hipSuccess && can)
                    (void) hipDeviceEnablePeerAccess(hip_devices[q], 0);
            }
        (void) hipGetLastError();
    }
    if (rc == ALGA_OK && transport == ALGA_TRANSPORT_RCCL) {
        std::string err;
        if (!m->rccl.load(err)) rc = ALGA_ERR_UNSUPPORTED;
        else {
            m->comm.assign((size_t) n_ranks, nullptr);
            std::vector<int> devs(hip_devices, hip_devices + n_ranks);
            if (m->rccl.CommInitAll(m->comm.data(), n_ranks, devs.data()) != 0) { m->comm.clear(); rc = ALGA_ERR_HIP; }
        }
    }
    m->transport = transport;
    if (prev >= 0) (void) hipSetDevice(prev);
    if (rc != ALGA_OK) { alga_multi_destroy(m); return rc; }
    *out = m;
    return ALGA_OK;
}

void alga_multi_destroy(alga_multi *m) {
    if (!m) return;
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    for (nccl_comm_t c : m->comm) if (c && m->rccl.CommDestroy) (void) m->rccl.CommDestroy(c);
    if (!m->dev.empty() && m->gathered.p) { (void) hipSetDevice(m->dev[0]); (void) hipFree(m->gathered.p); }
    for (size_t r = 0; r < m->dev.size(); r++)
        for (std::vector<DevBuf> *v : {&m->rx_desc, &m->rx_pending, &m->rx_small, &m->rx_edges})
            if (r < v->size() && (*v)[r].p) { (void) hipSetDevice(m->dev[r]); (void) hipFree((*v)[r].p); (*v)[r].p = nullptr; }
    for (size_t r = 0; r < m->stream.size(); r++) { (void) hipSetDevice(m->dev[r]); (void) hipStreamSynchronize(m->stream[r]); (void) hipStreamDestroy(m->stream[r]); }
    for (alga_engine *e : m->eng) alga_engine_destroy(e);
    if (m->rccl.lib) (void) dlclose(m->rccl.lib);
    if (prev >= 0) (void) hipSetDevice(prev);
    delete m;
}

int alga_multi_set_option(alga_multi *m, const char *name, int64_t value) {
    if (!m || !name) return ALGA_ERR_INVALID_ARGUMENT;
    if (!strcmp(name, "form")) {
        if (value < ALGA_MULTI_FORM_AUTO || value > ALGA_MULTI_FORM_BUCKET_SHARDED) return mfail(m, ALGA_ERR_INVALID_ARGUMENT, "option form: 0 auto, 1 replicated, 2 bucket-sharded");
        m->form = (int) value;
        return ALGA_OK;
    }
    return mfail(m, ALGA_ERR_INVALID_ARGUMENT, "unknown option");
}

const char *alga_multi_last_error(const alga_multi *m) { return m ? m->err.c_str() : "no multi-GPU handle"; }

alga_engine *alga_multi_engine(alga_multi *m, int32_t rank) { return (m && rank >= 0 && rank < m->n) ? m->eng[(size_t) rank] : nullptr; }

int alga_multi_prefsuf_build_device(alga_multi *m, const alga_nodes *nodes_per_rank, const alga_prefsuf_params *p, const alga_edge **d_edges, uint64_t *n_edges) {
    if (!m) return ALGA_ERR_INVALID_ARGUMENT;
    m->err.clear();
    if (!nodes_per_rank || !p || !d_edges || !n_edges) return mfail(m, ALGA_ERR_INVALID_ARGUMENT, "arguments must not be NULL");
    *d_edges = nullptr; *n_edges = 0;
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    const int rc = run_build(m, nodes_per_rank, p, d_edges, n_edges);
    if (prev >= 0) (void) hipSetDevice(prev);
    return rc;
}

int alga_multi_prefsuf_build_host(alga_multi *m, const alga_nodes *nodes, const alga_prefsuf_params *p, alga_edge **edges, uint64_t *n_edges) {
    if (!m) return ALGA_ERR_INVALID_ARGUMENT;
    m->err.clear();
    if (!nodes || !p || !edges || !n_edges) return mfail(m, ALGA_ERR_INVALID_ARGUMENT, "arguments must not be NULL");
    *edges = nullptr; *n_edges = 0;
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    // 1. the node set on every GPU.  The rows cross PCIe ONCE: rank r brings up its 1 / N of the caller's row array over its own link (all ranks
    //    side by side), the slices are all-gathered between the GPUs (xGMI: RCCL, or peer copies), and every rank turns the complete raw buffer
    //    into the engine's layout itself (twin expansion / re-stride).  The lengths (a byte per node on the wire) go to every rank directly.
    //    (Round 4 uploaded the whole node set on every rank: N times the PCIe traffic for the same result.)
    const int N = m->n;
    std::vector<alga_nodes> dev((size_t) N);
    const double t0 = now_ms();
    int rc = ALGA_OK;
    {
        const bool twin = p->twin_rows != 0;
        const uint64_t rows = twin ? (uint64_t) nodes->n / 2 : (uint64_t) (nodes->n > 0 ? nodes->n : 0);
        const size_t row_words = nodes->stride_words > 0 ? (size_t) nodes->stride_words : 1;
        // equal slices of whole rows, padded so that the slice length in words is the same for every rank (the all-gather's unit)
        const uint64_t chunk_rows = (rows + (uint64_t) N - 1) / (uint64_t) N;
        m->rc.assign((size_t) N, ALGA_OK);
        m->rank_err.assign((size_t) N, std::string());
        m->declined.assign((size_t) N, 0);
        m->shard_declined.assign((size_t) N, 0);
        std::vector<uint32_t *> raw((size_t) N, nullptr);
        auto up = [&](int r) {
            Collectives co{m, r};
            alga_engine *e = m->eng[(size_t) r];
            (void) hipSetDevice(m->dev[(size_t) r]);
            const int urc = N == 1 ? (twin ? alga_upload_twin_nodes(e, nodes, &dev[(size_t) r]) : alga_upload_nodes(e, nodes, &dev[(size_t) r]))
                                   : alga_upload_nodes_phase(e, nodes, twin, 1, (uint64_t) r * chunk_rows, (uint64_t) (r + 1) * chunk_rows, chunk_rows * (uint64_t) N, &dev[(size_t) r], &raw[(size_t) r]);
            if (urc != ALGA_OK) co.fail(urc, std::string("upload: ") + alga_last_error(e));
            if (N == 1) return;
            Agreed ag = rendezvous(m);                     // every rank's raw buffer exists (or somebody failed)
            if (ag.failed) return;
            ag = co.all_gather_u32(raw[(size_t) r], raw.data(), (size_t) chunk_rows * row_words);
            if (ag.failed) return;
            const int frc = alga_upload_nodes_phase(e, nodes, twin, 2, 0, 0, 0, &dev[(size_t) r], nullptr);
            if (frc != ALGA_OK) co.fail(frc, std::string("upload (finish): ") + alga_last_error(e));
            (void) rendezvous(m);
        };
        if (!run_rank_threads(m, up)) rc = mfail(m, ALGA_ERR_OUT_OF_MEMORY, "cannot start a host thread per GPU");
        for (int r = 0; r < N && rc == ALGA_OK; r++)
            if (m->rc[(size_t) r] != ALGA_OK) rc = mfail(m, m->rc[(size_t) r], "rank " + std::to_string(r) + ": " + m->rank_err[(size_t) r]);
        m->mstats_upload_bytes_per_rank = (uint64_t) chunk_rows * row_words * 4;
    }
    const double t1 = now_ms();
    const alga_edge *d = nullptr;
    uint64_t E = 0;
    alga_edge *h = nullptr;
    if (rc == ALGA_OK) rc = run_build(m, dev.data(), p, &d, &E, &h);     // every rank brings its own range down: ms_gather is the download here
    if (rc == ALGA_OK) {
        if (N == 1 && !h) {                                 // (one rank: run_build leaves the list on the device)
            rc = alga_download_edges(m->eng[0], d, E, &h);
            if (rc != ALGA_OK) mfail(m, rc, alga_last_error(m->eng[0]));
        }
        if (rc == ALGA_OK) { *edges = h; *n_edges = E; }
    }
    m->mstats.ms_upload = t1 - t0; m->mstats.ms_download = m->mstats.ms_gather;
    if (prev >= 0) (void) hipSetDevice(prev);
    return rc;
}

int alga_multi_pkb_supplement_device(alga_multi *m, const alga_nodes *nodes_per_rank, const alga_pkb_params *p, const alga_edge *d_edges_rank0, uint64_t n_edges,
                                     const alga_edge **d_edges_out, uint64_t *n_edges_out) {
    if (!m) return ALGA_ERR_INVALID_ARGUMENT;
    m->err.clear();
    if (!nodes_per_rank || !p || !d_edges_out || !n_edges_out || (n_edges && !d_edges_rank0)) return mfail(m, ALGA_ERR_INVALID_ARGUMENT, "arguments must not be NULL");
    *d_edges_out = nullptr; *n_edges_out = 0;
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    const int rc = run_supplement(m, nodes_per_rank, p, d_edges_rank0, n_edges, d_edges_out, n_edges_out);
    if (prev >= 0) (void) hipSetDevice(prev);
    return rc;
}

void alga_multi_free_edges(alga_multi *m, alga_edge *edges) { if (m && !m->eng.empty()) alga_free_edges(m->eng[0], edges); }

int alga_multi_last_stats(const alga_multi *m, alga_multi_stats *out, alga_prefsuf_stats *per_rank /* n_ranks entries or NULL */) {
    if (!m || !out) return ALGA_ERR_INVALID_ARGUMENT;
    *out = m->mstats;
    if (per_rank) for (size_t r = 0; r < m->stats.size(); r++) per_rank[r] = m->stats[r];
    return ALGA_OK;
}

} // extern "C"
