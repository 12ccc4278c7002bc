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
        return true;
    }
};

// What every rank must see alike after a rendezvous: did any rank fail, did any rank decline the source-side form.  The LAST arriver of
// a barrier takes the snapshot under the barrier's mutex and every rank leaves with that one copy: a rank that runs ahead and fails (or
// resets its slot) in the next phase can no longer make two ranks read different answers and take different branches -- which would
// leave them in barriers of different phases for ever.
struct Agreed { bool failed = false, declined = false; };

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
    std::string err;
    Barrier bar;
    // per call, shared between the rank threads
    std::vector<int> rc;
    std::vector<std::string> rank_err;
    std::vector<alga_node_keys> keys;
    std::vector<const alga_edge *> d_edges;
    std::vector<uint64_t> counts;
    std::vector<char> declined;                            // rank r's build answered ALGA_ERR_UNSUPPORTED; written by r before the barrier, read by the barrier's snapshot only
    std::vector<alga_prefsuf_stats> stats;
    DevBuf gathered;                                       // rank 0's device: the complete edge list
    alga_multi_stats mstats{};
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

    // (the collectives end in a rendezvous and hand back its snapshot: what every rank agrees on at that point)
    Agreed all_gather_u32(uint32_t *mine, uint32_t *const *all /* all[q] = rank q's array */, size_t chunk) {
        hipStream_t s = m->stream[(size_t) r];
        int rc = ALGA_OK;
        if (m->transport == ALGA_TRANSPORT_RCCL) {
            rc = nccl(m->rccl.AllGather(mine + (size_t) r * chunk, mine, chunk, NCCL_UINT32, m->comm[(size_t) r], s), "ncclAllGather(keys)");
            if (rc == ALGA_OK) rc = hip(hipStreamSynchronize(s), "all-gather of the keys");
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

    Agreed gather_edges(alga_edge *out /* rank 0 */, const std::vector<uint64_t> &off) {
        hipStream_t s = m->stream[(size_t) r];
        int rc = ALGA_OK;
        if (m->transport == ALGA_TRANSPORT_RCCL) {
            // every rank's transfers inside one group (rank 0: its receives; a peer: its one send), so that no call blocks on its partner
            rc = nccl(m->rccl.GroupStart(), "ncclGroupStart");
            if (r == 0) {
                for (int q = 1; q < m->n && rc == ALGA_OK; q++)
                    if (m->counts[(size_t) q]) rc = nccl(m->rccl.Recv(out + off[(size_t) q], m->counts[(size_t) q] * 3, NCCL_INT32, q, m->comm[0], s), "ncclRecv(edges)");
            } else if (m->counts[(size_t) r]) {
                rc = nccl(m->rccl.Send(m->d_edges[(size_t) r], m->counts[(size_t) r] * 3, NCCL_INT32, 0, m->comm[(size_t) r], s), "ncclSend(edges)");
            }
            { const int rc2 = nccl(m->rccl.GroupEnd(), "ncclGroupEnd"); if (rc == ALGA_OK) rc = rc2; }
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

// one rank's part of a build; nodes_r: the node set on THIS rank's device.  Every branch that contains a rendezvous is taken on an
// `Agreed` snapshot (the same copy on every rank), never on a flag another rank may be rewriting.
void rank_main(alga_multi *m, int r, const alga_nodes *nodes_r, const alga_prefsuf_params *p, std::vector<uint64_t> *off, double *t_phase /* 5 */) {
    Collectives co{m, r};
    alga_engine *e = m->eng[(size_t) r];
    hipStream_t s = m->stream[(size_t) r];
    (void) hipSetDevice(m->dev[(size_t) r]);
    const int N = m->n;
    const int64_t n = nodes_r->n, chunk = shard_chunk(n, N);
    const int32_t b0 = (int32_t) std::min<int64_t>(n, (int64_t) r * chunk), b1 = (int32_t) std::min<int64_t>(n, (int64_t) (r + 1) * chunk);
    double t0 = now_ms();
    // ---- 2. keys of my nodes ----
    alga_node_keys k{};
    int rc = alga_prefsuf_keys_device(e, nodes_r, p, b0, b1, (void *) s, &k);
    if (rc != ALGA_OK) co.fail(rc, alga_last_error(e));
    m->keys[(size_t) r] = k;
    Agreed ag = rendezvous(m);
    bool shared = !ag.failed && (int64_t) N * chunk <= n + ALGA_KEY_ARRAY_SLACK;
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
    if (!ag.failed) {
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
        }
        (void) rendezvous(m);                              // (counts[r != 0] are not read after a fallback: run_build takes rank 0's list alone)
    } else if (!ag.failed) {
        if (r == 0) {
            uint64_t total = 0;
            for (int q = 0; q < N; q++) { (*off)[(size_t) q] = total; total += m->counts[(size_t) q]; }
            (*off)[(size_t) N] = total;
            if (total >= (1ull << 32) - 16) co.fail(ALGA_ERR_CAPACITY, "more than 2^32 edges");
            else if (N > 1) {
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
        if (!ag.failed && N > 1) (void) co.gather_edges((alga_edge *) m->gathered.p, *off);
    }
    double t4 = now_ms();
    if (r == 0) { t_phase[0] = t1 - t0; t_phase[1] = t2 - t1; t_phase[2] = t3 - t2; t_phase[3] = t4 - t3; t_phase[4] = t4 - t0; }
}

int run_build(alga_multi *m, const alga_nodes *per_rank, const alga_prefsuf_params *p, const alga_edge **d_edges, uint64_t *n_edges) {
    const int N = m->n;
    for (int r = 1; r < N; r++)
        if (per_rank[r].n != per_rank[0].n || per_rank[r].stride_words != per_rank[0].stride_words) return mfail(m, ALGA_ERR_INVALID_ARGUMENT, "the ranks hold different node sets");
    m->rc.assign((size_t) N, ALGA_OK); m->rank_err.assign((size_t) N, "");
    m->keys.assign((size_t) N, alga_node_keys{}); m->d_edges.assign((size_t) N, nullptr); m->counts.assign((size_t) N, 0);
    m->declined.assign((size_t) N, 0);
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
        rank_main(m, r, &per_rank[r], p, &off, t_phase);
    };
    std::vector<std::thread> th;
    bool started = true;
    try {
        for (int r = 1; r < N; r++) th.emplace_back(gated, r);
    } catch (...) { started = false; }
    { std::lock_guard<std::mutex> lk(gate.mu); gate.state = started ? 1 : 2; }
    gate.cv.notify_all();
    if (started) rank_main(m, 0, &per_rank[0], p, &off, t_phase);
    for (std::thread &x : th) x.join();
    if (!started) return mfail(m, ALGA_ERR_OUT_OF_MEMORY, "cannot start a host thread per GPU");
    for (int r = 0; r < N; r++)
        if (m->rc[(size_t) r] != ALGA_OK) return mfail(m, m->rc[(size_t) r], "rank " + std::to_string(r) + ": " + m->rank_err[(size_t) r]);
    const bool one = N == 1 || m->mstats.fell_back_to_one_gpu;
    *d_edges = one ? m->d_edges[0] : (const alga_edge *) m->gathered.p;
    *n_edges = one ? m->counts[0] : off[(size_t) N];
    m->mstats.ms_keys = t_phase[0]; m->mstats.ms_share = t_phase[1]; m->mstats.ms_build = t_phase[2]; m->mstats.ms_gather = t_phase[3]; m->mstats.ms_total = t_phase[4];
    m->mstats.n_ranks = N; m->mstats.transport = m->transport; m->mstats.edges = *n_edges;
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
    for (size_t r = 0; r < m->stream.size(); r++) { (void) hipSetDevice(m->dev[r]); (void) hipStreamSynchronize(m->stream[r]); (void) hipStreamDestroy(m->stream[r]); }
    for (alga_engine *e : m->eng) alga_engine_destroy(e);
    if (m->rccl.lib) (void) dlclose(m->rccl.lib);
    if (prev >= 0) (void) hipSetDevice(prev);
    delete m;
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
    // 1. the node set on every GPU: one upload per rank, each over its own PCIe link, side by side
    const int N = m->n;
    std::vector<alga_nodes> dev((size_t) N);
    std::vector<int> urc((size_t) N, ALGA_OK);
    const double t0 = now_ms();
    {
        std::vector<std::thread> th;
        auto up = [&](int r) { urc[(size_t) r] = alga_upload_nodes(m->eng[(size_t) r], nodes, &dev[(size_t) r]); };
        try { for (int r = 1; r < N; r++) th.emplace_back(up, r); } catch (...) { for (int r = (int) th.size() + 1; r < N; r++) urc[(size_t) r] = ALGA_ERR_OUT_OF_MEMORY; }
        up(0);
        for (std::thread &x : th) x.join();
    }
    int rc = ALGA_OK;
    for (int r = 0; r < N && rc == ALGA_OK; r++)
        if (urc[(size_t) r] != ALGA_OK) rc = mfail(m, urc[(size_t) r], "rank " + std::to_string(r) + ": upload: " + alga_last_error(m->eng[(size_t) r]));
    const double t1 = now_ms();
    const alga_edge *d = nullptr;
    uint64_t E = 0;
    if (rc == ALGA_OK) rc = run_build(m, dev.data(), p, &d, &E);
    const double t2 = now_ms();
    if (rc == ALGA_OK) {
        rc = alga_download_edges(m->eng[0], d, E, edges);
        if (rc != ALGA_OK) mfail(m, rc, alga_last_error(m->eng[0]));
        else *n_edges = E;
    }
    m->mstats.ms_upload = t1 - t0; m->mstats.ms_download = now_ms() - t2;
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
