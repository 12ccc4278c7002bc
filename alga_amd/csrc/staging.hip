// alga_amd/csrc/staging.hip -- host <-> HBM copies of the host-buffer entry points through pinned staging buffers.
//
// A hipMemcpy from pageable memory is staged by the runtime through one bounce buffer on one thread (~13 GB/s measured
// here); the link itself carries ~55 GB/s.  A few worker threads, each with its own pair of pinned buffers and its own
// stream, copy slices of the caller's array into pinned memory and enqueue the DMA while the next slice is being copied:
// the memcpy of the cores and the DMA overlap, and several memcpy streams run side by side.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <thread>
#include <vector>

#include "engine_internal.h"

namespace {

constexpr size_t STAGE_CHUNK = 8u << 20;        // bytes per pinned buffer
constexpr size_t STAGE_DIRECT = 4u << 20;       // below this a plain copy is as fast
constexpr size_t STAGE_COLD_MIN = 512u << 20;   // setting the staging up (128 MB of pinned memory, 8 streams) costs tens of ms: a first
                                                // copy smaller than this goes the plain way (13 GB/s) and leaves it for later

int ensure_staging(alga_engine *e) {
    if (e->stage_ready) return ALGA_OK;
    for (int t = 0; t < ALGA_STAGE_THREADS; t++) {
        for (int b = 0; b < 2; b++) {
            HIP_TRY(e, hipHostMalloc(&e->stage_pin[t][b], STAGE_CHUNK));
            HIP_TRY(e, hipEventCreateWithFlags(&e->stage_ev[t][b], hipEventDisableTiming));
        }
        HIP_TRY(e, hipStreamCreateWithFlags(&e->stage_stream[t], hipStreamNonBlocking));
    }
    e->stage_ready = true;
    return ALGA_OK;
}

// worker t moves chunks t, t + T, t + 2T, ... ; `to_device`: host -> pinned -> device, else device -> pinned -> host
void worker(alga_engine *e, int t, int T, char *dev, char *host, size_t bytes, bool to_device, hipError_t *err, const AlgaStageFill *fill) {
    (void) hipSetDevice(e->device);
    hipStream_t s = e->stage_stream[t];
    const size_t n_chunks = (bytes + STAGE_CHUNK - 1) / STAGE_CHUNK;
    int b = 0;
    bool used[2] = {false, false};
    size_t pend_off[2] = {0, 0}, pend_len[2] = {0, 0};
    auto drain = [&](int k) {                     // device -> host: the DMA into buffer k has landed, hand its bytes to the caller
        if (!used[k]) return;
        hipError_t r = hipEventSynchronize(e->stage_ev[t][k]);
        if (r != hipSuccess) { *err = r; return; }
        if (!to_device) memcpy(host + pend_off[k], e->stage_pin[t][k], pend_len[k]);
        used[k] = false;
    };
    for (size_t c = (size_t) t; c < n_chunks && *err == hipSuccess; c += (size_t) T, b ^= 1) {
        const size_t off = c * STAGE_CHUNK, len = std::min(STAGE_CHUNK, bytes - off);
        drain(b);                                 // buffer b is free again (its previous DMA is done)
        hipError_t r;
        if (to_device) {
            if (fill) (*fill)(e->stage_pin[t][b], off, len); else memcpy(e->stage_pin[t][b], host + off, len);
            r = hipMemcpyAsync(dev + off, e->stage_pin[t][b], len, hipMemcpyHostToDevice, s);
        } else {
            r = hipMemcpyAsync(e->stage_pin[t][b], dev + off, len, hipMemcpyDeviceToHost, s);
        }
        if (r == hipSuccess) r = hipEventRecord(e->stage_ev[t][b], s);
        if (r != hipSuccess) { *err = r; return; }
        used[b] = true; pend_off[b] = off; pend_len[b] = len;
    }
    drain(0); drain(1);
}

// is `p` memory the runtime can DMA from / to directly (hipHostMalloc / hipHostRegister: alga_host_alloc)?
bool is_pinned(const void *p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void) hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}

int staged_copy(alga_engine *e, void *dev, void *host, size_t bytes, bool to_device, const AlgaStageFill *fill = nullptr) {
    if (bytes == 0) return ALGA_OK;
    if (!fill && bytes >= STAGE_DIRECT && is_pinned(host)) {
        // the caller's buffer is pinned (alga_host_alloc): the DMA engine takes it as it is -- no copy through staging buffers, no worker threads
        HIP_TRY(e, to_device ? hipMemcpy(dev, host, bytes, hipMemcpyHostToDevice) : hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost));
        return ALGA_OK;
    }
    if (!fill && (bytes < STAGE_DIRECT || (!e->stage_ready && bytes < STAGE_COLD_MIN))) {
        HIP_TRY(e, to_device ? hipMemcpy(dev, host, bytes, hipMemcpyHostToDevice) : hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost));
        return ALGA_OK;
    }
    int rc = ensure_staging(e);
    if (rc) return rc;
    const int T = (int) std::min<size_t>(ALGA_STAGE_THREADS, (bytes + STAGE_CHUNK - 1) / STAGE_CHUNK);
    hipError_t errs[ALGA_STAGE_THREADS];
    std::vector<std::thread> th;
    // a std::thread that cannot be started throws: nothing may cross the C ABI, so the copy then goes the plain way
    bool started = true;
    try {
        th.reserve((size_t) T);
        for (int t = 0; t < T; t++) { errs[t] = hipSuccess; th.emplace_back(worker, e, t, T, (char *) dev, (char *) host, bytes, to_device, &errs[t], fill); }
    } catch (...) { started = false; }
    for (std::thread &x : th) x.join();
    if (!started) {
        if (fill) return alga_fail(e, ALGA_ERR_HIP, "staged copy: worker threads could not be started");
        HIP_TRY(e, to_device ? hipMemcpy(dev, host, bytes, hipMemcpyHostToDevice) : hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost));
        return ALGA_OK;
    }
    for (int t = 0; t < T; t++) if (errs[t] != hipSuccess) return alga_fail(e, ALGA_ERR_HIP, "staged host/device copy", errs[t]);
    return ALGA_OK;
}

} // namespace

// both block until the bytes are where they belong; the device side must not be in use by work still in flight
int alga_staged_h2d(alga_engine *e, void *d_dst, const void *h_src, size_t bytes) { return staged_copy(e, d_dst, const_cast<void *>(h_src), bytes, true); }
int alga_staged_d2h(alga_engine *e, void *h_dst, const void *d_src, size_t bytes) { return staged_copy(e, const_cast<void *>(d_src), h_dst, bytes, false); }
// host -> device where the bytes are MADE chunk by chunk (fill(pinned chunk, byte offset, bytes): e.g. lengths narrowed to a byte on their way)
int alga_staged_h2d_fill(alga_engine *e, void *d_dst, size_t bytes, const AlgaStageFill &fill) { return staged_copy(e, d_dst, nullptr, bytes, true, &fill); }

// host edge list of at least `bytes`: the spare one if it is large enough, else a new allocation
void *alga_host_list_take(alga_engine *e, size_t bytes) {
    if (bytes == 0) bytes = 16;
    void *p = nullptr;
    size_t cap = bytes;
    if (e->host_spare && e->host_spare_cap >= bytes) { p = e->host_spare; cap = e->host_spare_cap; e->host_spare = nullptr; e->host_spare_cap = 0; }
    else p = malloc(bytes);
    if (p) e->host_lists[p] = cap;
    return p;
}

void alga_host_list_give(alga_engine *e, void *p) {
    if (!p) return;
    if (!e) { free(p); return; }
    auto it = e->host_lists.find(p);
    const size_t cap = it == e->host_lists.end() ? 0 : it->second;
    if (it != e->host_lists.end()) e->host_lists.erase(it);
    if (cap > e->host_spare_cap) { free(e->host_spare); e->host_spare = p; e->host_spare_cap = cap; }
    else free(p);
}

void alga_staging_release(alga_engine *e) {
    free(e->host_spare); e->host_spare = nullptr; e->host_spare_cap = 0;
    if (!e->stage_ready) return;
    for (int t = 0; t < ALGA_STAGE_THREADS; t++) {
        for (int b = 0; b < 2; b++) {
            if (e->stage_pin[t][b]) (void) hipHostFree(e->stage_pin[t][b]);
            if (e->stage_ev[t][b]) (void) hipEventDestroy(e->stage_ev[t][b]);
        }
        if (e->stage_stream[t]) (void) hipStreamDestroy(e->stage_stream[t]);
    }
    e->stage_ready = false;
}
