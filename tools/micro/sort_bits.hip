// tools/micro/sort_bits.hip -- rocPRIM onesweep radix sort of 90.6 M (u32 key, u32 value) pairs with 8 .. 11 radix bits per pass
// (4 or 3 passes over 32 key bits): which configuration the index build of prefsuf_cluster.hip should ask for.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/sort_bits.hip -o /tmp/sort_bits && /tmp/sort_bits
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <cstring>
#include <vector>

template <class Config>
static float run(const char *name, uint32_t *k0, uint32_t *k1, uint32_t *v0, uint32_t *v1, size_t n, unsigned end_bit = 32) {
    size_t bytes = 0;
    if (rocprim::radix_sort_pairs<Config>(nullptr, bytes, k0, k1, v0, v1, n, 0u, end_bit, (hipStream_t) 0) != hipSuccess) { printf("%s: size query failed\n", name); return -1; }
    void *tmp = nullptr;
    if (hipMalloc(&tmp, bytes) != hipSuccess) return -1;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f;
    for (int it = 0; it < 5; it++) {
        hipEventRecord(a, 0);
        hipError_t e = rocprim::radix_sort_pairs<Config>(tmp, bytes, k0, k1, v0, v1, n, 0u, end_bit, (hipStream_t) 0);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        if (e != hipSuccess) { printf("%s: %s\n", name, hipGetErrorString(e)); hipFree(tmp); return -1; }
        float ms = 0; hipEventElapsedTime(&ms, a, b);
        if (it && ms < best) best = ms;
    }
    // sortedness of the last result
    std::vector<uint32_t> h(1 << 20);
    hipMemcpy(h.data(), k1 + n / 2, h.size() * 4, hipMemcpyDeviceToHost);
    bool ok = true;
    for (size_t i = 1; i < h.size(); i++) ok = ok && h[i - 1] <= h[i];
    printf("%-44s %.3f ms  temp %.1f MB  sorted %d\n", name, best, bytes / 1e6, (int) ok);
    hipFree(tmp);
    return best;
}

__global__ void fill(uint32_t *k, uint32_t *v, size_t n) {
    for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) {
        uint64_t x = i * 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
        k[i] = (uint32_t) x; v[i] = (uint32_t) i;
    }
}

int main() {
    const size_t n = 90621096;
    uint32_t *k0, *k1, *v0, *v1;
    hipMalloc(&k0, n * 4); hipMalloc(&k1, n * 4); hipMalloc(&v0, n * 4); hipMalloc(&v1, n * 4);
    fill<<<4096, 256>>>(k0, v0, n);
    hipDeviceSynchronize();
    using namespace rocprim;
    run<default_config>("default (1024 x 16, 8 bits, match)", k0, k1, v0, v1, n);
#ifdef CFG_T
    run<radix_sort_config<default_config, default_config,
        radix_sort_onesweep_config<kernel_config<CFG_T, CFG_I>, kernel_config<CFG_T, CFG_I>, CFG_B, block_radix_rank_algorithm::CFG_A>>>("variant", k0, k1, v0, v1, n);
#endif
    return 0;
}
