// tools/hipstart.cpp -- what a process pays for the HIP runtime alone (no code of this repository): the floor under alga_hip's
// start-up.  hipcc --offload-arch=gfx950 -O2 -o /tmp/hipstart tools/hipstart.cpp && /tmp/hipstart
// MI355X box of this round: hipInit 190-215 ms + first stream 20 ms = 239 ms (first run after boot: 352 ms).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k(int *p) { if (p) *p = 1; }
int main() {
    using clk = std::chrono::steady_clock;
    auto t0 = clk::now();
    auto ms = [&](clk::time_point a) { return std::chrono::duration<double, std::milli>(clk::now() - a).count(); };
    hipInit(0);                      double a = ms(t0); auto t1 = clk::now();
    hipSetDevice(0);                 double b = ms(t1); auto t2 = clk::now();
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking); double c = ms(t2); auto t3 = clk::now();
    int *d; hipMalloc(&d, 4);        double e = ms(t3); auto t4 = clk::now();
    hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, s, d); hipStreamSynchronize(s); double f = ms(t4); auto t5 = clk::now();
    void *h; hipHostMalloc(&h, 16 << 20, 0); double g = ms(t5);
    printf("hipInit %.1f  hipSetDevice %.1f  stream %.1f  first hipMalloc %.1f  first launch+sync %.1f  hipHostMalloc(16MB) %.1f  total %.1f ms\n", a, b, c, e, f, g, ms(t0));
    return 0;
}
