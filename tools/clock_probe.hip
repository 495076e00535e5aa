// dev tool: what does reading a clock cost inside a kernel?  (wall_clock64 = s_memrealtime, clock64 = s_memtime)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(long long* out) {
    long long acc = 0;
    long long c0 = clock64();
    for (int i = 0; i < 1000; ++i) acc += wall_clock64();
    long long c1 = clock64();
    for (int i = 0; i < 1000; ++i) acc += clock64();
    long long c2 = clock64();
    long long w0 = wall_clock64();
    for (int i = 0; i < 1000; ++i) { __builtin_amdgcn_s_sleep(1); }
    long long w1 = wall_clock64();
    long long c3 = clock64();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = c2 - c1; out[2] = acc; out[3] = w1 - w0; out[4] = c3 - c2; }
}
int main() {
    long long* d; hipMalloc(&d, 64); long long h[8];
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d); hipMemcpy(h, d, 64, hipMemcpyDeviceToHost); }
    printf("1000 x wall_clock64: %lld shader cycles (%.1f per read); 1000 x clock64: %lld (%.1f per read); 1000 x s_sleep 1: %lld wall ticks = %lld shader cycles -> %.1f MHz shader clock\n",
           h[0], h[0] / 1000.0, h[1], h[1] / 1000.0, h[3], h[4], h[4] / (h[3] / 100.0));
    return 0;
}
