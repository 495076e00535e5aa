// dev probe: what ds_read_b64_tr_b16 returns for per-lane addresses (gfx950).   hipcc --offload-arch=gfx950 -O2 -o tools/tr_probe tools/tr_probe.hip
// LDS holds u16 element i at byte 2 i.  Each lane of a 16-lane group p = l & 15 passes the address of 4 contiguous u16:
//     addr = group * BS + (p >> 2) * RS + (p & 3) * 8        (a [4 rows][16 columns] block with row stride RS bytes)
// Hypothesis: lane p, element j  <-  row j, column p of its group's block  =  u16 index (group * BS + j * RS) / 2 + p.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(unsigned short* out, int RS, int BS) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x, p = l & 15, g = l >> 4;
    const unsigned addr = (unsigned)(size_t)lds + g * BS + (p >> 2) * RS + (p & 3) * 8;
    unsigned long long v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)(v >> (16 * j));
}
int main() {
    unsigned short* d; unsigned short h[256];
    hipMalloc(&d, sizeof(h));
    const int cases[3][2] = {{32, 128}, {80, 1024}, {528, 32}};   // contiguous block; padded rows, far blocks; wide rows, adjacent column blocks
    for (auto& c : cases) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, c[0], c[1]);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) bad += h[l * 4 + j] != (unsigned short)(((l >> 4) * c[1] + j * c[0]) / 2 + (l & 15));
        printf("RS=%d BS=%d: %s; lane 0: %d %d %d %d  lane 5: %d %d %d %d  lane 17: %d %d %d %d\n", c[0], c[1], bad ? "DIFFERS from the hypothesis" : "matches the hypothesis",
               h[0], h[1], h[2], h[3], h[20], h[21], h[22], h[23], h[68], h[69], h[70], h[71]);
    }
    return 0;
}
