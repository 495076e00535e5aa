// handoff_probe.hip -- measures the per-hop latency of the granule hand-off the generation chain is built on
// (a ring of persistent workgroups passing a token), for several store/load flavours, payload sizes and
// placements.  Dev tool: results are recorded in DESIGN.md / profiles/.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef unsigned long long u64;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ u64 ld_sc1(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// mode 0: sc1 store / sc1 load (agent scope atomics)   mode 1: plain store / sc1 load (valid only same-XCD)
// mode 2: sc1 store / sc1 load, but only ONE doorbell granule is polled, then the payload is read once
__global__ __launch_bounds__(256) void ring_probe(u64* gran, const int* pos_of_block, int n, int laps, int payload, int mode,
                                                  int* xcc, unsigned* fail, long long timeout_ticks, int sleep_arg) {
    const int b = blockIdx.x, pos = pos_of_block[b], tid = threadIdx.x;
    if (tid == 0) xcc[b] = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | 20) & 0xf;
    if (pos < 0) return;  // bystander block (placement 2)
    u64* mine = gran + (size_t)pos * payload;
    const u64* prev = gran + (size_t)((pos + n - 1) % n) * payload;
    const long long t0 = wall_clock64();
    __shared__ float acc[256];
    for (int lap = 0; lap < laps; ++lap) {
        const unsigned want = (pos == 0) ? (unsigned)lap : (unsigned)(lap + 1);
        float sum = 0.f;
        if (!(pos == 0 && lap == 0)) {
            if (mode == 2) {
                if (tid == 0) {
                    unsigned spins = 0;
                    while ((unsigned)(ld_sc1(prev + payload - 1) >> 32) != want) {
                        if ((++spins & 255u) == 0 && (wall_clock64() - t0 > timeout_ticks || *(volatile unsigned*)fail)) { *fail = 1; break; }
                        if (sleep_arg) __builtin_amdgcn_s_sleep(1);
                    }
                }
                __syncthreads();
            }
            for (int i = tid; i < payload; i += 256) {
                unsigned spins = 0;
                u64 v;
                while ((unsigned)((v = ld_sc1(prev + i)) >> 32) != want) {
                    if ((++spins & 255u) == 0 && (wall_clock64() - t0 > timeout_ticks || *(volatile unsigned*)fail)) { *fail = 1; break; }
                    if (sleep_arg) __builtin_amdgcn_s_sleep(1);
                }
                sum += __uint_as_float((unsigned)v);
            }
        }
        acc[tid] = sum;
        __syncthreads();
        if (*(volatile unsigned*)fail) return;
        const float val = acc[(tid + 1) & 255] + 1.0f;  // a token that depends on the data
        const u64 g = ((u64)(unsigned)(lap + 1) << 32) | (u64)__float_as_uint(val);
        for (int i = tid; i < payload; i += 256) {
            if (mode == 1) mine[i] = g; else st_sc1(mine + i, g);
        }
        __syncthreads();
    }
}

int main(int argc, char** argv) {
    int dev = 0;
    CHECK(hipSetDevice(dev));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, dev));
    int khz = 100000;
    (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev);
    printf("device %s CUs %d wallclock %d kHz\n", prop.gcnArchName, prop.multiProcessorCount, khz);
    const int laps = 2000;
    u64* gran; int* dpos; int* dxcc; unsigned* dfail;
    const int maxn = 256, maxpay = 1024;
    CHECK(hipMalloc(&gran, (size_t)maxn * maxpay * 8));
    CHECK(hipMalloc(&dpos, maxn * 4)); CHECK(hipMalloc(&dxcc, maxn * 4)); CHECK(hipMalloc(&dfail, 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    // placements: 0 = consecutive ring positions on consecutive blocks (every hop crosses XCDs if block b -> XCD b%8)
    //             1 = ring positions grouped per XCD (pos = (b%8)*cnt + b/8): 7 of n hops cross
    //             2 = whole ring on ONE XCD: n <= 32, grid = 8n, only blocks b%8==0 take part (the rest exit at once)
    for (int place = 0; place < 3; ++place)
        for (int n : {8, 32, 208})
            for (int payload : {1, 128, 640})
                for (int mode : {0, 2, 1}) {
                    if (place == 2 && n > 32) continue;
                    if (mode == 1 && place != 2) continue;
                    const int grid = place == 2 ? n * 8 : n;
                    std::vector<int> pos(grid, -1);
                    if (place == 0) for (int b = 0; b < n; ++b) pos[b] = b;
                    else if (place == 1) {
                        std::vector<int> cnt(8, 0), pre(9, 0);
                        for (int b = 0; b < n; ++b) cnt[b % 8]++;
                        for (int x = 0; x < 8; ++x) pre[x + 1] = pre[x] + cnt[x];
                        for (int b = 0; b < n; ++b) pos[b] = pre[b % 8] + b / 8;
                    } else for (int b = 0; b < grid; ++b) pos[b] = (b % 8 == 0) ? b / 8 : -1;
                    const int launch_grid = grid;
                    CHECK(hipMemcpy(dpos, pos.data(), grid * 4, hipMemcpyHostToDevice));
                    CHECK(hipMemset(gran, 0, (size_t)maxn * maxpay * 8));
                    CHECK(hipMemset(dfail, 0, 4));
                    for (int rep = 0; rep < 2; ++rep) {
                        CHECK(hipMemset(gran, 0, (size_t)maxn * maxpay * 8));
                        CHECK(hipEventRecord(e0));
                        hipLaunchKernelGGL(ring_probe, dim3(launch_grid), dim3(256), 0, 0, gran, dpos, n, laps, payload, mode, dxcc, dfail,
                                           (long long)khz * 3000, 1);
                        CHECK(hipEventRecord(e1));
                        CHECK(hipEventSynchronize(e1));
                    }
                    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                    unsigned f; CHECK(hipMemcpy(&f, dfail, 4, hipMemcpyDeviceToHost));
                    std::vector<int> xcc(grid);
                    CHECK(hipMemcpy(xcc.data(), dxcc, grid * 4, hipMemcpyDeviceToHost));
                    int agree = 0;
                    for (int b = 0; b < grid; ++b) agree += (xcc[b] == b % 8);
                    printf("place %d n %3d payload %4d mode %d : %8.3f us/hop  (%s; xcc==b%%8 for %d/%d blocks)\n", place, n, payload, mode,
                           ms * 1e3 / ((double)laps * n), f ? "TIMEOUT" : "ok", agree, grid);
                    fflush(stdout);
                }
    return 0;
}
