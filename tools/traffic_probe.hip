// traffic_probe.hip -- calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE (KiB per dispatch) on gfx950 against KNOWN byte counts for the access
// patterns the generation chain's hand-offs are made of: 16-byte accesses per lane (64 lanes = 1 KiB contiguous per wave instruction),
// write-through (sc1) or L2-resident (plain) stores, L1-bypassing (sc1) loads; every line touched once (streaming) or a small region
// re-used (resident in an XCD's L2).  Each kernel moves exactly BYTES bytes (printed); run it under
//     rocprofv3 --kernel-trace --pmc WRITE_SIZE -- tools/traffic_probe      and      ... --pmc FETCH_SIZE -- tools/traffic_probe
// and compare the counters of the six dispatches with it (tools/collect_profiles.sh does, into profiles/r04_pmc_calibration.txt).
//     hipcc --offload-arch=gfx950 -O3 -o tools/traffic_probe tools/traffic_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));

static __device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
// every workgroup walks its own contiguous slice; `wrap` bytes: the slice folds back onto its first `wrap` bytes (0 = never)
template <int AUX>   // 16 = sc1 (write-through), 0 = plain
__global__ __launch_bounds__(256) void k_store(char* buf, long long bytes_per_wg, long long wrap, int tag) {
    const __amdgpu_buffer_rsrc_t rs = rsrc(buf);
    const long long base = (long long)blockIdx.x * (wrap ? wrap : bytes_per_wg);
    const v4i d = {tag, tag + 1, tag + 2, tag + 3};
    for (long long off = (long long)threadIdx.x * 16; off < bytes_per_wg; off += 256 * 16) {
        const long long o = wrap ? off % wrap : off;
        __builtin_amdgcn_raw_buffer_store_b128(d, rs, (unsigned)(base + o), 0, AUX);
    }
}
__global__ __launch_bounds__(256) void k_load(const char* buf, long long bytes_per_wg, long long wrap, int* sink) {
    const __amdgpu_buffer_rsrc_t rs = rsrc(buf);
    const long long base = (long long)blockIdx.x * (wrap ? wrap : bytes_per_wg);
    int acc = 0;
    for (long long off = (long long)threadIdx.x * 16; off < bytes_per_wg; off += 256 * 16) {
        const long long o = wrap ? off % wrap : off;
        const v4i v = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)(base + o), 0, 16);   // sc1
        acc += v.x ^ v.w;
    }
    if (acc == 0x12345678) sink[0] = acc;
}

// the x' partials are polled with 8-byte loads (global_load_dwordx2 ... sc1)
__global__ __launch_bounds__(256) void k_load8(const unsigned long long* buf, long long bytes_per_wg, long long wrap, int* sink) {
    const unsigned long long* base = buf + (long long)blockIdx.x * (wrap ? wrap : bytes_per_wg) / 8;
    unsigned long long acc = 0;
    for (long long off = (long long)threadIdx.x * 8; off < bytes_per_wg; off += 256 * 8) {
        const long long o = wrap ? off % wrap : off;
        acc += __hip_atomic_load(base + o / 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (acc == 0x12345678ull) sink[0] = (int)acc;
}

int main() {
    CHECK(hipSetDevice(0));
    const int wgs = 1024;
    const long long per_wg = 1 << 20;                 // 1 MiB per workgroup: 1 GiB per kernel
    const long long total = (long long)wgs * per_wg;
    const long long wrap = 4096;                      // re-use: 4 KiB per workgroup = 4 MiB in all (half a MiB per XCD's L2)
    char* buf; int* sink;
    CHECK(hipMalloc(&buf, total)); CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(buf, 0, total));
    CHECK(hipDeviceSynchronize());
    printf("every kernel below moves %lld bytes = %lld KiB in 16-byte accesses per lane\n", total, total / 1024);
    printf("1 k_store<16> streaming, write-through\n2 k_store<0>  streaming, plain\n3 k_store<16> 4 KiB per workgroup re-used, write-through\n"
           "4 k_store<0>  4 KiB per workgroup re-used, plain\n5 k_load      streaming, sc1\n6 k_load      4 KiB per workgroup re-used, sc1\n"
           "7 k_load8     streaming, 8 bytes per lane, sc1\n");
    hipLaunchKernelGGL(k_store<16>, dim3(wgs), dim3(256), 0, 0, buf, per_wg, 0LL, 1);
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_store<0>, dim3(wgs), dim3(256), 0, 0, buf, per_wg, 0LL, 2);
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_store<16>, dim3(wgs), dim3(256), 0, 0, buf, per_wg, wrap, 3);
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_store<0>, dim3(wgs), dim3(256), 0, 0, buf, per_wg, wrap, 4);
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_load, dim3(wgs), dim3(256), 0, 0, buf, per_wg, 0LL, sink);
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_load, dim3(wgs), dim3(256), 0, 0, buf, per_wg, wrap, sink);
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_load8, dim3(wgs), dim3(256), 0, 0, reinterpret_cast<const unsigned long long*>(buf), per_wg, 0LL, sink);
    CHECK(hipDeviceSynchronize());
    printf("done\n");
    return 0;
}
