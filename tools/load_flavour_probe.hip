// load_flavour_probe.hip -- which cache-policy bits make the cheapest CORRECT hand-off between two workgroups of one XCD?
// Dev tool (hipcc --offload-arch=gfx950 -O3).  The generation chain polls 8-byte {value, tag} granules with agent-scope
// (sc1) loads: ~0.46 us per round trip in the kernel's stamps.  This probe times
//   (a) the raw round trip of a dependent load per flavour (one wave, the line resident in this XCD's L2), and
//   (b) a token ring of workgroups on ONE XCD (and, for reference, a ring that crosses XCDs on every hop), us per hop,
// for the load flavours  sc1 | sc0 | sc0+sc1 | nt | sc0+nt | plain after buffer_inv sc1 | sc0 after buffer_inv sc1 | plain
// and the store flavours plain | sc1 | sc0.  A flavour that lets the poll hit a stale L1/L2 line shows up as TIMEOUT.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
typedef int v2i __attribute__((ext_vector_type(2)));

static __device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
static __device__ char* rs_base;  // the granule buffer, for the flavours that need a pointer
template <int LF>
static __device__ __forceinline__ v2i ld(__amdgpu_buffer_rsrc_t rs, unsigned off) {
    if constexpr (LF == 0) return __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, 16);       // sc1
    else if constexpr (LF == 1) return __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, 1);   // sc0
    else if constexpr (LF == 2) return __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, 17);  // sc0 sc1
    else if constexpr (LF == 3) return __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, 2);   // nt
    else if constexpr (LF == 4) return __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, 3);   // sc0 nt
    else if constexpr (LF == 5) { asm volatile("buffer_inv sc1" ::: "memory"); return __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, 0); }
    else if constexpr (LF == 6) { asm volatile("buffer_inv sc1" ::: "memory"); return __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, 1); }
    else if constexpr (LF == 8) {  // an atomic executed AT the L2: fetch_add(0)
        unsigned long long* p = reinterpret_cast<unsigned long long*>(rs_base + off);
        const unsigned long long v = __hip_atomic_fetch_add(p, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return v2i{(int)(unsigned)v, (int)(unsigned)(v >> 32)};
    }
    else return __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, 0);                          // plain
}
template <int SF>
static __device__ __forceinline__ void st(__amdgpu_buffer_rsrc_t rs, unsigned off, v2i v) {
    if constexpr (SF == 0) __builtin_amdgcn_raw_buffer_store_b64(v, rs, off, 0, 0);
    else if constexpr (SF == 1) __builtin_amdgcn_raw_buffer_store_b64(v, rs, off, 0, 16);
    else if constexpr (SF == 2) __builtin_amdgcn_raw_buffer_store_b64(v, rs, off, 0, 1);
    else if constexpr (SF == 4) __builtin_amdgcn_raw_buffer_store_b64(v, rs, off, 0, 2);   // nt
    else if constexpr (SF == 5) __builtin_amdgcn_raw_buffer_store_b64(v, rs, off, 0, 3);   // sc0 nt
    else if constexpr (SF == 6) __builtin_amdgcn_raw_buffer_store_b64(v, rs, off, 0, 18);  // sc1 nt
    else {  // an atomic exchange executed AT the L2, result unused
        unsigned long long* p = reinterpret_cast<unsigned long long*>(rs_base + off);
        (void)__hip_atomic_exchange(p, ((unsigned long long)(unsigned)v.y << 32) | (unsigned)v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
static const char* SFN[] = {"plain", "sc1", "sc0", "xchg", "nt", "sc0+nt", "sc1+nt"};
static const char* LFN[] = {"sc1", "sc0", "sc0+sc1", "nt", "sc0+nt", "inv+plain", "inv+sc0", "plain", "fetch_add0"};

// (a) raw dependent-load round trip: one wave, the same line over and over (lane 0's result feeds the next address)
template <int LF>
__global__ void rtt_probe(int* buf, long long* out, int iters) {
    const __amdgpu_buffer_rsrc_t rs = rsrc(buf);
    unsigned off = threadIdx.x * 8;
    int acc = 0;
    const long long t0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
        const v2i v = ld<LF>(rs, off);
        acc += v.x;
        off = threadIdx.x * 8 + (unsigned)(v.y & 8);  // dependency (the buffer holds zeros: same address)
    }
    const long long t1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = acc; }
}

// (a2) store acknowledge time: store, then wait until the counter says it is complete
template <int SF>
__global__ void store_ack_probe(int* buf, long long* out, int iters) {
    const __amdgpu_buffer_rsrc_t rs = rsrc(buf);
    const unsigned off = threadIdx.x * 8;
    const long long t0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
        st<SF>(rs, off, v2i{i, i});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const long long t1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = 0; }
}
template <int SF>
static void run_store_ack(int* buf, long long* dout, int khz) {
    const int iters = 4000;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(store_ack_probe<SF>, dim3(1), dim3(64), 0, 0, buf, dout, iters);
        CHECK(hipDeviceSynchronize());
    }
    long long h[2];
    CHECK(hipMemcpy(h, dout, 16, hipMemcpyDeviceToHost));
    printf("ack  store %-5s : %.3f us from issue to vmcnt == 0\n", SFN[SF], (double)h[0] / (khz * 1e-3) / iters);
    fflush(stdout);
}

// (b) token ring: position pos waits for its predecessor's 128 granules of lap `want`, sums them through LDS, publishes its own
template <int LF, int SF>
__global__ __launch_bounds__(256) void ring_probe(int* gran, const int* pos_of_block, int n, int laps, unsigned* fail, long long timeout_ticks) {
    const int pos = pos_of_block[blockIdx.x], tid = threadIdx.x;
    if (pos < 0) return;
    constexpr int PAY = 128;
    const __amdgpu_buffer_rsrc_t rs = rsrc(gran);
    const unsigned mine = (unsigned)pos * PAY * 8, prev = (unsigned)((pos + n - 1) % n) * PAY * 8;
    const long long t0 = wall_clock64();
    __shared__ float acc[256];
    for (int lap = 0; lap < laps; ++lap) {
        const int want = (pos == 0) ? lap : lap + 1;
        float sum = 0.f;
        if (!(pos == 0 && lap == 0) && tid < PAY) {
            unsigned spins = 0;
            v2i v;
            while ((v = ld<LF>(rs, prev + tid * 8)).y != want) {
                if ((++spins & 255u) == 0 && (wall_clock64() - t0 > timeout_ticks || *(volatile unsigned*)fail)) { *fail = 1; break; }
            }
            sum = __int_as_float(v.x);
        }
        acc[tid] = sum;
        __syncthreads();
        if (*(volatile unsigned*)fail) return;
        const float val = acc[(tid + 1) & 127] + 1.0f;
        if (tid < PAY) st<SF>(rs, mine + tid * 8, v2i{__float_as_int(val), lap + 1});
        __syncthreads();
    }
}

template <int LF>
static void run_rtt(int* buf, long long* dout, int khz) {
    const int iters = 4000;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(rtt_probe<LF>, dim3(1), dim3(64), 0, 0, buf, dout, iters);
        CHECK(hipDeviceSynchronize());
    }
    long long h[2];
    CHECK(hipMemcpy(h, dout, 16, hipMemcpyDeviceToHost));
    printf("rtt  load %-9s : %.3f us per dependent load\n", LFN[LF], (double)h[0] / (khz * 1e-3) / iters);
    fflush(stdout);
}

template <int LF, int SF>
static void run_ring(int* gran, int* dpos, unsigned* dfail, int n, bool one_xcd, int khz) {
    const int laps = 2000, grid = one_xcd ? n * 8 : n;
    std::vector<int> pos(grid, -1);
    for (int b = 0; b < grid; ++b) pos[b] = one_xcd ? ((b % 8 == 0) ? b / 8 : -1) : b;
    CHECK(hipMemcpy(dpos, pos.data(), grid * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipMemset(gran, 0, (size_t)256 * 128 * 8));
        CHECK(hipMemset(dfail, 0, 4));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((ring_probe<LF, SF>), dim3(grid), dim3(256), 0, 0, gran, dpos, n, laps, dfail, (long long)khz * 300);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
    }
    unsigned f;
    CHECK(hipMemcpy(&f, dfail, 4, hipMemcpyDeviceToHost));
    printf("ring %s n %2d  store %-5s load %-9s : %8.3f us/hop  %s\n", one_xcd ? "one-XCD  " : "cross-XCD", n, SFN[SF], LFN[LF],
           ms * 1e3 / ((double)laps * n), f ? "TIMEOUT (stale line)" : "ok");
    fflush(stdout);
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
}

int main() {
    CHECK(hipSetDevice(0));
    int khz = 100000;
    (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0);
    int *gran, *dpos;
    unsigned* dfail;
    long long* dout;
    CHECK(hipMalloc(&gran, (size_t)256 * 128 * 8));
    CHECK(hipMalloc(&dpos, 2048 * 4));
    CHECK(hipMalloc(&dfail, 4));
    CHECK(hipMalloc(&dout, 16));
    CHECK(hipMemset(gran, 0, (size_t)256 * 128 * 8));
    run_rtt<0>(gran, dout, khz); run_rtt<1>(gran, dout, khz); run_rtt<2>(gran, dout, khz); run_rtt<3>(gran, dout, khz);
    run_rtt<4>(gran, dout, khz); run_rtt<5>(gran, dout, khz); run_rtt<6>(gran, dout, khz); run_rtt<7>(gran, dout, khz);
    { char* b = reinterpret_cast<char*>(gran); CHECK(hipMemcpyToSymbol(HIP_SYMBOL(rs_base), &b, sizeof(b))); }
    run_store_ack<0>(gran, dout, khz); run_store_ack<1>(gran, dout, khz); run_store_ack<2>(gran, dout, khz);
    for (int n : {16}) {
        run_ring<0, 0>(gran, dpos, dfail, n, true, khz);
        run_ring<0, 1>(gran, dpos, dfail, n, true, khz);
        run_ring<1, 0>(gran, dpos, dfail, n, true, khz);
        run_ring<1, 2>(gran, dpos, dfail, n, true, khz);
        run_ring<2, 0>(gran, dpos, dfail, n, true, khz);
        run_ring<3, 0>(gran, dpos, dfail, n, true, khz);
        run_ring<4, 0>(gran, dpos, dfail, n, true, khz);
        run_ring<5, 0>(gran, dpos, dfail, n, true, khz);
        run_ring<6, 0>(gran, dpos, dfail, n, true, khz);
        run_ring<7, 0>(gran, dpos, dfail, n, true, khz);
        run_ring<0, 4>(gran, dpos, dfail, n, true, khz);
        run_ring<0, 5>(gran, dpos, dfail, n, true, khz);
        run_ring<0, 6>(gran, dpos, dfail, n, true, khz);
        run_ring<3, 4>(gran, dpos, dfail, n, true, khz);
        run_ring<0, 6>(gran, dpos, dfail, n, false, khz);
        run_ring<0, 3>(gran, dpos, dfail, n, true, khz);
        run_ring<8, 0>(gran, dpos, dfail, n, true, khz);
        run_ring<8, 3>(gran, dpos, dfail, n, true, khz);
        run_ring<0, 3>(gran, dpos, dfail, n, false, khz);
        run_ring<8, 3>(gran, dpos, dfail, n, false, khz);
        run_ring<0, 1>(gran, dpos, dfail, n, false, khz);
        run_ring<2, 1>(gran, dpos, dfail, n, false, khz);
        run_ring<1, 1>(gran, dpos, dfail, n, false, khz);
        run_ring<5, 1>(gran, dpos, dfail, n, false, khz);
    }
    return 0;
}
