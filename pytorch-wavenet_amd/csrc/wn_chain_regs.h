// wn_chain_regs.h -- device helpers of the register-resident generation chain (gfx950, device only): the per-lane weight-image
// shapes (WnV2Shape), DPP reductions, LDS dot products on register weights, the branch-free exp of the gated unit, LDS-only
// barriers, XCD queries, wall-clock stamps, wave-level reductions / scans for the sampler and the LDS layout of a workgroup.
//
// History: rounds 1-2 had 256-thread kernels built on these (wn_generate_kernel_v2 single stream, wn_generate_kernel_v2m multi
// stream, two chains sharing the CUs).  Round 3 moved every instantiated shape to the wave-specialised kernel (wn_kernel_v3.h) and
// removed them; the names of the shape / layout structs keep their "V2" for the weight images' sake (same per-lane layout).
#ifndef WN_CHAIN_REGS_H
#define WN_CHAIN_REGS_H

#include "wn_kernel.h"

template <int R_, int DC_, int S_, int EC_>
struct WnV2Shape {
    static constexpr int R = R_, DC = DC_, S = S_, EC = EC_, C = 256;
    static constexpr int G1 = 2 * DC;     // filter+gate rows of this slice
    static constexpr int T1 = 256 / G1;   // lanes per fg row
    static constexpr int K1 = R / T1;     // channels per lane per tap
    static constexpr int T2 = 256 / R;    // lanes per residual row
    static constexpr int K2 = DC / T2;    // z channels per lane
    static constexpr int RS = S / 256;    // skip rows per lane (full DC reduction each)
    static constexpr int T3 = 256 / EC;   // lanes per end_conv_1 row
    static constexpr int K3 = S / T3;     // skip channels per lane
    // per-lane register images (floats), stored striped in HBM: image[j*256 + tid]
    static constexpr int NWL = 2 * K1 + K2 + RS * DC + 2 + RS;  // w1 | w0 | w2 | w3 | bias_fg, bias_res | bias_skip[RS]
    static constexpr int NWH = K3 + EC + 2;                     // end1 slice | end2 row | b1 | b2
    static __host__ __device__ constexpr int xpad(int ch) { return ch + 4 * (ch / K1); }    // LDS index of x[ch]
    static __host__ __device__ constexpr int skpad(int i) { return i + 4 * (i / K3); }      // LDS index of skip[i]
    static_assert(G1 <= 256 && 256 % G1 == 0 && T1 <= 16, "fg rows must tile 256 lanes");
    static_assert(R <= 256 && 256 % R == 0 && T2 <= 16, "residual rows must tile 256 lanes");
    static_assert(R % T1 == 0 && DC % T2 == 0 && S % 256 == 0 && 256 % EC == 0 && S % T3 == 0 && T3 <= 16, "shape");
};

// ---- DPP butterflies: after wn_reduce<T> every lane of an aligned T-lane group holds the group sum
template <int CTRL>
static __device__ __forceinline__ float wn_dpp(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
template <int LEVEL>  // partner at distance LEVEL inside a group whose lanes all hold the same value
static __device__ __forceinline__ float wn_partner(float v) {
    if constexpr (LEVEL == 1) return wn_dpp<0xB1>(v);        // quad_perm [1,0,3,2]
    else if constexpr (LEVEL == 2) return wn_dpp<0x4E>(v);   // quad_perm [2,3,0,1]
    else if constexpr (LEVEL == 4) return wn_dpp<0x141>(v);  // row_half_mirror
    else if constexpr (LEVEL == 8) return wn_dpp<0x140>(v);  // row_mirror
    else return __shfl_xor(v, LEVEL);
}
template <int T>
static __device__ __forceinline__ float wn_reduce(float v) {
    if constexpr (T >= 2) v += wn_partner<1>(v);
    if constexpr (T >= 4) v += wn_partner<2>(v);
    if constexpr (T >= 8) v += wn_partner<4>(v);
    if constexpr (T >= 16) v += wn_partner<8>(v);
    return v;
}


// dot(w[0..K), x[0..K)) with x in LDS: all float4 reads issued up front (one LDS latency, not K/16 of them), then
// four independent FMA chains (K % 4 == 0); else a plain chain
typedef float wn_f2 __attribute__((ext_vector_type(2)));
template <int K>
static __device__ __forceinline__ float wn_dot_lds(const float (&w)[K], const float* x, float init) {
    if constexpr (K % 4 == 0) {
        float4 v[K / 4];
        const float4* x4 = reinterpret_cast<const float4*>(x);
#pragma unroll
        for (int k = 0; k < K / 4; ++k) v[k] = x4[k];
        // the four chains as two PACKED chains (v_pk_fma_f32: two fp32 FMAs per lane and instruction; each element's arithmetic and
        // the final summation order are those of four scalar chains: bit-identical)
        wn_f2 a01 = {init, 0.f}, a23 = {0.f, 0.f};
#pragma unroll
        for (int k = 0; k < K / 4; ++k) {
            a01 = __builtin_elementwise_fma(wn_f2{w[4 * k], w[4 * k + 1]}, wn_f2{v[k].x, v[k].y}, a01);
            a23 = __builtin_elementwise_fma(wn_f2{w[4 * k + 2], w[4 * k + 3]}, wn_f2{v[k].z, v[k].w}, a23);
        }
        return (a01.x + a01.y) + (a23.x + a23.y);
    } else {
        float a = init;
#pragma unroll
        for (int k = 0; k < K; ++k) a += w[k] * x[k];
        return a;
    }
}

// e^x with the accuracy of the library expf (product x*log2(e) carried in two floats, v_exp_f32 on the reduced
// argument, v_ldexp_f32) but branch-free and without its range clamps -- v_exp/v_ldexp saturate to 0 / inf by
// themselves -- so that the two exponentials of a gated unit schedule as two interleaved dependency chains.
static __device__ __forceinline__ float wn_exp(float x) {
    const float p = x * 1.44269504088896341f;
    float lo = fmaf(x, 1.44269504088896341f, -p);  // exact rounding error of the product
    lo = fmaf(x, 1.92596299112661746e-8f, lo);     // + x * (log2(e) - float(log2(e)))
    const float n = rintf(p);
    return ldexpf(__builtin_amdgcn_exp2f((p - n) + lo), (int)n);
}

// diagnostics: wall-clock stamp k of this workgroup's step: 0 start, 1 input staged, 2 x' published, 3 done,
// 4 filter/gate sums ready, 5 z staged
#define WN_STAMPS 8
// Stamps are parked in LDS (one ds_write, no vector-memory traffic on the critical path) and flushed to HBM by
// wn_stamp_flush at the end of the step.
static __device__ __forceinline__ void wn_stamp(const WnRun& r, long long* park, long long item, int k, bool cx_single = false) {
    if (r.prof && item < r.prof_items && threadIdx.x == 0) {
        park[k] = (long long)wall_clock64();
        if (k == 0 && cx_single) park[6] = (long long)clock64();  // shader clock, to read the effective MHz off the stamps
    }
}
static __device__ __forceinline__ void wn_stamp_flush(const WnRun& r, const long long* park, int w, long long item) {
    if (r.prof && item < r.prof_items && threadIdx.x == 0) {
        long long* dst = r.prof + ((size_t)w * r.prof_items + item) * WN_STAMPS;
#pragma unroll
        for (int k = 0; k < 8; ++k) dst[k] = park[k];
    }
}

// XCC id of this workgroup's CU (HW_REG_XCC_ID, 4 bits)
static __device__ __forceinline__ int wn_xcc_id() { return (int)(__builtin_amdgcn_s_getreg(((4 - 1) << 11) | 20) & 0xf); }

// true iff every chain position in [first, first+count) reports the same XCC as `mine` (bounded wait for their entry)
static __device__ bool wn_same_xcd(WnCtx& cx, int mine, int first, int count) {
    bool same = true;
    for (int q = first; q < first + count; ++q) {
        unsigned v = 0, spins = 0;
        while ((v = __hip_atomic_load(cx.p->xcc_tab + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0u) {
            if ((++spins & 63u) == 0u && (long long)wall_clock64() - cx.t_start > cx.r->timeout_ticks) return false;
            __builtin_amdgcn_s_sleep(8);
        }
        same = same && ((int)v - 1 == mine);
    }
    return same;
}

// Workgroup barrier for data exchanged through LDS ONLY.  __syncthreads() also drains every outstanding vector
// memory operation (s_waitcnt vmcnt(0)): with request loads, queue taps and write-through stores in flight that
// costs ~1.4 us per step in the multi-stream pipeline.  Here only the LDS counter is waited for.
static __device__ __forceinline__ void wn_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// barrier that also tells every lane whether any lane gave up a wait (rare): one s_barrier, one LDS word
static __device__ __forceinline__ bool wn_barrier_failed(WnCtx& cx, int* flag) {  // NOT volatile: a volatile generic
    // pointer is compiled to FLAT accesses, and a flat load waits for every outstanding vector-memory operation
    if (cx.fail) *flag = 1;
    wn_lds_barrier();
    return *flag != 0;
}

// ---- wave-level helpers for the sampler: 16-lane rows with DPP butterflies, the 4 rows combined through
// v_readlane (uniform values) -- no ds_bpermute chains (a 6-step __shfl reduction costs ~0.3 us of pure latency).
static __device__ __forceinline__ float wn_lane_f(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }
static __device__ __forceinline__ float wn_wave_max(float v) {
    v = fmaxf(v, wn_dpp<0xB1>(v)); v = fmaxf(v, wn_dpp<0x4E>(v)); v = fmaxf(v, wn_dpp<0x141>(v)); v = fmaxf(v, wn_dpp<0x140>(v));
    return fmaxf(fmaxf(wn_lane_f(v, 0), wn_lane_f(v, 16)), fmaxf(wn_lane_f(v, 32), wn_lane_f(v, 48)));
}
static __device__ __forceinline__ float wn_wave_sum(float v) {
    v += wn_dpp<0xB1>(v); v += wn_dpp<0x4E>(v); v += wn_dpp<0x141>(v); v += wn_dpp<0x140>(v);
    return (wn_lane_f(v, 0) + wn_lane_f(v, 16)) + (wn_lane_f(v, 32) + wn_lane_f(v, 48));
}
template <int CTRL>
static __device__ __forceinline__ int wn_dpp_i(int x) { return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xf, 0xf, true); }
static __device__ __forceinline__ int wn_wave_min_i(int v) {
    v = min(v, wn_dpp_i<0xB1>(v)); v = min(v, wn_dpp_i<0x4E>(v)); v = min(v, wn_dpp_i<0x141>(v)); v = min(v, wn_dpp_i<0x140>(v));
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
template <int N>  // value held by lane-N of the same 16-lane row, 0.0 where there is none (row_shr:N)
static __device__ __forceinline__ double wn_row_shr_f64(double v) {
    const int lo = wn_dpp_i<0x110 + N>(__double2loint(v)), hi = wn_dpp_i<0x110 + N>(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
static __device__ __forceinline__ double wn_lane_d(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}

// LDS layout (floats) of a chain workgroup: G streams are processed per pipeline item
template <class SH, int G>
struct WnV2LdsM {
    static constexpr int XR = SH::R + 4 * SH::T1, SKP = SH::S + 4 * SH::T3, DCP = (SH::DC + 3) & ~3;
    static constexpr int xs = 0;                   // [2][G][XR]
    static constexpr int zs = xs + 2 * G * XR;     // [G][DCP]
    static constexpr int xo = zs + G * DCP;        // [G][XR]  queue taps of the item
    static constexpr int sk = xo + G * XR;         // [G][SKP] head
    static constexpr int ev = sk + G * SKP;        // [G][EC]  head
    static constexpr int smp = ev + G * SH::EC;    // sampler scratch (64 floats); [48] fail flag, [52..] flags
    static constexpr int park = smp + 64;          // 8 parked int64 stamps
    static constexpr int pre = park + 16;          // [n_streams][256]
    static __host__ __device__ int floats(int n_streams) { return pre + n_streams * 256; }
};

#endif  // WN_CHAIN_REGS_H
