// wn_kernel_v2.h -- latency-optimised generation chain for gfx950 (device only).
//
// Same chain, same hand-off protocol and same HBM buffers as the generic kernel (wn_kernel.h, which stays
// the fallback for arbitrary shapes), re-organised around what the
// MI355X hand-off probe measured (profiles/r01_handoff_probe.txt): a granule hop costs 0.45-0.75 us no
// matter what, so everything else must get off the critical path x[t] -> filter/gate -> z -> residual -> x'.
//
//   * WEIGHTS LIVE IN REGISTERS.  One workgroup = 4 waves, one per SIMD, so each lane owns up to 512 VGPRs;
//     a layer slice is 100-200 floats per lane (cfg3: 148).  No LDS traffic for weights at all; LDS only
//     carries activations (x, z, skip sum) between the lanes of the workgroup.
//   * compile-time shapes (template <R, DC, S, EC>, k = 2, C = 256): every matvec is a fully unrolled FMA
//     chain on register operands; row sums are finished with DPP butterflies (no LDS, no barrier).
//   * the tap-0 half of the dilated conv (W[:,:,0] . x[t-d]) does not depend on the token: it is computed
//     right after the previous step of the stream (off the critical path) and parked in LDS (prebuf).
//   * polls are batched: a lane issues all its granule loads, then checks all tags (one round trip per
//     retry instead of one per granule).
//   * 2 workgroup barriers per step on the critical path (x staged, z staged); skip 1x1, queue push and the
//     next step's tap-0 run after x' has been published.
//   * the queue rings stay in HBM in the reference's DilatedQueue layout (wn_export_queue still works);
//     they are only touched off the critical path.
#ifndef WN_KERNEL_V2_H
#define WN_KERNEL_V2_H

#include "wn_kernel.h"


#if !defined(WN_EXPERIMENT) && (defined(WN_MULTI_SLEEP) || defined(WN_ABL))
#error "WN_ABL / WN_MULTI_SLEEP are experiment switches: they need -DWN_EXPERIMENT (never set by build.py)"
#endif
#ifndef WN_MULTI_SLEEP
#define WN_MULTI_SLEEP 0
#endif
#ifndef WN_ABL
#define WN_ABL 0  // timing ablations for tools/ablate.sh (results are wrong when != 0): 1 no fg dot, 2 cheap gating, 3 no res dot, 5 all
#endif

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

// z = tanh(f) * sigmoid(g) with tanh(f) = 2*sigmoid(2f) - 1 (absolute error ~1e-7, the size of the fp32 rounding of
// the dot products that consume z) and 1-ulp reciprocals
static __device__ __forceinline__ float wn_gate(float f, float g) {
    const float e1 = wn_exp(-2.0f * f), e2 = wn_exp(-g);
    const float r1 = __builtin_amdgcn_rcpf(1.0f + e1), r2 = __builtin_amdgcn_rcpf(1.0f + e2);
    return fmaf(2.0f, r1, -1.0f) * r2;
}

static __device__ __forceinline__ wn_u64 wn_ld_granule(const wn_u64* g) {
    return __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Batched poll with a compile-time count: loads the N granules base[j*stride] together until every tag matches;
// returns their sum in the fixed order j = 0..N-1.
template <int N, int SLEEP = 0>  // SLEEP > 0: s_sleep between retries (multi-stream fallback: keep poll pressure off the fabric)
static __device__ __forceinline__ float wn_poll_fixed(WnCtx& cx, const wn_u64* base, size_t stride, uint32_t tag, int where,
                                                      long long e, int s) {
    if (cx.fail) return 0.f;
    unsigned spins = 0;
    for (;;) {
        wn_u64 v[N];
#pragma unroll
        for (int j = 0; j < N; ++j) v[j] = wn_ld_granule(base + (size_t)j * stride);
        bool ok = true;
#pragma unroll
        for (int j = 0; j < N; ++j) ok = ok && ((uint32_t)(v[j] >> 32) == tag);
        if (ok) {
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < N; ++j) sum += __uint_as_float((uint32_t)v[j]);
            return sum;
        }
        if ((++spins & 127u) == 0u) {  // the wall clock (s_memrealtime, hundreds of cycles) is only read on this slow path: the
            // spin bound of a wait runs from its first check, not from a time stamp taken on every item
            if (__hip_atomic_load(cx.p->status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { cx.fail = 1; return 0.f; }
            const long long now = (long long)wall_clock64();
            if (spins == 128u) cx.t_start = now;
            else if (now - cx.t_start > cx.r->timeout_ticks) { wn_give_up(cx, where, e, s); return 0.f; }
        }
        if constexpr (SLEEP > 0) __builtin_amdgcn_s_sleep(SLEEP);
    }
}

// n is block-uniform and small: dispatch to the unrolled form
template <int NMAX>
static __device__ __forceinline__ float wn_poll_sum(WnCtx& cx, const wn_u64* base, size_t stride, int n, uint32_t tag,
                                                    int where, long long e, int s) {
    switch (n) {
        case 1: return wn_poll_fixed<1>(cx, base, stride, tag, where, e, s);
        case 2: return wn_poll_fixed<2>(cx, base, stride, tag, where, e, s);
        case 4: return wn_poll_fixed<4>(cx, base, stride, tag, where, e, s);
        case 8: return wn_poll_fixed<(NMAX >= 8 ? 8 : 1)>(cx, base, stride, tag, where, e, s);
        case 16: return wn_poll_fixed<(NMAX >= 16 ? 16 : 1)>(cx, base, stride, tag, where, e, s);
        default: {
            float sum = 0.f;  // odd fan-in: one granule at a time
            for (int j = 0; j < n; ++j) sum += wn_poll_fixed<1>(cx, base + (size_t)j * stride, 0, tag, where, e, s);
            return sum;
        }
    }
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

// LDS layout (floats) of the v2 kernel
template <class SH>
struct WnV2Lds {
    // k-slices that different lanes of a wave read with ds_read_b128 are spaced K+4 floats apart: at a power-of-two
    // spacing they start on the same banks (8-way conflicts on the head's 256-byte slices, 2-way on the layer's)
    static constexpr int XR = SH::R + 4 * SH::T1;        // padded length of one x buffer
    static constexpr int SKP = SH::S + 4 * SH::T3;       // padded length of the head's skip vector
    static constexpr int xs = 0;                         // [2][XR]
    static constexpr int zs = xs + 2 * XR;               // [DC]
    static constexpr int sk = zs + ((SH::DC + 3) & ~3);  // [SKP]  head
    static constexpr int ev = sk + SKP;                  // [EC]   head
    static constexpr int smp = ev + SH::EC;              // sampler scratch: 64 floats (8-byte aligned); [48] = fail flag
    static constexpr int park = smp + 64;                // 8 parked int64 stamps
    static constexpr int xo = park + 16;                 // [XR] multi-stream: queue tap x[t+1-d] of the current item
    static constexpr int pre = xo + XR;                 // [n_streams][256]
    static __host__ __device__ int floats(int n_streams) { return pre + n_streams * 256; }
    // single-stream kernel: layer 0 keeps start_conv^T ([C][R]) behind the pre buffer when it fits (p.start_in_lds)
    static __host__ __device__ int floats_with_start(int n_streams) { return floats(n_streams) + 256 * SH::R; }
};

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

// Idle workgroups must not hammer the fabric with polls while the token is far away: if the previous wait
// for this stream was long, sleep through most of it before the first poll.
static __device__ __forceinline__ void wn_presleep(long long wait_ticks) {
    // 100 MHz ticks; s_sleep 32 ~ 2048 clocks ~ 1 us.  Sleep 3/4 of the previous wait beyond the first 4 us.
    long long n = (wait_ticks - 400) * 3 / 400;
    for (long long i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(32);
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

// ---- sampler: C = 256 classes, one per lane.  Same arithmetic as wn_sample (wn_kernel.h) /
// wavenet_model.py:280-294.  Returns the class index (uniform over the block).
static __device__ __forceinline__ int wn_sample_v2(WnCtx& cx, float* scratch, float logit, double u, bool greedy, float temperature) {
    const WnRun& r = *cx.r;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    float* fsc = scratch;                                    // [0..3] wave max, [4..7] wave sum
    int* isc = reinterpret_cast<int*>(scratch + 8);          // [0..3] wave argmax, [4..7] wave count
    double* dsc = reinterpret_cast<double*>(scratch + 24);   // [0..3] wave totals
    float x = logit;
    if (r.reg) x -= r.reg[tid];
    if (!greedy) x = x / temperature;
    const float wm = wn_wave_max(x);
    if (lane == 0) fsc[wv] = wm;
    wn_lds_barrier();
    const float gm = fmaxf(fmaxf(fsc[0], fsc[1]), fmaxf(fsc[2], fsc[3]));
    if (greedy) {  // first index of the maximum (torch.max semantics)
        const int wa = wn_wave_min_i(x == gm ? tid : 0x7fffffff);
        if (lane == 0) isc[wv] = wa;
        wn_lds_barrier();
        const int ga = min(min(isc[0], isc[1]), min(isc[2], isc[3]));
        wn_lds_barrier();
        return ga;
    }
    const float p = expf(x - gm);
    const float ws = wn_wave_sum(p);
    if (lane == 0) fsc[4 + wv] = ws;
    wn_lds_barrier();
    const float tot = ((fsc[4] + fsc[5]) + fsc[6]) + fsc[7];
    const float inv = 1.0f / tot;
    double run = (double)(p * inv);  // inclusive float64 scan (np.cumsum): rows by DPP, rows joined by readlane
    run += wn_row_shr_f64<1>(run);
    run += wn_row_shr_f64<2>(run);
    run += wn_row_shr_f64<4>(run);
    run += wn_row_shr_f64<8>(run);
    const double r0 = wn_lane_d(run, 15), r1 = wn_lane_d(run, 31), r2 = wn_lane_d(run, 47), r3 = wn_lane_d(run, 63);
    const int row = lane >> 4;
    run += row == 0 ? 0. : row == 1 ? r0 : row == 2 ? r0 + r1 : (r0 + r1) + r2;
    if (lane == 0) dsc[wv] = ((r0 + r1) + r2) + r3;
    wn_lds_barrier();
    double base = 0.;
    for (int w = 0; w < wv; ++w) base += dsc[w];
    const double total = ((dsc[0] + dsc[1]) + dsc[2]) + dsc[3];
    const bool le = (base + run) / total <= u;  // searchsorted(cdf / cdf[-1], u, side='right')
    const int cnt = __popcll(__ballot(le));
    if (lane == 0) isc[4 + wv] = cnt;
    wn_lds_barrier();
    int idx = isc[4] + isc[5] + isc[6] + isc[7];
    if (idx > 255) idx = 255;
    wn_lds_barrier();
    return idx;
}

// ------------------------------------------------------------------------------------------------
template <class SH>
static __device__ void wn_v2_layer(const WnPlan& p, const WnRun& r, WnCtx& cx, float* lds, int l, int c) {
    constexpr int R = SH::R, DC = SH::DC, S = SH::S, T1 = SH::T1, K1 = SH::K1, T2 = SH::T2, K2 = SH::K2, RS = SH::RS;
    using L = WnV2Lds<SH>;
    const int tid = threadIdx.x;
    const int ns = p.n_streams, P = p.P, NL = p.NL;
    // ---- register-resident weights of this slice
    float w1[K1], w0[K1], w2[K2], w3[RS][DC], bskip[RS];
    const float* img = p.blobs + (size_t)cx.w * (SH::NWL * 256) + tid;
    {
        int j = 0;
#pragma unroll
        for (int k = 0; k < K1; ++k) w1[k] = img[(size_t)(j++) * 256];
#pragma unroll
        for (int k = 0; k < K1; ++k) w0[k] = img[(size_t)(j++) * 256];
#pragma unroll
        for (int k = 0; k < K2; ++k) w2[k] = img[(size_t)(j++) * 256];
#pragma unroll
        for (int q = 0; q < RS; ++q)
#pragma unroll
            for (int k = 0; k < DC; ++k) w3[q][k] = img[(size_t)(j++) * 256];
    }
    const float bfg = img[(size_t)(2 * K1 + K2 + RS * DC) * 256];
    const float bres = img[(size_t)(2 * K1 + K2 + RS * DC + 1) * 256];
#pragma unroll
    for (int q = 0; q < RS; ++q) bskip[q] = img[(size_t)(2 * K1 + K2 + RS * DC + 2 + q) * 256];

    const int kq1 = tid % T1, grp = tid / T1, ch = grp >> 1, is_gate = grp & 1;
    const int kq2 = tid % T2, row2 = tid / T2;
    const int d = p.dil[l];
    const int ML = d + 1;  // (k-1)*d + 1, k = 2
    float* xs = lds + L::xs;
    float* zs = lds + L::zs;
    float* pre = lds + L::pre;
    float* smp = lds + L::smp;
    int* failflag = reinterpret_cast<int*>(smp + 48);
    long long* park = reinterpret_cast<long long*>(lds + L::park);
    int* locflags = reinterpret_cast<int*>(smp + 52);
    if (tid == 0) {
        *failflag = 0;
        const int mine = wn_xcc_id();
        __hip_atomic_store(p.xcc_tab + cx.w, (unsigned)(mine + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int lx = 0, lsk = 0;
        if (p.allow_plain) {
            if (l < NL - 1) {
                lx = wn_same_xcd(cx, mine, (l + 1) * P, P);         // x' partials feed every slice of layer l+1
                lsk = wn_same_xcd(cx, mine, (l + 1) * P + c, 1);    // the skip lane feeds slice c of layer l+1
            } else {
                lsk = wn_same_xcd(cx, mine, NL * P, p.PA);          // ... or every head workgroup
            }
        }
        locflags[0] = lx; locflags[1] = lsk;
    }
    __syncthreads();
    const bool local_x = locflags[0] != 0, local_s = locflags[1] != 0;

    // layer 0: the start_conv column gather (wavenet_model.py:127) reads LDS instead of L2/HBM when the table fits
    const float* start_tab = p.start_t;
    if (l == 0 && p.start_in_lds) {
        float* st = lds + L::floats(ns);
        for (int i = tid; i < 256 * R; i += 256) st[i] = p.start_t[i];
        start_tab = st;
    }
    // tap-0 contribution for the first evaluation of every stream: x[t_base - d] from the queue (zeros after reset)
    for (int s = 0; s < ns; ++s) {
        const float* ring = p.rings + p.ring_off[l] + ((size_t)c * ns + s) * (size_t)ML * R;
        long long pos = (r.t_base - d) % ML;
        if (pos < 0) pos += ML;
        float acc = kq1 == 0 ? bfg : 0.f;
#pragma unroll
        for (int k = 0; k < K1; ++k) acc += w0[k] * ring[(size_t)pos * R + kq1 * K1 + k];
        pre[s * 256 + tid] = acc;
    }
    __syncthreads();

    const long long n_it = r.n_eval + (l == 0 ? 1 : 0);
    int buf = 0;
    long long last_wait = 0;
    int tmod = (int)(r.t_base % ML);  // queue slot of x[t]; kept incrementally (a 64-bit modulo costs ~100 scalar ops)
    for (long long e = 0; e < n_it; ++e, tmod = (tmod + 1 == ML) ? 0 : tmod + 1) {
        const bool prime = e < r.n_given - 1;
        const uint32_t tag = (uint32_t)(e + 1);
        const int tapmod = tmod + 2 >= ML ? tmod + 2 - ML : tmod + 2;  // slot of x[t+1-d]: (t+1-d) mod (d+1) = (t+2) mod (d+1)
        for (int s = 0; s < ns; ++s, buf ^= 1) {
            float* xb = xs + buf * L::XR;
            const long long t_begin = (long long)wall_clock64();
            cx.t_start = t_begin;  // the spin bound is per hand-off wait, not per job
            const long long item = e * ns + s;
            wn_stamp(r, park, item, 0);
            // ---- 1. layer input x[t]
            if (l == 0) {
                int idx;
                if (e == 0) {
                    idx = r.first[(size_t)s * r.n_given];
                } else {
                    if (ns == 1) wn_presleep(last_wait);
                    const float logit = wn_poll_sum<16>(cx, p.gl + (size_t)s * 256 + tid, (size_t)ns * 256, p.PA, (uint32_t)e,
                                                        WN_W_LOGITS, e, s);
                    if (wn_barrier_failed(cx, failflag)) return;
                    last_wait = (long long)wall_clock64() - t_begin;
                    if (e < r.n_given) {
                        idx = r.first[(size_t)s * r.n_given + e];
                    } else {
                        const long long g = e - r.n_given;
                        if (r.dbg_logits && c == 0) r.dbg_logits[((size_t)s * r.num_samples + g) * 256 + tid] = logit;
                        const float temp = r.stream_temps ? r.stream_temps[s] : r.temperature;
                        const bool greedy = r.greedy != 0 || !(temp > 0.f);
                        const double u = greedy ? 0. : r.uniforms[(size_t)s * r.num_samples + g];
                        idx = wn_sample_v2(cx, smp, logit, u, greedy, temp);
                        if (c == 0 && tid == 0) r.out_idx[(size_t)s * r.num_samples + g] = idx;
                    }
                }
                if (e == r.n_eval) continue;
                if (tid < R) xb[SH::xpad(tid)] = start_tab[(size_t)idx * R + tid] + (p.start_b ? p.start_b[tid] : 0.f);
                wn_lds_barrier();
            } else {
                if (tid < R) {
                    if (ns == 1) wn_presleep(last_wait);
                    const wn_u64* g = p.gx + (((size_t)(l - 1) * P) * ns + s) * R + tid;
                    xb[SH::xpad(tid)] = wn_poll_sum<8>(cx, g, (size_t)ns * R, P, tag, WN_W_X, e, s);
                    last_wait = (long long)wall_clock64() - t_begin;
                }
                if (wn_barrier_failed(cx, failflag)) return;
            }
            wn_stamp(r, park, item, 1);
            // ---- 2. filter/gate: tap 1 on x[t] + parked tap 0, tanh * sigmoid   (wavenet_model.py:147-151)
            const float xres = (c == 0 && kq2 == 0) ? xb[SH::xpad(row2)] : 0.f;  // newest tap for the residual add, fetched early
            float acc = (WN_ABL == 1 || WN_ABL == 5) ? pre[s * 256 + tid] + w1[0] : wn_dot_lds<K1>(w1, xb + kq1 * (K1 + 4), pre[s * 256 + tid]);
            acc = wn_reduce<T1>(acc);
            const float other = wn_partner<T1>(acc);  // the gate (resp. filter) row of the same channel
            const float fv = is_gate ? other : acc, gv = is_gate ? acc : other;
            wn_stamp(r, park, item, 4);
            const float z = (WN_ABL == 2 || WN_ABL == 5) ? fv * gv * 0.001f : wn_gate(fv, gv);
            if (!is_gate && kq1 == 0) zs[ch] = z;
            wn_lds_barrier();
            wn_stamp(r, park, item, 5);
            // ---- 3. residual 1x1 partial, published at once                      (wavenet_model.py:164-165)
            if (l < NL - 1) {
                float a2 = (WN_ABL == 3 || WN_ABL == 5) ? zs[kq2 * K2] * w2[0] : wn_dot_lds<K2>(w2, zs + kq2 * K2, 0.f);
                a2 = wn_reduce<T2>(a2);
                if (kq2 == 0) wn_publish_at(p.gx + ((size_t)cx.w * ns + s) * R + row2, tag, (a2 + bres) + xres, local_x);
            }
            wn_stamp(r, park, item, 2);
            // ---- 4. skip 1x1 partial on this lane of the running skip sum          (wavenet_model.py:154-162)
            wn_u64* gs = p.gs + ((size_t)cx.w * ns + s) * S;
            if (!prime) {
                const wn_u64* gin = p.gs + (((size_t)(l - 1) * P + c) * ns + s) * S + tid;  // only read when l > 0
                wn_u64 sv[RS];
                if (l > 0) {  // (requested only now: the compiler joins every outstanding load at the top of the gated
                              //  unit with s_waitcnt vmcnt(0), an earlier request would stall the critical path)
#pragma unroll
                    for (int q = 0; q < RS; ++q) sv[q] = wn_ld_granule(gin + 256 * q);
                }
                float a3[RS];
#pragma unroll
                for (int q = 0; q < RS; ++q) a3[q] = bskip[q];
#pragma unroll
                for (int k = 0; k < DC; ++k) {
                    const float zk = zs[k];
#pragma unroll
                    for (int q = 0; q < RS; ++q) a3[q] += w3[q][k] * zk;
                }
#pragma unroll
                for (int q = 0; q < RS; ++q) {
                    if (l > 0) {
                        float v;
                        if ((uint32_t)(sv[q] >> 32) == tag) v = __uint_as_float((uint32_t)sv[q]);
                        else v = wn_poll_fixed<1>(cx, gin + 256 * q, 0, tag, WN_W_SKIN, e, s);
                        a3[q] += v;
                    }
                    wn_publish_at(gs + tid + 256 * q, tag, a3[q], local_s);
                }
            } else if (l == NL - 1) {
#pragma unroll
                for (int q = 0; q < RS; ++q) wn_publish_at(gs + tid + 256 * q, tag, 0.f, local_s);
            }
            // ---- 5. queue push (wavenet_modules.py:55-57) and the next step's tap 0 on x[t+1-d]
            {
                float* ring = p.rings + p.ring_off[l] + ((size_t)c * ns + s) * (size_t)ML * R;
                if (tid < R) ring[(size_t)tmod * R + tid] = xb[SH::xpad(tid)];
                float a0 = kq1 == 0 ? bfg : 0.f;
                if (d == 1) {
                    a0 = wn_dot_lds<K1>(w0, xb + kq1 * (K1 + 4), a0);
                } else {
                    const float* xo = ring + (size_t)tapmod * R + kq1 * K1;
#pragma unroll
                    for (int k = 0; k < K1; ++k) a0 += w0[k] * xo[k];
                }
                pre[s * 256 + tid] = a0;
            }
            wn_stamp(r, park, item, 3);
            wn_stamp_flush(r, park, cx.w, item);
        }
    }
}

template <class SH>
static __device__ void wn_v2_head(const WnPlan& p, const WnRun& r, WnCtx& cx, float* lds, int h) {
    constexpr int S = SH::S, EC = SH::EC, T3 = SH::T3, K3 = SH::K3, QS = S / 256;
    using L = WnV2Lds<SH>;
    const int tid = threadIdx.x, ns = p.n_streams, P = p.P, NL = p.NL;
    float w4[K3], w5[EC];
    const float* img = p.blobs + (size_t)NL * P * (SH::NWL * 256) + (size_t)h * (SH::NWH * 256) + tid;
#pragma unroll
    for (int k = 0; k < K3; ++k) w4[k] = img[(size_t)k * 256];
#pragma unroll
    for (int k = 0; k < EC; ++k) w5[k] = img[(size_t)(K3 + k) * 256];
    const float b1 = img[(size_t)(K3 + EC) * 256], b2 = img[(size_t)(K3 + EC + 1) * 256];
    const int kq3 = tid % T3, row3 = tid / T3;
    float* sk = lds + L::sk;
    float* ev = lds + L::ev;
    int* failflag = reinterpret_cast<int*>(lds + L::smp + 48);
    long long* park = reinterpret_cast<long long*>(lds + L::park);
    int* locflags = reinterpret_cast<int*>(lds + L::smp + 52);
    if (tid == 0) {
        *failflag = 0;
        const int mine = wn_xcc_id();
        __hip_atomic_store(p.xcc_tab + cx.w, (unsigned)(mine + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        locflags[0] = p.allow_plain ? (int)wn_same_xcd(cx, mine, 0, P) : 0;  // partial logits feed every slice of layer 0
    }
    __syncthreads();
    const bool local_l = locflags[0] != 0;
    long long last_wait = 0;
    for (long long e = 0; e < r.n_eval; ++e) {
        const bool prime = e < r.n_given - 1;
        const uint32_t tag = (uint32_t)(e + 1);
        for (int s = 0; s < ns; ++s) {
            const long long t_begin = (long long)wall_clock64();
            cx.t_start = t_begin;
            const long long item = e * ns + s;
            wn_stamp(r, park, item, 0);
            if (ns == 1) wn_presleep(last_wait);
            // the P lanes of the running skip sum, all rows of this thread in flight together
            const wn_u64* gin = p.gs + (((size_t)(NL - 1) * P) * ns + s) * S + tid;
            if (P == 4) {
                // first pass: everything at once; rows whose granules were late are re-polled individually
                wn_u64 v[QS][4];
#pragma unroll
                for (int q = 0; q < QS; ++q)
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) v[q][cc] = wn_ld_granule(gin + 256 * q + (size_t)cc * ns * S);
#pragma unroll
                for (int q = 0; q < QS; ++q) {
                    bool ok = true;
                    float sum = 0.f;
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) { ok = ok && ((uint32_t)(v[q][cc] >> 32) == tag); sum += __uint_as_float((uint32_t)v[q][cc]); }
                    if (!ok) sum = wn_poll_fixed<4>(cx, gin + 256 * q, (size_t)ns * S, tag, WN_W_HEAD, e, s);
                    sk[SH::skpad(tid + 256 * q)] = sum > 0.f ? sum : 0.f;  // relu(skip)  wavenet_model.py:167
                }
            } else {
#pragma unroll
                for (int q = 0; q < QS; ++q) {
                    const float v = wn_poll_sum<8>(cx, gin + 256 * q, (size_t)ns * S, P, tag, WN_W_HEAD, e, s);
                    sk[SH::skpad(tid + 256 * q)] = v > 0.f ? v : 0.f;
                }
            }
            if (wn_barrier_failed(cx, failflag)) return;
            last_wait = (long long)wall_clock64() - t_begin;
            wn_stamp(r, park, item, 1);
            wn_u64* gl = p.gl + ((size_t)h * ns + s) * 256;
            if (!prime) {
                float a = wn_dot_lds<K3>(w4, sk + kq3 * (K3 + 4), 0.f);
                a = wn_reduce<T3>(a);
                if (kq3 == 0) {
                    const float v = a + b1;
                    ev[row3] = v > 0.f ? v : 0.f;  // relu(end_conv_1)  :168
                }
                wn_lds_barrier();
                wn_publish_at(gl + tid, tag, wn_dot_lds<EC>(w5, ev, b2), local_l);  // partial end_conv_2  :169
            } else {
                wn_publish_at(gl + tid, tag, 0.f, local_l);
            }
            wn_stamp(r, park, item, 2);
            wn_lds_barrier();
            wn_stamp(r, park, item, 3);
            wn_stamp_flush(r, park, cx.w, item);
        }
    }
}


// ================================================================================================
// Multi-stream variant (n_streams >= 2): the same chain run as a PIPELINE over streams.  Per (step, stream) item a
// workgroup must not pay any memory round trip, so everything it will need for the NEXT item is requested while
// it works on the current one and only checked (tags) when due:
//   * the x' partials / skip lane / skip lanes of the next item are loaded into registers one item ahead; a stale
//     tag falls back to the polling loop (start-up, pipeline bubbles);
//   * the queue tap x[t+1-d] of the current stream is requested at the top of the item and consumed in its tail;
//   * sampling is moved off L0 onto NSMP dedicated sampler workgroups (chain positions after the head): they turn
//     partial logits into a class index per stream and publish it as an index granule gi[s]; L0 only gathers.
// LDS layout (floats) of the multi-stream kernel: G streams are processed per pipeline item
template <class SH, int G, bool W0L = false>
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
    // tap-0 weights (used off the critical path, in the tail) can live in LDS instead of registers: [K1/4][256] float4 --
    // the layer role then fits 256 VGPRs, so that two workgroups (of two independent chains) can share a CU
    static constexpr bool w0_lds = W0L;
    static_assert(!W0L || SH::K1 % 4 == 0, "tap-0 weights in LDS are stored as float4");
    static constexpr int w0_floats = w0_lds ? SH::K1 * 256 : 0;
    static __host__ __device__ int floats(int n_streams) { return pre + n_streams * 256 + w0_floats; }
};

// dot of one weight vector held in LDS as float4s strided by 256 lanes with G LDS vectors
template <int K, int G>
static __device__ __forceinline__ void wn_dot_lds_w(const float* w4, const float* x, int xstride, const float (&init)[G], float (&out)[G]) {
    float a[G][4];
#pragma unroll
    for (int g = 0; g < G; ++g) { a[g][0] = init[g]; a[g][1] = a[g][2] = a[g][3] = 0.f; }
#pragma unroll
    for (int k = 0; k < K / 4; ++k) {
        const float4 w = *reinterpret_cast<const float4*>(w4 + k * 1024);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float4 v = reinterpret_cast<const float4*>(x + g * xstride)[k];
            a[g][0] += w.x * v.x; a[g][1] += w.y * v.y; a[g][2] += w.z * v.z; a[g][3] += w.w * v.w;
        }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) out[g] = (a[g][0] + a[g][1]) + (a[g][2] + a[g][3]);
}

// dot of one register weight vector with G LDS vectors (G items share every weight operand)
template <int K, int G>
static __device__ __forceinline__ void wn_dot_lds_g(const float (&w)[K], const float* x, int xstride, const float (&init)[G], float (&out)[G]) {
    static_assert(K % 4 == 0 || G >= 1, "");
    if constexpr (K % 4 == 0) {
        float4 v[G][K / 4];
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int k = 0; k < K / 4; ++k) v[g][k] = reinterpret_cast<const float4*>(x + g * xstride)[k];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float a0 = init[g], a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
            for (int k = 0; k < K / 4; ++k) {
                a0 += w[4 * k] * v[g][k].x; a1 += w[4 * k + 1] * v[g][k].y; a2 += w[4 * k + 2] * v[g][k].z; a3 += w[4 * k + 3] * v[g][k].w;
            }
            out[g] = (a0 + a1) + (a2 + a3);
        }
    } else {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float a = init[g];
#pragma unroll
            for (int k = 0; k < K; ++k) a += w[k] * x[g * xstride + k];
            out[g] = a;
        }
    }
}

template <class SH, int P, int G, bool W0LDS>
static __device__ void wn_v2_layer_multi(const WnPlan& p, const WnRun& r, WnCtx& cx, float* lds, int l, int c) {
    constexpr int R = SH::R, DC = SH::DC, S = SH::S, T1 = SH::T1, K1 = SH::K1, T2 = SH::T2, K2 = SH::K2, RS = SH::RS;
    using L = WnV2LdsM<SH, G, W0LDS>;
    const int tid = threadIdx.x;
    const int ns = p.n_streams, NL = p.NL, ni = ns / G;  // ni items (of G streams) per evaluation
    constexpr bool W0L = L::w0_lds;
    float w1[K1], w0[W0L ? 1 : K1], w2[K2], w3[RS][DC], bskip[RS];
    float* w0s = lds + L::pre + p.n_streams * 256 + tid * 4;  // W0L: this lane's float4 k of tap 0 is w0s[k * 1024 .. +3]
    const float* img = p.blobs + (size_t)cx.w * (SH::NWL * 256) + tid;
    {
        int j = 0;
#pragma unroll
        for (int k = 0; k < K1; ++k) w1[k] = img[(size_t)(j++) * 256];
#pragma unroll
        for (int k = 0; k < K1; ++k) {
            const float v = img[(size_t)(j++) * 256];
            if constexpr (W0L) w0s[(k / 4) * 1024 + (k % 4)] = v;
            else w0[k] = v;
        }
#pragma unroll
        for (int k = 0; k < K2; ++k) w2[k] = img[(size_t)(j++) * 256];
#pragma unroll
        for (int q = 0; q < RS; ++q)
#pragma unroll
            for (int k = 0; k < DC; ++k) w3[q][k] = img[(size_t)(j++) * 256];
    }
    const float bfg = img[(size_t)(2 * K1 + K2 + RS * DC) * 256];
    const float bres = img[(size_t)(2 * K1 + K2 + RS * DC + 1) * 256];
#pragma unroll
    for (int q = 0; q < RS; ++q) bskip[q] = img[(size_t)(2 * K1 + K2 + RS * DC + 2 + q) * 256];

    const int kq1 = tid % T1, grp = tid / T1, ch = grp >> 1, is_gate = grp & 1;
    const int kq2 = tid % T2, row2 = tid / T2;
    const int d = p.dil[l];
    const int ML = d + 1;
    float* xs = lds + L::xs;
    float* zs = lds + L::zs;
    float* xol = lds + L::xo;
    float* pre = lds + L::pre;
    float* smp = lds + L::smp;
    int* failflag = reinterpret_cast<int*>(smp + 48);
    int* locflags = reinterpret_cast<int*>(smp + 52);
    long long* park = reinterpret_cast<long long*>(lds + L::park);
    if (tid == 0) {
        *failflag = 0;
        const int mine = wn_xcc_id();
        __hip_atomic_store(p.xcc_tab + cx.w, (unsigned)(mine + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int lx = 0, lsk = 0;
        if (p.allow_plain) {
            if (l < NL - 1) {
                lx = wn_same_xcd(cx, mine, (l + 1) * P, P);
                lsk = wn_same_xcd(cx, mine, (l + 1) * P + c, 1);
            } else {
                lsk = wn_same_xcd(cx, mine, NL * P, p.PA);
            }
        }
        locflags[0] = lx; locflags[1] = lsk;
    }
    __syncthreads();
    const bool local_x = locflags[0] != 0, local_s = locflags[1] != 0;

    for (int s = 0; s < ns; ++s) {  // tap 0 of the first evaluation of every stream
        const float* ring = p.rings + p.ring_off[l] + ((size_t)c * ns + s) * (size_t)ML * R;
        long long pos = (r.t_base - d) % ML;
        if (pos < 0) pos += ML;
        float acc = kq1 == 0 ? bfg : 0.f;
#pragma unroll
        for (int k = 0; k < K1; ++k) acc += (W0L ? w0s[(k / 4) * 1024 + (k % 4)] : w0[W0L ? 0 : k]) * ring[(size_t)pos * R + kq1 * K1 + k];
        pre[s * 256 + tid] = acc;
    }
    __syncthreads();

    // One-item-ahead request registers.  The request code is branch-free with a compile-time load count: a load
    // destination that is merged across control flow gets copied, and the copy waits for the load.
    wn_u64 nx[G][P];  // l > 0, tid < R: the P x' partials of each stream of the next item;  l == 0: the index granules
    const wn_u64* xbase = l == 0 ? p.gi : p.gx + ((size_t)(l - 1) * P) * ns * R + (tid < R ? tid : 0);
    const size_t xstep_s = l == 0 ? 1 : R, xstep_j = l == 0 ? 0 : (size_t)ns * R;
    const wn_u64* sbase = p.gs + (((size_t)(l > 0 ? l - 1 : 0) * P + c) * ns) * S + tid;
    auto request = [&](int it2) {  // issue the loads for item it2 (streams it2*G .. it2*G+G-1)
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int j = 0; j < P; ++j) nx[g][j] = wn_ld_granule(xbase + (size_t)(it2 * G + g) * xstep_s + (size_t)j * xstep_j);
    };
    request(0);

    int buf = 0;
    long long misses = 0;
    int tmod = (int)(r.t_base % ML);  // queue slot of x[t], kept incrementally
    for (long long e = 0; e < r.n_eval; ++e, tmod = (tmod + 1 == ML) ? 0 : tmod + 1) {
        const bool prime = e < r.n_given - 1;
        const uint32_t tag = (uint32_t)(e + 1);
        const int tapmod = tmod + 2 >= ML ? tmod + 2 - ML : tmod + 2;  // slot of x[t+1-d]
        for (int it = 0; it < ni; ++it, buf ^= 1) {
            const int s0 = it * G;
            float* xb = xs + buf * (G * L::XR);  // [G][XR]
            cx.t_start = (long long)wall_clock64();
            const long long item = e * ni + it;
            wn_stamp(r, park, item, 0);
            float* ring0 = p.rings + p.ring_off[l] + ((size_t)c * ns + s0) * (size_t)ML * R;  // rings of consecutive streams are ML*R apart
            // ---- 1. layer inputs x[t] of the G streams from the registers requested one item ago
            if (l == 0) {
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const int s = s0 + g;
                    int idx;
                    if (e == 0) {
                        idx = r.first[(size_t)s * r.n_given];
                    } else {
                        wn_u64 gv = nx[g][0];
                        if ((uint32_t)(gv >> 32) != (uint32_t)e) {
                            unsigned spins = 0;
                            while ((uint32_t)((gv = wn_ld_granule(p.gi + s)) >> 32) != (uint32_t)e) {
                                if ((++spins & 127u) == 0u) {
                                    if (__hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { cx.fail = 1; break; }
                                    if ((long long)wall_clock64() - cx.t_start > r.timeout_ticks) { wn_give_up(cx, WN_W_LOGITS, e, s); break; }
                                }
                            }
                        }
                        idx = (int)(uint32_t)gv & 255;
                    }
                    if (tid < R) xb[g * L::XR + SH::xpad(tid)] = p.start_t[(size_t)idx * R + tid] + (p.start_b ? p.start_b[tid] : 0.f);
                }
            } else if (tid < R) {
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    bool ok = true;
                    float sum = 0.f;
#pragma unroll
                    for (int j = 0; j < P; ++j) { ok = ok && ((uint32_t)(nx[g][j] >> 32) == tag); sum += __uint_as_float((uint32_t)nx[g][j]); }
                    if (!ok) {
                        sum = wn_poll_fixed<P, WN_MULTI_SLEEP>(cx, p.gx + (((size_t)(l - 1) * P) * ns + s0 + g) * R + tid, (size_t)ns * R, tag, WN_W_X, e, s0 + g);
                        if (tid == 0) ++misses;
                    }
                    xb[g * L::XR + SH::xpad(tid)] = sum;
                }
            }
            wn_stamp(r, park, item, 4);
            // this item's skip lanes: the upstream published them a little after the x' we just consumed, so loads
            // issued now land while the gated units compute and are checked in the tail
            wn_u64 sk_now[G][RS];
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int q = 0; q < RS; ++q) sk_now[g][q] = wn_ld_granule(sbase + (size_t)(s0 + g) * S + 256 * q);
            // queue taps x[t+1-d], consumed in the tail: ONE coalesced row load per stream by the first R lanes.
            // Requested only now: vector loads return in order, an HBM miss ahead of the polls would stall every poll.
            float xo_v[G];
#pragma unroll
            for (int g = 0; g < G; ++g) xo_v[g] = (d != 1 && tid < R) ? ring0[((size_t)g * ML + tapmod) * R + tid] : 0.f;
            if (wn_barrier_failed(cx, failflag)) return;
            wn_stamp(r, park, item, 1);
            // ---- 2. filter/gate of the G streams: every weight operand is used G times
            float xres[G], pin[G], acc[G];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                xres[g] = (c == 0 && kq2 == 0) ? xb[g * L::XR + SH::xpad(row2)] : 0.f;
                pin[g] = pre[(s0 + g) * 256 + tid];
            }
            wn_dot_lds_g<K1, G>(w1, xb + kq1 * (K1 + 4), L::XR, pin, acc);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const float a = wn_reduce<T1>(acc[g]);
                const float other = wn_partner<T1>(a);
                const float fv = is_gate ? other : a, gv = is_gate ? a : other;
                const float z = wn_gate(fv, gv);
                if (!is_gate && kq1 == 0) zs[g * L::DCP + ch] = z;
            }
            wn_lds_barrier();
            // ---- 3. residual partials
            if (l < NL - 1) {
                float zero[G], a2[G];
#pragma unroll
                for (int g = 0; g < G; ++g) zero[g] = 0.f;
                wn_dot_lds_g<K2, G>(w2, zs + kq2 * K2, L::DCP, zero, a2);
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const float v = wn_reduce<T2>(a2[g]);
                    if (kq2 == 0) wn_publish_at(p.gx + ((size_t)cx.w * ns + s0 + g) * R + row2, tag, (v + bres) + xres[g], local_x);
                }
            }
            wn_stamp(r, park, item, 2);
            // Request the NEXT item's x' partials only now: in the stage-bound steady state the upstream slice starts
            // that item less than one gated unit before us, so a request issued at staging time would come back stale
            // (measured: 71 % misses) and put a full poll round trip on the next item's critical path.
            request(it + 1 < ni ? it + 1 : 0);
            // ---- 4. skip partials on this lane of the running skip sums
            if (!prime) {
                float a3[G][RS];
#pragma unroll
                for (int g = 0; g < G; ++g)
#pragma unroll
                    for (int q = 0; q < RS; ++q) a3[g][q] = bskip[q];
#pragma unroll
                for (int k = 0; k < DC; ++k) {
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        const float zk = zs[g * L::DCP + k];
#pragma unroll
                        for (int q = 0; q < RS; ++q) a3[g][q] += w3[q][k] * zk;
                    }
                }
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    wn_u64* gs = p.gs + ((size_t)cx.w * ns + s0 + g) * S;
#pragma unroll
                    for (int q = 0; q < RS; ++q) {
                        if (l > 0) {
                            float v;
                            if ((uint32_t)(sk_now[g][q] >> 32) == tag) v = __uint_as_float((uint32_t)sk_now[g][q]);
                            else v = wn_poll_fixed<1>(cx, p.gs + (((size_t)(l - 1) * P + c) * ns + s0 + g) * S + tid + 256 * q, 0, tag, WN_W_SKIN, e, s0 + g);
                            a3[g][q] += v;
                        }
                        wn_publish_at(gs + tid + 256 * q, tag, a3[g][q], local_s);
                    }
                }
            } else if (l == NL - 1) {
#pragma unroll
                for (int g = 0; g < G; ++g)
#pragma unroll
                    for (int q = 0; q < RS; ++q) wn_publish_at(p.gs + ((size_t)cx.w * ns + s0 + g) * S + tid + 256 * q, tag, 0.f, local_s);
            }
            // ---- 5. queue pushes and the next step's tap 0
            {
                float a0i[G], a0[G];
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    if (tid < R) ring0[((size_t)g * ML + tmod) * R + tid] = xb[g * L::XR + SH::xpad(tid)];
                    a0i[g] = kq1 == 0 ? bfg : 0.f;
                }
                const float* xsrc = xb;
                if (d != 1) {
#pragma unroll
                    for (int g = 0; g < G; ++g)
                        if (tid < R) xol[g * L::XR + SH::xpad(tid)] = xo_v[g];
                    wn_lds_barrier();
                    xsrc = xol;
                }
                if constexpr (W0L) wn_dot_lds_w<K1, G>(w0s, xsrc + kq1 * (K1 + 4), L::XR, a0i, a0);
                else wn_dot_lds_g<K1, G>(w0, xsrc + kq1 * (K1 + 4), L::XR, a0i, a0);
#pragma unroll
                for (int g = 0; g < G; ++g) pre[(s0 + g) * 256 + tid] = a0[g];
            }
            wn_stamp(r, park, item, 3);
            if (r.prof && tid == 0) park[5] = misses;
            wn_stamp_flush(r, park, cx.w, item);
        }
    }
}

template <class SH, int P, int G>
static __device__ void wn_v2_head_multi(const WnPlan& p, const WnRun& r, WnCtx& cx, float* lds, int h) {
    constexpr int S = SH::S, EC = SH::EC, T3 = SH::T3, K3 = SH::K3, QS = S / 256;
    using L = WnV2LdsM<SH, G>;
    const int tid = threadIdx.x, ns = p.n_streams, NL = p.NL, ni = ns / G;
    float w4[K3], w5[EC];
    const float* img = p.blobs + (size_t)NL * P * (SH::NWL * 256) + (size_t)h * (SH::NWH * 256) + tid;
#pragma unroll
    for (int k = 0; k < K3; ++k) w4[k] = img[(size_t)k * 256];
#pragma unroll
    for (int k = 0; k < EC; ++k) w5[k] = img[(size_t)(K3 + k) * 256];
    const float b1 = img[(size_t)(K3 + EC) * 256], b2 = img[(size_t)(K3 + EC + 1) * 256];
    const int kq3 = tid % T3, row3 = tid / T3;
    float* sk = lds + L::sk;
    float* ev = lds + L::ev;
    int* failflag = reinterpret_cast<int*>(lds + L::smp + 48);
    int* locflags = reinterpret_cast<int*>(lds + L::smp + 52);
    long long* park = reinterpret_cast<long long*>(lds + L::park);
    if (tid == 0) {
        *failflag = 0;
        const int mine = wn_xcc_id();
        __hip_atomic_store(p.xcc_tab + cx.w, (unsigned)(mine + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        locflags[0] = p.allow_plain ? (int)wn_same_xcd(cx, mine, NL * P + p.PA, p.n_smp) : 0;  // logits feed the samplers
    }
    __syncthreads();
    const bool local_l = locflags[0] != 0;
    wn_u64 ng[G][QS][P];
    const wn_u64* gbase = p.gs + ((size_t)(NL - 1) * P) * ns * S + tid;
    auto request = [&](int it2) {  // branch-free, compile-time count (see wn_v2_layer_multi)
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int q = 0; q < QS; ++q)
#pragma unroll
                for (int j = 0; j < P; ++j) ng[g][q][j] = wn_ld_granule(gbase + (size_t)(it2 * G + g) * S + 256 * q + (size_t)j * ns * S);
    };
    request(0);
    for (long long e = 0; e < r.n_eval; ++e) {
        const bool prime = e < r.n_given - 1;
        const uint32_t tag = (uint32_t)(e + 1);
        for (int it = 0; it < ni; ++it) {
            const int s0 = it * G;
            cx.t_start = (long long)wall_clock64();
            const long long item = e * ni + it;
            wn_stamp(r, park, item, 0);
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int q = 0; q < QS; ++q) {
                    bool ok = true;
                    float sum = 0.f;
#pragma unroll
                    for (int j = 0; j < P; ++j) { ok = ok && ((uint32_t)(ng[g][q][j] >> 32) == tag); sum += __uint_as_float((uint32_t)ng[g][q][j]); }
                    if (!ok) sum = wn_poll_fixed<P>(cx, p.gs + (((size_t)(NL - 1) * P) * ns + s0 + g) * S + tid + 256 * q, (size_t)ns * S, tag, WN_W_HEAD, e, s0 + g);
                    sk[g * L::SKP + SH::skpad(tid + 256 * q)] = sum > 0.f ? sum : 0.f;
                }
            request(it + 1 < ni ? it + 1 : 0);
            if (wn_barrier_failed(cx, failflag)) return;
            wn_stamp(r, park, item, 1);
            if (!prime) {
                float zero[G], a[G];
#pragma unroll
                for (int g = 0; g < G; ++g) zero[g] = 0.f;
                wn_dot_lds_g<K3, G>(w4, sk + kq3 * (K3 + 4), L::SKP, zero, a);
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const float v = wn_reduce<T3>(a[g]) + b1;
                    if (kq3 == 0) ev[g * EC + row3] = v > 0.f ? v : 0.f;
                }
                wn_lds_barrier();
                float bb[G], o[G];
#pragma unroll
                for (int g = 0; g < G; ++g) bb[g] = b2;
                wn_dot_lds_g<EC, G>(w5, ev, EC, bb, o);
#pragma unroll
                for (int g = 0; g < G; ++g) wn_publish_at(p.gl + ((size_t)h * ns + s0 + g) * 256 + tid, tag, o[g], local_l);
            } else {
#pragma unroll
                for (int g = 0; g < G; ++g) wn_publish_at(p.gl + ((size_t)h * ns + s0 + g) * 256 + tid, tag, 0.f, local_l);
            }
            wn_stamp(r, park, item, 2);
            wn_lds_barrier();
            wn_stamp(r, park, item, 3);
            wn_stamp_flush(r, park, cx.w, item);
        }
    }
}

// Sampler workgroup j of n_smp (multi-stream only): for its streams, turns the head's partial logits of evaluation
// e-1 into the class index that enters evaluation e (teacher forced while priming) and publishes it as gi[s].
static __device__ void wn_v2_sampler(const WnPlan& p, const WnRun& r, WnCtx& cx, float* lds_smp, int j) {
    const int tid = threadIdx.x, ns = p.n_streams;
    int* failflag = reinterpret_cast<int*>(lds_smp + 48);
    int* locflags = reinterpret_cast<int*>(lds_smp + 52);
    if (tid == 0) {
        *failflag = 0;
        const int mine = wn_xcc_id();
        __hip_atomic_store(p.xcc_tab + cx.w, (unsigned)(mine + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        locflags[0] = p.allow_plain ? (int)wn_same_xcd(cx, mine, 0, p.P) : 0;  // the index feeds every slice of layer 0
    }
    __syncthreads();
    const bool local_i = locflags[0] != 0;
    long long* park = reinterpret_cast<long long*>(lds_smp + 64);  // the 8 stamp slots behind the sampler scratch (both LDS layouts)
    for (long long e = 1; e <= r.n_eval; ++e) {
        for (int s = j; s < ns; s += p.n_smp) {
            const long long item = (e - 1) * ns + s;  // stamps (diagnostics): 0 start of the wait, 1 logits complete, 2 index published
            wn_stamp(r, park, item, 0);
            const float logit = wn_poll_sum<16>(cx, p.gl + (size_t)s * 256 + tid, (size_t)ns * 256, p.PA, (uint32_t)e, WN_W_LOGITS, e, s);
            if (wn_barrier_failed(cx, failflag)) return;
            wn_stamp(r, park, item, 1);
            int idx;
            if (e < r.n_given) {
                idx = r.first[(size_t)s * r.n_given + e];
            } else {
                const long long g = e - r.n_given;
                if (r.dbg_logits) r.dbg_logits[((size_t)s * r.num_samples + g) * 256 + tid] = logit;
                const float temp = r.stream_temps ? r.stream_temps[s] : r.temperature;
                const bool greedy = r.greedy != 0 || !(temp > 0.f);
                const double u = greedy ? 0. : r.uniforms[(size_t)s * r.num_samples + g];
                idx = wn_sample_v2(cx, lds_smp, logit, u, greedy, temp);
                if (tid == 0) r.out_idx[(size_t)s * r.num_samples + g] = idx;
            }
            if (e < r.n_eval && tid == 0) {
                const wn_u64 v = ((wn_u64)(uint32_t)e << 32) | (wn_u64)(uint32_t)idx;
                if (local_i) __hip_atomic_store(p.gi + s, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else __hip_atomic_store(p.gi + s, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            wn_stamp(r, park, item, 2);
            wn_stamp_flush(r, park, cx.w, item);
            wn_lds_barrier();
        }
    }
}

// Shapes whose layer role fits 256 VGPRs once the tap-0 weights live in LDS: two workgroups -- of two independent chains,
// see wn_runtime.hip -- can then share a CU (2 waves per SIMD) and fill each other's hand-off waits.
static constexpr bool wn_v2m_shareable(int R, int DC, int S, int EC) {
    return (R / (256 / (2 * DC))) % 4 == 0 && !(R == 128 && EC == 64) && S < 1024;
}

template <int R, int DC, int S, int EC, int P, int G, bool W0LDS>
__global__ __launch_bounds__(WN_THREADS, W0LDS ? 2 : 1) void wn_generate_kernel_v2m(WnPlan p, WnRun r) {
    using SH = WnV2Shape<R, DC, S, EC>;
    extern __shared__ __attribute__((aligned(16))) float wn_lds2m[];
    const int w = p.wg_map[blockIdx.x];
    if (w < 0) return;
    WnCtx cx;
    cx.p = &p; cx.r = &r; cx.lds = wn_lds2m; cx.w = w; cx.fail = 0;
    cx.t_start = (long long)wall_clock64();
    const int n_layer_wg = p.NL * p.P;
    if (w < n_layer_wg) wn_v2_layer_multi<SH, P, G, W0LDS>(p, r, cx, wn_lds2m, w / P, w % P);
    else if (w < n_layer_wg + p.PA) wn_v2_head_multi<SH, P, G>(p, r, cx, wn_lds2m, w - n_layer_wg);
    else wn_v2_sampler(p, r, cx, wn_lds2m + WnV2LdsM<SH, G>::smp, w - n_layer_wg - p.PA);
}

template <int R, int DC, int S, int EC>
__global__ __launch_bounds__(WN_THREADS) void wn_generate_kernel_v2(WnPlan p, WnRun r) {
    using SH = WnV2Shape<R, DC, S, EC>;
    extern __shared__ __attribute__((aligned(16))) float wn_lds2[];
    const int w = p.wg_map[blockIdx.x];
    if (w < 0) return;  // bystander block of the XCD-aligned placement
    WnCtx cx;
    cx.p = &p; cx.r = &r; cx.lds = wn_lds2; cx.w = w; cx.fail = 0;
    cx.t_start = (long long)wall_clock64();
    const int n_layer_wg = p.NL * p.P;
    if (w < n_layer_wg) wn_v2_layer<SH>(p, r, cx, wn_lds2, w / p.P, w % p.P);
    else wn_v2_head<SH>(p, r, cx, wn_lds2, w - n_layer_wg);
}

#endif  // WN_KERNEL_V2_H
