// wn_kernel_v4.h -- the generation chain for SMALL channel shapes and few streams: several consecutive layers per workgroup
// (gfx950, device only; kernel variant 4).
//
// Variant 3 (wn_kernel_v3.h) gives every layer a workgroup of its own: a token crosses NL + 2 workgroup boundaries per audio sample,
// each an L2 round trip of ~0.36-0.45 us, and for the small shapes of BASELINE configs[0] / [1] and of the reference's only trained
// model (train_script.py:17-25) that is all there is: cfg2 single stream 25.2 us per sample = 33 stages x 0.76 us, 2.5 % of the HBM
// roofline; cfg1 1.4 %.  A cfg2 layer is 147 KB of fp32 weights, a cfg1 layer 53 KB -- three to five of them fit ONE CU's register file.
// Here a "stack" workgroup of 512 threads holds LPW consecutive layers: the layer-to-layer hand-off inside it is an LDS write, one
// LDS-only barrier and an LDS read (wavenet_model.py:131-165 as one instruction stream); only every LPW-th boundary is a hand-off
// through L2.  Same protocol as variant 3 with an unsplit stack (P = 1): x' granules gx[w][stream][R] between stack workgroups, the
// running skip sum as ONE lane gs[w][stream][S] that every stack workgroup adds its layers' skip convs to (16-byte pairs, the
// layout of wn_kernel_v3.h), head and sampler workgroups are variant 3's own roles (wn_v3_head<SH, 1>, wn_v3_sampler): the last
// stack workgroup's lane feeds the heads, the samplers feed stack workgroup 0.
//
// Per item (evaluation e, stream s) of a stack workgroup holding the layers l0 .. l0 + nl - 1 -- all 512 lanes take part in every phase:
//   input      lanes t < R poll the upstream's x' granules (workgroup 0: the sampler's start_conv row), stage x            [barrier]
//   per layer  FG:  lane (channel c = t / 8, slice kq = t % 8): packed {filter, gate} FMAs of tap 1 on x[kq K .. ) + the parked
//                   tap-0 sums, 8-lane DPP reduction, tanh * sigmoid -> z                       (wavenet_model.py:147-151) [barrier]
//              RES: lane (row t / 8, slice kq): residual 1x1 on z + x[t]  -> the next layer's x, or -- last layer of the workgroup --
//                   published at once as x' granules (rows 2j, 2j+1 as one 16-byte store)       (:164-165)                 [barrier]
//   tail       (nobody downstream waits for it, except the head for the LAST workgroup's skip lane)
//              queue push of every layer's x[t], request of the taps x[t+1-d] of the next timestep (HBM rings in the reference's
//              DilatedQueue layout: wn_export_queue, wn_prime, continuation see the same state as with variant 3; d = 1: the tap is
//              x[t] itself)                                                                     (wavenet_modules.py:55-72)
//              skip 1x1 of every layer on its z (lane (row t / 2, half t % 2)), + the upstream lane, published (:154-162)
//              taps staged                                                                                                 [barrier]
//              tap-0 half of the dilated convs for the next timestep (weights from LDS: off the token's path), parked      [barrier]
// Weights: tap 1, residual and skip rows in registers (cfg2: 60 per layer and lane), tap 0 in LDS (float4 rows, lane-linear).
// Every spin is bounded like variant 3's (timeout per hand-off wait + the abort word).
#ifndef WN_KERNEL_V4_H
#define WN_KERNEL_V4_H

#include "wn_kernel_v3.h"

#ifndef WN_V4_PAIR_GATE
#define WN_V4_PAIR_GATE 2   // round 6: each lane of a pair evaluates ONE factor of the gated unit (one exp, one reciprocal) instead of both (see the FG window):
                            // 0 never, 1 always, 2 where the FG window fills all eight waves (D = 64: two waves per SIMD share the transcendental unit --
                            // cfg2 x 1 49.2 -> 50.9 k samples/s, x 4 197 -> 203 k; at D = 32 half the waves idle in that window and the select + DPP move are
                            // pure latency: cfg1 x 1 133 -> 126 k, the train_script shape 49.0 -> 47.7 k; same bits either way: profiles/r06_v4_pair_gate.txt)
#endif
#ifndef WN_THREADS_V4
#define WN_THREADS_V4 512   // (also wn_stacked_table.h: the host side plans with it)
#endif

template <int R_, int D_, int S_>
struct WnV4Shape {
    static constexpr int R = R_, D = D_, S = S_;
    static constexpr int TK = 8;           // lanes that share a filter/gate channel, or a residual row
    static constexpr int KF = R / TK;      // x elements per lane and tap
    static constexpr int KR = D / TK;      // z elements per lane of a residual row
    static constexpr int RS = S / 256;     // skip rows per row-lane (rows r + 256 q)
    static constexpr int KS = D / 2;       // z elements per lane of a skip row (two lanes per row-lane)
    static constexpr int XP = R + 4 * (R / 32), ZP = D + 4 * (D / 32);   // padded vectors: + 4 floats per 32 (16-byte reads of 8 slices hit 8 bank groups)
    static __host__ __device__ constexpr int xpad(int ch) { return ch + 4 * (ch >> 5); }
    // per-lane weight image of ONE layer (floats, striped over the 512 lanes: image[j * 512 + t]):
    //   w1 {f,g}[KF] | wr[KR] | ws[RS * KS] | bres | bskip[RS] | bf, bg | w0 {f,g}[KF]        (w0 goes to LDS, the rest to registers)
    static constexpr int O_W1 = 0, O_WR = 2 * KF, O_WS = O_WR + KR, O_BRES = O_WS + RS * KS, O_BSKIP = O_BRES + 1, O_B0 = O_BSKIP + RS,
                         O_W0 = O_B0 + 2, NWPL = O_W0 + 2 * KF;
    static_assert((R == 32 || R == 64) && (D == 32 || D == 64), "8 lanes per channel / row on 512 lanes; slices read as float4");
    static_assert(S % 256 == 0 && (RS == 1 || RS % 2 == 0), "skip rows come in pairs per row-lane, or one");
};

// LDS of a stack workgroup (floats)
template <class V, int LPW>
struct WnV4Lds {
    static constexpr int xl = 0;                              // [LPW + 1][XP]  the layers' inputs of the item (xl[0]: the workgroup's)
    static constexpr int zl = xl + (LPW + 1) * V::XP;         // [LPW][ZP]
    static constexpr int xo = zl + LPW * V::ZP;               // [LPW][XP]      taps x[t+1-d]
    static constexpr int misc = xo + LPW * V::XP;             // [0] fail flag, [1], [2] locality flags; [8 .. 24): eight parked int64 stamps
    static constexpr int w0 = (misc + 32 + 3) & ~3;           // float4 [LPW][KF / 2][512]  tap-0 weights
    static constexpr int pre = w0 + LPW * 2 * V::KF * WN_THREADS_V4;   // [n_streams][LPW][2 D]  parked tap-0 sums {f, g} per channel (+ bias)
    static __host__ __device__ int floats(int ns) { return pre + ns * LPW * 2 * V::D; }
};

// tanh(f) * sigmoid(g) = (2 s(2f) - 1) s(g), s(v) = 1 / (1 + e^-v).  e^-v is taken as exp2(-v log2 e) -- two operations on the token's path
// instead of the eight of wn_exp (variant 3 carries the product -v log2 e in two floats).  The single rounding of that product moves
// e^-v by |v| 6e-8 relative, and s damps it by s (1 - s) <= 1/4 and exponentially beyond |v| ~ 2: |ds| <= max |v| s (1 - s) 6e-8 = 1.4e-8,
// a quarter of an fp32 ulp of the result -- below the rounding of the sums that feed it (the parity tests' bars are unchanged).
static __device__ __forceinline__ float wn_v4_gate(float f, float g) {
    const float rf = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(f * -2.88539008177792681f));   // -2 log2(e)
    const float rg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(g * -1.44269504088896341f));
    return fmaf(2.0f, rf, -1.0f) * rg;
}
static __device__ __forceinline__ float wn_ld_sc1(const float* p) {   // served by the L2, never by this CU's L1
    return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

template <class V, int LPW>
static __device__ void wn_v4_stack(const WnPlan& p, const WnRun& r, WnCtx& cx, float* lds, int w) {
    constexpr int R = V::R, D = V::D, S = V::S, KF = V::KF, KR = V::KR, KS = V::KS, RS = V::RS, XP = V::XP, ZP = V::ZP;
    constexpr int NPL = RS / 2, ODD = RS % 2, NPR = NPL + ODD, NWS = RS * KS / 2;
    static_assert(LPW * R <= WN_THREADS_V4, "one lane per (layer, element) of the queue pushes");
    using L = WnV4Lds<V, LPW>;
    const int t = threadIdx.x, ns = p.n_streams, NL = p.NL;
    const int l0 = w * LPW;
    const int nl = NL - l0 < LPW ? NL - l0 : LPW;   // layers this workgroup holds
    const bool last_wg = l0 + nl == NL;
    const int g8 = t >> 3, kq = t & 7;              // filter/gate channel (and residual row), slice
    const bool fg_on = g8 < D, rs_on = g8 < R;
    const int c8 = fg_on ? g8 : 0, r8 = rs_on ? g8 : 0;
    const float gate_c = (kq & 1) ? -1.44269504088896341f : -2.88539008177792681f;   // -log2 e (gate factor), -2 log2 e (filter factor): WN_V4_PAIR_GATE
    const int sr = t >> 1, kh = t & 1;              // skip row-lane (rows sr + 256 q), half of z
    const int tl = t / R, tr = t % R;               // (layer, element) this lane pushes / taps
    float* xl = lds + L::xl;
    float* zl = lds + L::zl;
    float* xo = lds + L::xo;
    float* pre = lds + L::pre;
    int* failflag = reinterpret_cast<int*>(lds + L::misc);
    int* locflags = failflag + 1;
    long long* park = reinterpret_cast<long long*>(lds + L::misc + 8);
    // diagnostics (wn_profile_next): stamps of an item -- 0 start of the wait, 1 input staged, 2 first layer done, 3 x' published (layers
    // done), 4 skip lane published, 5 item done (tap-0 sums of the next timestep parked)
    auto stamp = [&](long long item, int k) {
        if (r.prof && item < r.prof_items && t == 0) park[k] = (long long)wall_clock64();
    };
    float4* w0l = reinterpret_cast<float4*>(lds + L::w0) + t;   // [(li * KF / 2 + k4) * 512]

    // ---- weights: registers (tap 1, residual, skip, biases), LDS (tap 0)
    const float* img = p.blobs + (size_t)w * ((size_t)LPW * V::NWPL * WN_THREADS_V4) + t;
    wn_f2 w1[LPW][KF], ws[LPW][NWS], b0[LPW];
    float wr[LPW][KR], bres[LPW], bskip[LPW][RS];
#pragma unroll
    for (int li = 0; li < LPW; ++li) {
        const float* im = img + (size_t)li * V::NWPL * WN_THREADS_V4;
#pragma unroll
        for (int k = 0; k < KF; ++k) w1[li][k] = wn_f2{im[(size_t)(V::O_W1 + 2 * k) * WN_THREADS_V4], im[(size_t)(V::O_W1 + 2 * k + 1) * WN_THREADS_V4]};
#pragma unroll
        for (int k = 0; k < KR; ++k) wr[li][k] = im[(size_t)(V::O_WR + k) * WN_THREADS_V4];
#pragma unroll
        for (int k = 0; k < NWS; ++k) ws[li][k] = wn_f2{im[(size_t)(V::O_WS + 2 * k) * WN_THREADS_V4], im[(size_t)(V::O_WS + 2 * k + 1) * WN_THREADS_V4]};
        bres[li] = im[(size_t)V::O_BRES * WN_THREADS_V4];
#pragma unroll
        for (int q = 0; q < RS; ++q) bskip[li][q] = im[(size_t)(V::O_BSKIP + q) * WN_THREADS_V4];
        b0[li] = wn_f2{im[(size_t)V::O_B0 * WN_THREADS_V4], im[(size_t)(V::O_B0 + 1) * WN_THREADS_V4]};
#pragma unroll
        for (int k4 = 0; k4 < KF / 2; ++k4)
            w0l[(size_t)(li * (KF / 2) + k4) * WN_THREADS_V4] = float4{im[(size_t)(V::O_W0 + 4 * k4) * WN_THREADS_V4], im[(size_t)(V::O_W0 + 4 * k4 + 1) * WN_THREADS_V4],
                                                                       im[(size_t)(V::O_W0 + 4 * k4 + 2) * WN_THREADS_V4], im[(size_t)(V::O_W0 + 4 * k4 + 3) * WN_THREADS_V4]};
    }
    // ---- per layer: dilation, ring, queue slot of x[t]
    int dil[LPW], ML[LPW], tmod[LPW];
    float* ring[LPW];
#pragma unroll
    for (int li = 0; li < LPW; ++li) {
        const int l = l0 + li < NL ? l0 + li : NL - 1;
        dil[li] = p.dil[l];
        ML[li] = dil[li] + 1;
        tmod[li] = (int)(r.t_base % ML[li]);
        ring[li] = p.rings + p.ring_off[l];   // P = 1: stream s at + s * ML * R
    }
    if (t == 0) {
        *failflag = 0;
        const int mine = wn_xcc_id();
        __hip_atomic_store(p.xcc_tab + cx.w, (unsigned)(mine + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int lx = 0, lsk = 0;
        if (p.allow_plain) {
            if (!last_wg) { lx = wn_same_xcd(cx, mine, w + 1, 1); lsk = lx; }
            else lsk = wn_same_xcd(cx, mine, p.n_lw, p.PA * p.HR);
        }
        locflags[0] = lx; locflags[1] = lsk;
    }
    __syncthreads();
    const bool local_x = locflags[0] != 0, local_s = locflags[1] != 0;
    const int n_prime = (int)(r.n_given - 1);
    const __amdgpu_buffer_rsrc_t rs_gx = wn_rsrc(p.gx), rs_gs = wn_rsrc(p.gs);

    // tap-0 sums of stream s for every layer from the staged taps xo, parked (+ bias) for the filter/gate window of the next timestep
    auto tap0_dots = [&](int s) {
#pragma unroll
        for (int li = 0; li < LPW; ++li) {
            if (li < nl) {
                const float4* x4 = reinterpret_cast<const float4*>(xo + li * XP + V::xpad(kq * KF));
                wn_f2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f};
#pragma unroll
                for (int k4 = 0; k4 < KF / 4; ++k4) {
                    const float4 xv = x4[k4];
                    const float4 wa = w0l[(size_t)(li * (KF / 2) + 2 * k4) * WN_THREADS_V4], wb = w0l[(size_t)(li * (KF / 2) + 2 * k4 + 1) * WN_THREADS_V4];
                    a0 = __builtin_elementwise_fma(wn_f2{wa.x, wa.y}, wn_f2{xv.x, xv.x}, a0);
                    a1 = __builtin_elementwise_fma(wn_f2{wa.z, wa.w}, wn_f2{xv.y, xv.y}, a1);
                    a0 = __builtin_elementwise_fma(wn_f2{wb.x, wb.y}, wn_f2{xv.z, xv.z}, a0);
                    a1 = __builtin_elementwise_fma(wn_f2{wb.z, wb.w}, wn_f2{xv.w, xv.w}, a1);
                }
                const float f = wn_reduce<8>(a0.x + a1.x) + b0[li].x, g = wn_reduce<8>(a0.y + a1.y) + b0[li].y;
                if (fg_on && kq == 0) *reinterpret_cast<wn_f2*>(pre + ((size_t)s * LPW + li) * 2 * D + 2 * c8) = wn_f2{f, g};
            }
        }
    };
    // ---- prologue: the first evaluation's tap x[t_base - d] sits in slot (t_base + 1) mod (d + 1) of the rings (zeros after a reset)
    for (int s = 0; s < ns; ++s) {
#pragma unroll
        for (int li = 0; li < LPW; ++li)
            if (tl == li && li < nl) {
                const int slot = tmod[li] + 1 == ML[li] ? 0 : tmod[li] + 1;
                xo[li * XP + V::xpad(tr)] = ring[li][((size_t)s * ML[li] + slot) * R + tr];
            }
        wn_lds_barrier();
        tap0_dots(s);
        wn_lds_barrier();
    }

    const wn_u64* gin = (w == 0 ? p.g0 : p.gx + (size_t)(w - 1) * ns * R) + (t < R ? t : 0);   // + s * R: this lane's input granule
    for (long long e = 0; e < r.n_eval; ++e) {
        const uint32_t tag = (uint32_t)(e + 1);
        const bool prime = e < n_prime;
        for (int s = 0; s < ns; ++s) {
            const long long item = e * ns + s;
            stamp(item, 0);
            // ---- input x[t] of the workgroup's first layer
            if (t < R) {
                float xin = 0.f;
                if (w == 0 && e == 0) {   // the first evaluation's input is a given sample: start_conv row gather (wavenet_model.py:127, 256-257)
                    const int idx = r.first[(size_t)s * r.n_given];
                    xin = p.start_t[(size_t)idx * R + t] + (p.start_b ? p.start_b[t] : 0.f);
                } else if (!cx.fail) {
                    const wn_u64* g = gin + (size_t)s * R;
                    bool ok = false;
                    unsigned spins = 0;
                    for (;;) {
                        if (!ok) {
                            const wn_u64 v = wn_peek(g);
                            if ((uint32_t)(v >> 32) == tag) { xin = wn_granule_value(v); ok = true; }
                        }
                        if (__builtin_amdgcn_ballot_w64(!ok) == 0) break;   // the polling lanes leave together
                        if ((++spins & 127u) == 0u) {
                            if (__hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { cx.fail = 1; break; }
                            const long long now = (long long)wall_clock64();
                            if (spins == 128u) cx.t_start = now;
                            else if (now - cx.t_start > r.timeout_ticks) { wn_give_up(cx, w == 0 ? WN_W_LOGITS : WN_W_X, e, s); break; }
                        }
                    }
                }
                xl[V::xpad(t)] = xin;
            }
            const int fail_in = wn_barrier_flag(cx, failflag);
            stamp(item, 1);
            // ---- the layers of this workgroup, one after the other            (wavenet_model.py:131-165)
#pragma unroll
            for (int li = 0; li < LPW; ++li) {
                if (li < nl) {
                    const float* x = xl + li * XP;
                    // filter/gate: tap 1 on x[t] + the parked tap-0 sums, tanh * sigmoid       (:147-151)
                    float4 xv[KF / 4];
#pragma unroll
                    for (int k4 = 0; k4 < KF / 4; ++k4) xv[k4] = reinterpret_cast<const float4*>(x + V::xpad(kq * KF))[k4];
                    const wn_f2 pf = *reinterpret_cast<const wn_f2*>(pre + ((size_t)s * LPW + li) * 2 * D + 2 * c8);
                    const float xres = x[V::xpad(r8)];
                    wn_f2 acc[4];   // four independent chains (KF = 8: two FMAs deep)
                    acc[0] = kq == 0 ? pf : wn_f2{0.f, 0.f};
                    acc[1] = acc[2] = acc[3] = wn_f2{0.f, 0.f};
#pragma unroll
                    for (int k4 = 0; k4 < KF / 4; ++k4) {
                        acc[0] = __builtin_elementwise_fma(w1[li][4 * k4], wn_f2{xv[k4].x, xv[k4].x}, acc[0]);
                        acc[1] = __builtin_elementwise_fma(w1[li][4 * k4 + 1], wn_f2{xv[k4].y, xv[k4].y}, acc[1]);
                        acc[2] = __builtin_elementwise_fma(w1[li][4 * k4 + 2], wn_f2{xv[k4].z, xv[k4].z}, acc[2]);
                        acc[3] = __builtin_elementwise_fma(w1[li][4 * k4 + 3], wn_f2{xv[k4].w, xv[k4].w}, acc[3]);
                    }
                    const wn_f2 asum = (acc[0] + acc[1]) + (acc[2] + acc[3]);
                    const float f = wn_reduce<8>(asum.x), g = wn_reduce<8>(asum.y);
                    float z;
                    if constexpr (WN_V4_PAIR_GATE == 1 || (WN_V4_PAIR_GATE == 2 && 8 * D >= WN_THREADS_V4)) {
                        // one exp and one reciprocal per lane: the even lane of a pair takes the filter factor 2 s(2 f) - 1, the odd lane the gate factor
                        // s(g), each gets the other's from its neighbour (one DPP move) -- the same two numbers multiplied as in wn_v4_gate: the same bits
                        const float rc = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(((kq & 1) ? g : f) * gate_c));
                        const float fac = (kq & 1) ? rc : fmaf(2.0f, rc, -1.0f);
                        z = fac * wn_partner<1>(fac);
                    } else {
                        z = wn_v4_gate(f, g);
                    }
                    if (fg_on && kq == 0) zl[li * ZP + V::xpad(c8)] = z;
                    if (li == 0 && fail_in) return;
                    wn_lds_barrier();
                    // residual 1x1 + x[t]                                                      (:164-165; the network's last layer has no consumer)
                    if (l0 + li < NL - 1) {
                        float4 zv[KR / 4];
#pragma unroll
                        for (int k4 = 0; k4 < KR / 4; ++k4) zv[k4] = reinterpret_cast<const float4*>(zl + li * ZP + V::xpad(kq * KR))[k4];
                        wn_f2 b = {0.f, 0.f}, b2 = {0.f, 0.f};
#pragma unroll
                        for (int k4 = 0; k4 < KR / 4; ++k4) {
                            b = __builtin_elementwise_fma(wn_f2{wr[li][4 * k4], wr[li][4 * k4 + 1]}, wn_f2{zv[k4].x, zv[k4].y}, b);
                            b2 = __builtin_elementwise_fma(wn_f2{wr[li][4 * k4 + 2], wr[li][4 * k4 + 3]}, wn_f2{zv[k4].z, zv[k4].w}, b2);
                        }
                        const float xn = (wn_reduce<8>((b.x + b.y) + (b2.x + b2.y)) + bres[li]) + xres;
                        if (li + 1 < nl) {
                            if (rs_on && kq == 0) xl[(li + 1) * XP + V::xpad(r8)] = xn;
                            wn_lds_barrier();
                        } else {
                            // the workgroup's output: rows 2j and 2j+1 sit on lanes 16j and 16j+8 -- one 16-byte store {x'(2j), tag, x'(2j+1), tag}
                            const float xn1 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(xn), 0x108, 0xf, 0xf, true));  // row_shl:8
                            if (rs_on && (t & 15) == 0) wn_st_pair(rs_gx, (unsigned)((((size_t)w * ns + s) * R + r8) * 8), tag, xn, xn1, local_x);
                        }
                    }
                    if (li == 0) stamp(item, 2);
                }
            }
            stamp(item, 3);
            // ---- tail.  Queue pushes and the taps of the next timestep first: their round trips run next to the skip dots.
            float tapv = 0.f;
#pragma unroll
            for (int li = 0; li < LPW; ++li)
                if (tl == li && li < nl) {
                    const float xv = xl[li * XP + V::xpad(tr)];
                    float* q = ring[li] + (size_t)s * ML[li] * R + tr;
                    q[(size_t)tmod[li] * R] = xv;                                                   // enqueue (wavenet_modules.py:55-57)
                    const int tap = tmod[li] + 2 >= ML[li] ? tmod[li] + 2 - ML[li] : tmod[li] + 2;  // slot of x[t+1-d]
                    tapv = dil[li] == 1 ? xv : wn_ld_sc1(q + (size_t)tap * R);                      // (pushed >= one timestep ago by this workgroup)
                }
            // skip 1x1 of every layer of the workgroup on the running lane               (:154-162)
            {
                const unsigned base_up = (unsigned)((((size_t)(w > 0 ? w - 1 : 0) * ns + s) * (size_t)S) * 8);
                const unsigned base_me = (unsigned)((((size_t)w * ns + s) * (size_t)S) * 8);
                constexpr unsigned ODD_BASE = 2048u * (unsigned)(RS - 1);
                const unsigned lane16 = (unsigned)sr * 16, odd_ld = ODD_BASE + (unsigned)(sr & ~1) * 8, odd_st = ODD_BASE + (unsigned)sr * 8;
                const bool work = !prime;
                wn_v4i up[NPR];
                if (work && w > 0) {   // (requested now, looked at after the dots: the upstream publishes its lane right after its x')
#pragma unroll
                    for (int h2 = 0; h2 < NPL; ++h2) up[h2] = wn_ld_pair(rs_gs, base_up + h2 * 4096 + lane16);
                    if constexpr (ODD) up[NPL] = wn_ld_pair(rs_gs, base_up + odd_ld);
                }
                float a3[RS];
#pragma unroll
                for (int q = 0; q < RS; ++q) a3[q] = 0.f;
                if (work) {
                    // (independent chains: a single accumulator over the workgroup's layers is one dependent chain of LPW KS / 2 packed FMAs --
                    //  on the token's path in the LAST workgroup, whose lane the head waits for)
                    wn_f2 ap[NPL > 0 ? NPL : 1][2], ao[4];
#pragma unroll
                    for (int h2 = 0; h2 < NPL; ++h2) ap[h2][0] = ap[h2][1] = wn_f2{0.f, 0.f};
#pragma unroll
                    for (int k = 0; k < 4; ++k) ao[k] = wn_f2{0.f, 0.f};
#pragma unroll
                    for (int li = 0; li < LPW; ++li) {
                        if (li < nl) {
                            const float4* z4 = reinterpret_cast<const float4*>(zl + li * ZP + V::xpad(kh * KS));
#pragma unroll
                            for (int c4 = 0; c4 < KS / 4; c4 += 4) {   // 16 z values at a time (KS = 16 or 32: never across a 32-float pad)
                                float4 zv[4];
#pragma unroll
                                for (int k = 0; k < 4; ++k) zv[k] = z4[c4 + k];
#pragma unroll
                                for (int k = 0; k < 4; ++k) {
                                    const int kk = 4 * (c4 + k);
                                    if constexpr (ODD) {   // one row: {w[2j], w[2j+1]} . {z[2j], z[2j+1]}
                                        ao[k] = __builtin_elementwise_fma(ws[li][kk / 2], wn_f2{zv[k].x, zv[k].y}, ao[k]);
                                        ao[k] = __builtin_elementwise_fma(ws[li][kk / 2 + 1], wn_f2{zv[k].z, zv[k].w}, ao[k]);
                                    }
#pragma unroll
                                    for (int h2 = 0; h2 < NPL; ++h2) {   // rows 2h, 2h+1 side by side: {w[2h][k], w[2h+1][k]} . z[k]
                                        ap[h2][0] = __builtin_elementwise_fma(ws[li][h2 * KS + kk], wn_f2{zv[k].x, zv[k].x}, ap[h2][0]);
                                        ap[h2][1] = __builtin_elementwise_fma(ws[li][h2 * KS + kk + 1], wn_f2{zv[k].y, zv[k].y}, ap[h2][1]);
                                        ap[h2][0] = __builtin_elementwise_fma(ws[li][h2 * KS + kk + 2], wn_f2{zv[k].z, zv[k].z}, ap[h2][0]);
                                        ap[h2][1] = __builtin_elementwise_fma(ws[li][h2 * KS + kk + 3], wn_f2{zv[k].w, zv[k].w}, ap[h2][1]);
                                    }
                                }
                            }
#pragma unroll
                            for (int h2 = 0; h2 < NPL; ++h2) ap[h2][0] += wn_f2{bskip[li][2 * h2], bskip[li][2 * h2 + 1]};
                            if constexpr (ODD) ao[0].x += bskip[li][RS - 1];
                        }
                    }
                    // the two halves of z (lanes 2 sr, 2 sr + 1), then the upstream lane
#pragma unroll
                    for (int h2 = 0; h2 < NPL; ++h2) {
                        const wn_f2 v = ap[h2][0] + ap[h2][1];
                        a3[2 * h2] = v.x + wn_partner<1>(v.x);
                        a3[2 * h2 + 1] = v.y + wn_partner<1>(v.y);
                    }
                    if constexpr (ODD) {
                        const wn_f2 v2 = (ao[0] + ao[1]) + (ao[2] + ao[3]);
                        const float v = v2.x + v2.y;
                        a3[RS - 1] = v + wn_partner<1>(v);
                    }
                    if (w > 0) {
#pragma unroll
                        for (int h2 = 0; h2 < NPL; ++h2) {
                            wn_v4i v = up[h2];
                            if ((uint32_t)v.y != tag || (uint32_t)v.w != tag) v = wn_poll_pair(cx, rs_gs, base_up + h2 * 4096 + lane16, tag, WN_W_SKIN, e, s);
                            a3[2 * h2] += __int_as_float(v.x);
                            a3[2 * h2 + 1] += __int_as_float(v.z);
                        }
                        if constexpr (ODD) {
                            wn_v4i v = up[NPL];
                            if ((uint32_t)v.y != tag || (uint32_t)v.w != tag) v = wn_poll_pair(cx, rs_gs, base_up + odd_ld, tag, WN_W_SKIN, e, s);
                            a3[RS - 1] += __int_as_float((sr & 1) ? v.z : v.x);
                        }
                    }
                }
                if (work || last_wg) {   // (priming: only the head's lane is kept moving, with zeros)
#pragma unroll
                    for (int h2 = 0; h2 < NPL; ++h2)
                        if (kh == 0) wn_st_pair(rs_gs, base_me + h2 * 4096 + lane16, tag, a3[2 * h2], a3[2 * h2 + 1], local_s);
                    if constexpr (ODD) {
                        const float nb = wn_dpp<0x4E>(a3[RS - 1]);   // quad_perm [2,3,0,1]: row sr + 1 lives two lanes up
                        if ((t & 3) == 0) wn_st_pair(rs_gs, base_me + odd_st, tag, a3[RS - 1], nb, local_s);
                    }
                }
            }
            stamp(item, 4);
            if (tl < nl) xo[tl * XP + V::xpad(tr)] = tapv;
            wn_lds_barrier();
            tap0_dots(s);   // the tap-0 half of the dilated convs of timestep t + 1
            wn_lds_barrier();
            stamp(item, 5);
            if (r.prof && item < r.prof_items && t == 0) {
                long long* dst = r.prof + ((size_t)cx.w * r.prof_items + item) * WN_STAMPS;
#pragma unroll
                for (int k = 0; k < 6; ++k) dst[k] = park[k];
            }
        }
#pragma unroll
        for (int li = 0; li < LPW; ++li) tmod[li] = tmod[li] + 1 == ML[li] ? 0 : tmod[li] + 1;
    }
}

template <int R, int D, int S, int EC, int LPW>
__global__ __launch_bounds__(WN_THREADS_V4) void wn_generate_kernel_v4(WnPlan p, WnRun r) {
    using V = WnV4Shape<R, D, S>;
    using SH = WnV2Shape<R, D, S, EC>;   // head / sampler roles: variant 3's, on an unsplit stack
    extern __shared__ __attribute__((aligned(16))) float wn_lds4[];
    const int w = p.wg_map[blockIdx.x];
    if (w < 0) return;
    WnCtx cx;
    cx.p = &p; cx.r = &r; cx.lds = wn_lds4; cx.w = w; cx.fail = 0;
    cx.t_start = (long long)wall_clock64();
    if (wn_not_resident(cx, wn_lds4)) return;   // (every workgroup of the job is resident from here on)
    if (w < p.n_lw) {
        wn_v4_stack<V, LPW>(p, r, cx, wn_lds4, w);
        return;
    }
    if (threadIdx.x >= WN_THREADS) return;  // the head role is a 256-thread role, the sampler role a one-wave (or, collecting many head slices, four-wave) role
    if (w < p.n_lw + p.PA * p.HR) wn_v3_head<SH, 1>(p, r, cx, wn_lds4, w - p.n_lw);
    else if (p.PA >= 8)   // many head slices: the sampler collects them with four waves (the head's staging area is free in this workgroup)
        wn_v3_sampler<SH, true>(p, r, cx, wn_lds4 + WnV3Lds<SH, 1>::smp, wn_lds4 + WnV3Lds<SH, 1>::pre, w - p.n_lw - p.PA * p.HR, wn_lds4 + WnV3Lds<SH, 1>::sk);
    else if (threadIdx.x < 64) wn_v3_sampler<SH>(p, r, cx, wn_lds4 + WnV3Lds<SH, 1>::smp, wn_lds4 + WnV3Lds<SH, 1>::pre, w - p.n_lw - p.PA * p.HR);
}

#endif  // WN_KERNEL_V4_H
