// wn_kernel_v3.h -- multi-stream generation chain with WAVE-SPECIALISED layer workgroups (gfx950, device only).
//
// Same chain, same hand-off granules, same HBM buffers and the same per-lane weight images as the multi-stream kernel of
// wn_kernel_v2.h; what changes is WHO inside a layer workgroup does what.  Measured on cfg3 (profiles/r02_sweep_one_chain.txt):
// a pipeline item keeps a 256-thread layer workgroup busy ~1.15 us, of which only ~0.47 us (stage x, filter/gate, gate, residual,
// publish x') is on the token's critical path; the other ~0.7 us (skip 1x1 + the running skip lane, queue push, queue tap, the
// next step's tap-0 half of the dilated conv) is work nobody downstream is waiting for -- but it sits in the same instruction
// stream, so with 64 tokens in flight the chain saturates at 64 x 1.87 us per timestep while its 53 stages could turn a token
// around in 58 us (16 tokens: 58.2 us per timestep).  Here a layer workgroup has 512 threads = two wave groups, one wave of
// each per SIMD:
//     waves 0-3  "critical":  poll x' partials -> stage x -> [A] -> filter/gate dot (tap 1) + parked tap 0 -> tanh*sigmoid -> z
//                             -> [B] -> residual partial -> publish x' -> request the next item's inputs
//     waves 4-7  "tail":      [B] -> skip partial on the running skip lane -> publish; queue push; stage the queue tap
//                             -> [A] -> tap-0 half of the NEXT timestep of this stream -> park it
// [A] and [B] are the two LDS-only workgroup barriers of an item; the tail group works on item i between B(i) and B(i+1) while
// the critical group is already polling / staging item i+1.  The critical waves hold only w1 (tap 1) and the residual slice,
// the tail waves tap 0 and the skip slice: both fit 256 VGPRs, nothing lives in LDS but activations and the parked tap-0 sums.
// ONE chain serves all streams (no second copy of the weights, no second set of hand-off buffers).
//
// LDS hazards (i = item index; x is double buffered, everything else single):
//   xs[buf(i)]   written by critical before A(i); read by critical (A(i)..A(i+1)), by tail in chunk 1 (queue push) and chunk 2
//                (d = 1: x[t] is the tap); next written for item i+2 after B(i+1), i.e. after tail's chunk 2 of item i.
//   zs           written between A(i) and B(i); read by both groups between B(i) and A(i+1).
//   xo           written by tail in chunk 1 (B(i)..A(i+1)), read in chunk 2 (A(i+1)..B(i+1)).
//   pre[s]       written by tail in chunk 2 of item (e, s), read by critical in item (e+1, s) = i + n_streams >= i + 2 (the
//                kernel is used for n_streams >= 2 only).
#ifndef WN_KERNEL_V3_H
#define WN_KERNEL_V3_H

#include "wn_kernel_v2.h"

#define WN_THREADS_V3 512
#ifndef WN_V3_PRIO
#define WN_V3_PRIO 1  // critical waves run at a higher static wave priority than the tail waves they share a SIMD with
#endif

template <class SH>
struct WnV3Lds {
    using M = WnV2LdsM<SH, 1, false>;  // the head / sampler roles are those of wn_kernel_v2.h: keep their offsets
    static constexpr int XR = M::XR, SKP = M::SKP, DCP = M::DCP;
    static constexpr int xs = M::xs, zs = M::zs, xo = M::xo, sk = M::sk, ev = M::ev, smp = M::smp, park = M::park;
    static constexpr int park_t = M::pre;     // 16 floats: stamps of the tail group
    static constexpr int pre = park_t + 16;   // [n_streams][256]
    static __host__ __device__ int floats(int n_streams) { return pre + n_streams * 256; }
};

// Shapes whose roles fit the 256 VGPRs a 512-thread workgroup leaves each lane: the tail group holds tap 0 + the skip slice
// (K1 + RS*DC floats), a head workgroup its end_conv_1 slice + end_conv_2 rows (K3 + EC); the rest is working set.
template <class SH>
static constexpr bool wn_v3_fits() { return SH::K1 + SH::RS * SH::DC <= 170 && SH::K3 + SH::EC <= 150; }

template <class SH, int P>
static __device__ void wn_v3_layer(const WnPlan& p, const WnRun& r, WnCtx& cx, float* lds, int l, int c) {
    constexpr int R = SH::R, DC = SH::DC, S = SH::S, T1 = SH::T1, K1 = SH::K1, T2 = SH::T2, K2 = SH::K2, RS = SH::RS;
    using L = WnV3Lds<SH>;
    const int tid = threadIdx.x, t = tid & 255;
    const bool tail = tid >= 256;  // wave-uniform
    const int ns = p.n_streams, NL = p.NL;
    const float* img = p.blobs + (size_t)cx.w * (SH::NWL * 256) + t;  // image rows: w1[K1] w0[K1] w2[K2] w3[RS][DC] bfg bres bskip[RS]
    const int kq1 = t % T1, grp = t / T1, ch = grp >> 1, is_gate = grp & 1;
    const int kq2 = t % T2, row2 = t / T2;
    const int d = p.dil[l];
    const int ML = d + 1;
    float* xs = lds + L::xs;
    float* zs = lds + L::zs;
    float* xol = lds + L::xo;
    float* pre = lds + L::pre;
    float* smp = lds + L::smp;
    int* failflag = reinterpret_cast<int*>(smp + 48);
    int* locflags = reinterpret_cast<int*>(smp + 52);
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
    const int n_prime = (int)(r.n_given - 1);

    if (!tail) {
        // ================================================================== critical group
#if WN_V3_PRIO
        __builtin_amdgcn_s_setprio(3);
#endif
        float w1[K1], w2[K2];
#pragma unroll
        for (int k = 0; k < K1; ++k) w1[k] = img[(size_t)k * 256];
#pragma unroll
        for (int k = 0; k < K2; ++k) w2[k] = img[(size_t)(2 * K1 + k) * 256];
        const float bres = img[(size_t)(2 * K1 + K2 + RS * DC + 1) * 256];
        long long* park = reinterpret_cast<long long*>(lds + L::park);
        // one-item-ahead request registers (branch-free, compile-time load count: see wn_v2_layer_multi)
        wn_u64 nx[P];
        const wn_u64* xbase = l == 0 ? p.gi : p.gx + ((size_t)(l - 1) * P) * ns * R + (t < R ? t : 0);
        const size_t xstep_s = l == 0 ? 1 : R, xstep_j = l == 0 ? 0 : (size_t)ns * R;
        auto request = [&](int s2) {
#pragma unroll
            for (int j = 0; j < P; ++j) nx[j] = wn_ld_granule(xbase + (size_t)s2 * xstep_s + (size_t)j * xstep_j);
        };
        request(0);
        int buf = 0;
        long long misses = 0;
        for (long long e = 0; e < r.n_eval; ++e) {
            const uint32_t tag = (uint32_t)(e + 1);
            for (int s = 0; s < ns; ++s, buf ^= 1) {
                float* xb = xs + buf * L::XR;
                cx.t_start = (long long)wall_clock64();
                const long long item = e * ns + s;
                wn_stamp(r, park, item, 0);
                // ---- 1. layer input x[t]
                if (l == 0) {
                    int idx;
                    if (e == 0) {
                        idx = r.first[(size_t)s * r.n_given];
                    } else {
                        wn_u64 gv = nx[0];
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
                    if (t < R) xb[SH::xpad(t)] = p.start_t[(size_t)idx * R + t] + (p.start_b ? p.start_b[t] : 0.f);
                } else if (t < R) {
                    bool ok = true;
                    float sum = 0.f;
#pragma unroll
                    for (int j = 0; j < P; ++j) { ok = ok && ((uint32_t)(nx[j] >> 32) == tag); sum += __uint_as_float((uint32_t)nx[j]); }
                    if (!ok) {
                        sum = wn_poll_fixed<P, WN_MULTI_SLEEP>(cx, p.gx + (((size_t)(l - 1) * P) * ns + s) * R + t, (size_t)ns * R, tag, WN_W_X, e, s);
                        if (t == 0) ++misses;
                    }
                    xb[SH::xpad(t)] = sum;
                }
                wn_stamp(r, park, item, 4);
                if (wn_barrier_failed(cx, failflag)) return;  // ---- A(i): x staged
                wn_stamp(r, park, item, 1);
                // ---- 2. filter/gate: tap 1 on x[t] + parked tap 0, tanh * sigmoid   (wavenet_model.py:147-151)
                const float xres = (c == 0 && kq2 == 0) ? xb[SH::xpad(row2)] : 0.f;
                float acc = wn_dot_lds<K1>(w1, xb + kq1 * (K1 + 4), pre[s * 256 + t]);
                acc = wn_reduce<T1>(acc);
                const float other = wn_partner<T1>(acc);
                const float fv = is_gate ? other : acc, gv = is_gate ? acc : other;
                const float z = wn_gate(fv, gv);
                if (!is_gate && kq1 == 0) zs[ch] = z;
                if (wn_barrier_failed(cx, failflag)) return;  // ---- B(i): z staged
                wn_stamp(r, park, item, 5);
                // ---- 3. residual 1x1 partial, published at once                      (wavenet_model.py:164-165)
                if (l < NL - 1) {
                    float a2 = wn_dot_lds<K2>(w2, zs + kq2 * K2, 0.f);
                    a2 = wn_reduce<T2>(a2);
                    if (kq2 == 0) wn_publish_at(p.gx + ((size_t)cx.w * ns + s) * R + row2, tag, (a2 + bres) + xres, local_x);
                }
                wn_stamp(r, park, item, 2);
                request(s + 1 < ns ? s + 1 : 0);
                wn_stamp(r, park, item, 3);
                if (r.prof && tid == 0) park[5] = misses;
                wn_stamp_flush(r, park, cx.w, item);
            }
        }
        if (wn_barrier_failed(cx, failflag)) return;  // A(N), B(N): the tail group's last chunk 2 runs between them
        (void)wn_barrier_failed(cx, failflag);
        return;
    }

    // ====================================================================== tail group
    float w0[K1], w3[RS][DC], bskip[RS];
#pragma unroll
    for (int k = 0; k < K1; ++k) w0[k] = img[(size_t)(K1 + k) * 256];
#pragma unroll
    for (int q = 0; q < RS; ++q)
#pragma unroll
        for (int k = 0; k < DC; ++k) w3[q][k] = img[(size_t)(2 * K1 + K2 + q * DC + k) * 256];
    const float bfg = img[(size_t)(2 * K1 + K2 + RS * DC) * 256];
#pragma unroll
    for (int q = 0; q < RS; ++q) bskip[q] = img[(size_t)(2 * K1 + K2 + RS * DC + 2 + q) * 256];
    long long* park_t = reinterpret_cast<long long*>(lds + L::park_t);

    for (int s = 0; s < ns; ++s) {  // tap 0 of the first evaluation of every stream: x[t_base - d] from the queue (zeros after reset)
        const float* ring = p.rings + p.ring_off[l] + ((size_t)c * ns + s) * (size_t)ML * R;
        long long pos = (r.t_base - d) % ML;
        if (pos < 0) pos += ML;
        float acc = kq1 == 0 ? bfg : 0.f;
#pragma unroll
        for (int k = 0; k < K1; ++k) acc += w0[k] * ring[(size_t)pos * R + kq1 * K1 + k];
        pre[s * 256 + t] = acc;
    }
    // one-item-ahead requests: the upstream slice's skip lane and this stream's queue tap x[t+1-d]
    wn_u64 sk_nx[RS];
    float xo_nx;
    const wn_u64* sbase = p.gs + (((size_t)(l > 0 ? l - 1 : 0) * P + c) * ns) * S + t;
    float* rings_l = p.rings + p.ring_off[l] + (size_t)c * ns * (size_t)ML * R;  // stream s: + s * ML * R
    auto request_t = [&](int s2, int tapmod2) {
#pragma unroll
        for (int q = 0; q < RS; ++q) sk_nx[q] = wn_ld_granule(sbase + (size_t)s2 * S + 256 * q);
        xo_nx = (d != 1 && t < R) ? rings_l[((size_t)s2 * ML + tapmod2) * R + t] : 0.f;
    };
    int tmod = (int)(r.t_base % ML);  // queue slot of x[t], kept incrementally
    {
        const int tapmod0 = tmod + 2 >= ML ? tmod + 2 - ML : tmod + 2;
        request_t(0, tapmod0);
    }
    if (wn_barrier_failed(cx, failflag)) return;  // A(0)
    int buf = 0;
    for (long long e = 0; e < r.n_eval; ++e, tmod = (tmod + 1 == ML) ? 0 : tmod + 1) {
        const bool prime = e < n_prime;
        const uint32_t tag = (uint32_t)(e + 1);
        for (int s = 0; s < ns; ++s, buf ^= 1) {
            if (wn_barrier_failed(cx, failflag)) return;  // ---- B(i): z of this item staged
            const float* xb = xs + buf * L::XR;
            cx.t_start = (long long)wall_clock64();
            const long long item = e * ns + s;
            if (r.prof && item < r.prof_items && tid == 256) park_t[0] = cx.t_start;
            // ---- chunk 1a: skip 1x1 partial on this lane of the running skip sum          (wavenet_model.py:154-162)
            wn_u64* gs = p.gs + ((size_t)cx.w * ns + s) * S;
            if (!prime) {
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
                        if ((uint32_t)(sk_nx[q] >> 32) == tag) v = __uint_as_float((uint32_t)sk_nx[q]);
                        else v = wn_poll_fixed<1>(cx, sbase + (size_t)s * S + 256 * q, 0, tag, WN_W_SKIN, e, s);
                        a3[q] += v;
                    }
                    wn_publish_at(gs + t + 256 * q, tag, a3[q], local_s);
                }
            } else if (l == NL - 1) {
#pragma unroll
                for (int q = 0; q < RS; ++q) wn_publish_at(gs + t + 256 * q, tag, 0.f, local_s);
            }
            // ---- chunk 1b: queue push (wavenet_modules.py:55-57), stage the tap x[t+1-d] requested one item ago
            if (t < R) {
                rings_l[((size_t)s * ML + tmod) * R + t] = xb[SH::xpad(t)];
                if (d != 1) xol[SH::xpad(t)] = xo_nx;
            }
            {   // requests for the next item (the last item re-requests a valid address; the values are never used)
                const bool wrap = s + 1 == ns;
                const int s2 = wrap ? 0 : s + 1;
                const int tmod2 = wrap ? ((tmod + 1 == ML) ? 0 : tmod + 1) : tmod;
                request_t(s2, tmod2 + 2 >= ML ? tmod2 + 2 - ML : tmod2 + 2);
            }
            if (wn_barrier_failed(cx, failflag)) return;  // ---- A(i+1): the tap is staged (and the critical group has x of item i+1)
            // ---- chunk 2: tap-0 half of the dilated conv for the NEXT timestep of this stream, parked for the critical group
            {
                const float* xsrc = d == 1 ? xb : xol;
                pre[s * 256 + t] = wn_dot_lds<K1>(w0, xsrc + kq1 * (K1 + 4), kq1 == 0 ? bfg : 0.f);
            }
            if (r.prof && item < r.prof_items && tid == 256) {
                long long* dst = r.prof + ((size_t)cx.w * r.prof_items + item) * WN_STAMPS;
                dst[6] = park_t[0];
                dst[7] = (long long)wall_clock64();
            }
        }
    }
    (void)wn_barrier_failed(cx, failflag);  // B(N)
}

template <int R, int DC, int S, int EC, int P>
__global__ __launch_bounds__(WN_THREADS_V3) void wn_generate_kernel_v3m(WnPlan p, WnRun r) {
    using SH = WnV2Shape<R, DC, S, EC>;
    extern __shared__ __attribute__((aligned(16))) float wn_lds3m[];
    const int w = p.wg_map[blockIdx.x];
    if (w < 0) return;
    WnCtx cx;
    cx.p = &p; cx.r = &r; cx.lds = wn_lds3m; cx.w = w; cx.fail = 0;
    cx.t_start = (long long)wall_clock64();
    const int n_layer_wg = p.NL * p.P;
    if (w < n_layer_wg) {
        wn_v3_layer<SH, P>(p, r, cx, wn_lds3m, w / P, w % P);
        return;
    }
    if (threadIdx.x >= WN_THREADS) return;  // head and sampler roles are 256-thread roles (wn_kernel_v2.h)
    if (w < n_layer_wg + p.PA) wn_v2_head_multi<SH, P, 1>(p, r, cx, wn_lds3m, w - n_layer_wg);
    else wn_v2_sampler(p, r, cx, wn_lds3m + WnV3Lds<SH>::smp, w - n_layer_wg - p.PA);
}

#endif  // WN_KERNEL_V3_H
