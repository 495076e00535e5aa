// wn_stacked.hip -- translation unit of the stacked-layer generation kernels (variant 4, wn_kernel_v4.h): their instantiations, the host-side
// packers of their per-lane weight images and their launchers, behind wn_v4_table() (wn_stacked_table.h: why this is a unit of its own).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "wn_kernel.h"
#include "wn_kernel_v3.h"
#include "wn_kernel_v4.h"
#include "wn_stacked_table.h"

template <int R, int D, int S, int EC, int LPW>
static void wn_pack_v4(const WnPlan& pl, const WnHostWeights& w, std::vector<float>& out) {
    using V = WnV4Shape<R, D, S>;
    using SH = WnV2Shape<R, D, S, EC>;
    constexpr int KF = V::KF, KR = V::KR, KS = V::KS, RS = V::RS, T = WN_THREADS_V4;
    const int NL = pl.NL, n_stack = pl.n_lw, E = pl.E;
    const size_t per_wg = (size_t)LPW * V::NWPL * T;
    out.assign((size_t)n_stack * per_wg + (size_t)pl.PA * SH::NWH * 256, 0.f);
    for (int wg = 0; wg < n_stack; ++wg)
        for (int li = 0; li < LPW; ++li) {
            const int l = wg * LPW + li;
            if (l >= NL) continue;   // (a last workgroup with fewer layers: zeros)
            float* img = out.data() + (size_t)wg * per_wg + (size_t)li * V::NWPL * T;
            const float* fw = w.filter_w + (size_t)l * D * R * 2;
            const float* gw = w.gate_w + (size_t)l * D * R * 2;
            const float* rw = w.res_w + (size_t)l * R * D;
            const float* sw = w.skip_w + (size_t)l * S * D;
            for (int t = 0; t < T; ++t) {
                const int g8 = t >> 3, kq = t & 7, sr = t >> 1, kh = t & 1;
                auto put = [&](int j, float v) { img[(size_t)j * T + t] = v; };
                if (g8 < D) {   // filter / gate rows of channel g8 on x[kq KF .. ): pairs {f, g}; tap 1 = x[t], tap 0 = x[t-d] (Appendix A item 1)
                    for (int k = 0; k < KF; ++k) {
                        const size_t at = ((size_t)g8 * R + kq * KF + k) * 2;
                        put(V::O_W1 + 2 * k, fw[at + 1]); put(V::O_W1 + 2 * k + 1, gw[at + 1]);
                        put(V::O_W0 + 2 * k, fw[at + 0]); put(V::O_W0 + 2 * k + 1, gw[at + 0]);
                    }
                    if (pl.has_bias && kq == 0) { put(V::O_B0, w.filter_b[(size_t)l * D + g8]); put(V::O_B0 + 1, w.gate_b[(size_t)l * D + g8]); }
                }
                if (g8 < R) {   // residual row g8 on z[kq KR .. )
                    for (int k = 0; k < KR; ++k) put(V::O_WR + k, rw[(size_t)g8 * D + kq * KR + k]);
                    if (pl.has_bias && kq == 0) put(V::O_BRES, w.res_b[(size_t)l * R + g8]);
                }
                // skip rows sr + 256 q on z[kh KS .. ): rows 2h, 2h+1 side by side; a single row in natural order
                if (RS % 2 == 0) {
                    for (int h = 0; h < RS / 2; ++h)
                        for (int k = 0; k < KS; ++k) {
                            put(V::O_WS + 2 * (h * KS + k), sw[(size_t)(sr + 256 * (2 * h)) * D + kh * KS + k]);
                            put(V::O_WS + 2 * (h * KS + k) + 1, sw[(size_t)(sr + 256 * (2 * h + 1)) * D + kh * KS + k]);
                        }
                } else {
                    for (int k = 0; k < KS; ++k) put(V::O_WS + k, sw[(size_t)sr * D + kh * KS + k]);
                }
                if (pl.has_bias && kh == 0)
                    for (int q = 0; q < RS; ++q) put(V::O_BSKIP + q, w.skip_b[(size_t)l * S + sr + 256 * q]);
            }
        }
    for (int h = 0; h < pl.PA; ++h) {   // head images: variant 3's (wn_pack_v2)
        float* img = out.data() + (size_t)n_stack * per_wg + (size_t)h * SH::NWH * 256;
        for (int tid = 0; tid < 256; ++tid) {
            const int kq3 = tid % SH::T3, row3 = tid / SH::T3, e = h * EC + row3;
            int j = 0;
            for (int k = 0; k < SH::K3; ++k) img[(size_t)(j++) * 256 + tid] = w.end1_w[(size_t)e * S + kq3 * SH::K3 + k];
            for (int k = 0; k < EC; ++k) img[(size_t)(j++) * 256 + tid] = w.end2_w[(size_t)tid * E + h * EC + k];
            img[(size_t)(j++) * 256 + tid] = kq3 == 0 ? w.end1_b[e] : 0.f;
            img[(size_t)(j++) * 256 + tid] = h == 0 ? w.end2_b[tid] : 0.f;
        }
    }
}

template <int R, int D, int S, int EC, int LPW>
static WnV4Entry wn_v4_entry() {
    using V = WnV4Shape<R, D, S>;
    using SH = WnV2Shape<R, D, S, EC>;
    WnV4Entry e;
    e.R = R; e.D = D; e.S = S; e.EC = EC; e.LPW = LPW; e.nwpl = V::NWPL; e.nwh = SH::NWH;
    e.pack = wn_pack_v4<R, D, S, EC, LPW>;
    e.fn = (const void*)wn_generate_kernel_v4<R, D, S, EC, LPW>;
    e.lds_pre_head = WnV3Lds<SH, 1>::pre;
    e.lds_floats = [](int ns) {
        int need = WnV4Lds<V, LPW>::floats(ns);
        const int head = WnV3Lds<SH, 1>::pre + (SH::K3 > 100 ? SH::K3 : EC) * 256;   // the head lanes' LDS-resident weights (wn_v3_head)
        const int smp = WnV3Lds<SH, 1>::pre + 256 * R;                               // start_conv^T in the sampler workgroups
        if (head > need) need = head;
        if (smp * 4 <= WN_LDS_MAX_BYTES && smp > need) need = smp;
        return need;
    };
    e.launch = [](int grid, size_t lds, hipStream_t st, const WnPlan& p, const WnRun& r) {
        hipLaunchKernelGGL((wn_generate_kernel_v4<R, D, S, EC, LPW>), dim3(grid), dim3(WN_THREADS_V4), lds, st, p, r);
    };
    return e;
}

const std::vector<WnV4Entry>& wn_v4_table() {
    static const std::vector<WnV4Entry> t = {
        wn_v4_entry<64, 64, 256, 64, 3>(),     // cfg2 (BASELINE configs[1]): 10 stack workgroups
        wn_v4_entry<32, 32, 256, 64, 5>(),     // cfg1 (configs[0]): 2
        wn_v4_entry<32, 32, 1024, 32, 2>(),    // train_script.py:17-25, the reference's only trained model shape: 15
    };
    return t;
}

