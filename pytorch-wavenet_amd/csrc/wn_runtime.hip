// wn_runtime.hip -- C ABI of libwn_mi355.so (include/wn_abi.h): handle, planner glue, HIP launch.
//
//   hipcc --offload-arch=gfx950 -shared ... wn_runtime.hip    -> libwn_mi355.so
// One backend: gfx950.  (Host logic above this ABI is tested without a GPU against tests/double, a host-memory test double
// of include/wn_abi.h built on the C oracle -- it shares no code with this file.)
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/wn_abi.h"
#include "wn_kernel.h"
#include "wn_kernel_v3.h"
#include "wn_stacked_table.h"
#include "wn_forward.h"
#include "wn_gate.h"
#include "wn_optim.h"

static thread_local char g_err[512] = "";
// Development overrides (WN_KERNEL, WN_V3_MODE, WN_CHAINS, WN_SAMPLERS, WN_NO_LOCAL_STORES) pick another kernel or form than the
// planner would: they exist for A/B runs and for the tests that pin a form, and are IGNORED unless WN_TESTING=1 is set as well -- a
// stray variable in a production environment must not silently change what runs (wn_get_info reports the form that does).
static thread_local int g_dev_env_used = 0;
static thread_local int g_create_depth = 0;  // wn_create is building the member handles of a rounds front
static const char* wn_dev_env(const char* name) {
    const char* t = getenv("WN_TESTING");
    if (!t || t[0] != '1') return nullptr;
    const char* v = getenv(name);
    if (v) g_dev_env_used = 1;
    return v;
}

static int wn_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// One "NN" product C = A . B^T of wn_forward.h.  bn != NULL: bf16 operands (B given as [N][K] bf16 -- or as two [N][ldb] halves
// bn / bn1 --, A rounded while staged), fp32 accumulation; else fp32 operands.  Products with N % 256 == 0 take the 128 x 256 tile.
static void wn_launch_nn(hipStream_t st, int epi, const WnGemmArgs& a, const unsigned short* bn = nullptr, const unsigned short* bn1 = nullptr, int ldb = 0) {
    const bool wide = bn && a.N % 256 == 0 && epi != WN_EPI_GATE_BWD;
    const unsigned mt = (unsigned)((a.M + 127) / 128), nt = (unsigned)(wide ? a.N / 256 : (a.N + 127) / 128);
    const dim3 grid(mt * nt);   // 1-D (M / 128 can exceed a grid's y limit): row tiles fastest, then column tiles
    if (bn) {
        WnGemmArgsBf16 b;
        b.g = a; b.bn = bn; b.bn1 = bn1; b.ldb = ldb;
        if (wide) {
            if (epi == WN_EPI_GATE && a.a_bf16) hipLaunchKernelGGL((wn_fwd_gemm_bf16<WN_EPI_GATE, 8, true>), grid, dim3(512), 0, st, b);   // (bf16-stored A: the shadow of x)
            else if (epi == WN_EPI_GATE) hipLaunchKernelGGL((wn_fwd_gemm_bf16<WN_EPI_GATE, 8>), grid, dim3(512), 0, st, b);
            else if (a.a_bf16) hipLaunchKernelGGL((wn_fwd_gemm_bf16<WN_EPI_PLAIN, 8, true>), grid, dim3(512), 0, st, b);   // (bf16-stored A: the grouped skip product)
            else hipLaunchKernelGGL((wn_fwd_gemm_bf16<WN_EPI_PLAIN, 8>), grid, dim3(512), 0, st, b);
        } else {
            if (epi == WN_EPI_GATE) hipLaunchKernelGGL((wn_fwd_gemm_bf16<WN_EPI_GATE, 4>), grid, dim3(256), 0, st, b);   // (never with a bf16-stored A: wn_train_layout_ws)
            else if (epi == WN_EPI_GATE_BWD) hipLaunchKernelGGL((wn_fwd_gemm_bf16<WN_EPI_GATE_BWD, 4>), grid, dim3(256), 0, st, b);
            else if (a.a_bf16) hipLaunchKernelGGL((wn_fwd_gemm_bf16<WN_EPI_PLAIN, 4, true>), grid, dim3(256), 0, st, b);   // (bf16-stored A: the residual and dx products)
            else hipLaunchKernelGGL((wn_fwd_gemm_bf16<WN_EPI_PLAIN, 4>), grid, dim3(256), 0, st, b);
        }
        return;
    }
    if (epi == WN_EPI_GATE) hipLaunchKernelGGL(wn_fwd_gemm<WN_EPI_GATE>, grid, dim3(256), 0, st, a);
    else if (epi == WN_EPI_GATE_BWD) hipLaunchKernelGGL(wn_fwd_gemm<WN_EPI_GATE_BWD>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(wn_fwd_gemm<WN_EPI_PLAIN>, grid, dim3(256), 0, st, a);
}

// One forward layer in one launch (wn_fwd_layer_bf16): `a` = the filter/gate product's arguments (bf16 operands, c_bf16 = 1; a.c.base may be
// NULL: z is not stored), `r` = the residual product's (its bias, cin, c, c_h are used).  Returns false when the shape is not the fused
// kernel's (the caller launches the two products).  WN_NO_FUSED_LAYER=1 (with WN_TESTING=1) switches it off for A/B runs.
static bool wn_fused_layer_enabled() { const char* off = wn_dev_env("WN_NO_FUSED_LAYER"); return !(off && off[0] == '1'); }
static bool wn_launch_layer(hipStream_t st, const WnGemmArgs& a, const unsigned short* bn_fg, const WnGemmArgs& r, const unsigned short* bn_res) {
    if (!bn_fg || !bn_res || !a.a_bf16 || a.N != 256 || r.N != 128 || r.K != 128 || a.K % 32 != 0 || a.k_split % 32 != 0 || !a.c_bf16 || a.relu_a || r.relu_a || r.relu_c ||
        r.mask || r.cin_skip_lo || r.M != a.M || r.rows_per_batch != a.rows_per_batch) return false;   // (x from its bf16 shadow, z kept as bf16, the same rows in both products)
    if (!wn_fused_layer_enabled()) return false;
    WnGemmArgsBf16 b;
    b.g = a; b.bn = bn_fg; b.bn1 = nullptr; b.ldb = 0;
    WnLayerArgs la;
    la.bn = bn_res; la.bias = r.bias; la.cin = r.cin; la.c = r.c; la.c_h = r.c_h; la.N = r.N;
    const dim3 grid(wn_layer_grid(a.M));
    hipLaunchKernelGGL(wn_fwd_layer_bf16, grid, dim3(512), 0, st, b, la);
    return true;
}

// The backward's fused pair (wn_bwd_layer_bf16): `a` = the dx product of layer l (bf16-stored A in two views, weight banks bn / bn1 with row
// length ldb), `b` = the gate-derivative product of layer l - 1, whose A operand is `a`'s output.  Returns false when the shapes are not
// the fused kernel's (the caller launches the two products).
static bool wn_launch_bwd_layer(hipStream_t st, const WnGemmArgs& a, const unsigned short* bn, const unsigned short* bn1, int ldb,
                                const WnGemmArgs& b, const unsigned short* bn_res) {
    if (!bn || !bn_res || !a.a_bf16 || a.N != 128 || a.K % 32 != 0 || a.bias || a.relu_a || a.relu_c || a.mask || a.c_h || a.c_bf16 ||
        b.N != 128 || b.K != 128 || !b.c_bf16 || !b.gate_packed || b.M != a.M || b.rows_per_batch != a.rows_per_batch) return false;
    if (b.a0.base != a.c.base || b.a0.t0 != a.c.t0 || b.a0.batch_stride != a.c.batch_stride || b.a0.row_stride != a.c.row_stride) return false;   // (the same rows)
    if (!wn_fused_layer_enabled()) return false;
    { const char* off = wn_dev_env("WN_NO_FUSED_BWD"); if (off && off[0] == '1') return false; }   // (A/B: the forward's fused layer alone)
    WnGemmArgsBf16 x, y;
    x.g = a; x.bn = bn; x.bn1 = bn1; x.ldb = ldb;
    y.g = b; y.bn = bn_res; y.bn1 = nullptr; y.ldb = 0;
    hipLaunchKernelGGL(wn_bwd_layer_bf16, dim3(wn_layer_grid(a.M)), dim3(512), 0, st, x, y);
    return true;
}

// ------------------------------------------------------------------------------------------------ runtime shim
static const char* g_hip_what = "";
static int rt_hip(hipError_t e, const char* what) {
    if (e == hipSuccess) return 0;
    g_hip_what = what;
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return WN_E_HIP;
}
static void* rt_malloc(size_t n) {
    void* p = nullptr;
    if (hipMalloc(&p, n ? n : 1) != hipSuccess) return nullptr;
    return p;
}
static void rt_free(void* p) { if (p) (void)hipFree(p); }
static int rt_h2d(void* d, const void* h, size_t n) { return rt_hip(hipMemcpy(d, h, n, hipMemcpyHostToDevice), "hipMemcpy H2D"); }
static int rt_d2h(void* h, const void* d, size_t n) { return rt_hip(hipMemcpy(h, d, n, hipMemcpyDeviceToHost), "hipMemcpy D2H"); }
static int rt_memset_async(void* d, int v, size_t n, void* stream) {
    return rt_hip(hipMemsetAsync(d, v, n, (hipStream_t)stream), "hipMemsetAsync");
}
static int rt_sync(void* stream) { return rt_hip(hipStreamSynchronize((hipStream_t)stream), "hipStreamSynchronize"); }

// The persistent kernel: one workgroup per chain position, alive for the whole generate_fast() job.
__global__ __launch_bounds__(WN_THREADS) void wn_generate_kernel(WnPlan p, WnRun r) {
    extern __shared__ __attribute__((aligned(16))) float wn_lds[];
    const int w = p.wg_map[blockIdx.x];
    if (w < 0) return;
    WnCtx cx;
    cx.p = &p; cx.r = &r; cx.lds = wn_lds; cx.w = w; cx.fail = 0;
    cx.t_start = (long long)wall_clock64();
    if (wn_not_resident(cx, wn_lds)) return;   // (every workgroup of the job is resident from here on)
    wn_load_lds(p, w, wn_lds);
    cx.t_start = (long long)wall_clock64();
    const int n_layer_wg = p.NL * p.P;
    if (w < n_layer_wg) {
        const int l = w / p.P, c = w % p.P;
        const long long n_it = r.n_eval + (l == 0 ? 1 : 0);  // L0 runs one sample-only iteration at the end
        const int ML = (p.k - 1) * p.dil[l] + 1;
        int tmod = (int)(r.t_base % ML);  // queue slot of x[t], kept incrementally (a 64-bit modulo is ~100 scalar instructions)
        for (long long e = 0; e < n_it; ++e, tmod = tmod + 1 == ML ? 0 : tmod + 1)
            for (int s = 0; s < p.n_streams; ++s) {
                cx.t_start = (long long)wall_clock64();  // the spin bound is per hand-off wait, not per job
                if (!wn_layer_item(cx, l, c, e, s, tmod)) return;
            }
    } else {
        const int h = w - n_layer_wg;
        for (long long e = 0; e < r.n_eval; ++e)
            for (int s = 0; s < p.n_streams; ++s) {
                cx.t_start = (long long)wall_clock64();
                if (!wn_head_item(cx, h, e, s)) return;
            }
    }
}


// ------------------------------------------------------------------------------------------------ register-resident shapes
// Shapes the wave-specialised kernel (variant 3, wn_kernel_v3.h) is instantiated for: (R, D/P, S, E/PA, P).  Anything else runs on the
// generic LDS-resident kernel above.  (Rounds 1-2 also had 256-thread kernels on these shapes -- "variant 2", one chain or several
// sharing the CUs two by two; round 3 moved every shape to variant 3 and removed them.)
struct WnV2Entry {
    int R, DC, S, EC, nwl, nwh;
    int Pm;  // layer split the kernel is compiled for (its input poll is unrolled over it)
    void (*pack)(const WnPlan& pl, const WnHostWeights& w, std::vector<float>& out);
    // [form]: 0 = one stream per pipeline item of a layer workgroup, 1 = two (wn_v3_mode), 2 = two + skip-lane slot re-use (wn_v3_slots_for); NULL where
    // the shape has no such form
    const void* fn_v3[3];
    int (*lds_floats_v3)(int ns, int g2);
    int lds_pre_v3;  // float offset of the per-stream area = what head / sampler workgroups use in front of their own tables
    void (*launch_v3)(int form, int grid, size_t lds, hipStream_t st, const WnPlan& p, const WnRun& r);
};

template <class SH>
static void wn_pack_v2(const WnPlan& pl, const WnHostWeights& w, std::vector<float>& out) {
    constexpr int R = SH::R, DC = SH::DC, S = SH::S, EC = SH::EC, T1 = SH::T1, K1 = SH::K1, T2 = SH::T2, K2 = SH::K2, RS = SH::RS,
                  T3 = SH::T3, K3 = SH::K3;
    const int D = pl.D, E = pl.E, C = pl.C, P = pl.P, NL = pl.NL;
    out.assign((size_t)NL * P * SH::NWL * 256 + (size_t)pl.PA * SH::NWH * 256, 0.f);
    for (int l = 0; l < NL; ++l)
        for (int c = 0; c < P; ++c) {
            float* img = out.data() + ((size_t)l * P + c) * SH::NWL * 256;
            const float* fw = w.filter_w + (size_t)l * D * R * 2;
            const float* gw = w.gate_w + (size_t)l * D * R * 2;
            const float* rw = w.res_w + (size_t)l * R * D;
            const float* sw = w.skip_w + (size_t)l * S * D;
            for (int tid = 0; tid < 256; ++tid) {
                int j = 0;
                const int kq1 = tid % T1, grp = tid / T1, ch = c * DC + (grp >> 1), gate = grp & 1;
                const float* cw = gate ? gw : fw;
                for (int k = 0; k < K1; ++k) img[(size_t)(j++) * 256 + tid] = cw[((size_t)ch * R + kq1 * K1 + k) * 2 + 1];  // tap 1: x[t]
                for (int k = 0; k < K1; ++k) img[(size_t)(j++) * 256 + tid] = cw[((size_t)ch * R + kq1 * K1 + k) * 2 + 0];  // tap 0: x[t-d]
                const int kq2 = tid % T2, row2 = tid / T2;
                for (int k = 0; k < K2; ++k) img[(size_t)(j++) * 256 + tid] = rw[(size_t)row2 * D + c * DC + kq2 * K2 + k];
                for (int q = 0; q < RS; ++q)
                    for (int k = 0; k < DC; ++k) img[(size_t)(j++) * 256 + tid] = sw[(size_t)(tid + 256 * q) * D + c * DC + k];
                float bfg = 0.f, bres = 0.f;
                if (pl.has_bias) {
                    if (kq1 == 0) bfg = (gate ? w.gate_b : w.filter_b)[(size_t)l * D + ch];
                    if (c == 0 && kq2 == 0) bres = w.res_b[(size_t)l * R + row2];
                }
                img[(size_t)(j++) * 256 + tid] = bfg;
                img[(size_t)(j++) * 256 + tid] = bres;
                for (int q = 0; q < RS; ++q)
                    img[(size_t)(j++) * 256 + tid] = (pl.has_bias && c == 0) ? w.skip_b[(size_t)l * S + tid + 256 * q] : 0.f;
            }
        }
    for (int h = 0; h < pl.PA; ++h) {
        float* img = out.data() + (size_t)NL * P * SH::NWL * 256 + (size_t)h * SH::NWH * 256;
        for (int tid = 0; tid < 256; ++tid) {
            const int kq3 = tid % T3, row3 = tid / T3, e = h * EC + row3;
            int j = 0;
            for (int k = 0; k < K3; ++k) img[(size_t)(j++) * 256 + tid] = w.end1_w[(size_t)e * S + kq3 * K3 + k];
            for (int k = 0; k < EC; ++k) img[(size_t)(j++) * 256 + tid] = w.end2_w[(size_t)tid * E + h * EC + k];
            img[(size_t)(j++) * 256 + tid] = kq3 == 0 ? w.end1_b[e] : 0.f;
            img[(size_t)(j++) * 256 + tid] = h == 0 ? w.end2_b[tid] : 0.f;
        }
    }
    (void)C;
}

template <int R, int DC, int S, int EC, int PM, int SK = 0>
static WnV2Entry wn_v2_entry() {
    using SH = WnV2Shape<R, DC, S, EC>;
    WnV2Entry e;
    e.R = R; e.DC = DC; e.S = S; e.EC = EC; e.nwl = SH::NWL; e.nwh = SH::NWH; e.Pm = PM;
    e.pack = wn_pack_v2<SH>;
    e.fn_v3[0] = e.fn_v3[1] = e.fn_v3[2] = nullptr; e.lds_floats_v3 = nullptr; e.launch_v3 = nullptr; e.lds_pre_v3 = 0;
    static_assert(wn_v3_fits<SH, PM>(), "every table entry runs the wave-specialised kernel");
    {
        e.fn_v3[0] = (const void*)wn_generate_kernel_v3m<R, DC, S, EC, PM, 1>;
        if constexpr (wn_v3_g2_fits<SH>()) e.fn_v3[1] = (const void*)wn_generate_kernel_v3m<R, DC, S, EC, PM, 2>;
        if constexpr (wn_v3_g2_fits<SH>() && SK > 0) e.fn_v3[2] = (const void*)wn_generate_kernel_v3m<R, DC, S, EC, PM, 2, SK>;
        e.lds_pre_v3 = WnV3Lds<SH, 1>::pre;
        e.lds_floats_v3 = [](int ns, int g2) {
            int lay = WnV3Lds<SH, 1>::floats(ns);
            if constexpr (wn_v3_g2_fits<SH>()) { if (g2) lay = WnV3Lds<SH, 2>::floats(ns); }
            const int head = WnV3Lds<SH, 1>::pre + (SH::K3 > 100 ? SH::K3 : EC) * 256;  // + the head lanes' LDS-resident weights (wn_v3_head)
            const int smp = WnV3Lds<SH, 1>::pre + 256 * R;    // + start_conv^T in the sampler workgroups (wn_v3_sampler), when it fits
            int need = lay > head ? lay : head;
            if (smp * 4 <= WN_LDS_MAX_BYTES && smp > need) need = smp;
            return need;
        };
        e.launch_v3 = [](int form, int grid, size_t lds, hipStream_t st, const WnPlan& p, const WnRun& r) {
            if constexpr (wn_v3_g2_fits<SH>() && SK > 0) {
                if (form == 2) { hipLaunchKernelGGL((wn_generate_kernel_v3m<R, DC, S, EC, PM, 2, SK>), dim3(grid), dim3(WN_THREADS_V3), lds, st, p, r); return; }
            }
            if constexpr (wn_v3_g2_fits<SH>()) {
                if (form >= 1) { hipLaunchKernelGGL((wn_generate_kernel_v3m<R, DC, S, EC, PM, 2>), dim3(grid), dim3(WN_THREADS_V3), lds, st, p, r); return; }
            }
            hipLaunchKernelGGL((wn_generate_kernel_v3m<R, DC, S, EC, PM, 1>), dim3(grid), dim3(WN_THREADS_V3), lds, st, p, r);
        };
    }
    return e;
}

static const std::vector<WnV2Entry>& wn_v2_table() {
    static const std::vector<WnV2Entry> t = {
        wn_v2_entry<128, 32, 512, 32, 4, 4>(),   // cfg3: P=4, PA=8 (a 64-row head slice is the slowest pipeline stage); + the slot re-use form (96 streams and more)
        wn_v2_entry<64, 64, 256, 64, 1>(),    // cfg2: P=1, PA=4
        wn_v2_entry<32, 32, 256, 64, 1>(),    // cfg1: P=1, PA=4
        wn_v2_entry<32, 16, 1024, 32, 2>(),   // train_script.py chaconne shape, split two ways (64 skip weights per lane), PA=16 (end_conv_1 slice in LDS)
        wn_v2_entry<64, 32, 256, 64, 2>(),    // cfg2 split in two (layer_split=2)
        wn_v2_entry<16, 16, 256, 32, 1>(),    // small test shape
        wn_v2_entry<16, 16, 256, 32, 2>(),    // ... and its two-slice form
    };
    return t;
}

// picks an instantiated shape for this model; returns its index or -1
static int wn_v2_choose(const WnPlan& pl, int n_cu, int n_smp, int forced_P, int forced_PA, int* outP, int* outPA) {
    if (pl.k != 2 || pl.C != 256) return -1;
    const std::vector<WnV2Entry>& t = wn_v2_table();
    for (size_t i = 0; i < t.size(); ++i) {
        const WnV2Entry& e = t[i];
        if (e.R != pl.R || e.S != pl.S || pl.D % e.DC || pl.E % e.EC) continue;
        const int P = pl.D / e.DC, PA = pl.E / e.EC;
        if (P != e.Pm || PA > 16) continue;  // the kernel is compiled per layer split
        if ((forced_P > 0 && forced_P != P) || (forced_PA > 0 && forced_PA != PA)) continue;
        if (pl.NL * P + PA + n_smp > n_cu) continue;
        *outP = P; *outPA = PA;
        return (int)i;
    }
    return -1;
}

// dedicated sampler workgroups of a multi-stream chain (each serves the streams s = j mod n): 4 by default
static int wn_sampler_count(int n_streams) {
    int n = 4;
    const char* e = wn_dev_env("WN_SAMPLERS");
    if (e && atoi(e) >= 1 && atoi(e) <= 16) n = atoi(e);
    return n_streams < n ? n_streams : n;
}

// Throughput form of the wave-specialised kernel (bit 0: two streams per pipeline item of a layer workgroup; bit 1: two replicas
// of the head workgroups, each serving every other stream).  One stream per item is the latency-optimal form; with more streams
// than the chain can turn around in one trip the stages' fixed costs per item -- the request round trip, two workgroup barriers,
// the LDS and DPP latencies of the dot products -- bound the throughput, and two streams per item share them (every weight
// operand is used twice) at the price of a longer trip through each stage.  WN_V3_MODE = 0..3 pins a form (A/B runs, tests).
// (the rule itself is host-only arithmetic in wn_plan.h: wn_v3_mode_for, tests/test_plan_host.py)
static int wn_v3_mode(int n_streams, int n_layers) { return wn_v3_mode_for(n_streams, wn_dev_env("WN_V3_MODE"), n_layers); }

// true iff the wave-specialised kernel (variant 3) serves this configuration with ONE chain: an instantiated shape, at least
// two streams, the parked tap-0 sums of all streams fit the LDS next to the activations, one CU per workgroup
static bool wn_v3_applicable(const wn_config* cfg, int n_cu, int* out_vi, int* outP, int* outPA) {
    const char* force = wn_dev_env("WN_KERNEL");  // "generic" pins the LDS-resident kernel (A/B runs, tests)
    if (force && !strcmp(force, "generic")) return false;
    if (cfg->n_streams < WN_V3_MIN_STREAMS) return false;
    WnPlan pl;
    memset(&pl, 0, sizeof(pl));
    pl.layers = cfg->layers; pl.blocks = cfg->blocks; pl.NL = cfg->layers * cfg->blocks;
    pl.R = cfg->residual_channels; pl.D = cfg->dilation_channels; pl.S = cfg->skip_channels; pl.E = cfg->end_channels;
    pl.C = cfg->classes; pl.k = cfg->kernel_size; pl.n_streams = cfg->n_streams;
    const int n_smp = wn_sampler_count(cfg->n_streams);
    int P = 0, PA = 0;
    const int vi = wn_v2_choose(pl, n_cu, n_smp, cfg->layer_split, cfg->head_split, &P, &PA);
    if (vi < 0) return false;
    if (wn_v2_table()[vi].lds_floats_v3(cfg->n_streams, (wn_v3_mode(cfg->n_streams, cfg->layers * cfg->blocks) & 1) && wn_v2_table()[vi].fn_v3[1]) * 4 > WN_LDS_MAX_BYTES) return false;
    if (out_vi) *out_vi = vi;
    if (outP) *outP = P;
    if (outPA) *outPA = PA;
    return true;
}

// true iff the stacked kernel serves this configuration: an instantiated shape exactly, few streams, no pinned split
static bool wn_v4_applicable(const wn_config* cfg, int n_cu, int* out_vi, int* outPA) {
    const char* force = wn_dev_env("WN_KERNEL");  // "v3" / "generic" pin another kernel (A/B runs, tests); "v4" lifts the stream limit
    if (force && (!strcmp(force, "generic") || !strcmp(force, "v3"))) return false;
    const bool forced = force && !strcmp(force, "v4");
    if (cfg->kernel_size != 2 || cfg->classes != 256 || cfg->layer_split > 0 || cfg->head_split > 0) return false;
    if (cfg->n_streams < 1 || cfg->n_streams > 16) return false;
    const std::vector<WnV4Entry>& t = wn_v4_table();
    for (size_t i = 0; i < t.size(); ++i) {
        const WnV4Entry& e = t[i];
        if (e.R != cfg->residual_channels || e.D != cfg->dilation_channels || e.S != cfg->skip_channels || cfg->end_channels % e.EC) continue;
        const int PA = cfg->end_channels / e.EC, NL = cfg->layers * cfg->blocks, n_stack = (NL + e.LPW - 1) / e.LPW;
        if (PA > 16) continue;
        // The stacked chain is a short pipeline (n_stack + 2 stages, each busy ~3 us per item): it saturates at about half a token per stage,
        // beyond that the one-layer-per-workgroup pipeline of variant 3 wins (measured, profiles/r04_v4_vs_v3_streams.txt: cfg2 -- 10 stack
        // workgroups -- up to 6 streams, cfg1 -- 2 -- up to 2, the train_script shape -- 15 -- up to 8).
        if (!forced && cfg->n_streams > wn_v4_stream_limit(n_stack)) continue;
        const int n_smp = wn_sampler_count(cfg->n_streams);
        if (n_stack + PA + n_smp > n_cu) continue;
        if (e.lds_floats(cfg->n_streams) * 4 > WN_LDS_MAX_BYTES) continue;
        if (out_vi) *out_vi = (int)i;
        if (outPA) *outPA = PA;
        return true;
    }
    return false;
}

// ------------------------------------------------------------------------------------------------ handle
struct WnTrainLay {
    long long N, L, out_len;
    std::vector<long long> need;          // need[l] = trailing time steps of layer l's input the loss depends on (and that exist: wn_forward_geometry)
    std::vector<long long> zlo;           // zlo[l] = leading output rows of layer l whose tap x(t - d) is one of the reference's pad zeros
    std::vector<size_t> x, th, sg;        // per layer offsets (floats) into the training workspace
    std::vector<size_t> zb;               // per skip block: Z_b, the block's z matrices side by side -- row (n, t) holds [z_first(n, t) | ... | z_{first+cnt-1}(n, t)] (see wn_train_layout_ws)
    std::vector<long long> zb_rows;       //   rows per batch entry of Z_b (the block's first layer has the most)
    std::vector<size_t> xh;               // bf16 step: the bf16 shadow of x[l] (same shape; offsets in floats, N * L * R / 2 floats each); empty otherwise
    size_t skip, ev, dzg, bskip_total, res_o, skip_o, w1_o, w2_o, fgb0, fgb1, dskip, de, dz, dfg, dfg2, dxa, dxb, colsum_tmp, idx, total;
    size_t dskip_h;                       // bf16 step: the bf16 shadow of dskip (a matrix operand twice per skip block: the dzg product and the skip weight gradient); offset in floats
    size_t bw, bt_fg, bt_res, bt_skip, bt_w1, bt_w2;  // bf16 operand banks (offsets in floats)
    int G, nblk;  // layers per skip block, blocks
    bool bf16;  // the saved forward ran with bf16 operands: so does its backward
};
struct WnDetWs { float* buf = nullptr; size_t floats = 0; };
struct wn_handle {
    wn_config cfg;
    WnPlan plan;
    bool have_weights;
    bool pending;
    bool broken;   // rounds front: a launch failed after some rounds had started; the rounds' queue times diverged -> wn_reset
    void* last_stream;
    long long t_base;  // evaluations since the last reset
    int n_cu, wall_khz;
    int variant;   // 1 = generic LDS-resident kernel, 3 = wave-specialised register-resident kernel (2: the 256-thread kernels of rounds 1-2, removed)
    int v2_index;  // row of wn_v2_table()
    int lds_bytes;
    int v3_mode;   // variant 3: streams per pipeline item (wn_v3_mode)
    int v3_slots = 0;   // variant 3: skip-lane slots re-used per in-flight item (wn_v3_slots_for; 0 = one slot per stream)
    int dev_overrides = 0;  // a development override was in effect when this handle was planned (wn_dev_env)
    // Zero padding: a channel shape the wave-specialised kernel is not compiled for runs as the next instantiated shape that holds it,
    // its weights padded with zeros (wn_pad_config).  cfg / plan then carry the PADDED channel counts; these are the caller's.
    bool padded = false;
    int user_R = 0, user_D = 0, user_S = 0, user_E = 0;
    int export_R = 0;       // rows wn_export_queue returns (0: plan.R); set on a padded handle and on the member handles of its rounds
    // Rounds: more streams than ONE chain holds (its LDS parks a tap-0 sum per stream: cfg3 ~150 streams) are served in rounds of up to
    // WN_V3_ROUND_STREAMS streams, one round after the other on the caller's stream: this handle is then only a front that routes
    // every call to its member handles (`chains`, one complete engine per round).
    std::vector<wn_handle*> chains;
    std::vector<int> chain_first;  // first stream of round i (chain_first[n] = n_streams)
    bool rounds;
    bool shares_weights = false;   // member i > 0 of a rounds front: its weight images, start_conv^T and GEMM banks ARE member 0's (immutable during a
                                   // job, identical for every member -- the image layout depends on the channel shape, not on the stream count): not freed here
    // owned device allocations
    float *d_blobs, *d_start_t, *d_start_b, *d_rings;
    int32_t *d_dil, *d_wg_map;
    int64_t* d_ring_off;
    wn_u64* d_gran;
    uint32_t* d_status;
    size_t blob_floats, ring_floats, gran_count;
    long long* d_prof;
    // batched forward (wn_forward): GEMM-ready weight banks and a workspace that grows on demand
    float* d_fw; size_t fw_floats; bool fw_ok;
    size_t fw_off_fg, fw_off_bfg, fw_off_res, fw_off_bres, fw_off_skip, fw_off_bskip, fw_off_bskip_total, fw_off_w1, fw_off_b1, fw_off_w2, fw_off_b2;
    size_t fw_off_start_t, fw_off_start_b;  // training only: wn_forward / wn_prime read d_start_t / d_start_b
    float* d_ws; size_t ws_floats;
    unsigned short* d_fwb; size_t fwb_elems; bool fwb_ok; int fw_bf16;  // bf16 copies of the forward banks, [N][K] row-major
    size_t fwb_off_fg, fwb_off_res, fwb_off_skip, fwb_off_w1, fwb_off_w2;
    int prof_items;      // stamps requested for the next job (0 = off)
    int prof_recorded;   // stamps held in d_prof
    std::vector<int64_t> ring_off;
    std::vector<int32_t> dil;
    float* d_tws; size_t tws_floats;  // training workspace (saved activations + backward temporaries)
    float* d_xent = nullptr; size_t xent_rows = 0;  // wn_train_loss: per-row losses
    WnTrainLay train; bool train_valid;
    // admission of persistent jobs (wn_gate.h)
    // wn_train_backward: the weight-gradient products run on a second stream next to the activation-gradient chain (wn_train.inl)
    hipStream_t side_stream = nullptr;
    std::vector<hipEvent_t> events;
    // deterministic weight / bias gradients (wn_train_set_deterministic; default from WN_DETERMINISTIC=1 at wn_create): partial tiles of the row splits
    // in a workspace per stream + an ordered reduction instead of fp32 atomics (wn_train.inl)
    bool deterministic = false;
    WnDetWs det_ws[2];
    char busid[32] = "";
    std::shared_ptr<WnGateTicket> gate;   // the booking of the job in flight (released by the host function behind the kernel, or in wn_wait)
    int gate_shared = -1, gate_waited_ms = 0;
    int gate_need = 0;   // resident workgroups on the fullest XCD (wn_create)
    int resident_ms = 0; // bound of the last job's residency barrier
    int wg_per_cu = 0;   // workgroups of the job's kernel one CU holds (hipOccupancyMaxActiveBlocksPerMultiprocessor at wn_create)
    long long last_n_eval = 0;   // evaluations the job in flight advances the queues by (rolled back when it never started: WN_E_BUSY)
};

// CUs per XCD a job of this handle needs -- the workgroups that stay resident on the fullest XCD (blocks are dispatched round-robin over
// the XCDs; the padding blocks of the layer-aligned placement exit at once and hold nothing) -- and what an XCD has
static void wn_gate_numbers(const wn_handle* h, int* need, int* cap) {
    const int n_xcd = h->n_cu % 8 == 0 ? 8 : 1;
    *cap = h->n_cu / n_xcd;
    *need = h->gate_need > 0 ? h->gate_need : (h->plan.n_blocks + n_xcd - 1) / n_xcd;
}
static void wn_gate_host_release(void* user) {   // runs on the runtime's callback thread once the job's kernel has finished
    std::shared_ptr<WnGateTicket>* t = static_cast<std::shared_ptr<WnGateTicket>*>(user);
    wn_gate_release(*t);
    delete t;
}

extern "C" int wn_abi_version(void) { return WN_ABI_VERSION; }
extern "C" const char* wn_last_error(void) { return g_err; }

extern "C" void wn_destroy(wn_handle* h) {
    if (!h) return;
    if (!h->chains.empty()) {
        for (wn_handle* c : h->chains) wn_destroy(c);
        (void)hipSetDevice(h->cfg.device_id);
        delete h;
        return;
    }
    (void)hipSetDevice(h->cfg.device_id);
    if (h->pending) (void)hipStreamSynchronize((hipStream_t)h->last_stream);
    if (h->side_stream) { (void)hipStreamSynchronize(h->side_stream); (void)hipStreamDestroy(h->side_stream); }
    for (hipEvent_t e : h->events) (void)hipEventDestroy(e);
    wn_gate_release(h->gate);
    if (!h->shares_weights) { rt_free(h->d_blobs); rt_free(h->d_start_t); rt_free(h->d_start_b); rt_free(h->d_fw); rt_free(h->d_fwb); }
    rt_free(h->d_rings); rt_free(h->d_dil);
    rt_free(h->d_wg_map); rt_free(h->d_ring_off); rt_free(h->d_gran); rt_free(h->d_status); rt_free(h->d_prof); rt_free(h->d_ws);
    rt_free(h->d_tws);
    rt_free(h->d_xent);
    rt_free(h->det_ws[0].buf); rt_free(h->det_ws[1].buf);
    delete h;
}

static int wn_create_impl(const wn_config* cfg, wn_handle** out);

// Zero padding of the channel counts.  Extra residual channels stay 0 for ever (zero rows in start_conv and the residual convs), extra
// dilation channels give tanh(0) * sigmoid(0) = 0, extra skip / end channels give relu(0) = 0 against zero weights: the padded network
// computes the same logits, and every product only gains exact zeros.  So a kernel_size-2, 256-class model whose channel shape is not
// in the table of the wave-specialised kernel (e.g. 48 / 48 / 300 / 200) runs as the cheapest instantiated shape that holds it instead
// of on the generic LDS-resident kernel (last measured 6-13x slower).  Not applied when the caller pins layer_split / head_split.
static bool wn_pad_config(const wn_config* cfg, int n_cu, wn_config* padded) {
    if (cfg->kernel_size != 2 || cfg->classes != 256 || cfg->layer_split > 0 || cfg->head_split > 0 || (cfg->reserved[0] & WN_CFG_NO_PADDING)) return false;
    wn_config probe = *cfg;
    if (probe.n_streams > WN_V3_ROUND_STREAMS) probe.n_streams = WN_V3_ROUND_STREAMS;
    if (wn_v4_applicable(&probe, n_cu, nullptr, nullptr) || wn_v3_applicable(&probe, n_cu, nullptr, nullptr, nullptr)) return false;   // served as it is
    std::vector<WnShapeRow> rows;
    for (const WnV2Entry& e : wn_v2_table()) rows.push_back(WnShapeRow{e.R, e.DC, e.S, e.EC, e.Pm});
    int dims[4];
    const int pick = wn_pad_pick(rows.data(), (int)rows.size(), cfg->residual_channels, cfg->dilation_channels, cfg->skip_channels, cfg->end_channels,
                                 cfg->layers * cfg->blocks, [&](int R2, int D2, int S2, int E2) {
                                     wn_config c = probe;
                                     c.residual_channels = R2; c.dilation_channels = D2; c.skip_channels = S2; c.end_channels = E2;
                                     return wn_v3_applicable(&c, n_cu, nullptr, nullptr, nullptr);
                                 }, dims);
    if (pick < 0) return false;
    *padded = *cfg;
    padded->residual_channels = dims[0]; padded->dilation_channels = dims[1]; padded->skip_channels = dims[2]; padded->end_channels = dims[3];
    return true;
}

extern "C" int wn_create(const wn_config* cfg, wn_handle** out) {
    g_err[0] = 0;
    if (!cfg || !out) return wn_fail(WN_E_BADARG, "wn_create: NULL argument");
    *out = nullptr;
    if (!g_create_depth) g_dev_env_used = 0;
    if (!g_create_depth && cfg->layers >= 1 && cfg->blocks >= 1 && cfg->dilation_channels >= 1 && cfg->residual_channels >= 1 && cfg->skip_channels >= 1 &&
        cfg->end_channels >= 1 && cfg->n_streams >= 1 && cfg->layers <= 24) {
        int ndev = 0, n_cu = 0;
        if (hipGetDeviceCount(&ndev) == hipSuccess && cfg->device_id >= 0 && cfg->device_id < ndev &&
            hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, cfg->device_id) == hipSuccess) {
            wn_config eff;
            if (wn_pad_config(cfg, n_cu, &eff)) {
                const int rc = wn_create_impl(&eff, out);
                if (rc == WN_OK && (*out)->variant >= 3) {
                    wn_handle* h = *out;
                    h->padded = true;
                    h->user_R = cfg->residual_channels; h->user_D = cfg->dilation_channels; h->user_S = cfg->skip_channels; h->user_E = cfg->end_channels;
                    h->export_R = h->user_R;
                    for (wn_handle* c : h->chains) c->export_R = h->user_R;
                    return WN_OK;
                }
                if (rc == WN_OK) { wn_destroy(*out); *out = nullptr; }   // (not the kernel the padding was for: plan the caller's shape instead)
                g_err[0] = 0;
            }
        }
    }
    return wn_create_impl(cfg, out);
}

static int wn_create_impl(const wn_config* cfg, wn_handle** out) {
    g_err[0] = 0;
    if (!cfg || !out) return wn_fail(WN_E_BADARG, "wn_create: NULL argument");
    *out = nullptr;
    if (cfg->layers < 1 || cfg->blocks < 1 || cfg->dilation_channels < 1 || cfg->residual_channels < 1 ||
        cfg->skip_channels < 1 || cfg->end_channels < 1 || cfg->classes < 2 || cfg->n_streams < 1)
        return wn_fail(WN_E_BADARG, "wn_create: non-positive dimension in wn_config");
    if (cfg->kernel_size < 1) return wn_fail(WN_E_BADARG, "wn_create: kernel_size must be >= 1");
    if (cfg->layers > 24) return wn_fail(WN_E_UNSUPPORTED, "wn_create: layers > 24 (dilation 2^layers overflows the queue)");
    if (cfg->layer_split < 0 || cfg->head_split < 0 || (cfg->reserved[0] & ~WN_CFG_NO_PADDING) || cfg->reserved[1] || cfg->reserved[2])
        return wn_fail(WN_E_BADARG, "wn_create: negative split / non-zero reserved field");
    int n_cu = 256, wall_khz = 100000;
    {
        int ndev = 0;
        int rc = rt_hip(hipGetDeviceCount(&ndev), "hipGetDeviceCount");
        if (rc) return rc;
        if (cfg->device_id < 0 || cfg->device_id >= ndev) return wn_fail(WN_E_BADARG, "wn_create: device_id %d of %d", cfg->device_id, ndev);
        if ((rc = rt_hip(hipSetDevice(cfg->device_id), "hipSetDevice"))) return rc;
        hipDeviceProp_t prop;
        if ((rc = rt_hip(hipGetDeviceProperties(&prop, cfg->device_id), "hipGetDeviceProperties"))) return rc;
        n_cu = prop.multiProcessorCount;
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
            return wn_fail(WN_E_UNSUPPORTED, "wn_create: device %d is %s; this library is built for gfx950 (MI355X) only",
                           cfg->device_id, prop.gcnArchName);
        int khz = 0;
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, cfg->device_id) == hipSuccess && khz > 0) wall_khz = khz;
    }
    char busid[32] = "";
    if (hipDeviceGetPCIBusId(busid, (int)sizeof(busid), cfg->device_id) != hipSuccess || !busid[0]) snprintf(busid, sizeof(busid), "dev%d", cfg->device_id);
    {   // rounds of the wave-specialised chain (see wn_handle::rounds)
        const char* ce = wn_dev_env("WN_CHAINS");
        const bool off = ce && ce[0] == '1';
        wn_config probe = *cfg;
        probe.n_streams = WN_V3_ROUND_STREAMS;
        if (!off && cfg->n_streams > WN_V3_ROUND_STREAMS && !wn_v3_applicable(cfg, n_cu, nullptr, nullptr, nullptr) &&
            wn_v3_applicable(&probe, n_cu, nullptr, nullptr, nullptr)) {
            const std::vector<int> sizes = wn_v3_round_sizes(cfg->n_streams, WN_V3_ROUND_STREAMS);
            std::vector<wn_handle*> cs;
            std::vector<int> firsts(1, 0);
            int rc = WN_OK;
            for (size_t i = 0; i < sizes.size() && rc == WN_OK; ++i) {
                wn_config part = *cfg;
                const int n = sizes[i];
                part.n_streams = n;
                wn_handle* c = nullptr;
                ++g_create_depth;
                rc = wn_create(&part, &c);
                --g_create_depth;
                if (rc == WN_OK) {
                    cs.push_back(c);
                    firsts.push_back(firsts.back() + n);
                    if (c->variant != 3) rc = WN_E_UNSUPPORTED;
                }
            }
            if (rc == WN_OK) {
                wn_handle* c0 = cs[0];
                wn_handle* f = new wn_handle();
                f->cfg = *cfg;
                f->plan = c0->plan;
                f->plan.n_streams = cfg->n_streams;
                f->have_weights = false; f->pending = false; f->last_stream = nullptr; f->t_base = 0;
                f->n_cu = n_cu; f->wall_khz = wall_khz; f->variant = c0->variant; f->v2_index = c0->v2_index; f->lds_bytes = c0->lds_bytes;
                f->v3_mode = c0->v3_mode; f->v3_slots = c0->v3_slots;
                f->d_blobs = f->d_start_t = f->d_start_b = f->d_rings = nullptr;
                f->d_dil = f->d_wg_map = nullptr; f->d_ring_off = nullptr; f->d_gran = nullptr; f->d_status = nullptr;
                f->d_prof = nullptr; f->prof_items = 0; f->prof_recorded = 0;
                f->d_fw = nullptr; f->fw_floats = 0; f->fw_ok = false; f->d_ws = nullptr; f->ws_floats = 0;
                f->d_fwb = nullptr; f->fwb_elems = 0; f->fwb_ok = false; f->fw_bf16 = 0;
                f->d_tws = nullptr; f->tws_floats = 0; f->train_valid = false;
                f->blob_floats = f->ring_floats = f->gran_count = 0;
                f->dil = c0->dil;
                f->chains = cs;
                f->chain_first = firsts;
                f->rounds = true;
                *out = f;
                return WN_OK;
            }
            for (wn_handle* c : cs) wn_destroy(c);
            if (rc != WN_E_UNSUPPORTED) return rc;
            g_err[0] = 0;
        }
    }
    wn_handle* h = new wn_handle();
    memset(&h->plan, 0, sizeof(h->plan));
    h->cfg = *cfg;
    h->v3_mode = 0; h->rounds = false;
    { const char* de = getenv("WN_DETERMINISTIC"); h->deterministic = de && de[0] == '1'; }
    h->have_weights = false; h->pending = false; h->last_stream = nullptr; h->t_base = 0;
    h->n_cu = n_cu; h->wall_khz = wall_khz;
    memcpy(h->busid, busid, sizeof(busid));
    h->d_blobs = h->d_start_t = h->d_start_b = h->d_rings = nullptr;
    h->d_dil = h->d_wg_map = nullptr; h->d_ring_off = nullptr; h->d_gran = nullptr; h->d_status = nullptr;
    h->d_prof = nullptr; h->prof_items = 0; h->prof_recorded = 0;
    h->d_fw = nullptr; h->fw_floats = 0; h->fw_ok = false; h->d_ws = nullptr; h->ws_floats = 0;
    h->d_fwb = nullptr; h->fwb_elems = 0; h->fwb_ok = false; h->fw_bf16 = 0;
    h->d_tws = nullptr; h->tws_floats = 0; h->train_valid = false;
    WnPlan& pl = h->plan;
    pl.layers = cfg->layers; pl.blocks = cfg->blocks; pl.NL = cfg->layers * cfg->blocks;
    pl.R = cfg->residual_channels; pl.D = cfg->dilation_channels; pl.S = cfg->skip_channels; pl.E = cfg->end_channels;
    pl.C = cfg->classes; pl.k = cfg->kernel_size; pl.has_bias = cfg->bias ? 1 : 0; pl.n_streams = cfg->n_streams;
    pl.HR = 1;
    h->variant = 1; h->v2_index = -1;
    {
        int P2 = 0, PA2 = 0;
        int vi3 = -1, vi4 = -1;
        if (wn_v4_applicable(cfg, n_cu, &vi4, &PA2)) {
            const WnV4Entry& ve = wn_v4_table()[vi4];
            h->variant = 4; h->v2_index = vi4;
            wn_plan_geometry(pl, 1, PA2);
            pl.LPW = ve.LPW;
            pl.n_lw = (pl.NL + ve.LPW - 1) / ve.LPW;
            pl.n_smp = wn_sampler_count(cfg->n_streams);
            pl.n_wg = pl.n_lw + PA2 + pl.n_smp;
            pl.start_in_lds = (ve.lds_pre_head + 256 * pl.R) * 4 <= WN_LDS_MAX_BYTES ? 1 : 0;
            h->lds_bytes = ve.lds_floats(pl.n_streams) * 4;
            pl.lds_floats = h->lds_bytes / 4;
        } else if (wn_v3_applicable(cfg, n_cu, &vi3, &P2, &PA2)) {
            h->variant = 3; h->v2_index = vi3;
            wn_plan_geometry(pl, P2, PA2);
            pl.n_smp = wn_sampler_count(cfg->n_streams);  // sampling runs on dedicated workgroups (they also feed layer 0)
            pl.n_wg += pl.n_smp;
            pl.start_in_lds = (wn_v2_table()[vi3].lds_pre_v3 + 256 * pl.R) * 4 <= WN_LDS_MAX_BYTES ? 1 : 0;  // the samplers' copy of start_conv^T
            h->v3_mode = wn_v3_mode(pl.n_streams, pl.NL);
            if (!wn_v2_table()[vi3].fn_v3[1]) h->v3_mode &= ~1;  // (shapes whose filter/gate slices cannot be halved: one stream per item only)
            if ((h->v3_mode & 2) && pl.NL * P2 + 2 * PA2 + pl.n_smp <= n_cu) {  // a second set of head workgroups
                pl.HR = 2;
                pl.n_wg += PA2;
            } else {
                h->v3_mode &= 1;
            }
            h->v3_slots = wn_v3_slots_for(pl.n_streams, h->v3_mode, wn_v2_table()[vi3].fn_v3[2] != nullptr, wn_dev_env("WN_V3_SLOTS"), pl.NL);
            h->lds_bytes = wn_v2_table()[vi3].lds_floats_v3(pl.n_streams, h->v3_mode & 1) * 4;
            if (pl.n_streams <= 4 && !(h->v3_mode & 1)) h->lds_bytes = WN_LDS_MAX_BYTES;  // 1-4 streams (8: measured level, profiles/archive/r03_few_stream_lds_queues.txt): the layers' dilation queues live in LDS where they fit (wn_v3_layer)
            pl.lds_floats = h->lds_bytes / 4;
        }
    }
    if (h->variant == 1) {
        const std::string why = wn_plan_choose(pl, n_cu, cfg->layer_split, cfg->head_split);
        if (!why.empty()) {
            delete h;
            return wn_fail(WN_E_UNSUPPORTED, "wn_create: %s", why.c_str());
        }
        h->lds_bytes = (pl.lds_floats + 8) * 4;  // + the fail-flag word (wn_any_failed)
    }
    // tables
    h->dil.resize(pl.NL); h->ring_off.resize(pl.NL);
    int64_t off = 0;
    for (int l = 0; l < pl.NL; ++l) {
        const int d = 1 << (l % pl.layers);  // wavenet_model.py:72,108-110
        h->dil[l] = d;
        h->ring_off[l] = off;
        const int64_t ML = (int64_t)(pl.k - 1) * d + 1;  // wavenet_model.py:78
        off += (int64_t)pl.P * pl.n_streams * ML * pl.R;
    }
    h->ring_floats = (size_t)off;
    std::vector<int32_t> wg_map;
    pl.n_blocks = pl.n_wg;
    pl.allow_plain = 0;
    if ((h->variant == 3 || h->variant == 4) && n_cu % 8 == 0 &&
        wn_make_wg_map_layers(h->variant == 4 ? pl.n_lw : pl.NL, pl.P, pl.PA * pl.HR, pl.n_smp, 8, n_cu / 8, wg_map, &pl.n_blocks)) {
        const char* np = wn_dev_env("WN_NO_LOCAL_STORES");
        pl.allow_plain = (np && np[0] == '1') ? 0 : 1;
    } else {
        wn_make_wg_map(pl.n_wg, 8, wg_map);
    }
    {   // what a job of this handle keeps RESIDENT per XCD (block b lands on XCD b % 8; padding blocks exit at once and hold nothing)
        int per_xcd[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const int nx = n_cu % 8 == 0 ? 8 : 1;
        for (int b = 0; b < pl.n_blocks; ++b)
            if (wg_map[b] >= 0) per_xcd[b % nx]++;
        h->gate_need = 0;
        for (int x = 0; x < nx; ++x) h->gate_need = per_xcd[x] > h->gate_need ? per_xcd[x] : h->gate_need;
    }
    const size_t n_lw = h->variant == 4 ? (size_t)pl.n_lw : (size_t)pl.NL * pl.P;
    const size_t gx_n = n_lw * pl.n_streams * pl.R, gs_n = n_lw * pl.n_streams * pl.S, gl_n = (size_t)pl.PA * pl.n_streams * pl.C;
    const size_t g0_n = h->variant >= 3 ? (size_t)pl.n_streams * pl.R : 0;
    h->gran_count = gx_n + gs_n + gl_n + (size_t)pl.n_streams + g0_n;
    h->blob_floats = n_lw * pl.blob_layer_floats + (size_t)pl.PA * pl.blob_head_floats;
    pl.n_lw = (int32_t)n_lw; pl.head_blob_off = (int64_t)n_lw * pl.blob_layer_floats;
    if (h->variant != 4) pl.LPW = 1;
    if (h->variant == 4) {
        const WnV4Entry& ve = wn_v4_table()[h->v2_index];
        pl.head_blob_off = (int64_t)n_lw * ve.LPW * ve.nwpl * WN_THREADS_V4;
        h->blob_floats = (size_t)pl.head_blob_off + (size_t)pl.PA * ve.nwh * 256;
    }
    if (h->variant == 3) {
        const WnV2Entry& ve = wn_v2_table()[h->v2_index];
        h->blob_floats = n_lw * (size_t)ve.nwl * 256 + (size_t)pl.PA * ve.nwh * 256;
        pl.head_blob_off = (int64_t)n_lw * ve.nwl * 256;
    }
    h->d_blobs = (float*)rt_malloc(h->blob_floats * 4);
    h->d_start_t = (float*)rt_malloc((size_t)pl.C * pl.R * 4);
    h->d_start_b = (float*)rt_malloc((size_t)pl.R * 4);
    h->d_rings = (float*)rt_malloc(h->ring_floats * 4);
    h->d_dil = (int32_t*)rt_malloc((size_t)pl.NL * 4);
    h->d_ring_off = (int64_t*)rt_malloc((size_t)pl.NL * 8);
    h->d_wg_map = (int32_t*)rt_malloc((size_t)pl.n_blocks * 4);
    h->d_gran = (wn_u64*)rt_malloc(h->gran_count * 8);
    h->d_status = (uint32_t*)rt_malloc((size_t)(8 + pl.n_wg) * 4);
    if (!h->d_blobs || !h->d_start_t || !h->d_start_b || !h->d_rings || !h->d_dil || !h->d_ring_off || !h->d_wg_map ||
        !h->d_gran || !h->d_status) {
        const double q_mb = h->ring_floats * 4e-6, g_mb = h->gran_count * 8e-6, w_mb = h->blob_floats * 4e-6;
        wn_destroy(h);
        return wn_fail(WN_E_NOMEM, "wn_create: device allocation failed (queues %.1f MB, hand-off %.1f MB, weights %.1f MB)", q_mb, g_mb, w_mb);
    }
    int rc = 0;
    rc = rc ? rc : rt_h2d(h->d_dil, h->dil.data(), (size_t)pl.NL * 4);
    rc = rc ? rc : rt_h2d(h->d_ring_off, h->ring_off.data(), (size_t)pl.NL * 8);
    rc = rc ? rc : rt_h2d(h->d_wg_map, wg_map.data(), (size_t)pl.n_blocks * 4);
    rc = rc ? rc : rt_memset_async(h->d_rings, 0, h->ring_floats * 4, nullptr);
    rc = rc ? rc : rt_memset_async(h->d_status, 0, (size_t)(8 + pl.n_wg) * 4, nullptr);
    rc = rc ? rc : rt_sync(nullptr);
    if (rc) { wn_destroy(h); return rc; }
    pl.blobs = h->d_blobs; pl.start_t = h->d_start_t; pl.start_b = nullptr;
    pl.dil = h->d_dil; pl.ring_off = h->d_ring_off; pl.wg_map = h->d_wg_map; pl.rings = h->d_rings;
    pl.gx = h->d_gran; pl.gs = h->d_gran + gx_n; pl.gl = h->d_gran + gx_n + gs_n; pl.gi = h->d_gran + gx_n + gs_n + gl_n;
    pl.g0 = pl.gi + pl.n_streams;
    pl.status = h->d_status;
    pl.xcc_tab = h->d_status + 8;
    rc = rt_hip(hipFuncSetAttribute(h->variant == 4 ? wn_v4_table()[h->v2_index].fn : h->variant == 3 ? wn_v2_table()[h->v2_index].fn_v3[h->v3_slots ? 2 : (h->v3_mode & 1)] : (const void*)wn_generate_kernel,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, h->lds_bytes),
                "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
    if (rc) { wn_destroy(h); return rc; }
    {   // residency is a requirement, not a hope: what the hardware can keep resident of THIS kernel with THIS much LDS, against what the plan needs per XCD
        const void* fn = h->variant == 4 ? wn_v4_table()[h->v2_index].fn : h->variant == 3 ? wn_v2_table()[h->v2_index].fn_v3[h->v3_slots ? 2 : (h->v3_mode & 1)] : (const void*)wn_generate_kernel;
        const int threads = h->variant == 4 ? WN_THREADS_V4 : h->variant == 3 ? WN_THREADS_V3 : WN_THREADS;
        int per_cu = 0;
        rc = rt_hip(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, threads, (size_t)h->lds_bytes), "hipOccupancyMaxActiveBlocksPerMultiprocessor");
        if (rc) { wn_destroy(h); return rc; }
        int need = 0, cap = 0;
        wn_gate_numbers(h, &need, &cap);
        h->wg_per_cu = per_cu;
        if ((long long)per_cu * cap < need) {
            const int n_wg = pl.n_wg;
            wn_destroy(h);
            return wn_fail(WN_E_UNSUPPORTED, "wn_create: the plan keeps %d workgroups resident on one XCD (%d in all), the device holds %d x %d of this kernel there",
                           need, n_wg, cap, per_cu);
        }
    }
    h->dev_overrides = g_dev_env_used;
    *out = h;
    return WN_OK;
}

static int wn_load_weights_impl(wn_handle* h, const wn_weight_ptrs* w);

// the caller's weights copied into zero-filled arrays of the padded shape
static int wn_load_weights_padded(wn_handle* h, const wn_weight_ptrs* w) {
    const WnPlan& pl = h->plan;
    const int R = h->user_R, D = h->user_D, S = h->user_S, E = h->user_E, R2 = pl.R, D2 = pl.D, S2 = pl.S, E2 = pl.E, C = pl.C, NL = pl.NL, k = pl.k;
    if (!w->start_w || !w->filter_w || !w->gate_w || !w->res_w || !w->skip_w || !w->end1_w || !w->end1_b || !w->end2_w || !w->end2_b)
        return wn_fail(WN_E_BADARG, "wn_load_weights: a mandatory weight pointer is NULL");
    if (pl.has_bias && (!w->start_b || !w->filter_b || !w->gate_b || !w->res_b || !w->skip_b))
        return wn_fail(WN_E_BADARG, "wn_load_weights: cfg.bias=1 but a stack bias pointer is NULL");
    // dst[n][a][b] (extents A2, B2, inner run of `in` floats) <- src[n][a][b] (extents A, B)
    auto pad3 = [](const float* src, int n, int A, int B, int A2, int B2, int in) {
        std::vector<float> dst((size_t)n * A2 * B2 * in, 0.f);
        for (int i = 0; i < n; ++i)
            for (int a = 0; a < A; ++a)
                memcpy(dst.data() + (((size_t)i * A2 + a) * B2) * in, src + (((size_t)i * A + a) * B) * in, (size_t)B * in * 4);
        return dst;
    };
    std::vector<float> start_w = pad3(w->start_w, 1, R, C, R2, C, 1);
    std::vector<float> filter_w = pad3(w->filter_w, NL, D, R, D2, R2, k), gate_w = pad3(w->gate_w, NL, D, R, D2, R2, k);
    std::vector<float> res_w = pad3(w->res_w, NL, R, D, R2, D2, 1), skip_w = pad3(w->skip_w, NL, S, D, S2, D2, 1);
    std::vector<float> end1_w = pad3(w->end1_w, 1, E, S, E2, S2, 1), end1_b = pad3(w->end1_b, 1, 1, E, 1, E2, 1);
    std::vector<float> end2_w = pad3(w->end2_w, 1, C, E, C, E2, 1);
    std::vector<float> start_b, filter_b, gate_b, res_b, skip_b;
    wn_weight_ptrs p = *w;
    p.start_w = start_w.data(); p.filter_w = filter_w.data(); p.gate_w = gate_w.data(); p.res_w = res_w.data(); p.skip_w = skip_w.data();
    p.end1_w = end1_w.data(); p.end1_b = end1_b.data(); p.end2_w = end2_w.data();
    if (pl.has_bias) {
        start_b = pad3(w->start_b, 1, 1, R, 1, R2, 1); filter_b = pad3(w->filter_b, NL, 1, D, 1, D2, 1); gate_b = pad3(w->gate_b, NL, 1, D, 1, D2, 1);
        res_b = pad3(w->res_b, NL, 1, R, 1, R2, 1); skip_b = pad3(w->skip_b, NL, 1, S, 1, S2, 1);
        p.start_b = start_b.data(); p.filter_b = filter_b.data(); p.gate_b = gate_b.data(); p.res_b = res_b.data(); p.skip_b = skip_b.data();
    }
    return wn_load_weights_impl(h, &p);   // (the padded arrays are this handle's shape; the handle's state is not touched on the way)
}

extern "C" int wn_load_weights(wn_handle* h, const wn_weight_ptrs* w) {
    if (h && w && h->padded) { g_err[0] = 0; return wn_load_weights_padded(h, w); }
    return wn_load_weights_impl(h, w);
}

// weights in the handle's own (possibly padded) channel shape
static int wn_load_weights_impl(wn_handle* h, const wn_weight_ptrs* w) {
    if (h && !h->chains.empty()) {
        // Rounds: ONE copy of the weights.  Member 0 packs and uploads; the others point at its images and banks (round 3 uploaded the
        // weight images, start_conv^T and both GEMM banks into every member: 4 x 120 MB at cfg3 x 512 streams).
        wn_handle* c0 = h->chains[0];
        { int rc = wn_load_weights_impl(c0, w); if (rc) return rc; }
        for (size_t i = 1; i < h->chains.size(); ++i) {
            wn_handle* c = h->chains[i];
            if (c->blob_floats != c0->blob_floats || c->plan.P != c0->plan.P || c->plan.PA != c0->plan.PA || c->variant != c0->variant || c->v2_index != c0->v2_index)
                return wn_fail(WN_E_STATE, "wn_load_weights: the rounds of this handle were planned with different geometries");
            { int rc = rt_hip(hipSetDevice(c->cfg.device_id), "hipSetDevice"); if (rc) return rc; }
            if (c->pending) { int rc = wn_wait(c); if (rc) return rc; }
            if (!c->shares_weights) { rt_free(c->d_blobs); rt_free(c->d_start_t); rt_free(c->d_start_b); rt_free(c->d_fw); rt_free(c->d_fwb); }
            c->shares_weights = true;
            c->d_blobs = c0->d_blobs; c->d_start_t = c0->d_start_t; c->d_start_b = c0->d_start_b;
            c->plan.blobs = c0->plan.blobs; c->plan.start_t = c0->plan.start_t; c->plan.start_b = c0->plan.start_b;
            c->d_fw = c0->d_fw; c->fw_floats = c0->fw_floats; c->fw_ok = c0->fw_ok;
            c->fw_off_fg = c0->fw_off_fg; c->fw_off_bfg = c0->fw_off_bfg; c->fw_off_res = c0->fw_off_res; c->fw_off_bres = c0->fw_off_bres;
            c->fw_off_skip = c0->fw_off_skip; c->fw_off_bskip = c0->fw_off_bskip; c->fw_off_bskip_total = c0->fw_off_bskip_total;
            c->fw_off_w1 = c0->fw_off_w1; c->fw_off_b1 = c0->fw_off_b1; c->fw_off_w2 = c0->fw_off_w2; c->fw_off_b2 = c0->fw_off_b2;
            c->fw_off_start_t = c0->fw_off_start_t; c->fw_off_start_b = c0->fw_off_start_b;
            c->d_fwb = c0->d_fwb; c->fwb_elems = c0->fwb_elems; c->fwb_ok = c0->fwb_ok;
            c->fwb_off_fg = c0->fwb_off_fg; c->fwb_off_res = c0->fwb_off_res; c->fwb_off_skip = c0->fwb_off_skip; c->fwb_off_w1 = c0->fwb_off_w1; c->fwb_off_w2 = c0->fwb_off_w2;
            c->have_weights = true;
        }
        h->have_weights = true;
        return WN_OK;
    }
    g_err[0] = 0;
    if (!h || !w) return wn_fail(WN_E_BADARG, "wn_load_weights: NULL argument");
    if (!w->start_w || !w->filter_w || !w->gate_w || !w->res_w || !w->skip_w || !w->end1_w || !w->end1_b || !w->end2_w || !w->end2_b)
        return wn_fail(WN_E_BADARG, "wn_load_weights: a mandatory weight pointer is NULL");
    const WnPlan& pl = h->plan;
    if (pl.has_bias && (!w->start_b || !w->filter_b || !w->gate_b || !w->res_b || !w->skip_b))
        return wn_fail(WN_E_BADARG, "wn_load_weights: cfg.bias=1 but a stack bias pointer is NULL");
    { int rc = rt_hip(hipSetDevice(h->cfg.device_id), "hipSetDevice"); if (rc) return rc; }
    if (h->pending) { int rc = wn_wait(h); if (rc) return rc; }
    WnHostWeights hw = {w->start_w, w->start_b, w->filter_w, w->filter_b, w->gate_w, w->gate_b, w->res_w,
                        w->res_b, w->skip_w, w->skip_b, w->end1_w, w->end1_b, w->end2_w, w->end2_b};
    std::vector<float> blobs;
    if (h->variant == 4) wn_v4_table()[h->v2_index].pack(pl, hw, blobs);
    else if (h->variant == 3) wn_v2_table()[h->v2_index].pack(pl, hw, blobs);
    else
        wn_pack_blobs(pl, hw, blobs);
    if (blobs.size() != h->blob_floats) return wn_fail(WN_E_STATE, "wn_load_weights: internal blob size mismatch");
    std::vector<float> st((size_t)pl.C * pl.R);
    for (int r = 0; r < pl.R; ++r)
        for (int c = 0; c < pl.C; ++c) st[(size_t)c * pl.R + r] = w->start_w[(size_t)r * pl.C + c];
    int rc = rt_h2d(h->d_blobs, blobs.data(), blobs.size() * 4);
    rc = rc ? rc : rt_h2d(h->d_start_t, st.data(), st.size() * 4);
    if (pl.has_bias) {
        rc = rc ? rc : rt_h2d(h->d_start_b, w->start_b, (size_t)pl.R * 4);
        h->plan.start_b = h->d_start_b;
    } else {
        h->plan.start_b = nullptr;
    }
    if (rc) return rc;
    {   // GEMM-ready banks for wn_forward: B^T [K][N] row-major per layer (see wn_forward.h)
        const int R = pl.R, D = pl.D, S = pl.S, E = pl.E, C = pl.C, NL = pl.NL;
        h->fw_ok = pl.k == 2 && R % 32 == 0 && D % 32 == 0 && S % 32 == 0 && E % 32 == 0 && C % 32 == 0;
        if (h->fw_ok) {
            size_t o = 0;
            h->fw_off_fg = o; o += (size_t)NL * 2 * R * 2 * D;
            h->fw_off_bfg = o; o += (size_t)NL * 2 * D;
            h->fw_off_res = o; o += (size_t)NL * D * R;
            h->fw_off_bres = o; o += (size_t)NL * R;
            h->fw_off_skip = o; o += (size_t)NL * D * S;
            h->fw_off_bskip = o; o += (size_t)NL * S;
            h->fw_off_bskip_total = o; o += (size_t)S;
            h->fw_off_w1 = o; o += (size_t)S * E;
            h->fw_off_b1 = o; o += (size_t)E;
            h->fw_off_w2 = o; o += (size_t)E * C;
            h->fw_off_b2 = o; o += (size_t)C;
            h->fw_off_start_t = o; o += (size_t)C * R;
            h->fw_off_start_b = o; o += (size_t)R;
            std::vector<float> fw(o, 0.f);
            for (int l = 0; l < NL; ++l) {
                float* fg = fw.data() + h->fw_off_fg + (size_t)l * 2 * R * 2 * D;
                for (int ch = 0; ch < D; ++ch) {
                    const int nf = 64 * (ch / 32) + (ch % 32), ng = nf + 32;  // column order [F(32) | G(32)] per 32-channel group
                    for (int tap = 0; tap < 2; ++tap)
                        for (int r = 0; r < R; ++r) {
                            fg[(size_t)(tap * R + r) * 2 * D + nf] = w->filter_w[(((size_t)l * D + ch) * R + r) * 2 + tap];
                            fg[(size_t)(tap * R + r) * 2 * D + ng] = w->gate_w[(((size_t)l * D + ch) * R + r) * 2 + tap];
                        }
                    if (pl.has_bias) {
                        fw[h->fw_off_bfg + (size_t)l * 2 * D + nf] = w->filter_b[(size_t)l * D + ch];
                        fw[h->fw_off_bfg + (size_t)l * 2 * D + ng] = w->gate_b[(size_t)l * D + ch];
                    }
                }
                for (int dch = 0; dch < D; ++dch) {
                    for (int r = 0; r < R; ++r) fw[h->fw_off_res + ((size_t)l * D + dch) * R + r] = w->res_w[((size_t)l * R + r) * D + dch];
                    for (int sc = 0; sc < S; ++sc) fw[h->fw_off_skip + ((size_t)l * D + dch) * S + sc] = w->skip_w[((size_t)l * S + sc) * D + dch];
                }
                if (pl.has_bias) {
                    for (int r = 0; r < R; ++r) fw[h->fw_off_bres + (size_t)l * R + r] = w->res_b[(size_t)l * R + r];
                    for (int sc = 0; sc < S; ++sc) {
                        fw[h->fw_off_bskip + (size_t)l * S + sc] = w->skip_b[(size_t)l * S + sc];
                        fw[h->fw_off_bskip_total + sc] += w->skip_b[(size_t)l * S + sc];  // the grouped skip GEMM adds all biases once
                    }
                }
            }
            for (int sc = 0; sc < S; ++sc)
                for (int e = 0; e < E; ++e) fw[h->fw_off_w1 + (size_t)sc * E + e] = w->end1_w[(size_t)e * S + sc];
            for (int e = 0; e < E; ++e) fw[h->fw_off_b1 + e] = w->end1_b[e];
            for (int e = 0; e < E; ++e)
                for (int c = 0; c < C; ++c) fw[h->fw_off_w2 + (size_t)e * C + c] = w->end2_w[(size_t)c * E + e];
            for (int c = 0; c < C; ++c) fw[h->fw_off_b2 + c] = w->end2_b[c];
            memcpy(fw.data() + h->fw_off_start_t, st.data(), st.size() * 4);
            if (pl.has_bias) memcpy(fw.data() + h->fw_off_start_b, w->start_b, (size_t)R * 4);
            if (h->fw_floats != o) { rt_free(h->d_fw); h->d_fw = (float*)rt_malloc(o * 4); h->fw_floats = o; }
            if (!h->d_fw) return wn_fail(WN_E_NOMEM, "wn_load_weights: forward weight banks (%.1f MB)", o * 4e-6);
            rc = rt_h2d(h->d_fw, fw.data(), o * 4);
            if (rc) return rc;
            // bf16 copies for wn_set_forward_precision(1): B as [N][K] row-major (K contiguous), K a multiple of 64
            const int G = pl.layers < NL ? pl.layers : NL;
            h->fwb_ok = R % 64 == 0 && D % 64 == 0 && S % 64 == 0 && E % 64 == 0 && NL % G == 0;
            if (h->fwb_ok) {
                auto bf = [](float x) -> unsigned short {
                    unsigned u; memcpy(&u, &x, 4);
                    u += 0x7fffu + ((u >> 16) & 1u);
                    return (unsigned short)(u >> 16);
                };
                size_t ob = 0;
                h->fwb_off_fg = ob; ob += (size_t)NL * 2 * D * 2 * R;
                h->fwb_off_res = ob; ob += (size_t)NL * R * D;
                h->fwb_off_skip = ob; ob += (size_t)NL * D * S;
                h->fwb_off_w1 = ob; ob += (size_t)E * S;
                h->fwb_off_w2 = ob; ob += (size_t)C * E;
                std::vector<unsigned short> wb(ob, 0);
                for (int l = 0; l < NL; ++l) {
                    for (int n = 0; n < 2 * D; ++n)  // packed column n of layer l = row n here; k = tap*R + ch
                        for (int k = 0; k < 2 * R; ++k)
                            wb[h->fwb_off_fg + ((size_t)l * 2 * D + n) * 2 * R + k] = bf(fw[h->fw_off_fg + (size_t)l * 2 * R * 2 * D + (size_t)k * 2 * D + n]);
                    for (int r = 0; r < R; ++r)
                        for (int dch = 0; dch < D; ++dch) wb[h->fwb_off_res + ((size_t)l * R + r) * D + dch] = bf(w->res_w[((size_t)l * R + r) * D + dch]);
                    const int blk = l / G, li = l % G;  // skip banks are grouped per block: [block][S][G*D]
                    for (int sc = 0; sc < S; ++sc)
                        for (int dch = 0; dch < D; ++dch)
                            wb[h->fwb_off_skip + ((size_t)blk * S + sc) * G * D + (size_t)li * D + dch] = bf(w->skip_w[((size_t)l * S + sc) * D + dch]);
                }
                for (size_t i = 0; i < (size_t)E * S; ++i) wb[h->fwb_off_w1 + i] = bf(w->end1_w[i]);
                for (size_t i = 0; i < (size_t)C * E; ++i) wb[h->fwb_off_w2 + i] = bf(w->end2_w[i]);
                if (h->fwb_elems != ob) { rt_free(h->d_fwb); h->d_fwb = (unsigned short*)rt_malloc(ob * 2); h->fwb_elems = ob; }
                if (!h->d_fwb) return wn_fail(WN_E_NOMEM, "wn_load_weights: bf16 forward banks");
                rc = rt_h2d(h->d_fwb, wb.data(), ob * 2);
                if (rc) return rc;
            }
        }
    }
    h->have_weights = true;
    return WN_OK;
}

extern "C" int wn_reset(wn_handle* h, void* hip_stream) {
    g_err[0] = 0;
    if (!h) return wn_fail(WN_E_BADARG, "wn_reset: NULL handle");
    if (!h->chains.empty()) {
        if (h->pending) { int rc = wn_wait(h); if (rc) return rc; }
        for (wn_handle* c : h->chains) { int rc = wn_reset(c, hip_stream); if (rc) return rc; }
        h->t_base = 0;
        h->broken = false;
        return WN_OK;
    }
    { int rc = rt_hip(hipSetDevice(h->cfg.device_id), "hipSetDevice"); if (rc) return rc; }
    if (h->pending) { int rc = wn_wait(h); if (rc) return rc; }
    h->t_base = 0;
    return rt_memset_async(h->d_rings, 0, h->ring_floats * 4, hip_stream);
}


extern "C" int wn_generate(wn_handle* h, const wn_generate_args* a) {
    g_err[0] = 0;
    if (!h || !a) return wn_fail(WN_E_BADARG, "wn_generate: NULL argument");
    if (!h->have_weights) return wn_fail(WN_E_STATE, "wn_generate: wn_load_weights has not been called");
    if (!h->chains.empty()) {   // rounds: member i owns streams [chain_first[i], chain_first[i + 1]); one after the other on the caller's stream
        if (a->n_given < 1 || a->num_samples < 0) return wn_fail(WN_E_BADARG, "wn_generate: n_given must be >= 1 and num_samples >= 0");
        if (!a->first_samples) return wn_fail(WN_E_BADARG, "wn_generate: first_samples is NULL");
        { int rc = rt_hip(hipSetDevice(h->cfg.device_id), "hipSetDevice"); if (rc) return rc; }
        if (h->pending) { int rc = wn_wait(h); if (rc) return rc; }
        if (h->broken) return wn_fail(WN_E_STATE, "wn_generate: an earlier job failed half way through its rounds; call wn_reset");
        int rc = WN_OK;
        for (size_t i = 0; i < h->chains.size(); ++i) {
            wn_generate_args b = *a;
            const size_t s0 = (size_t)h->chain_first[i];
            b.first_samples = a->first_samples + s0 * (size_t)a->n_given;
            if (a->uniforms) b.uniforms = a->uniforms + s0 * (size_t)a->num_samples;
            if (a->out_idx) b.out_idx = a->out_idx + s0 * (size_t)a->num_samples;
            if (a->dbg_logits) b.dbg_logits = a->dbg_logits + s0 * (size_t)a->num_samples * h->plan.C;
            if (a->stream_temperatures) b.stream_temperatures = a->stream_temperatures + s0;
            if (i == 0 && h->prof_items > 0) { h->chains[0]->prof_items = h->prof_items; h->prof_items = 0; }
            rc = wn_generate(h->chains[i], &b);
            if (rc) {
                if (i > 0) {  // rounds 0..i-1 are enqueued: drain them, then refuse further jobs until the queues are reset
                    char msg[sizeof(g_err)];
                    memcpy(msg, g_err, sizeof(msg));
                    h->pending = true; h->last_stream = a->hip_stream;
                    (void)wn_wait(h);
                    h->broken = true;
                    memcpy(g_err, msg, sizeof(msg));
                }
                return rc;
            }
        }
        h->pending = true;
        h->last_stream = a->hip_stream;
        h->t_base = h->chains[0]->t_base;
        return WN_OK;
    }
    if (a->n_given < 1 || a->num_samples < 0) return wn_fail(WN_E_BADARG, "wn_generate: n_given must be >= 1 and num_samples >= 0");
    if (!a->first_samples) return wn_fail(WN_E_BADARG, "wn_generate: first_samples is NULL");
    if (a->num_samples > 0 && !a->out_idx) return wn_fail(WN_E_BADARG, "wn_generate: out_idx is NULL");
    if (a->flags != 0 || a->reserved != 0) return wn_fail(WN_E_BADARG, "wn_generate: flags/reserved must be 0");
    const bool greedy = (!(a->temperature > 0.f) && !a->stream_temperatures) || a->uniforms == nullptr;
    const long long n_eval = a->n_given - 1 + a->num_samples;
    if (n_eval + 1 >= 0xFFFFFFFFll || (n_eval + 1) * (long long)h->plan.n_streams >= 0xFFFFFFFFll)   // (re-used slots count pipeline items, not evaluations)
        return wn_fail(WN_E_BADARG, "wn_generate: job too long for 32-bit hand-off tags");
    { int rc = rt_hip(hipSetDevice(h->cfg.device_id), "hipSetDevice"); if (rc) return rc; }
    if (h->pending) { int rc = wn_wait(h); if (rc) return rc; }
    if (n_eval == 0) return WN_OK;
    WnRun r;
    memset(&r, 0, sizeof(r));
    r.first = a->first_samples; r.n_given = a->n_given; r.num_samples = a->num_samples; r.n_eval = n_eval;
    r.t_base = h->t_base; r.temperature = a->temperature; r.greedy = greedy ? 1 : 0; r.reg = a->regularizer;
    r.uniforms = a->uniforms; r.out_idx = a->out_idx; r.dbg_logits = a->dbg_logits; r.stream_temps = a->stream_temperatures;
    const long long ms = a->timeout_ms > 0 ? a->timeout_ms : 10000;
    r.timeout_ticks = ms * (long long)h->wall_khz;
    {   // the start-up residency barrier has a bound of its own: a job that finds CUs taken by another kernel starts when they free up
        const char* re = getenv("WN_RESIDENT_TIMEOUT_MS");
        const long long rms = re && atoll(re) > 0 ? atoll(re) : 60000;
        r.resident_ticks = rms * (long long)h->wall_khz;
        h->resident_ms = (int)(rms > 0x7fffffff ? 0x7fffffff : rms);
    }
    {   // CUs this stream may use right now (a CU mask on the stream, or the process-wide one): fewer than the job keeps resident = it could never start
        uint32_t mask[32];
        memset(mask, 0, sizeof(mask));
        if (hipExtStreamGetCUMask((hipStream_t)a->hip_stream, 32, mask) == hipSuccess) {
            int cus = 0;
            for (int i = 0; i < 32; ++i) cus += __builtin_popcount(mask[i]);
            const int per_cu = h->wg_per_cu > 0 ? h->wg_per_cu : 1;
            if (cus > 0 && (long long)cus * per_cu < h->plan.n_wg)
                return wn_fail(WN_E_UNSUPPORTED, "wn_generate: the stream's CU mask leaves %d compute units (%d workgroup%s of this kernel each), the job keeps %d "
                               "workgroups resident; nothing was launched", cus, per_cu, per_cu == 1 ? "" : "s", h->plan.n_wg);
        } else {
            (void)hipGetLastError();
        }
    }
    if (h->prof_items > 0) {
        rt_free(h->d_prof);
        const size_t nb = (size_t)h->plan.n_wg * h->prof_items * 8 * sizeof(long long);
        h->d_prof = (long long*)rt_malloc(nb);
        if (!h->d_prof) return wn_fail(WN_E_NOMEM, "wn_generate: profile buffer");
        int prc = rt_memset_async(h->d_prof, 0, nb, a->hip_stream);
        if (prc) return prc;
        r.prof = h->d_prof; r.prof_items = h->prof_items;
        h->prof_recorded = h->prof_items;
        h->prof_items = 0;
    }
    {   // admission: a persistent job only runs once ALL its workgroups are resident (wn_gate.h)
        const char* off = wn_dev_env("WN_NO_DEVICE_GATE");
        wn_gate_release(h->gate);
        h->gate.reset();
        if (!(off && off[0] == '1')) {
            int need = 0, cap = 0;
            wn_gate_numbers(h, &need, &cap);
            const char* te = getenv("WN_GATE_TIMEOUT_MS");
            const long long gate_ms = te && atoll(te) > 0 ? atoll(te) : 600000;
            long long waited = 0;
            int shared = 0;
            if (wn_gate_acquire(h->busid, cap, need, a->hip_stream, gate_ms, &h->gate, &waited, &shared))
                return wn_fail(WN_E_TIMEOUT, "wn_generate: device %s stayed booked by other persistent jobs for %lld ms (this job needs %d of %d CUs per XCD); "
                               "nothing was launched", h->busid, waited, need, cap);
            h->gate_shared = shared; h->gate_waited_ms = (int)(waited > 0x7fffffff ? 0x7fffffff : waited);
        }
    }
    // hand-off words restart at tag 1 every call: zero them (and the status word) ahead of the launch
    int rc = rt_memset_async(h->d_gran, 0, h->gran_count * 8, a->hip_stream);
    rc = rc ? rc : rt_memset_async(h->d_status, 0, (size_t)(8 + h->plan.n_wg) * 4, a->hip_stream);
    if (rc) { wn_gate_release(h->gate); h->gate.reset(); return rc; }
    if (h->variant == 4)
        wn_v4_table()[h->v2_index].launch(h->plan.n_blocks, (size_t)h->lds_bytes, (hipStream_t)a->hip_stream, h->plan, r);
    else if (h->variant == 3)
        wn_v2_table()[h->v2_index].launch_v3(h->v3_slots ? 2 : (h->v3_mode & 1), h->plan.n_blocks, (size_t)h->lds_bytes, (hipStream_t)a->hip_stream, h->plan, r);
    else
        hipLaunchKernelGGL(wn_generate_kernel, dim3(h->plan.n_blocks), dim3(WN_THREADS), (size_t)h->lds_bytes,
                           (hipStream_t)a->hip_stream, h->plan, r);
    rc = rt_hip(hipGetLastError(), "launch wn_generate_kernel");
    if (rc) { wn_gate_release(h->gate); h->gate.reset(); return rc; }
    if (h->gate) {   // the booking goes back the moment the kernel is done (a caller that never waits must not keep the device closed)
        std::shared_ptr<WnGateTicket>* t = new std::shared_ptr<WnGateTicket>(h->gate);
        if (hipLaunchHostFunc((hipStream_t)a->hip_stream, wn_gate_host_release, t) != hipSuccess) { (void)hipGetLastError(); delete t; }  // (wn_wait returns it then)
    }
    h->pending = true;
    h->last_stream = a->hip_stream;
    h->t_base += n_eval;
    h->last_n_eval = n_eval;
    return WN_OK;
}

extern "C" int wn_wait(wn_handle* h) {
    if (!h) return wn_fail(WN_E_BADARG, "wn_wait: NULL handle");
    if (!h->pending) return WN_OK;
    h->pending = false;
    if (!h->chains.empty()) {
        // Every round is a kernel of its own with its own residency barrier: "nothing ran, repeat the call" (WN_E_BUSY) is only true of the JOB when
        // it is true of EVERY round.  A round that gave up while another ran to completion leaves the rounds' queue times apart -- a repeated call
        // would advance the completed rounds twice -- so a mixed outcome closes the handle until wn_reset, like a launch that failed half way.
        int first_rc = WN_OK, n_busy = 0, n_ok = 0;
        char msg[sizeof(g_err)] = "";
        for (wn_handle* c : h->chains) {
            const int rc = wn_wait(c);
            if (rc == WN_E_BUSY) ++n_busy;
            else if (rc == WN_OK) ++n_ok;
            if (rc && (!first_rc || (first_rc == WN_E_BUSY && rc != WN_E_BUSY))) { first_rc = rc; memcpy(msg, g_err, sizeof(msg)); }   // (a hard error outranks BUSY)
        }
        if (!first_rc) return WN_OK;
        if (n_busy == (int)h->chains.size()) {   // no round started: every member rolled its own queue time back, the parent follows
            h->t_base = h->chains[0]->t_base;
            memcpy(g_err, msg, sizeof(msg));
            return WN_E_BUSY;
        }
        h->broken = true;
        if (first_rc == WN_E_BUSY)
            return wn_fail(WN_E_STATE, "wn_generate: %d of the job's %d rounds never became resident (WN_RESIDENT_TIMEOUT_MS) while %d ran: the rounds' queues are "
                           "out of step -- call wn_reset", n_busy, (int)h->chains.size(), n_ok);
        memcpy(g_err, msg, sizeof(msg));
        return first_rc;
    }
    int rc = rt_sync(h->last_stream);
    wn_gate_release(h->gate);
    h->gate.reset();
    if (rc) return rc;
    uint32_t st[8];
    rc = rt_d2h(st, h->d_status, 32);
    if (rc) return rc;
    if (st[0] != 0 && st[4] == 5) {   // WN_W_RESIDENT: the job never started -- no workgroup entered the chain, queues and rings are as they were
        h->t_base -= h->last_n_eval;
        h->last_n_eval = 0;
        return wn_fail(WN_E_BUSY,
                       "wn_generate: only %u of the job's %d workgroups had become resident after %d ms (WN_RESIDENT_TIMEOUT_MS): the device's compute units "
                       "are held by other kernels (or masked); nothing was generated and the queues are unchanged -- the call can be repeated",
                       st[2], h->plan.n_wg, h->resident_ms);
    }
    if (st[0] != 0) {
        static const char* where[] = {"?", "partial logits (head -> L0)", "x partials (layer -> layer)", "skip lane (layer -> layer)",
                                      "skip lanes (last layer -> head)"};
        return wn_fail(WN_E_TIMEOUT,
                       "wn_generate: hand-off wait gave up at chain position %u (eval %u, stream %u) waiting for %s; "
                       "queues are in an undefined state -- call wn_reset",
                       st[1], st[2], st[3], where[st[4] < 5 ? st[4] : 0]);
    }
    return WN_OK;
}

extern "C" int wn_get_info(wn_handle* h, wn_info* out) {
    if (!h || !out) return wn_fail(WN_E_BADARG, "wn_get_info: NULL argument");
    if (!h->chains.empty()) {  // both chains have the same geometry; sizes and workgroups add up
        int rc = wn_get_info(h->chains[0], out);
        if (rc) return rc;
        for (size_t i = 1; i < h->chains.size(); ++i) {
            wn_info ci;
            if ((rc = wn_get_info(h->chains[i], &ci))) return rc;
            out->n_workgroups += ci.n_workgroups; out->queue_bytes += ci.queue_bytes;
            if (!h->chains[i]->shares_weights) out->weight_bytes += ci.weight_bytes;   // (after wn_load_weights the rounds share member 0's)
            out->handoff_bytes += ci.handoff_bytes;
        }
        out->n_chains = (int)h->chains.size();
        for (wn_handle* c : h->chains) if (c->gate_waited_ms > out->gate_waited_ms) out->gate_waited_ms = c->gate_waited_ms;
        return WN_OK;
    }
    const WnPlan& pl = h->plan;
    memset(out, 0, sizeof(*out));
    out->abi_version = WN_ABI_VERSION;
    out->n_layers = pl.NL; out->layer_split = pl.P; out->head_split = pl.PA; out->n_workgroups = pl.n_wg;
    out->lds_bytes = h->lds_bytes; out->n_compute_units = h->n_cu; out->kernel_variant = h->variant;
    out->receptive_field = 1 + pl.blocks * (pl.k - 1) * ((1 << pl.layers) - 1);
    out->weight_bytes = (int64_t)h->blob_floats * 4 + (int64_t)pl.C * pl.R * 4;
    out->queue_bytes = (int64_t)h->ring_floats * 4;
    out->handoff_bytes = (int64_t)h->gran_count * 8;
    out->evals_done = h->t_base;
    out->n_chains = 1;
    out->streams_per_item = (h->variant == 3 && (h->v3_mode & 1)) ? 2 : 1;
    out->head_replicas = pl.HR;
    out->n_samplers = pl.n_smp;
    out->dev_overrides = h->dev_overrides;
    out->layers_per_workgroup = h->variant == 4 ? pl.LPW : 1;
    if (h->variant == 4) out->n_workgroups = pl.n_wg;
    out->gate_shared = h->gate_shared; out->gate_waited_ms = h->gate_waited_ms;
    { int need = 0, cap = 0; wn_gate_numbers(h, &need, &cap); out->gate_need_per_xcd = need; }
    out->forward_native = (h->have_weights && h->fw_ok) ? 1 : 0;
    out->workgroups_per_cu = h->wg_per_cu; out->resident_timeout_ms = h->resident_ms; out->skip_lane_slots = h->variant == 3 ? h->v3_slots : 0;
    return WN_OK;
}

extern "C" int wn_export_queue(wn_handle* h, int32_t layer, int32_t stream, float* host_data, int32_t* in_pos, int32_t* out_pos) {
    g_err[0] = 0;
    if (!h || !host_data) return wn_fail(WN_E_BADARG, "wn_export_queue: NULL argument");
    if (!h->chains.empty()) {
        if (stream < 0 || stream >= h->plan.n_streams) return wn_fail(WN_E_BADARG, "wn_export_queue: index out of range");
        if (h->pending) { int rc = wn_wait(h); if (rc) return rc; }
        size_t i = 0;
        while (stream >= h->chain_first[i + 1]) ++i;
        return wn_export_queue(h->chains[i], layer, stream - h->chain_first[i], host_data, in_pos, out_pos);
    }
    const WnPlan& pl = h->plan;
    if (layer < 0 || layer >= pl.NL || stream < 0 || stream >= pl.n_streams) return wn_fail(WN_E_BADARG, "wn_export_queue: index out of range");
    { int rc = rt_hip(hipSetDevice(h->cfg.device_id), "hipSetDevice"); if (rc) return rc; }
    if (h->pending) { int rc = wn_wait(h); if (rc) return rc; }
    const int d = h->dil[layer];
    const int ML = (pl.k - 1) * d + 1;
    std::vector<float> tmp((size_t)ML * pl.R);
    // slice c = 0 holds the layer's queue (all P slices keep identical copies)
    int rc = rt_d2h(tmp.data(), h->d_rings + h->ring_off[layer] + (size_t)stream * ML * pl.R, tmp.size() * 4);
    if (rc) return rc;
    const int R_out = h->export_R ? h->export_R : pl.R;   // (zero padding: the caller's channels are the first ones)
    for (int r = 0; r < R_out; ++r)
        for (int m = 0; m < ML; ++m) host_data[(size_t)r * ML + m] = tmp[(size_t)m * pl.R + r];  // (slot, R) -> (R, max_length)
    if (in_pos) *in_pos = (int32_t)(h->t_base % ML);  // one enqueue + one dequeue per evaluation
    if (out_pos) *out_pos = (int32_t)(h->t_base % ML);
    return WN_OK;
}

// Diagnostics (not part of the reference surface): wall-clock stamps of the first n_items (eval, stream) steps of
// every workgroup of the NEXT wn_generate, 4 stamps each (start, input staged, x' published, done), 100 MHz ticks.
extern "C" int wn_profile_next(wn_handle* h, int32_t n_items) {
    if (!h || n_items < 0) return wn_fail(WN_E_BADARG, "wn_profile_next: bad argument");
    if (!h->chains.empty()) return wn_profile_next(h->chains[0], n_items);  // stamps of the first chain
    h->prof_items = n_items;
    return WN_OK;
}

extern "C" int wn_profile_read(wn_handle* h, int64_t* host_out, int64_t capacity) {
    if (!h || !host_out) return wn_fail(WN_E_BADARG, "wn_profile_read: NULL argument");
    if (!h->chains.empty()) {
        if (h->pending) { int rc = wn_wait(h); if (rc) return rc; }
        return wn_profile_read(h->chains[0], host_out, capacity);
    }
    if (h->pending) { int rc = wn_wait(h); if (rc) return rc; }
    const int64_t n = (int64_t)h->plan.n_wg * h->prof_recorded * 8;
    if (!h->d_prof || n == 0 || capacity < n) return wn_fail(WN_E_STATE, "wn_profile_read: nothing recorded / buffer too small (%lld)", (long long)n);
    return rt_d2h(host_out, h->d_prof, (size_t)n * 8);
}

// Time geometry of WaveNetModel.forward() for clips of L samples (wavenet_modules.py:10-39 `dilate`, wavenet_model.py:125-196).
// In absolute time every layer's sequence ends at L (a kernel-size-2 dilated conv drops its input's first d positions).  Where the
// length of a layer's input is not a multiple of its dilation the reference left-pads it with ZERO ACTIVATIONS (wavenet_modules.py:24-27),
// so layer l's input lives on [a[l], L) preceded by pad[l] = (-(L - a[l])) mod d zeros, and its output on [a[l+1], L) with
// a[l+1] = a[l] - pad[l] + d.  With L >= receptive_field + output_length - 1 none of the returned positions can see a pad zero (the
// regime of rounds 1-3); shorter clips can: the tap x(t - d) then reads as zero for t - d < a[l] (row windows of the GEMMs' A views).
//   rows[l] = trailing positions of layer l's input that are computed = min(rows[l+1] + d, L - a[l]);   zlo[l] = leading output rows of
//   layer l whose tap is a pad zero.
// Returns WN_E_UNSUPPORTED where the reference itself has no defined result: a layer left with no output position, the skip
// un-dilation quirk at a per-row length of 1, fewer than output_length final positions (its view fails).
static int wn_forward_geometry(const wn_handle* h, long long L, long long out_len, WnFwdGeom& g, const char* who) {
    const std::string why = wn_forward_geometry_host(h->dil.data(), h->plan.NL, L, out_len, g);   // (wn_plan.h: plain host arithmetic, tested with g++)
    if (!why.empty()) return wn_fail(WN_E_UNSUPPORTED, "%s: %s", who, why.c_str());
    return WN_OK;
}

// WaveNetModel.forward() for one-hot inputs (class indices), see wn_forward.h.  Asynchronous on hip_stream.
extern "C" int wn_forward(wn_handle* h, const int32_t* indices, int64_t N, int64_t L, int64_t out_len, float* logits, void* hip_stream) {
    g_err[0] = 0;
    if (!h || !indices || !logits) return wn_fail(WN_E_BADARG, "wn_forward: NULL argument");
    if (!h->chains.empty()) return wn_forward(h->chains[0], indices, N, L, out_len, logits, hip_stream);  // every chain holds the weights
    if (!h->have_weights) return wn_fail(WN_E_STATE, "wn_forward: wn_load_weights has not been called");
    if (N < 1 || out_len < 1) return wn_fail(WN_E_BADARG, "wn_forward: N and output_length must be >= 1");
    const WnPlan& pl = h->plan;
    const int R = pl.R, D = pl.D, S = pl.S, E = pl.E, C = pl.C, NL = pl.NL;
    if (!h->fw_ok) return wn_fail(WN_E_UNSUPPORTED, "wn_forward: needs kernel_size 2 and channel counts that are multiples of 32");
    if ((long long)N * L >= 0x7fffffffll) return wn_fail(WN_E_UNSUPPORTED, "wn_forward: N*L must stay below 2^31 rows");
    { int rc = rt_hip(hipSetDevice(h->cfg.device_id), "hipSetDevice"); if (rc) return rc; }
    WnFwdGeom geo;
    { int rc = wn_forward_geometry(h, L, out_len, geo, "wn_forward"); if (rc) return rc; }
    const std::vector<long long>& need = geo.rows;
    const size_t x_fl = (size_t)N * L * R, z_fl = (size_t)N * need[1 < NL ? 1 : NL] * D > (size_t)N * need[NL] * D ? (size_t)N * need[1 < NL ? 1 : NL] * D : (size_t)N * need[NL] * D;
    // The skip sum over layers is accumulated G layers at a time: the gate epilogue also drops z (last output_length rows)
    // into column block (l mod G) of ZG [N*out_len][G*D], and one GEMM with K = G*D adds the group to SKIP -- instead of a
    // read-modify-write of the whole SKIP matrix per layer (1.4 GB per layer at config 5).
    const int G = pl.layers < NL ? pl.layers : NL;
    const size_t skip_fl = (size_t)N * out_len * S, e_fl = (size_t)N * out_len * E, zg_fl = (size_t)N * out_len * G * D;
    // bf16 operands at the 128 / 128 shape: a layer is ONE launch (wn_fwd_layer_bf16: z goes from the gate epilogue to the residual product
    // through LDS and is never stored), its matrix operand reads of x take a bf16 shadow written next to x (WnGemmArgs::c_h), z on the skip
    // rows (zg) is stored as bf16.  Same roundings as the two-launch form (every value is rounded to bf16 once, where it becomes an operand).
    const bool fuse = h->fw_bf16 && h->fwb_ok && R == 128 && D == 128 && wn_fused_layer_enabled();
    const size_t xh_fl = fuse ? ((x_fl + 1) / 2 + 63) / 64 * 64 : 0;
    const size_t total = 2 * x_fl + z_fl + skip_fl + e_fl + zg_fl + 2 * xh_fl;
    if (h->ws_floats < total) {
        if (h->pending) { int rc = wn_wait(h); if (rc) return rc; }
        rt_free(h->d_ws);
        h->d_ws = (float*)rt_malloc(total * 4);
        h->ws_floats = h->d_ws ? total : 0;
        if (!h->d_ws) return wn_fail(WN_E_NOMEM, "wn_forward: workspace of %.1f MB", total * 4e-6);
    }
    float* xa = h->d_ws; float* xb = xa + x_fl; float* z = xb + x_fl; float* skip = z + z_fl; float* ev = skip + skip_fl;
    float* zg = ev + e_fl;
    unsigned short* xha = fuse ? reinterpret_cast<unsigned short*>(zg + zg_fl) : nullptr;
    unsigned short* xhb = fuse ? reinterpret_cast<unsigned short*>(zg + zg_fl + xh_fl) : nullptr;
    hipStream_t st = (hipStream_t)hip_stream;
    {
        const long long rows = N * L;
        const long long work = rows * (R / 4);
        hipLaunchKernelGGL(wn_fwd_start, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, st, indices, h->d_start_t,
                           pl.has_bias ? h->d_start_b : nullptr, xa, rows, R, xha);
    }
    const bool bf16 = h->fw_bf16 && h->fwb_ok;
    auto launch = [&](int epi, const WnGemmArgs& a, const unsigned short* bn) { wn_launch_nn(st, epi, a, bf16 ? bn : nullptr); };
    const unsigned short* fwb = h->d_fwb;
    const float* fw = h->d_fw;
    float* xin = xa; float* xout = xb;
    unsigned short* xhin = xha; unsigned short* xhout = xhb;
    for (int l = 0; l < NL; ++l) {
        const long long d = h->dil[l], rows = need[l + 1], t0 = L - rows;
        const int gi = l % G;
        WnGemmArgs a;
        memset(&a, 0, sizeof(a));
        // z = gate([x(t-d) | x(t)] . Wfg^T)
        const float* xop = fuse ? reinterpret_cast<const float*>(xhin) : xin;   // (fuse: the bf16 shadow, the row maps count bf16 elements)
        a.a0 = WnRowMap{xop, (long long)L * R, R, t0 - d};
        a.a_skip_lo[0] = (int)geo.zlo[l];   // (short clips: the reference's left zero padding stands in for x(t - d) there)
        a.a1 = WnRowMap{xop, (long long)L * R, R, t0};
        a.a_bf16 = fuse ? 1 : 0;
        a.k_split = R; a.K = 2 * R; a.bt = fw + h->fw_off_fg + (size_t)l * 2 * R * 2 * D; a.N = 2 * D;
        a.bias = pl.has_bias ? fw + h->fw_off_bfg + (size_t)l * 2 * D : nullptr;
        a.c = WnRowMap{z, rows * D, D, 0};
        a.c_bf16 = fuse ? 1 : 0;   // (fuse: z and zg hold bf16)
        a.c2 = WnRowMap{fuse ? reinterpret_cast<float*>(reinterpret_cast<unsigned short*>(zg) + (size_t)gi * D) : zg + (size_t)gi * D, out_len * (long long)G * D, (long long)G * D, 0};
        a.c2_first_row = (int)(rows - out_len);
        a.M = N * rows; a.rows_per_batch = (int)rows;
        WnGemmArgs ar;  // x' = z . Wres^T + x(t)   (the last layer's residual output is never consumed, also upstream)
        memset(&ar, 0, sizeof(ar));
        if (l < NL - 1) {
            ar.a0 = ar.a1 = WnRowMap{z, rows * D, D, 0};
            ar.a_bf16 = fuse ? 1 : 0;
            ar.k_split = D; ar.K = D; ar.bt = fw + h->fw_off_res + (size_t)l * D * R; ar.N = R;
            ar.bias = pl.has_bias ? fw + h->fw_off_bres + (size_t)l * R : nullptr;
            ar.cin = WnRowMap{xin, (long long)L * R, R, t0};
            ar.c = WnRowMap{xout, (long long)L * R, R, t0};
            ar.c_h = fuse ? xhout : nullptr;
            ar.M = N * rows; ar.rows_per_batch = (int)rows;
        }
        bool fused = false;
        if (fuse && l < NL - 1) {
            WnGemmArgs af = a;
            af.c.base = nullptr;   // z itself is not stored: nothing reads it again
            fused = wn_launch_layer(st, af, fwb + h->fwb_off_fg + (size_t)l * 2 * D * 2 * R, ar, fwb + h->fwb_off_res + (size_t)l * R * D);
        }
        if (!fused) {
            launch(WN_EPI_GATE, a, bf16 ? fwb + h->fwb_off_fg + (size_t)l * 2 * D * 2 * R : nullptr);
            if (l < NL - 1) launch(WN_EPI_PLAIN, ar, bf16 ? fwb + h->fwb_off_res + (size_t)l * R * D : nullptr);
        }
        if (gi == G - 1 || l == NL - 1) {  // skip (+)= ZG . [Wskip of the group's layers]^T   (K = layers_in_group * D)
            const int first = l - gi, cnt = gi + 1;
            memset(&a, 0, sizeof(a));
            a.a0 = a.a1 = WnRowMap{zg, out_len * (long long)G * D, (long long)G * D, 0};
            a.a_bf16 = fuse ? 1 : 0;
            a.k_split = cnt * D; a.K = cnt * D; a.bt = fw + h->fw_off_skip + (size_t)first * D * S; a.N = S;
            a.bias = (pl.has_bias && first == 0) ? fw + h->fw_off_bskip_total : nullptr;
            if (first > 0) a.cin = WnRowMap{skip, out_len * S, S, 0};
            a.c = WnRowMap{skip, out_len * S, S, 0};
            a.M = N * out_len; a.rows_per_batch = (int)out_len;
            launch(WN_EPI_PLAIN, a, bf16 ? fwb + h->fwb_off_skip + (size_t)(first / G) * S * G * D : nullptr);
        }
        float* t = xin; xin = xout; xout = t;
        unsigned short* th = xhin; xhin = xhout; xhout = th;
    }
    {   // head: relu(skip) -> end_conv_1 (+b, relu) -> end_conv_2 (+b)     wavenet_model.py:167-169
        WnGemmArgs a;
        memset(&a, 0, sizeof(a));
        a.a0 = a.a1 = WnRowMap{skip, out_len * S, S, 0};
        a.k_split = S; a.K = S; a.bt = fw + h->fw_off_w1; a.N = E; a.bias = fw + h->fw_off_b1;
        a.c = WnRowMap{ev, out_len * E, E, 0};
        a.M = N * out_len; a.rows_per_batch = (int)out_len; a.relu_a = 1; a.relu_c = 1;
        launch(WN_EPI_PLAIN, a, bf16 ? fwb + h->fwb_off_w1 : nullptr);
        memset(&a, 0, sizeof(a));
        a.a0 = a.a1 = WnRowMap{ev, out_len * E, E, 0};
        a.k_split = E; a.K = E; a.bt = fw + h->fw_off_w2; a.N = C; a.bias = fw + h->fw_off_b2;
        a.c = WnRowMap{logits, out_len * C, C, 0};
        a.M = N * out_len; a.rows_per_batch = (int)out_len;
        launch(WN_EPI_PLAIN, a, bf16 ? fwb + h->fwb_off_w2 : nullptr);
    }
    return rt_hip(hipGetLastError(), "wn_forward launches");
}

// Batched (teacher-forced) priming: the n_prime = n_given - 1 priming evaluations of generate_fast (wavenet_model.py:259-269)
// as GEMMs over all given positions at once instead of one chain pass per sample: the layer
// inputs of the whole window are computed with the forward kernels (no skip / head work -- the reference discards those
// outputs) and the newest d+1 columns of every layer are written straight into the queues.  Requires freshly reset queues
// (queue time 0); activations before the stream start are zero at every layer, like DilatedQueue.reset().
extern "C" int wn_prime(wn_handle* h, const int32_t* first_samples, int64_t n_prime, int64_t row_stride, void* hip_stream) {
    g_err[0] = 0;
    if (!h || !first_samples) return wn_fail(WN_E_BADARG, "wn_prime: NULL argument");
    if (!h->chains.empty()) {
        if (n_prime < 0 || row_stride < n_prime) return wn_fail(WN_E_BADARG, "wn_prime: bad n_prime / row_stride");
        if (h->pending) { int rc = wn_wait(h); if (rc) return rc; }
        for (size_t i = 0; i < h->chains.size(); ++i) {
            int rc = wn_prime(h->chains[i], first_samples + (size_t)h->chain_first[i] * (size_t)row_stride, n_prime, row_stride, hip_stream);
            if (rc) return rc;
        }
        h->t_base = h->chains[0]->t_base;
        return WN_OK;
    }
    if (!h->have_weights) return wn_fail(WN_E_STATE, "wn_prime: wn_load_weights has not been called");
    if (n_prime < 0 || row_stride < n_prime) return wn_fail(WN_E_BADARG, "wn_prime: bad n_prime / row_stride");
    if (n_prime == 0) return WN_OK;
    const WnPlan& pl = h->plan;
    const int R = pl.R, D = pl.D, NL = pl.NL, ns = pl.n_streams;
    if (!h->fw_ok) return wn_fail(WN_E_UNSUPPORTED, "wn_prime: needs kernel_size 2 and channel counts that are multiples of 32");
    if (h->pending) { int rc = wn_wait(h); if (rc) return rc; }
    if (h->t_base != 0) return wn_fail(WN_E_STATE, "wn_prime: queues must be freshly reset (queue time is %lld)", h->t_base);
    if (row_stride != n_prime && ns > 1) { /* strided rows are fine: handled by the gather below */ }
    { int rc = rt_hip(hipSetDevice(h->cfg.device_id), "hipSetDevice"); if (rc) return rc; }
    const long long n = n_prime;
    if ((long long)ns * n >= 0x7fffffffll) return wn_fail(WN_E_UNSUPPORTED, "wn_prime: too many rows");
    // q[i] = trailing positions of layer i's input that are needed (its own queue: d+1, and what the layers above need)
    std::vector<long long> q(NL + 1, 0);
    for (int l = NL - 1; l >= 0; --l) {
        const long long d = h->dil[l];
        long long v = q[l + 1] > 0 ? q[l + 1] + d : 0;
        if (v < d + 1) v = d + 1;
        q[l] = v < n ? v : n;
    }
    long long max_d = 1;
    for (int l = 0; l < NL; ++l) max_d = h->dil[l] > max_d ? h->dil[l] : max_d;
    const long long Lp = max_d, Lt = Lp + n;  // every stream's activation rows are preceded by Lp rows of zeros (t < 0)
    const size_t x_fl = (size_t)ns * Lt * R, z_fl = (size_t)ns * n * D;
    const size_t total = 2 * x_fl + z_fl;
    if (h->ws_floats < total) {
        rt_free(h->d_ws);
        h->d_ws = (float*)rt_malloc(total * 4);
        h->ws_floats = h->d_ws ? total : 0;
        if (!h->d_ws) return wn_fail(WN_E_NOMEM, "wn_prime: workspace of %.1f MB", total * 4e-6);
    }
    float* xa = h->d_ws; float* xb = xa + x_fl; float* z = xb + x_fl;
    hipStream_t st = (hipStream_t)hip_stream;
    int rc = rt_hip(hipMemsetAsync(xa, 0, 2 * x_fl * 4, st), "hipMemsetAsync(prime workspace)");
    if (rc) return rc;
    // x0 = start_conv column gather over all given positions; rows of stream s start at xa + s*Lt*R + Lp*R
    for (int s = 0; s < ns; ++s) {
        const long long work = n * (R / 4);
        hipLaunchKernelGGL(wn_fwd_start, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, st, first_samples + (size_t)s * row_stride,
                           h->d_start_t, pl.has_bias ? h->d_start_b : nullptr, xa + ((size_t)s * Lt + Lp) * R, n, R);
    }
    auto launch = [&](int epi, const WnGemmArgs& a) { wn_launch_nn(st, epi, a); };
    const float* fw = h->d_fw;
    float* xin = xa; float* xout = xb;
    for (int l = 0; l < NL; ++l) {
        const long long d = h->dil[l];
        const int ML = (int)d + 1;
        {   // queue of layer l <- newest min(d+1, n) columns of its input
            const int count = (int)(ML < n ? ML : n);
            const long long work = (long long)ns * count * (R / 4);
            hipLaunchKernelGGL(wn_fill_ring, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, st, xin + Lp * R, Lt * R,
                               h->d_rings + h->ring_off[l], R, ML, ns, pl.P, n, count);
        }
        const long long rows = q[l + 1];
        if (l == NL - 1 || rows <= 0) break;
        const long long t0 = n - rows;
        WnGemmArgs a;
        memset(&a, 0, sizeof(a));
        a.a0 = WnRowMap{xin + Lp * R, Lt * R, R, t0 - d};  // t - d may be negative: those rows are the zero prefix
        a.a1 = WnRowMap{xin + Lp * R, Lt * R, R, t0};
        a.k_split = R; a.K = 2 * R; a.bt = fw + h->fw_off_fg + (size_t)l * 2 * R * 2 * D; a.N = 2 * D;
        a.bias = pl.has_bias ? fw + h->fw_off_bfg + (size_t)l * 2 * D : nullptr;
        a.c = WnRowMap{z, rows * D, D, 0};
        a.M = ns * rows; a.rows_per_batch = (int)rows;
        launch(WN_EPI_GATE, a);
        memset(&a, 0, sizeof(a));
        a.a0 = a.a1 = WnRowMap{z, rows * D, D, 0};
        a.k_split = D; a.K = D; a.bt = fw + h->fw_off_res + (size_t)l * D * R; a.N = R;
        a.bias = pl.has_bias ? fw + h->fw_off_bres + (size_t)l * R : nullptr;
        a.cin = WnRowMap{xin + Lp * R, Lt * R, R, t0};
        a.c = WnRowMap{xout + Lp * R, Lt * R, R, t0};
        a.M = ns * rows; a.rows_per_batch = (int)rows;
        launch(WN_EPI_PLAIN, a);
        float* t = xin; xin = xout; xout = t;
    }
    rc = rt_hip(hipGetLastError(), "wn_prime launches");
    if (rc) return rc;
    h->t_base = n;
    return WN_OK;
}

// Operand precision of wn_forward's GEMMs: 0 = fp32 (default; matches the reference's fp32 forward to rounding),
// 1 = bf16 operands with fp32 accumulation (the residual stream and all sums stay fp32).  wn_prime always runs fp32.
extern "C" int wn_set_forward_precision(wn_handle* h, int32_t bf16) {
    g_err[0] = 0;
    if (!h) return wn_fail(WN_E_BADARG, "wn_set_forward_precision: NULL handle");
    if (!h->chains.empty()) return wn_set_forward_precision(h->chains[0], bf16);
    if (bf16 && !h->have_weights) return wn_fail(WN_E_STATE, "wn_set_forward_precision: load the weights first");
    if (bf16 && !h->fwb_ok) return wn_fail(WN_E_UNSUPPORTED, "wn_set_forward_precision: bf16 needs R, D, S, E to be multiples of 64");
    h->fw_bf16 = bf16 ? 1 : 0;
    return WN_OK;
}

#include "wn_train.inl"

extern "C" int wn_adam_step(const wn_adam_args* a) {
    g_err[0] = 0;
    if (!a || a->n_tensors < 0 || (a->n_tensors > 0 && (!a->sizes || !a->params || !a->grads || !a->exp_avg || !a->exp_avg_sq)) || !a->scratch)
        return wn_fail(WN_E_BADARG, "wn_adam_step: NULL argument");
    if (a->step < 1 || !(a->beta1 >= 0. && a->beta1 < 1.) || !(a->beta2 >= 0. && a->beta2 < 1.) || !(a->eps >= 0.))
        return wn_fail(WN_E_BADARG, "wn_adam_step: step must be >= 1, betas in [0, 1), eps >= 0");
    if (a->flags & ~(int64_t)(WN_ADAM_NORM_ONLY | WN_ADAM_NORM_KEEP | WN_ADAM_NORM_GIVEN)) return wn_fail(WN_E_BADARG, "wn_adam_step: unknown flags");
    if ((a->flags & WN_ADAM_NORM_ONLY) && (a->flags & WN_ADAM_NORM_GIVEN)) return wn_fail(WN_E_BADARG, "wn_adam_step: NORM_ONLY and NORM_GIVEN exclude each other");
    // (no handle: the caller's current device is left as it was -- torch's current device is process state the parameters' device must not change)
    struct DeviceGuard {
        int prev = -1;
        ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
    } guard;
    { int cur = -1; if (hipGetDevice(&cur) == hipSuccess && cur != a->device_id) guard.prev = cur; else (void)hipGetLastError(); }
    { int rc = rt_hip(hipSetDevice(a->device_id), "hipSetDevice"); if (rc) return rc; }
    hipStream_t st = (hipStream_t)a->hip_stream;
    const bool norm_only = (a->flags & WN_ADAM_NORM_ONLY) != 0;
    const bool clip = a->max_grad_norm > 0. || norm_only;
    double* acc = static_cast<double*>(a->scratch);
    WnAdamScalars k;   // every fp32 scalar is the double torch forms in Python, rounded once (torch/optim/adam.py: _multi_tensor_adam)
    const double bc1 = 1.0 - pow(a->beta1, (double)a->step), bc2 = 1.0 - pow(a->beta2, (double)a->step);
    k.neg_step = (float)(-(a->lr / bc1)); k.sqrt_bc2 = (float)sqrt(bc2);
    k.one_minus_b1 = (float)(1.0 - a->beta1); k.b2 = (float)a->beta2; k.one_minus_b2 = (float)(1.0 - a->beta2); k.eps = (float)a->eps;
    k.weight_decay = (float)a->weight_decay; k.max_norm = (clip && !norm_only) ? (float)a->max_grad_norm : 0.f;
    k.lerp_hi = k.one_minus_b1 >= 0.5f ? 1 : 0;
    // batches of up to WN_OPT_TENSORS tensors (skipping the ones without a gradient: torch's optimisers do)
    std::vector<WnOptBatch> batches;
    WnOptBatch b;
    memset(&b, 0, sizeof(b));
    auto flush = [&]() { if (b.n > 0) { batches.push_back(b); memset(&b, 0, sizeof(b)); } };
    for (int i = 0; i < a->n_tensors; ++i) {
        if (!a->grads[i] || a->sizes[i] <= 0) continue;
        if (!a->params[i] || !a->exp_avg[i] || !a->exp_avg_sq[i]) return wn_fail(WN_E_BADARG, "wn_adam_step: tensor %d has a gradient but no parameter / state pointer", i);
        const long long chunks = (a->sizes[i] + WN_OPT_CHUNK - 1) / WN_OPT_CHUNK;
        if (b.n == WN_OPT_TENSORS || (long long)b.chunk0[b.n] + chunks > 0x3fffffffll) flush();
        b.p[b.n] = static_cast<float*>(a->params[i]); b.g[b.n] = static_cast<float*>(a->grads[i]);
        b.m[b.n] = static_cast<float*>(a->exp_avg[i]); b.v[b.n] = static_cast<float*>(a->exp_avg_sq[i]);
        b.size[b.n] = a->sizes[i];
        b.chunk0[b.n + 1] = b.chunk0[b.n] + (int)chunks;
        b.n++;
    }
    flush();
    // The norm of a clipped step is the norm of ALL gradients that are clipped together (clip_grad_norm_(model.parameters())): a caller with several
    // parameter groups first adds every group's sum of squares into `scratch` (NORM_ONLY; NORM_KEEP from the second group on), then steps each group on
    // the total (NORM_GIVEN).  One group: one call, no flags.
    if (clip && !(a->flags & WN_ADAM_NORM_GIVEN)) {
        if (!(a->flags & WN_ADAM_NORM_KEEP)) {
            int rc = rt_hip(hipMemsetAsync(acc, 0, sizeof(double), st), "hipMemsetAsync(norm)");
            if (rc) return rc;
        }
        for (const WnOptBatch& bb : batches) hipLaunchKernelGGL(wn_opt_sumsq, dim3((unsigned)bb.chunk0[bb.n]), dim3(256), 0, st, bb, acc);
    }
    if (norm_only) return rt_hip(hipGetLastError(), "wn_adam_step launches");
    for (const WnOptBatch& bb : batches)
        hipLaunchKernelGGL(wn_opt_adam, dim3((unsigned)bb.chunk0[bb.n]), dim3(256), 0, st, bb, k, clip ? acc : nullptr, clip ? a->total_norm : nullptr);
    return rt_hip(hipGetLastError(), "wn_adam_step launches");
}
