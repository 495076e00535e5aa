// wn_forward.h -- batched (training-time) forward of the dilated-conv stack on the matrix cores (gfx950, device only).
//
// Reference: WaveNetModel.forward() = wavenet() with wavenet_dilate (wavenet_model.py:125-196, wavenet_modules.py:10-39).
// The reference turns every dilated conv into a dense k=2 conv by folding time into the batch dimension (two full
// tensor copies per layer) and evaluates the skip conv on every position of every layer.  Here each layer is three
// GEMMs on the time-major activation matrix X[(n,t)][channels] (fp32 in HBM, channels contiguous):
//
//     z      = tanh(F) * sigmoid(G),  [F | G] = [X(t-d) | X(t)] . Wfg^T (+b)      K = 2R  (the two taps are two row-shifted
//                                                                                 views of the same matrix: no copy)
//     X'     = z . Wres^T (+b) + X(t)                                            K = D
//     SKIP  += z(last output_length positions only) . Wskip^T (+b)               K = D
//
// evaluated only on the positions that can still reach the returned outputs (need_i = output_length + sum of the
// dilations above layer i), then  logits = W2 . relu(W1 . relu(SKIP) + b1) + b2  on the last output_length positions.
// The default GEMMs run on v_mfma_f32_32x32x2_f32: fp32 in, fp32 accumulate, bit-for-bit an fmaf chain (MI355X guide), so the
// result matches the reference's fp32 forward to rounding (tests: 1e-4).  wn_set_forward_precision selects the bf16 forms
// (v_mfma_f32_32x32x16_bf16, fp32 accumulation): operands rounded while they are staged, and in the training step the activations
// that only ever feed such operands stored as bf16 -- the products are HBM streams (K = 128-512), so the bytes are what they cost.
// The same file holds the backward products of the training step (wn_train.inl is their host side).
//
// Valid when every returned position has a full receptive field: L >= receptive_field + output_length - 1 (the
// reference's own training shape, train_script.py:39).  Shorter inputs hit the reference's zero-padding quirk
//  and are served by the torch path of the facade.
#ifndef WN_FORWARD_H
#define WN_FORWARD_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "wn_plan.h"

typedef float wn_f16v __attribute__((ext_vector_type(16)));

// address of logical row m of a (batch, time) matrix: base + (m / rows_per_batch) * batch_stride + (t0 + m % rows_per_batch) * row_stride
struct WnRowMap {
    const float* base;
    long long batch_stride, row_stride, t0;
};
static __device__ __forceinline__ const float* wn_row(const WnRowMap& r, long long m, int rows_per_batch) {
    // M < 2^31 (checked on the host): 32-bit division, a 64-bit one is ~100 instructions and the epilogue does 32 of them
    const unsigned q = (unsigned)m / (unsigned)rows_per_batch, rem = (unsigned)m - q * (unsigned)rows_per_batch;
    return r.base + (long long)q * r.batch_stride + (r.t0 + (long long)rem) * r.row_stride;
}

enum { WN_EPI_PLAIN = 0, WN_EPI_GATE = 1, WN_EPI_GATE_BWD = 2 };

// (workgroup -> tile mapping of the weight-gradient products: wn_tile_of, wn_plan.h)


struct WnGemmArgs {
    WnRowMap a0, a1;      // A = [a0 (k < k_split) | a1 (k >= k_split)], rows of K floats in total
    int k_split, K;       // K % 32 == 0, k_split % 32 == 0
    const float* bt;      // B^T [K][N] row-major (N = logical output columns, N % 32 == 0; % 64 with WN_EPI_GATE)
    int N;
    const float* bias;    // [N] or NULL
    WnRowMap cin;         // optional addend (residual / accumulate): row of N_out floats, base == NULL -> none
    WnRowMap c;           // output rows of N_out floats (N_out = N, or N/2 with WN_EPI_GATE)
    long long M;          // logical rows
    int rows_per_batch;
    int relu_a, relu_c;
    const float* mask;    // WN_EPI_PLAIN: optional [rows of N floats, laid out like c]: output forced to 0 where mask <= 0 (ReLU backward)
    float* gate_t;        // WN_EPI_GATE: optional [M][N/2] dense copies of tanh(F) and sigmoid(G) for the backward pass (row = m)
    float* gate_g;
    WnRowMap c2;          // WN_EPI_GATE only: rows whose index inside the batch entry is >= c2_first_row are ALSO written here
    int c2_first_row;     //   (at row index - c2_first_row): the z block the grouped skip GEMM consumes.  base == NULL -> off
    int gate_packed;      // WN_EPI_GATE: gate_t receives ONE dword per element, {bf16 tanh (low half), bf16 sigmoid (high half)}; gate_g unused
    // Row windows: view v of A reads as ZERO on the first a_skip_lo[v] and the last a_skip_hi[v] rows of every batch entry, cin adds
    // nothing on the first cin_skip_lo rows (their addresses are never formed into loads).  One product can then sum two
    // row-shifted views of a matrix whose shifts run off its ends (the backward's dx_l = dx' + dfg(t).W1 + dfg(t+d).W0).
    int a_skip_lo[2], a_skip_hi[2], cin_skip_lo;
    const float* bt1;     // optional: rows k >= k_split of B^T come from here (row k - k_split), so the two halves need not be adjacent
    // bf16 STORAGE (the bf16 training step's [dF|dG]; bf16 kernels only).  The row maps of a bf16 matrix point at unsigned short and
    // count their strides in bf16 elements.
    int a_bf16;           // A (both views) is stored as bf16: staged into LDS as it is          (wn_fwd_gemm_bf16<*, *, true>)
    int c_bf16;           // WN_EPI_GATE_BWD: c receives bf16 (round to nearest even); WN_EPI_GATE: c and c2 do; WN_EPI_PLAIN (the stand-alone bf16
                          // products only): c receives bf16 INSTEAD of fp32 (its row map counts bf16 elements) -- the skip path's share of dz (dzg)
    unsigned short* c_h;  // WN_EPI_PLAIN: optional bf16 COPY of the output (round to nearest even), laid out like c (same strides, in elements):
                          // the bf16 training step's shadow of the residual stream -- every matrix operand read of x takes half the bytes,
                          // with the bits the fp32-stored operand would be rounded to on its way to LDS
};
static __device__ __forceinline__ const float* wn_row_at(const WnRowMap& r, unsigned q, unsigned rem) {
    return r.base + (long long)q * r.batch_stride + (r.t0 + (long long)rem) * r.row_stride;
}

static __device__ __forceinline__ unsigned wn_pack_bf16(float lo, float hi) {  // two RNE-rounded bf16 in one dword
    unsigned a = __float_as_uint(lo), b = __float_as_uint(hi);
    a += 0x7fffu + ((a >> 16) & 1u);
    b += 0x7fffu + ((b >> 16) & 1u);
    return (a >> 16) | (b & 0xffff0000u);
}

// Epilogue of a wave's 32 x 128 strip (4 accumulator tiles; C/D layout of 32x32: col = lane & 31, row = (i & 3) + 8 * (i >> 2) +
// 4 * (lane >> 5)): rows mw.., logical columns nw.. .  WN_EPI_GATE: the 128 columns are [F(32) | G(32) | F(32) | G(32)] and the
// strip emits 64 columns of tanh(F+bf) * sigmoid(G+bg).
//
// Round 4: every global access of the epilogue is 16 bytes per lane.  In the MFMA layout a lane holds ONE column of 16 rows, so the
// straightforward epilogue (rounds 1-3) read cin / the saved gates and wrote C with 4-byte (bf16 outputs: 2-byte) accesses, 128 of them
// per lane and strip -- and those accesses, not the products, were half of the training step: with the epilogues compiled out the bf16
// config-5 step took 47.5 ms instead of 94.5 (profiles/r04_train_epilogue_experiment.txt); narrow vector-memory accesses retire per
// lane, not per byte (MI355X guide).  Each 32 x 32 tile now goes through a wave-private LDS tile (`stage`: the GEMM's own operand
// buffers, free once the K loop is done; row pitch 36 floats) and comes back row-wise: a lane takes 4 (fp32) or 8 (bf16-stored)
// consecutive columns of a row.  No barrier: a wave's LDS operations execute in order.
#define WN_EPI_PITCH 36
#define WN_EPI_TILE_FLOATS (32 * WN_EPI_PITCH)
static __device__ __forceinline__ void wn_epi_put(float* stage, const float (&v)[16], int lane) {
    float* dst = stage + (4 * (lane >> 5)) * WN_EPI_PITCH + (lane & 31);
#pragma unroll
    for (int i = 0; i < 16; ++i) dst[((i & 3) + 8 * (i >> 2)) * WN_EPI_PITCH] = v[i];
}
static __device__ __forceinline__ uint2 wn_pack_bf16x4(float4 a) { return make_uint2(wn_pack_bf16(a.x, a.y), wn_pack_bf16(a.z, a.w)); }
static __device__ __forceinline__ uint4 wn_pack_bf16x8(float4 a, float4 b) {
    return make_uint4(wn_pack_bf16(a.x, a.y), wn_pack_bf16(a.z, a.w), wn_pack_bf16(b.x, b.y), wn_pack_bf16(b.z, b.w));
}
// NTILES: accumulator tiles of the strip that exist (WN_EPI_PLAIN, WN_EPI_GATE_BWD; 32 columns each).  zl: the strip's rows of an LDS image
// [row][zld bf16] of what the strip emits -- z (WN_EPI_GATE with c_bf16; g.c.base may then be NULL: z not stored) or the plain output
// (WN_EPI_PLAIN) -- the A operand of the fused kernels' second product.
// CB16: WN_EPI_PLAIN honours g.c_bf16 (the stand-alone bf16 products; the fused kernels' plain epilogues always write fp32 and compile without the branch).
#ifndef WN_ABL_TN_PLAIN_STORE
#define WN_ABL_TN_PLAIN_STORE 0   // timing ablation of wn_bwd_gemm_tn_bf16 (tools/build_variant.py; results wrong): plain stores instead of the fp32 atomics
#endif
#ifndef WN_DZG_BF16
#define WN_DZG_BF16 1   // bf16 step: the skip path's share of dz -- dzg = dskip . Wskip of a block of layers, written by one product, read once by every layer's
                        // gate derivative -- is STORED as bf16 like every other product output that only feeds the next stage ([dF|dG], z, the gate pair): 8.9 GB
                        // of fp32 written and 8.9 GB read per config-5 step become 4.45 + 4.45.  One more rounding point of the bf16 step (oracle/bf16_step.py
                        // carries it); 0 keeps fp32 (A/B builds: host and kernels read the same switch).
#endif
template <int EPI, int NTILES = 4, bool CB16 = false>
static __device__ __forceinline__ void wn_gemm_epilogue(const WnGemmArgs& g, const wn_f16v (&acc)[NTILES], long long mw, int nw, int lane, float* stage,
                                                        unsigned short* zl = nullptr, int zld = 0) {
    const int col = lane & 31;
#ifdef WN_EPI_TIMING_SKIP   // timing experiment (results wrong): one store per lane instead of the strip's epilogue
    if (acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] == 12345.678f) const_cast<float*>(g.c.base)[lane] = 1.f;
    return;
#endif
    // row-wise lane roles: 4 columns per lane, 8 rows per pass (fp32 rows);  8 columns per lane, 16 rows per pass (bf16-stored rows)
    const int r4 = lane >> 3, c4 = 4 * (lane & 7);
    const int r8 = lane >> 2, c8 = 8 * (lane & 3);
    // (q, rem) of row m: M < 2^31 (checked on the host): 32-bit division, a 64-bit one is ~100 instructions
    auto split = [&](long long m, unsigned& q, unsigned& rem) { q = (unsigned)m / (unsigned)g.rows_per_batch; rem = (unsigned)m - q * (unsigned)g.rows_per_batch; };
    static_assert(EPI != WN_EPI_GATE || NTILES == 4, "the gate epilogue takes the whole 128-column strip");
    if constexpr (EPI == WN_EPI_GATE && NTILES == 4) {
        const int NH = g.N >> 1;   // channels per row of z / the gate pair
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int nf = nw + 64 * p + col, ng = nf + 32;  // logical columns of F and G
            if (nw + 64 * p + 32 >= g.N) continue;           // (wave-uniform)
            float z[16], th[16], sg[16];
            const float bf = g.bias ? g.bias[nf] : 0.f, bg = g.bias ? g.bias[ng] : 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float f = acc[2 * p][i] + bf, gg = acc[2 * p + 1][i] + bg;
                // tanh(f) = 2 sigmoid(2f) - 1 on the branch-free exp of the generation kernels (absolute error ~1e-7), 1-ulp reciprocals
                th[i] = fmaf(2.0f, __builtin_amdgcn_rcpf(1.0f + wn_exp(-2.0f * f)), -1.0f);
                sg[i] = __builtin_amdgcn_rcpf(1.0f + wn_exp(-gg));
                z[i] = th[i] * sg[i];
            }
            const int zc0 = (nw >> 1) + 32 * p;   // first of this tile's 32 channels
            // ---- z (and its copy on the skip rows)
            wn_epi_put(stage, z, lane);
            if (g.c_bf16) {   // the bf16 step stores z as bf16: it only ever feeds bf16 matrix operands
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int row = r8 + 16 * k;
                    const long long m = mw + row;
                    if (m >= g.M) continue;
                    unsigned q, rem;
                    split(m, q, rem);
                    const float* sp = stage + row * WN_EPI_PITCH + c8;
                    const uint4 zb = wn_pack_bf16x8(*reinterpret_cast<const float4*>(sp), *reinterpret_cast<const float4*>(sp + 4));
                    if (zl) *reinterpret_cast<uint4*>(zl + row * zld + zc0 + c8) = zb;
                    if (g.c.base) {
                        unsigned short* c16 = const_cast<unsigned short*>(reinterpret_cast<const unsigned short*>(g.c.base)) + (long long)q * g.c.batch_stride + (g.c.t0 + (long long)rem) * g.c.row_stride;
                        *reinterpret_cast<uint4*>(c16 + zc0 + c8) = zb;
                    }
                    if (g.c2.base && (int)rem >= g.c2_first_row) {
                        unsigned short* d16 = const_cast<unsigned short*>(reinterpret_cast<const unsigned short*>(g.c2.base)) + (long long)q * g.c2.batch_stride +
                                              (g.c2.t0 + (long long)rem - g.c2_first_row) * g.c2.row_stride;
                        *reinterpret_cast<uint4*>(d16 + zc0 + c8) = zb;
                    }
                }
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int row = r4 + 8 * k;
                    const long long m = mw + row;
                    if (m >= g.M) continue;
                    unsigned q, rem;
                    split(m, q, rem);
                    const float4 zv = *reinterpret_cast<const float4*>(stage + row * WN_EPI_PITCH + c4);
                    *reinterpret_cast<float4*>(const_cast<float*>(wn_row_at(g.c, q, rem)) + zc0 + c4) = zv;
                    if (g.c2.base && (int)rem >= g.c2_first_row)
                        *reinterpret_cast<float4*>(const_cast<float*>(g.c2.base) + (long long)q * g.c2.batch_stride + (g.c2.t0 + (long long)rem - g.c2_first_row) * g.c2.row_stride + zc0 + c4) = zv;
                }
            }
            // ---- the saved gates (the backward's inputs): one {bf16 tanh, bf16 sigmoid} dword per element, or two fp32 matrices
            if (g.gate_t) {
                if (g.gate_packed) {
                    float pk[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) pk[i] = __uint_as_float(wn_pack_bf16(th[i], sg[i]));
                    wn_epi_put(stage, pk, lane);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int row = r4 + 8 * k;
                        const long long m = mw + row;
                        if (m >= g.M) continue;
                        *reinterpret_cast<float4*>(g.gate_t + m * NH + zc0 + c4) = *reinterpret_cast<const float4*>(stage + row * WN_EPI_PITCH + c4);
                    }
                } else {
#pragma unroll
                    for (int which = 0; which < 2; ++which) {
                        wn_epi_put(stage, which ? sg : th, lane);
                        float* dstm = which ? g.gate_g : g.gate_t;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int row = r4 + 8 * k;
                            const long long m = mw + row;
                            if (m >= g.M) continue;
                            *reinterpret_cast<float4*>(dstm + m * NH + zc0 + c4) = *reinterpret_cast<const float4*>(stage + row * WN_EPI_PITCH + c4);
                        }
                    }
                }
            }
        }
    } else if constexpr (EPI == WN_EPI_GATE_BWD) {
        // The product is dz = dx' . Wres (N = D channels); the strip emits [dF | dG] = dz * {G (1 - T^2), T G (1 - G)} in the packed
        // [F(32) | G(32)] column order of Wfg^T (2N columns per row of c).  gate_t / gate_g are the forward's saved gates (INPUTS
        // here, row m, N columns; gate_packed as in the forward); c2 is the skip path's share of dz (READ here: rows >=
        // c2_first_row of a batch entry add c2's row (index - c2_first_row)).  dz itself never reaches HBM.
#pragma unroll
        for (int j = 0; j < NTILES; ++j) {
            const int ch0 = nw + 32 * j;
            if (ch0 >= g.N) continue;   // (wave-uniform)
            float d[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) d[i] = acc[j][i];
            wn_epi_put(stage, d, lane);
            if (g.c_bf16) {   // [dF|dG] STORED as bf16 (the bf16 step): 8 channels per lane
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int row = r8 + 16 * k;
                    const long long m = mw + row;
                    if (m >= g.M) continue;
                    unsigned q, rem;
                    split(m, q, rem);
                    const float* sp = stage + row * WN_EPI_PITCH + c8;
                    float dz[8], t[8], s2[8];
                    { const float4 a = *reinterpret_cast<const float4*>(sp), b = *reinterpret_cast<const float4*>(sp + 4);
                      dz[0] = a.x; dz[1] = a.y; dz[2] = a.z; dz[3] = a.w; dz[4] = b.x; dz[5] = b.y; dz[6] = b.z; dz[7] = b.w; }
                    if (g.c2.base && (int)rem >= g.c2_first_row) {
#if WN_DZG_BF16   // (dzg is stored as bf16: c2's row map counts bf16 elements; one 16-byte load)
                        const unsigned short* zrow = reinterpret_cast<const unsigned short*>(g.c2.base) + (long long)q * g.c2.batch_stride +
                                                     (g.c2.t0 + (long long)rem - g.c2_first_row) * g.c2.row_stride + ch0 + c8;
                        const uint4 a = *reinterpret_cast<const uint4*>(zrow);
                        dz[0] += __uint_as_float(a.x << 16); dz[1] += __uint_as_float(a.x & 0xffff0000u); dz[2] += __uint_as_float(a.y << 16); dz[3] += __uint_as_float(a.y & 0xffff0000u);
                        dz[4] += __uint_as_float(a.z << 16); dz[5] += __uint_as_float(a.z & 0xffff0000u); dz[6] += __uint_as_float(a.w << 16); dz[7] += __uint_as_float(a.w & 0xffff0000u);
#else
                        const float* zrow = g.c2.base + (long long)q * g.c2.batch_stride + (g.c2.t0 + (long long)rem - g.c2_first_row) * g.c2.row_stride + ch0 + c8;
                        const float4 a = *reinterpret_cast<const float4*>(zrow), b = *reinterpret_cast<const float4*>(zrow + 4);
                        dz[0] += a.x; dz[1] += a.y; dz[2] += a.z; dz[3] += a.w; dz[4] += b.x; dz[5] += b.y; dz[6] += b.z; dz[7] += b.w;
#endif
                    }
                    if (g.gate_packed) {
                        const unsigned* gp = reinterpret_cast<const unsigned*>(g.gate_t) + m * g.N + ch0 + c8;
                        const uint4 a = *reinterpret_cast<const uint4*>(gp), b = *reinterpret_cast<const uint4*>(gp + 4);
                        const unsigned ts[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
                        for (int e = 0; e < 8; ++e) { t[e] = __uint_as_float(ts[e] << 16); s2[e] = __uint_as_float(ts[e] & 0xffff0000u); }
                    } else {
                        const float* tp = g.gate_t + m * g.N + ch0 + c8;
                        const float* gp = g.gate_g + m * g.N + ch0 + c8;
#pragma unroll
                        for (int e = 0; e < 8; ++e) { t[e] = tp[e]; s2[e] = gp[e]; }
                    }
                    float df[8], dg[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) { df[e] = dz[e] * s2[e] * (1.f - t[e] * t[e]); dg[e] = dz[e] * t[e] * s2[e] * (1.f - s2[e]); }
                    unsigned short* c16 = const_cast<unsigned short*>(reinterpret_cast<const unsigned short*>(g.c.base)) +
                                          (long long)q * g.c.batch_stride + (g.c.t0 + (long long)rem) * g.c.row_stride + 2 * nw + 64 * j + c8;
                    *reinterpret_cast<uint4*>(c16) = wn_pack_bf16x8(float4{df[0], df[1], df[2], df[3]}, float4{df[4], df[5], df[6], df[7]});
                    *reinterpret_cast<uint4*>(c16 + 32) = wn_pack_bf16x8(float4{dg[0], dg[1], dg[2], dg[3]}, float4{dg[4], dg[5], dg[6], dg[7]});
                }
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int row = r4 + 8 * k;
                    const long long m = mw + row;
                    if (m >= g.M) continue;
                    unsigned q, rem;
                    split(m, q, rem);
                    float4 dzv = *reinterpret_cast<const float4*>(stage + row * WN_EPI_PITCH + c4);
                    if (g.c2.base && (int)rem >= g.c2_first_row) {
                        const float4 a = *reinterpret_cast<const float4*>(g.c2.base + (long long)q * g.c2.batch_stride + (g.c2.t0 + (long long)rem - g.c2_first_row) * g.c2.row_stride + ch0 + c4);
                        dzv.x += a.x; dzv.y += a.y; dzv.z += a.z; dzv.w += a.w;
                    }
                    float dz[4] = {dzv.x, dzv.y, dzv.z, dzv.w}, t[4], s2[4];
                    if (g.gate_packed) {
                        const uint4 a = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned*>(g.gate_t) + m * g.N + ch0 + c4);
                        const unsigned ts[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) { t[e] = __uint_as_float(ts[e] << 16); s2[e] = __uint_as_float(ts[e] & 0xffff0000u); }
                    } else {
                        const float4 a = *reinterpret_cast<const float4*>(g.gate_t + m * g.N + ch0 + c4), b = *reinterpret_cast<const float4*>(g.gate_g + m * g.N + ch0 + c4);
                        t[0] = a.x; t[1] = a.y; t[2] = a.z; t[3] = a.w; s2[0] = b.x; s2[1] = b.y; s2[2] = b.z; s2[3] = b.w;
                    }
                    float4 df, dg;
                    df.x = dz[0] * s2[0] * (1.f - t[0] * t[0]); dg.x = dz[0] * t[0] * s2[0] * (1.f - s2[0]);
                    df.y = dz[1] * s2[1] * (1.f - t[1] * t[1]); dg.y = dz[1] * t[1] * s2[1] * (1.f - s2[1]);
                    df.z = dz[2] * s2[2] * (1.f - t[2] * t[2]); dg.z = dz[2] * t[2] * s2[2] * (1.f - s2[2]);
                    df.w = dz[3] * s2[3] * (1.f - t[3] * t[3]); dg.w = dz[3] * t[3] * s2[3] * (1.f - s2[3]);
                    float* crow = const_cast<float*>(wn_row_at(g.c, q, rem)) + 2 * nw + 64 * j + c4;
                    *reinterpret_cast<float4*>(crow) = df;
                    *reinterpret_cast<float4*>(crow + 32) = dg;
                }
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < NTILES; ++j) {
            const int n0 = nw + 32 * j;
            if (n0 >= g.N) continue;   // (wave-uniform)
            float v16[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v16[i] = acc[j][i];
            wn_epi_put(stage, v16, lane);
            const int n = n0 + c4;
            float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
            if (g.bias) bias = *reinterpret_cast<const float4*>(g.bias + n);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int row = r4 + 8 * k;
                const long long m = mw + row;
                if (m >= g.M) continue;
                unsigned q, rem;
                split(m, q, rem);
                const long long coff = (long long)q * g.c.batch_stride + (g.c.t0 + (long long)rem) * g.c.row_stride;   // the row's offset in ELEMENTS of c (fp32 or bf16)
                float4 v = *reinterpret_cast<const float4*>(stage + row * WN_EPI_PITCH + c4);
                v.x += bias.x; v.y += bias.y; v.z += bias.z; v.w += bias.w;
                if (g.cin.base && (int)rem >= g.cin_skip_lo) {
                    const float4 a = *reinterpret_cast<const float4*>(wn_row_at(g.cin, q, rem) + n);
                    v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
                }
                if (g.relu_c) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                if (g.mask) {  // the mask shares the output's row layout
                    const float4 mk = *reinterpret_cast<const float4*>(g.mask + coff + n);
                    if (!(mk.x > 0.f)) v.x = 0.f;
                    if (!(mk.y > 0.f)) v.y = 0.f;
                    if (!(mk.z > 0.f)) v.z = 0.f;
                    if (!(mk.w > 0.f)) v.w = 0.f;
                }
                if (CB16 && g.c_bf16) *reinterpret_cast<uint2*>(const_cast<unsigned short*>(reinterpret_cast<const unsigned short*>(g.c.base)) + coff + n) = wn_pack_bf16x4(v);
                else *reinterpret_cast<float4*>(const_cast<float*>(g.c.base) + coff + n) = v;
                if (g.c_h) *reinterpret_cast<uint2*>(g.c_h + coff + n) = wn_pack_bf16x4(v);
                if (zl) *reinterpret_cast<uint2*>(zl + row * zld + n) = wn_pack_bf16x4(v);   // (WN_EPI_PLAIN: the output tile as the next product's bf16 A operand, column n of an image that starts at column 0)
            }
        }
    }
}

// C[M][N] (+)= A[M][K] . B^T[K][N]; 128 x 128 tile per workgroup, 4 waves, wave w owns rows 32w..32w+31 and all 128
// columns (4 accumulator tiles of 32x32).  WN_EPI_GATE: the 128 columns are [F(32) | G(32) | F(32) | G(32)] and the tile
// emits 64 columns of tanh(F+bf) * sigmoid(G+bg).
#ifndef WN_GEMM_KC
#define WN_GEMM_KC 16   // K chunk of the fp32 GEMMs.  Measured on the config-5 forward: 32 (65.8 KB LDS, 2 workgroups per CU)
                        // 89.4 ms; 16 (32.9 KB) with 3 per CU 78.4 ms, with 4 per CU (119 VGPRs) 71.2 ms; 8: 70.9 ms
#endif
template <int EPI>
#ifndef WN_GEMM_MINB
#define WN_GEMM_MINB 4
#endif
__global__ __launch_bounds__(256, WN_GEMM_MINB) void wn_fwd_gemm(WnGemmArgs g) {
    constexpr int TM = 128, TN = 128, KC = WN_GEMM_KC, AP = TM + 1;  // AP: padded row length of the transposed A chunk
    constexpr int NQ = KC / 8;          // float4 per thread per operand and chunk
    constexpr int BT = 256 / KC;        // threads per B row
    // (ONE block: the epilogue re-uses it as the waves' staging tiles once the K loop is done)
    __shared__ __attribute__((aligned(16))) float smem_f[2 * KC * AP + 2 * KC * TN + (2 * KC * AP) % 4];
    static_assert(2 * KC * AP + 2 * KC * TN >= 4 * WN_EPI_TILE_FLOATS, "the operand buffers hold the four waves' staging tiles");
    float (*a_t)[KC * AP] = reinterpret_cast<float (*)[KC * AP]>(smem_f);                                        // [2][k][row]
    float (*b_s)[KC * TN] = reinterpret_cast<float (*)[KC * TN]>(smem_f + ((2 * KC * AP + 3) & ~3));               // [2][k][col]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const unsigned mtiles = (unsigned)((g.M + TM - 1) / TM), tm_i = blockIdx.x % mtiles, tn_i = blockIdx.x / mtiles;  // row tiles fastest
    const long long m0 = (long long)tm_i * TM;
    const int n0 = (int)tn_i * TN;
    wn_f16v acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;

    // loader roles: A chunk = 128 rows x 32 floats: thread -> row tid/2, 16 floats (4 float4) ; B chunk = 32 x 128: 4 float4 each
    const int arow = tid >> 1, ahalf = tid & 1;
    const long long am = m0 + arow;
    const bool arow_ok = am < g.M;
    const unsigned aq = arow_ok ? (unsigned)am / (unsigned)g.rows_per_batch : 0u, arem = arow_ok ? (unsigned)am - aq * (unsigned)g.rows_per_batch : 0u;
    const bool ok0 = arow_ok && (int)arem >= g.a_skip_lo[0] && (int)arem < g.rows_per_batch - g.a_skip_hi[0];
    const bool ok1 = arow_ok && (int)arem >= g.a_skip_lo[1] && (int)arem < g.rows_per_batch - g.a_skip_hi[1];
    const float* a0p = wn_row_at(g.a0, aq, arem);
    const float* a1p = wn_row_at(g.a1, aq, arem);
    const int brow = tid / BT, bcol = (tid % BT) * (KC / 2);

    float4 va[NQ], vb[NQ];  // staging registers of the chunk in flight
    auto fetch = [&](int kc) {  // global -> registers (issued before the multiply of the current chunk)
        const int k0 = kc * KC;
        const bool first = k0 < g.k_split, ok = first ? ok0 : ok1;
        const float* src = first ? a0p + k0 : a1p + (k0 - g.k_split);
#pragma unroll
        for (int q = 0; q < NQ; ++q) va[q] = ok ? *reinterpret_cast<const float4*>(src + ahalf * (KC / 2) + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float* bsrc = ((first || !g.bt1) ? g.bt + (size_t)(k0 + brow) * g.N : g.bt1 + (size_t)(k0 - g.k_split + brow) * g.N) + n0 + bcol;
        // (ONE predicate for the thread's NQ loads -- N % 32 == 0 and bcol is a multiple of KC / 2 = 4 NQ floats, so they are in range together: with a
        //  predicate per load each one sat in a block of its own with a full wait behind it.  Round 6, profiles/r06_tn_loads.txt.)
        const bool okb = n0 + bcol < g.N;
#pragma unroll
        for (int q = 0; q < NQ; ++q) vb[q] = okb ? *reinterpret_cast<const float4*>(bsrc + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto stash = [&](int buf) {  // registers -> LDS (A transposed to [k][row])
        float* at = a_t[buf];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            float4 x = va[q];
            if (g.relu_a) { x.x = fmaxf(x.x, 0.f); x.y = fmaxf(x.y, 0.f); x.z = fmaxf(x.z, 0.f); x.w = fmaxf(x.w, 0.f); }
            const int k = ahalf * (KC / 2) + q * 4;
            at[(k + 0) * AP + arow] = x.x; at[(k + 1) * AP + arow] = x.y; at[(k + 2) * AP + arow] = x.z; at[(k + 3) * AP + arow] = x.w;
        }
        float* bs = b_s[buf] + brow * TN + bcol;
#pragma unroll
        for (int q = 0; q < NQ; ++q) *reinterpret_cast<float4*>(bs + q * 4) = vb[q];
    };

    const int nchunks = g.K / KC;
    fetch(0);
    stash(0);
    __syncthreads();
    for (int kc = 0; kc < nchunks; ++kc) {
        const int buf = kc & 1;
        if (kc + 1 < nchunks) fetch(kc + 1);  // lands while this chunk is multiplied
        const float* at = a_t[buf] + 32 * wv + (lane & 31);
        const float* bs = b_s[buf] + (lane & 31);
        const int kh = lane >> 5;
#pragma unroll
        for (int ks = 0; ks < KC / 2; ++ks) {
            const float a = at[(2 * ks + kh) * AP];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float b = bs[(2 * ks + kh) * TN + 32 * j];
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
            }
        }
        if (kc + 1 < nchunks) stash(buf ^ 1);
        __syncthreads();
    }

    wn_gemm_epilogue<EPI>(g, acc, m0 + 32 * wv, n0, lane, smem_f + wv * WN_EPI_TILE_FLOATS);   // (the loop's last barrier: nobody reads the operand buffers any more)
}


// ------------------------------------------------------------------------------------------------ bf16 operands
// Same GEMM with bf16 MFMA operands and fp32 accumulation (v_mfma_f32_32x32x16_bf16, 16x the fp32 matrix rate): the fp32
// activation rows are rounded to bf16 (RNE) while they are staged into LDS, the weights are pre-converted.  The residual
// stream X and all accumulators stay fp32.  Opt-in (wn_set_forward_precision): results differ from the fp32 reference at
// the 1e-2 level of the logit scale -- the usual bf16 training trade -- so the fp32 kernel stays the parity default.
typedef __bf16 wn_bf16x8 __attribute__((ext_vector_type(8)));

struct WnGemmArgsBf16 {
    WnGemmArgs g;              // as the fp32 kernel; g.bt / g.bt1 unused
    const unsigned short* bn;  // B as bf16 [N][ldb] row-major (K contiguous: the weights' natural (out, in) layout)
    const unsigned short* bn1; // optional: columns k >= k_split of B come from here ([N][ldb], column k - k_split)
    int ldb;                   // 0 -> K
};

#ifndef WN_NN_COL_ADJ
#define WN_NN_COL_ADJ 1   // column tiles of a row tile next to each other in dispatch order: grouped skip product 1176 -> 1097 us, dzg 1215 -> 1171 us
#endif
#ifndef WN_GEMM_BF16_KC
#define WN_GEMM_BF16_KC 32    // config-5 forward (round 2): 64 (73.7 KB LDS, 2 workgroups per CU) 52.7 ms; 32 (41 KB, 3 per CU, 152 VGPRs) 40.8 ms
#endif
#ifndef WN_GEMM_BF16_MINB
#define WN_GEMM_BF16_MINB 4   // 4-wave form: 4 workgroups per CU (exactly the CU's 160 KB of LDS, 128 VGPRs): 3 -> 4 took the residual / dx products from 346 to 314 us
                              // (the forms that convert an fp32-stored A on its way to LDS keep 3: at 128 VGPRs they spilled 2-3 registers -- round 6)
#endif

// WAVES = 4: 128 x 128 tile (wave w: rows 32w.., all 128 columns).  WAVES = 8: 128 x 256 tile (wave w: rows 32 (w & 3).., column half
// w >> 2) for products with N >= 256 -- these GEMMs are streams over A (K is 128-512, M is 350 k-500 k rows), and a 128-column
// tile makes every further column block re-read A from HBM (the gate product: 4 x 262 MB instead of 2 x); the wide tile stages A
// once for all 256 columns.
// A16: A is stored as bf16 (g.a_bf16): its pieces go to LDS as they are.
template <int EPI, int WAVES, bool A16 = false>
__global__ __launch_bounds__(64 * WAVES, WAVES == 4 ? (A16 ? WN_GEMM_BF16_MINB : 3) : 4) void wn_fwd_gemm_bf16(WnGemmArgsBf16 ga) {
    const WnGemmArgs& g = ga.g;
    constexpr int NT = 64 * WAVES, TM = 128, TN = 32 * WAVES, KC = WN_GEMM_BF16_KC, LD = KC + 8;  // LD: padded row length (bf16): 80-byte rows, conflict-free b128 reads
    constexpr int TPR = NT / TM;   // loader threads per A row (2 / 4)
    constexpr int HK = KC / TPR;   // k values per A loader thread and chunk (16 / 8)
    constexpr int HB = KC / 2;     // B: two loader threads per column, KC/2 bf16 each (NT == 2 TN)
    static_assert(HK % 8 == 0 && HB % 8 == 0, "loader pieces are 16-byte LDS stores");
    // (ONE block: the epilogue re-uses it as the waves' staging tiles once the K loop is done)
    __shared__ __attribute__((aligned(16))) unsigned short smem_h[2 * TM * LD + 2 * TN * LD];
    static_assert((2 * TM * LD + 2 * TN * LD) * 2 >= WAVES * WN_EPI_TILE_FLOATS * 4, "the operand buffers hold the waves' staging tiles");
    unsigned short (*a_s)[TM * LD] = reinterpret_cast<unsigned short (*)[TM * LD]>(smem_h);
    unsigned short (*b_s)[TN * LD] = reinterpret_cast<unsigned short (*)[TN * LD]>(smem_h + 2 * TM * LD);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wr = wv & 3, wc = wv >> 2;
#if WN_NN_COL_ADJ   // the column tiles of a row tile next to each other in dispatch order (different XCDs, same moment: the second read of the
                    // row tile's A is served by the memory-side cache; the XCD-local variant of this -- ids 8 apart -- measured slower)
    const unsigned ntiles = (unsigned)((g.N + TN - 1) / TN), tm_i = blockIdx.x / ntiles, tn_i = blockIdx.x % ntiles;
#else
    const unsigned mtiles = (unsigned)((g.M + TM - 1) / TM), tm_i = blockIdx.x % mtiles, tn_i = blockIdx.x / mtiles;  // row tiles fastest
#endif
    const long long m0 = (long long)tm_i * TM;
    const int n0 = (int)tn_i * TN;
    wn_f16v acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    const int lrow = tid / TPR, lpart = tid % TPR;
    const long long am = m0 + lrow;
    const bool arow_ok = am < g.M;
    const unsigned aq = arow_ok ? (unsigned)am / (unsigned)g.rows_per_batch : 0u, arem = arow_ok ? (unsigned)am - aq * (unsigned)g.rows_per_batch : 0u;
    const bool ok0 = arow_ok && (int)arem >= g.a_skip_lo[0] && (int)arem < g.rows_per_batch - g.a_skip_hi[0];
    const bool ok1 = arow_ok && (int)arem >= g.a_skip_lo[1] && (int)arem < g.rows_per_batch - g.a_skip_hi[1];
    // (A16: the same arithmetic in bf16 elements -- half the bytes per element)
    const float* a0p = A16 ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(g.a0.base) + (long long)aq * g.a0.batch_stride + (g.a0.t0 + (long long)arem) * g.a0.row_stride + lpart * HK)
                           : wn_row_at(g.a0, aq, arem) + lpart * HK;
    const float* a1p = A16 ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(g.a1.base) + (long long)aq * g.a1.batch_stride + (g.a1.t0 + (long long)arem) * g.a1.row_stride + lpart * HK)
                           : wn_row_at(g.a1, aq, arem) + lpart * HK;
    const int bcol = tid >> 1, bhalf = tid & 1;
    const bool bcol_ok = n0 + bcol < g.N;
    const int ldb = ga.ldb ? ga.ldb : g.K;
    const unsigned short* bp0 = ga.bn + (size_t)(n0 + bcol) * ldb + bhalf * HB;
    const unsigned short* bp1 = ga.bn1 ? ga.bn1 + (size_t)(n0 + bcol) * ldb + bhalf * HB : bp0 + g.k_split;

    // Staging registers of the chunk in flight.  (Two A chunks in flight per workgroup -- a second register set, chunk kc + 2 fetched
    // while kc + 1 waits to be staged -- measured level in both forms: profiles/archive/r03_train_step_experiments.txt.)
    float4 va0[HK / 4];   // A16: only the first HK / 8 hold data (HK bf16)
    uint4 vb[HB / 8];
    auto fetch_a = [&](int kc, float4 (&va)[HK / 4]) {
        const int k0 = kc * KC;
        const bool first = k0 < g.k_split, ok = first ? ok0 : ok1;
        if constexpr (A16) {
            const unsigned short* src = reinterpret_cast<const unsigned short*>(first ? a0p : a1p) + (first ? k0 : k0 - g.k_split);
#pragma unroll
            for (int q = 0; q < HK / 8; ++q) va[q] = ok ? *reinterpret_cast<const float4*>(src + q * 8) : make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            const float* src = first ? a0p + k0 : a1p + (k0 - g.k_split);
#pragma unroll
            for (int q = 0; q < HK / 4; ++q) va[q] = ok ? *reinterpret_cast<const float4*>(src + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto fetch_b = [&](int kc) {
        const int k0 = kc * KC;
        const uint4* bsrc = reinterpret_cast<const uint4*>(k0 < g.k_split ? bp0 + k0 : bp1 + (k0 - g.k_split));
#pragma unroll
        for (int q = 0; q < HB / 8; ++q) vb[q] = bcol_ok ? bsrc[q] : make_uint4(0u, 0u, 0u, 0u);
    };
    auto stash = [&](int buf, const float4 (&va)[HK / 4]) {
        uint4* ad = reinterpret_cast<uint4*>(a_s[buf] + lrow * LD + lpart * HK);
        if constexpr (A16) {
#pragma unroll
            for (int q = 0; q < HK / 8; ++q) reinterpret_cast<float4*>(ad)[q] = va[q];
        } else
#pragma unroll
        for (int q = 0; q < HK / 8; ++q) {
            float4 x = va[2 * q], y = va[2 * q + 1];
            if (g.relu_a) {
                x.x = fmaxf(x.x, 0.f); x.y = fmaxf(x.y, 0.f); x.z = fmaxf(x.z, 0.f); x.w = fmaxf(x.w, 0.f);
                y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f);
            }
            ad[q] = make_uint4(wn_pack_bf16(x.x, x.y), wn_pack_bf16(x.z, x.w), wn_pack_bf16(y.x, y.y), wn_pack_bf16(y.z, y.w));
        }
        uint4* bd = reinterpret_cast<uint4*>(b_s[buf] + bcol * LD + bhalf * HB);
#pragma unroll
        for (int q = 0; q < HB / 8; ++q) bd[q] = vb[q];
    };
    auto multiply = [&](int buf) {
        const unsigned short* ar = a_s[buf] + (32 * wr + (lane & 31)) * LD + 8 * (lane >> 5);
        const unsigned short* br = b_s[buf] + (128 * wc + (lane & 31)) * LD + 8 * (lane >> 5);
#pragma unroll
        for (int ks = 0; ks < KC / 16; ++ks) {
            const wn_bf16x8 a = *reinterpret_cast<const wn_bf16x8*>(ar + 16 * ks);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const wn_bf16x8 b = *reinterpret_cast<const wn_bf16x8*>(br + 32 * j * LD + 16 * ks);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
            }
        }
    };

    const int nchunks = g.K / KC;
    {
        fetch_a(0, va0);
        fetch_b(0);
        stash(0, va0);
        __syncthreads();
        for (int kc = 0; kc < nchunks; ++kc) {
            const int buf = kc & 1;
            if (kc + 1 < nchunks) { fetch_a(kc + 1, va0); fetch_b(kc + 1); }
            multiply(buf);
            if (kc + 1 < nchunks) stash(buf ^ 1, va0);
            __syncthreads();
        }
    }
    // (the epilogue's lane roles from an OPAQUE copy of the thread index: otherwise the compiler forms its row and column offsets in front of the K loop and
    //  carries them through it -- at the 128 registers of the four-workgroups-per-CU form that was 2-5 spilled registers in the loop)
    int te = threadIdx.x;
    asm volatile("" : "+v"(te));
    const int lane_e = te & 63, wv_e = te >> 6, wr_e = wv_e & 3, wc_e = wv_e >> 2;
    wn_gemm_epilogue<EPI, 4, true>(g, acc, m0 + 32 * wr_e, n0 + 128 * wc_e, lane_e, reinterpret_cast<float*>(smem_h) + wv_e * WN_EPI_TILE_FLOATS);   // (after the loop's last barrier)
}

// ---- One layer of the forward in ONE kernel (bf16 operands; the shape whose filter/gate product is ONE 256-column tile: D = 128, and
// R = 128): the filter/gate product of a 128-row tile (K = 2R over the two tap views of x), its gate epilogue -- z, the saved gates and
// the copy on the skip rows go out as in wn_fwd_gemm_bf16<WN_EPI_GATE, 8> --, and then the residual product of the SAME rows on the z tile
// where it lies: z (128 x 128 bf16) stays in LDS as the second product's A operand, Wres streams through LDS in four 32-column chunks,
// x' = z . Wres^T + bias + x leaves through the plain epilogue.  Unfused, z makes a round trip through HBM between two launches
// (0.13 GB per layer at config 5) and each launch pays its own tail (3250 workgroups = 3.2 rounds of the chip); the second product adds
// ~20 % to the tile's MFMA work and nothing to its global reads but the addend.  Same operands, same chunk order, same accumulation as
// the two kernels: the results are theirs bit for bit.  g.c.base == NULL: z itself is not stored (forward only: nothing reads it again).
// Row tile of a workgroup of the one-launch layer kernels.  Workgroups are handed to the eight XCDs round robin, so with tile = blockIdx neighbouring row tiles
// run on DIFFERENT XCDs -- and the layer kernels read every row twice: the tap view x(t - d) / [dF|dG](t + d) of a tile is the other view of the tile d / 128 further
// (or of itself and its neighbour for d < 128).  With one contiguous range of tiles per XCD, in dispatch order, both reads meet in one L2 a few microseconds apart
// instead of going out to the fabric twice.  grid = 8 * ceil(tiles / 8) workgroups; the ones past the end return.  (The mapping of workgroups to XCDs is not a
// contract: a different one costs the locality, nothing else.)
#ifndef WN_LAYER_XCD_RANGES
#define WN_LAYER_XCD_RANGES 1
#endif
static __device__ __forceinline__ long long wn_layer_tile(long long M) {
#if WN_LAYER_XCD_RANGES
    const unsigned tiles = (unsigned)((M + 127) / 128), per = (tiles + 7) / 8;
    const unsigned t = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    return (blockIdx.x >> 3) < per && t < tiles ? (long long)t : -1;
#else
    return (long long)blockIdx.x < (M + 127) / 128 ? (long long)blockIdx.x : -1;
#endif
}
static inline unsigned wn_layer_grid(long long M) {
    const unsigned tiles = (unsigned)((M + 127) / 128);
#if WN_LAYER_XCD_RANGES
    return 8u * ((tiles + 7) / 8);
#else
    return tiles;
#endif
}

struct WnLayerArgs {
    const unsigned short* bn;  // Wres as bf16 [R][D] row-major (K = D contiguous)
    const float* bias;         // [R] or NULL
    WnRowMap cin, c;           // the addend x_l and the output x_{l+1}: rows of R floats, row index as in the gate product
    unsigned short* c_h;       // optional bf16 copy of the output (WnGemmArgs::c_h)
    int N;                     // R
};
__global__ __launch_bounds__(512, 4) void wn_fwd_layer_bf16(WnGemmArgsBf16 ga, WnLayerArgs la) {
    constexpr bool A16 = true;   // x comes from its bf16 shadow (the callers have one wherever this kernel applies)
    const WnGemmArgs& g = ga.g;
    constexpr int NT = 512, TM = 128, TN = 256, KC = WN_GEMM_BF16_KC, LD = KC + 8;
    constexpr int TPR = NT / TM, HK = KC / TPR, HB = KC / 2;
    constexpr int D2 = 128, N2 = 128, LD2 = D2 + 8;   // the second product: K = D2 channels of z, N2 = R output columns
    static_assert(KC == 32 && HK % 8 == 0 && HB % 8 == 0, "loader pieces are 16-byte LDS stores");
    constexpr int STAGE_SHORTS = 8 * WN_EPI_TILE_FLOATS * 2;            // the eight waves' epilogue tiles (fp32) at the start of the block
    constexpr int OPER_SHORTS = 2 * TM * LD + 2 * TN * LD;              // the first product's operand buffers (same place)
    constexpr int ZOFF = STAGE_SHORTS;   // the z image [128][LD2] bf16 sits behind the staging tiles, over the (by then dead) operand buffers
    static_assert(2 * N2 * LD <= STAGE_SHORTS, "the second product's B chunks live where the staging tiles are");
    constexpr int SMEM_SHORTS = ZOFF + TM * LD2 > OPER_SHORTS ? ZOFF + TM * LD2 : OPER_SHORTS;   // 71 680 bytes: two workgroups per CU
    __shared__ __attribute__((aligned(16))) unsigned short smem_h[SMEM_SHORTS];
    unsigned short (*a_s)[TM * LD] = reinterpret_cast<unsigned short (*)[TM * LD]>(smem_h);
    unsigned short (*b_s)[TN * LD] = reinterpret_cast<unsigned short (*)[TN * LD]>(smem_h + 2 * TM * LD);
    unsigned short* z_s = smem_h + ZOFF;
    unsigned short (*b2_s)[N2 * LD] = reinterpret_cast<unsigned short (*)[N2 * LD]>(smem_h);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wr = wv & 3, wc = wv >> 2;
    const long long tile = wn_layer_tile(g.M);
    if (tile < 0) return;   // (block-uniform)
    const long long m0 = tile * TM;
    wn_f16v acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    {   // ---- first product: wn_fwd_gemm_bf16<WN_EPI_GATE, 8, A16> with ONE column tile (n0 = 0)
        const int lrow = tid / TPR, lpart = tid % TPR;
        const long long am = m0 + lrow;
        const bool arow_ok = am < g.M;
        const unsigned aq = arow_ok ? (unsigned)am / (unsigned)g.rows_per_batch : 0u, arem = arow_ok ? (unsigned)am - aq * (unsigned)g.rows_per_batch : 0u;
        const bool ok0 = arow_ok && (int)arem >= g.a_skip_lo[0] && (int)arem < g.rows_per_batch - g.a_skip_hi[0];
        const bool ok1 = arow_ok && (int)arem >= g.a_skip_lo[1] && (int)arem < g.rows_per_batch - g.a_skip_hi[1];
        const float* a0p = A16 ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(g.a0.base) + (long long)aq * g.a0.batch_stride + (g.a0.t0 + (long long)arem) * g.a0.row_stride + lpart * HK)
                               : wn_row_at(g.a0, aq, arem) + lpart * HK;
        const float* a1p = A16 ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(g.a1.base) + (long long)aq * g.a1.batch_stride + (g.a1.t0 + (long long)arem) * g.a1.row_stride + lpart * HK)
                               : wn_row_at(g.a1, aq, arem) + lpart * HK;
        const int bcol = tid >> 1, bhalf = tid & 1;
        const bool bcol_ok = bcol < g.N;
        const int ldb = ga.ldb ? ga.ldb : g.K;
        const unsigned short* bp0 = ga.bn + (size_t)bcol * ldb + bhalf * HB;
        const unsigned short* bp1 = ga.bn1 ? ga.bn1 + (size_t)bcol * ldb + bhalf * HB : bp0 + g.k_split;
        float4 va[HK / 4];
        uint4 vb[HB / 8];
        auto fetch = [&](int kc) {
            const int k0 = kc * KC;
            const bool first = k0 < g.k_split, ok = first ? ok0 : ok1;
            if constexpr (A16) {
                const unsigned short* src = reinterpret_cast<const unsigned short*>(first ? a0p : a1p) + (first ? k0 : k0 - g.k_split);
#pragma unroll
                for (int q = 0; q < HK / 8; ++q) va[q] = ok ? *reinterpret_cast<const float4*>(src + q * 8) : make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                const float* src = first ? a0p + k0 : a1p + (k0 - g.k_split);
#pragma unroll
                for (int q = 0; q < HK / 4; ++q) va[q] = ok ? *reinterpret_cast<const float4*>(src + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            const uint4* bsrc = reinterpret_cast<const uint4*>(first ? bp0 + k0 : bp1 + (k0 - g.k_split));
#pragma unroll
            for (int q = 0; q < HB / 8; ++q) vb[q] = bcol_ok ? bsrc[q] : make_uint4(0u, 0u, 0u, 0u);
        };
        auto stash = [&](int buf) {
            uint4* ad = reinterpret_cast<uint4*>(a_s[buf] + lrow * LD + lpart * HK);
            if constexpr (A16) {
#pragma unroll
                for (int q = 0; q < HK / 8; ++q) reinterpret_cast<float4*>(ad)[q] = va[q];
            } else {
#pragma unroll
                for (int q = 0; q < HK / 8; ++q) {
                    const float4 x = va[2 * q], y = va[2 * q + 1];
                    ad[q] = make_uint4(wn_pack_bf16(x.x, x.y), wn_pack_bf16(x.z, x.w), wn_pack_bf16(y.x, y.y), wn_pack_bf16(y.z, y.w));
                }
            }
            uint4* bd = reinterpret_cast<uint4*>(b_s[buf] + bcol * LD + bhalf * HB);
#pragma unroll
            for (int q = 0; q < HB / 8; ++q) bd[q] = vb[q];
        };
        const int nchunks = g.K / KC;
        fetch(0);
        stash(0);
        __syncthreads();
        for (int kc = 0; kc < nchunks; ++kc) {
            const int buf = kc & 1;
            if (kc + 1 < nchunks) fetch(kc + 1);
            const unsigned short* ar = a_s[buf] + (32 * wr + (lane & 31)) * LD + 8 * (lane >> 5);
            const unsigned short* br = b_s[buf] + (128 * wc + (lane & 31)) * LD + 8 * (lane >> 5);
#pragma unroll
            for (int ks = 0; ks < KC / 16; ++ks) {
                const wn_bf16x8 a = *reinterpret_cast<const wn_bf16x8*>(ar + 16 * ks);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const wn_bf16x8 b = *reinterpret_cast<const wn_bf16x8*>(br + 32 * j * LD + 16 * ks);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
                }
            }
            if (kc + 1 < nchunks) stash(buf ^ 1);
            __syncthreads();
        }
    }
    float* stage = reinterpret_cast<float*>(smem_h) + wv * WN_EPI_TILE_FLOATS;
    // ---- gate epilogue: z (bf16) to HBM where it is wanted, and ALWAYS into the LDS image (rows 32 wr .., this wave's 64 channels)
    wn_gemm_epilogue<WN_EPI_GATE>(g, acc, m0 + 32 * wr, 128 * wc, lane, stage, z_s + (32 * wr) * LD2, LD2);
    __syncthreads();   // z_s complete; the staging tiles are free: the second product's B chunks take their place
    // ---- second product: x' tile [128 rows][128 columns] = z_s . Wres^T, wave (wr, wc): rows 32 wr .., columns 64 wc ..
    wn_f16v acc2[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc2[j][i] = 0.f;
    {
        const int bcol = tid >> 2, bpart = tid & 3;   // 128 columns x 4 pieces of 8 bf16 per chunk
        const unsigned short* bp = la.bn + (size_t)bcol * D2 + bpart * 8;
        uint4 vb2 = *reinterpret_cast<const uint4*>(bp);
        *reinterpret_cast<uint4*>(b2_s[0] + bcol * LD + bpart * 8) = vb2;
        __syncthreads();
        const unsigned short* zr = z_s + (32 * wr + (lane & 31)) * LD2 + 8 * (lane >> 5);
#pragma unroll
        for (int kc = 0; kc < D2 / KC; ++kc) {
            const int buf = kc & 1;
            if (kc + 1 < D2 / KC) vb2 = *reinterpret_cast<const uint4*>(bp + (kc + 1) * KC);
            const unsigned short* br = b2_s[buf] + (64 * wc + (lane & 31)) * LD + 8 * (lane >> 5);
#pragma unroll
            for (int ks = 0; ks < KC / 16; ++ks) {
                const wn_bf16x8 a = *reinterpret_cast<const wn_bf16x8*>(zr + KC * kc + 16 * ks);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const wn_bf16x8 b = *reinterpret_cast<const wn_bf16x8*>(br + 32 * j * LD + 16 * ks);
                    acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc2[j], 0, 0, 0);
                }
            }
            if (kc + 1 < D2 / KC) *reinterpret_cast<uint4*>(b2_s[buf ^ 1] + bcol * LD + bpart * 8) = vb2;
            __syncthreads();
        }
    }
    WnGemmArgs g2;   // the plain epilogue's view of the second product (rows as in the first)
    g2.a0 = g2.a1 = WnRowMap{nullptr, 0, 0, 0};
    g2.k_split = g2.K = D2; g2.bt = nullptr; g2.N = la.N; g2.bias = la.bias; g2.cin = la.cin; g2.c = la.c; g2.M = g.M; g2.rows_per_batch = g.rows_per_batch;
    g2.relu_a = g2.relu_c = 0; g2.mask = nullptr; g2.gate_t = g2.gate_g = nullptr; g2.c2 = WnRowMap{nullptr, 0, 0, 0}; g2.c2_first_row = 0; g2.gate_packed = 0;
    g2.a_skip_lo[0] = g2.a_skip_lo[1] = g2.a_skip_hi[0] = g2.a_skip_hi[1] = 0; g2.cin_skip_lo = 0; g2.bt1 = nullptr; g2.a_bf16 = 0; g2.c_bf16 = 0; g2.c_h = la.c_h;
    wn_gemm_epilogue<WN_EPI_PLAIN, 2>(g2, acc2, m0 + 32 * wr, 64 * wc, lane, stage);   // (after the loop's last barrier: the B chunks are done with)
}

// ---- The backward's counterpart of wn_fwd_layer_bf16: the dx product of layer l and the gate-derivative product of layer l - 1 in ONE
// kernel (bf16 step, R = D = 128).  dx_l = dx' + [dF|dG](t) . Wfg(tap 1) + [dF|dG](t + d) . Wfg(tap 0) is a 128-row x 128-column tile
// (K = 4D over two row-windowed views, as wn_fwd_gemm_bf16<WN_EPI_PLAIN, 4, true>); layer l - 1's dz = dx_l . Wres is a product over the SAME
// rows with dx_l as its A operand -- the tile goes to HBM (the residual weight gradient and the next dx product's addend read it) AND
// stays in LDS as bf16, Wres(l - 1) streams through LDS, and the second product's epilogue is the gate derivative (WN_EPI_GATE_BWD:
// [dF|dG] of layer l - 1).  Unfused, dx_l (0.25 GB at config 5) is read back by the next launch and each launch pays its own tail.  Same
// operands, same chunk order, same accumulation: bit-identical to the two kernels.
__global__ __launch_bounds__(512, 4) void wn_bwd_layer_bf16(WnGemmArgsBf16 ga, WnGemmArgsBf16 gb) {
    const WnGemmArgs& g = ga.g;    // the dx product (A stored as bf16: [dF|dG] of layer l)
    const WnGemmArgs& g2 = gb.g;   // the gate-derivative product of layer l - 1 (its A operand is this kernel's output tile)
    constexpr int NT = 512, TM = 128, TN = 128, KC = WN_GEMM_BF16_KC, LD = KC + 8;
    constexpr int TPR = NT / TM, HK = KC / TPR;
    constexpr int K2 = 128, N2 = 128, LD2 = K2 + 8;
    static_assert(KC == 32 && HK == 8, "one 16-byte piece per loader thread and operand");
    constexpr int STAGE_SHORTS = 8 * WN_EPI_TILE_FLOATS * 2, OPER_SHORTS = 2 * TM * LD + 2 * TN * LD, ZOFF = STAGE_SHORTS;
    static_assert(2 * N2 * LD <= STAGE_SHORTS && OPER_SHORTS <= ZOFF + TM * LD2, "LDS plan of the fused kernels");
    __shared__ __attribute__((aligned(16))) unsigned short smem_h[ZOFF + TM * LD2];
    unsigned short (*a_s)[TM * LD] = reinterpret_cast<unsigned short (*)[TM * LD]>(smem_h);
    unsigned short (*b_s)[TN * LD] = reinterpret_cast<unsigned short (*)[TN * LD]>(smem_h + 2 * TM * LD);
    unsigned short* x_s = smem_h + ZOFF;
    unsigned short (*b2_s)[N2 * LD] = reinterpret_cast<unsigned short (*)[N2 * LD]>(smem_h);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wr = wv & 3, wc = wv >> 2;
    const long long tile = wn_layer_tile(g.M);
    if (tile < 0) return;   // (block-uniform)
    const long long m0 = tile * TM;
    const int bcol = tid >> 2, bpart = tid & 3;   // B loaders of both products: 128 columns x 4 pieces of 8 bf16 per chunk
    wn_f16v acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    {   // ---- first product (wave (wr, wc): rows 32 wr .., columns 64 wc ..)
        const int lrow = tid / TPR, lpart = tid % TPR;
        const long long am = m0 + lrow;
        const bool arow_ok = am < g.M;
        const unsigned aq = arow_ok ? (unsigned)am / (unsigned)g.rows_per_batch : 0u, arem = arow_ok ? (unsigned)am - aq * (unsigned)g.rows_per_batch : 0u;
        const bool ok0 = arow_ok && (int)arem >= g.a_skip_lo[0] && (int)arem < g.rows_per_batch - g.a_skip_hi[0];
        const bool ok1 = arow_ok && (int)arem >= g.a_skip_lo[1] && (int)arem < g.rows_per_batch - g.a_skip_hi[1];
        const unsigned short* a0p = reinterpret_cast<const unsigned short*>(g.a0.base) + (long long)aq * g.a0.batch_stride + (g.a0.t0 + (long long)arem) * g.a0.row_stride + lpart * HK;
        const unsigned short* a1p = reinterpret_cast<const unsigned short*>(g.a1.base) + (long long)aq * g.a1.batch_stride + (g.a1.t0 + (long long)arem) * g.a1.row_stride + lpart * HK;
        const bool bcol_ok = bcol < g.N;
        const int ldb = ga.ldb ? ga.ldb : g.K;
        const unsigned short* bp0 = ga.bn + (size_t)bcol * ldb + bpart * 8;
        const unsigned short* bp1 = ga.bn1 ? ga.bn1 + (size_t)bcol * ldb + bpart * 8 : bp0 + g.k_split;
        float4 va;
        uint4 vb;
        auto fetch = [&](int kc) {
            const int k0 = kc * KC;
            const bool first = k0 < g.k_split, ok = first ? ok0 : ok1;
            const unsigned short* src = (first ? a0p : a1p) + (first ? k0 : k0 - g.k_split);
            va = ok ? *reinterpret_cast<const float4*>(src) : make_float4(0.f, 0.f, 0.f, 0.f);
            const uint4* bsrc = reinterpret_cast<const uint4*>(first ? bp0 + k0 : bp1 + (k0 - g.k_split));
            vb = bcol_ok ? *bsrc : make_uint4(0u, 0u, 0u, 0u);
        };
        auto stash = [&](int buf) {
            *reinterpret_cast<float4*>(a_s[buf] + lrow * LD + lpart * HK) = va;
            *reinterpret_cast<uint4*>(b_s[buf] + bcol * LD + bpart * 8) = vb;
        };
        const int nchunks = g.K / KC;
        fetch(0);
        stash(0);
        __syncthreads();
        for (int kc = 0; kc < nchunks; ++kc) {
            const int buf = kc & 1;
            if (kc + 1 < nchunks) fetch(kc + 1);
            const unsigned short* ar = a_s[buf] + (32 * wr + (lane & 31)) * LD + 8 * (lane >> 5);
            const unsigned short* br = b_s[buf] + (64 * wc + (lane & 31)) * LD + 8 * (lane >> 5);
#pragma unroll
            for (int ks = 0; ks < KC / 16; ++ks) {
                const wn_bf16x8 a = *reinterpret_cast<const wn_bf16x8*>(ar + 16 * ks);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const wn_bf16x8 b = *reinterpret_cast<const wn_bf16x8*>(br + 32 * j * LD + 16 * ks);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
                }
            }
            if (kc + 1 < nchunks) stash(buf ^ 1);
            __syncthreads();
        }
    }
    float* stage = reinterpret_cast<float*>(smem_h) + wv * WN_EPI_TILE_FLOATS;
    // ---- dx_l: to HBM (fp32) and into the LDS image (bf16: what the second product's loader would round it to)
    wn_gemm_epilogue<WN_EPI_PLAIN, 2>(g, acc, m0 + 32 * wr, 64 * wc, lane, stage, x_s + (32 * wr) * LD2, LD2);
    __syncthreads();   // the image is complete, the staging tiles are free
    wn_f16v acc2[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc2[j][i] = 0.f;
    {   // ---- second product: dz tile = image . Wres(l - 1)  ([D][R] bf16, K = R contiguous)
        const unsigned short* bp = gb.bn + (size_t)bcol * K2 + bpart * 8;
        uint4 vb2 = *reinterpret_cast<const uint4*>(bp);
        *reinterpret_cast<uint4*>(b2_s[0] + bcol * LD + bpart * 8) = vb2;
        __syncthreads();
        const unsigned short* xr = x_s + (32 * wr + (lane & 31)) * LD2 + 8 * (lane >> 5);
#pragma unroll
        for (int kc = 0; kc < K2 / KC; ++kc) {
            const int buf = kc & 1;
            if (kc + 1 < K2 / KC) vb2 = *reinterpret_cast<const uint4*>(bp + (kc + 1) * KC);
            const unsigned short* br = b2_s[buf] + (64 * wc + (lane & 31)) * LD + 8 * (lane >> 5);
#pragma unroll
            for (int ks = 0; ks < KC / 16; ++ks) {
                const wn_bf16x8 a = *reinterpret_cast<const wn_bf16x8*>(xr + KC * kc + 16 * ks);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const wn_bf16x8 b = *reinterpret_cast<const wn_bf16x8*>(br + 32 * j * LD + 16 * ks);
                    acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc2[j], 0, 0, 0);
                }
            }
            if (kc + 1 < K2 / KC) *reinterpret_cast<uint4*>(b2_s[buf ^ 1] + bcol * LD + bpart * 8) = vb2;
            __syncthreads();
        }
    }
    wn_gemm_epilogue<WN_EPI_GATE_BWD, 2>(g2, acc2, m0 + 32 * wr, 64 * wc, lane, stage);   // [dF|dG] of layer l - 1
}

// out[i] = bf16(in[i]) (round to nearest even): the backward products' weight operands, [N][K] row-major, are the forward
// banks as they are
__global__ void wn_cvt_bf16(const float* in, unsigned short* out, long long n) {
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (i + 1 < n) *reinterpret_cast<unsigned*>(out + i) = wn_pack_bf16(in[i], in[i + 1]);
    else if (i < n) out[i] = (unsigned short)(wn_pack_bf16(in[i], 0.f) & 0xffffu);
}

// out[b][c][r] = bf16(in[b][r][c]): the forward products' weight operands ([N][K]) from the fp32 banks ([K][N])
__global__ void wn_cvt_bf16_transposed(const float* in, long long in_batch_stride, unsigned short* out, int rows, int cols) {
    __shared__ float tile[32][33];
    const float* ib = in + (long long)blockIdx.z * in_batch_stride;
    unsigned short* ob = out + (long long)blockIdx.z * rows * cols;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8)
        if (r0 + i < rows && c0 + tx < cols) tile[i][tx] = ib[(long long)(r0 + i) * cols + c0 + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
        if (c0 + i < cols && r0 + tx < rows) ob[(long long)(c0 + i) * rows + r0 + tx] = (unsigned short)(wn_pack_bf16(tile[tx][i], 0.f) & 0xffffu);
}

// x0[(n,t)][r] = start_conv.weight[r][idx[n][t]] (+ bias): the one-hot input makes start_conv a column gather (wavenet_model.py:127)
__global__ void wn_fwd_start(const int32_t* idx, const float* start_t, const float* start_b, float* x, long long rows, int R, unsigned short* xh = nullptr) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long row = i / (R / 4);
    const int q = (int)(i % (R / 4));
    if (row >= rows) return;
    float4 v = *reinterpret_cast<const float4*>(start_t + (size_t)idx[row] * R + q * 4);
    if (start_b) { v.x += start_b[q * 4]; v.y += start_b[q * 4 + 1]; v.z += start_b[q * 4 + 2]; v.w += start_b[q * 4 + 3]; }
    *reinterpret_cast<float4*>(x + row * R + q * 4) = v;
    if (xh) *reinterpret_cast<uint2*>(xh + row * R + q * 4) = wn_pack_bf16x4(v);   // (the bf16 step's shadow of x, see WnGemmArgs::c_h)
}


// ------------------------------------------------------------------------------------------------ backward pass
// Weight gradients: C[Ka][Nb] += sum_m A[m][ka] * B[m][nb]  ("TN": the contraction runs over the rows).  Grid
// (Ka/128, Nb/128, splits): every workgroup reduces its slice of the rows into a 128x128 register tile (4 waves x 4
// accumulator tiles, fp32 MFMA) and adds it to C with fp32 atomics (the order of the splits is not fixed: gradients are
// reproducible to rounding, not bitwise).  A may be a one-hot matrix given by class indices (start_conv's gradient).
struct WnGemmTnArgs {
    WnRowMap a;            // rows of >= ka0+128 floats (ignored when a_idx != NULL)
    const int32_t* a_idx;  // A[m][k] = (a_idx[row(m)] == k): one-hot rows addressed through `a` (base = NULL offset trick: see loader)
    WnRowMap b;
    int Ka, Nb;            // logical sizes (multiples of 32); C is [Ka][ldc]
    float* c;
    int ldc;
    long long M;
    int rows_per_batch;
    int relu_a;            // A := max(A, 0)
    long long rows_per_split;
    int tiles_ka, n_splits; // grid: 8 * (tiles_ka * tiles_nb) * ceil(n_splits / 8) workgroups (wn_tile_of: the tiles of a row split share an XCD)
    WnRowMap a1;           // ka_split > 0: columns ka >= ka_split of A are columns (ka - ka_split) of this second row view (ka_split % 128 == 0):
    int ka_split;          //   the two taps of the filter/gate weight gradient in one launch -- their workgroups read the same rows of B
    int b_bf16;            //   at the same time, so the second read is served by the caches instead of HBM
                           // b_bf16: B is STORED as bf16 (its base points at unsigned short, its strides count bf16 elements): the bf16
                           // step's [dF|dG] and z.  wn_bwd_gemm_tn_bf16<*, true> only.
    int c_trans, a_bf16;   // c_trans: C is stored transposed, element (ka, nb) at c[nb * ldc + ka] (operands swapped by the caller so that the
                           // bf16-stored one is B);  a_bf16: A is stored as bf16 (excludes relu_a, a_idx)
    int a_skip_lo;         // row window of view `a` (not a1): it reads as ZERO on the first a_skip_lo rows of every batch entry (their addresses are
                           // never formed into loads) -- the tap x(t - d) where the reference's left zero padding stands in for it
    float* part;           // deterministic mode (wn_train_set_deterministic): != NULL -- every row split STORES its partial tile at part[split][ka][nb]
};                         // (Nb columns per row) instead of adding it to C with atomics; wn_tn_reduce then adds the splits in their order

__global__ __launch_bounds__(256, WN_GEMM_MINB) void wn_bwd_gemm_tn(WnGemmTnArgs g) {
    constexpr int T = 128, KC = WN_GEMM_KC, NQ = KC / 8, LT = 256 / KC;  // NQ float4 per thread per operand, LT threads per row
    __shared__ float a_s[2][KC * T];  // [m][ka]
    __shared__ float b_s[2][KC * T];  // [m][nb]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    unsigned split, tile;
    if (!wn_tile_of(blockIdx.x, (unsigned)(g.tiles_ka * ((g.Nb + T - 1) / T)), (unsigned)g.n_splits, split, tile)) return;
    const int ka0 = (int)(tile % (unsigned)g.tiles_ka) * T, nb0 = (int)(tile / (unsigned)g.tiles_ka) * T;
    const long long m_begin = (long long)split * g.rows_per_split;
    long long m_end = m_begin + g.rows_per_split;
    if (m_end > g.M) m_end = g.M;
    if (m_begin >= m_end) return;
    wn_f16v acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    const int lrow = tid / LT, lcol = (tid % LT) * (KC / 2);  // loader: row of the chunk, KC/2 floats of its 128
    const bool second = g.ka_split > 0 && ka0 >= g.ka_split;
    const WnRowMap& amap = second ? g.a1 : g.a;
    const int kap0 = second ? ka0 - g.ka_split : ka0;          // first column of this tile inside its view
    float4 va[NQ], vb[NQ];
    constexpr unsigned WN_OOB = 0x80000000u;   // beyond the 2 GB window of wn_rsrc: the load returns zeros
    auto row_elem = [&](const WnRowMap& map, unsigned q, unsigned rem) -> long long { return (long long)q * map.batch_stride + (map.t0 + (long long)rem) * map.row_stride; };
    const unsigned q_wg = (unsigned)((unsigned long long)m_begin / (unsigned)g.rows_per_batch), rem_wg = (unsigned)m_begin - q_wg * (unsigned)g.rows_per_batch;
    const long long ea_wg = g.a_idx ? 0 : row_elem(amap, q_wg, rem_wg), eb_wg = row_elem(g.b, q_wg, rem_wg);   // descriptors start at the row split's first row
    const __amdgpu_buffer_rsrc_t rs_a = wn_rsrc(g.a_idx ? g.b.base : amap.base + ea_wg), rs_b = wn_rsrc(g.b.base + eb_wg);
    auto fetch = [&](long long mc) {
        const long long m = mc + lrow;
        const bool ok = m < m_end;
        const unsigned q_m = (unsigned)m / (unsigned)g.rows_per_batch, rem_m = (unsigned)m - q_m * (unsigned)g.rows_per_batch;
        if (g.a_idx) {
            int cls = -1;
            if (ok) {  // the row map of `a` addresses the index array: base holds no data, strides are in elements
                const unsigned q = (unsigned)m / (unsigned)g.rows_per_batch, rem = (unsigned)m - q * (unsigned)g.rows_per_batch;
                cls = g.a_idx[(long long)q * g.a.batch_stride + (g.a.t0 + (long long)rem) * g.a.row_stride];
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int k = ka0 + lcol + q * 4;
                va[q] = make_float4(cls == k ? 1.f : 0.f, cls == k + 1 ? 1.f : 0.f, cls == k + 2 ? 1.f : 0.f, cls == k + 3 ? 1.f : 0.f);
            }
        } else {
            // (unconditional, range-checked buffer loads -- an offset beyond the descriptor's window reads as zero -- instead of `ok ? *ptr : zero`: every
            //  predicated load sat in a block of its own with a full wait behind it, four round trips per 16-row chunk and thread.  Round 6, as in
            //  wn_bwd_gemm_tn_bf16: profiles/r06_tn_loads.txt.)
            const bool oka = ok && (second || (int)rem_m >= g.a_skip_lo);
            const unsigned offa = (unsigned)((row_elem(amap, q_m, rem_m) - ea_wg + kap0 + lcol) * 4);
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const wn_v4i got = __builtin_amdgcn_raw_buffer_load_b128(rs_a, (oka && ka0 + lcol + q * 4 < g.Ka) ? offa + 16u * q : WN_OOB, 0, 0);
                va[q] = make_float4(__int_as_float(got.x), __int_as_float(got.y), __int_as_float(got.z), __int_as_float(got.w));
            }
        }
        const unsigned offb = (unsigned)((row_elem(g.b, q_m, rem_m) - eb_wg + nb0 + lcol) * 4);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const wn_v4i got = __builtin_amdgcn_raw_buffer_load_b128(rs_b, (ok && nb0 + lcol + q * 4 < g.Nb) ? offb + 16u * q : WN_OOB, 0, 0);
            vb[q] = make_float4(__int_as_float(got.x), __int_as_float(got.y), __int_as_float(got.z), __int_as_float(got.w));
        }
    };
    auto stash = [&](int buf) {
        float* ad = a_s[buf] + lrow * T + lcol;
        float* bd = b_s[buf] + lrow * T + lcol;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            float4 x = va[q];
            if (g.relu_a) { x.x = fmaxf(x.x, 0.f); x.y = fmaxf(x.y, 0.f); x.z = fmaxf(x.z, 0.f); x.w = fmaxf(x.w, 0.f); }
            *reinterpret_cast<float4*>(ad + q * 4) = x;
            *reinterpret_cast<float4*>(bd + q * 4) = vb[q];
        }
    };
    fetch(m_begin);
    stash(0);
    __syncthreads();
    int buf = 0;
    for (long long mc = m_begin; mc < m_end; mc += KC, buf ^= 1) {
        if (mc + KC < m_end) fetch(mc + KC);
        const float* as = a_s[buf] + 32 * wv + (lane & 31);
        const float* bs = b_s[buf] + (lane & 31);
        const int kh = lane >> 5;
#pragma unroll
        for (int ks = 0; ks < KC / 2; ++ks) {
            const float a = as[(2 * ks + kh) * T];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bs[(2 * ks + kh) * T + 32 * j], acc[j], 0, 0, 0);
        }
        if (mc + KC < m_end) stash(buf ^ 1);
        __syncthreads();
    }
    const int col = lane & 31;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int ka = ka0 + 32 * wv + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
        if (ka >= g.Ka) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int nb = nb0 + 32 * j + col;
            if (nb >= g.Nb) continue;
            if (g.part) g.part[((size_t)split * g.Ka + ka) * g.Nb + nb] = acc[j][i];
            else unsafeAtomicAdd(g.c + (size_t)ka * g.ldc + nb, acc[j][i]);
        }
    }
}

// Deterministic mode: C[ka][nb] (or its transpose) = the row splits' partial tiles added in the order of the splits -- one thread per four columns.
// Also the second half of the bias gradients' column sums (Ka = 1, one "split" per 512-row block).
__global__ __launch_bounds__(256) void wn_tn_reduce(const float* part, int n_splits, int Ka, int Nb, float* c, int ldc, int c_trans) {
    const long long e4 = (long long)blockIdx.x * 256 + threadIdx.x, per = (long long)Ka * Nb;
    if (e4 * 4 >= per) return;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int sp = 0; sp < n_splits; ++sp) {
        const float4 v = *reinterpret_cast<const float4*>(part + (size_t)sp * per + e4 * 4);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const int ka = (int)(e4 * 4 / Nb), nb = (int)(e4 * 4 % Nb);   // (Nb is a multiple of 4: the four columns share a row)
    if (c_trans) { c[(size_t)nb * ldc + ka] = s.x; c[(size_t)(nb + 1) * ldc + ka] = s.y; c[(size_t)(nb + 2) * ldc + ka] = s.z; c[(size_t)(nb + 3) * ldc + ka] = s.w; }
    else *reinterpret_cast<float4*>(c + (size_t)ka * ldc + nb) = s;
}

// The same weight-gradient product with bf16 MATRIX OPERANDS (fp32 accumulation, fp32 atomics into the gradient): the rows of
// A and B are rounded to bf16 (RNE) on their way into LDS and multiplied with v_mfma_f32_32x32x16_bf16 (16x the fp32 matrix
// rate), which turns the product from matrix-pipe bound (104 TFLOP/s of fp32 MFMA) into a stream over the two operands.
// The contraction runs over the ROWS, so both operands are needed "transposed" ([column][8 consecutive rows] per lane): a loader
// thread reads a 4-column x 8-row patch (eight coalesced float4 loads), transposes it in registers and writes four 16-byte
// [column][row 0..7] vectors.  Used by wn_train_backward when the step runs with bf16 operands (wn_set_forward_precision);
// the one-hot product of start_conv stays on the fp32 kernel.
#ifndef WN_TN_BF16_MINB
#define WN_TN_BF16_MINB 3
#endif
// WAVES = 4: 128 x 128 tile of C.  WAVES = 8: 128 (Ka) x 256 (Nb) tile for Nb >= 256 (wave w: ka strip 32 (w & 3).., nb half w >> 2), so
// that A -- the layer input x in the filter/gate weight gradient, 262 MB -- is streamed once instead of once per 128 columns of B.
// B16: B is stored as bf16 rows.  Its chunk is copied to LDS as it is -- row-major [KC rows][TB columns], 16-byte pieces, no conversion,
// no register transpose -- and the MFMA operand (8 consecutive ROWS of one column per lane) is gathered by the transposing LDS read
// ds_read_b64_tr_b16: lane p of a 16-lane group passes the address of 4 contiguous bf16 of row (p >> 2), columns 4 (p & 3)..; it
// receives column p of that [4 rows][16 columns] block (checked on the device: tools/tr_probe.hip).  Two reads (rows +0..3, +4..7) make
// one operand.  Row stride TB * 2 + 64 bytes: the four rows of a block and the two column blocks of a 32-lane half land in 64
// different banks.
typedef short wn_s4 __attribute__((ext_vector_type(4)));
typedef short wn_s8 __attribute__((ext_vector_type(8)));
// A16: A is stored as bf16 (same image, same reads, on the A side).  Measured a little slower than an fp32-stored A (160 vs 148 us in the
// residual weight gradient) where B16 is a clear gain (211 -> 160 us in the filter/gate one), so where the choice exists the bf16-stored
// operand goes in as B (operands swapped, c_trans) -- but only for few row splits: a transposed tile of atomics touches 32x the cache
// lines (residual weight gradient, ~1000 splits: 598 us).
#ifndef WN_TN_DEEP
#define WN_TN_DEEP 0   // 1: two chunks in flight per workgroup where both operands are stored as bf16 (see the K loop) -- built in round 6, at the 128-register cap of the 256-column form it spills, at 256 registers (one workgroup per CU) it is level: off -- profiles/r06_tn_loads.txt
#endif
template <int WAVES, bool A16, bool B16>
__global__ __launch_bounds__(64 * WAVES, WAVES == 4 ? WN_TN_BF16_MINB : ((A16 && !(B16 && WN_TN_DEEP)) ? 4 : 2)) void wn_bwd_gemm_tn_bf16(WnGemmTnArgs g) {
    constexpr int T = 128, TB = 32 * WAVES, KC = 32, LD = KC + 8;  // fp32-stored operand: LDS rows [column][KC rows of the chunk] bf16, padded to 80 bytes
    constexpr int RSA = T * 2 + 64, RSB = TB * 2 + 64;             // bf16-stored operand: bytes per row of its row-major image
    constexpr int ASZ = A16 ? KC * RSA / 2 : T * LD, BSZ = B16 ? KC * RSB / 2 : TB * LD;   // shorts per buffer
    __shared__ __attribute__((aligned(16))) unsigned short a_s[2][ASZ];
    __shared__ __attribute__((aligned(16))) unsigned short b_s[2][BSZ];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wr = wv & 3, wc = wv >> 2;
    unsigned split, tile;
    if (!wn_tile_of(blockIdx.x, (unsigned)(g.tiles_ka * ((g.Nb + TB - 1) / TB)), (unsigned)g.n_splits, split, tile)) return;
    const int ka0 = (int)(tile % (unsigned)g.tiles_ka) * T, nb0 = (int)(tile / (unsigned)g.tiles_ka) * TB;
    const long long m_begin = (long long)split * g.rows_per_split;
    long long m_end = m_begin + g.rows_per_split;
    if (m_end > g.M) m_end = g.M;
    if (m_begin >= m_end) return;
    wn_f16v acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    // loader: threads 0-127 feed A, the next 128 (256) feed B, 128 columns each (WAVES = 8: the last 128 threads load nothing).
    // fp32-stored operand: unit = 8 rows (group mg) x 4 columns (group cg), transposed in registers on the way to LDS.
    // bf16-stored operand: loader thread lt of the operand copies the 16-byte pieces (row lt / PPR + 8 q, piece lt % PPR), q = 0..3
    // (PPR = pieces per row) -- consecutive threads, consecutive 16 bytes of a row.
    const int role = tid >> 7;                    // 0: A;  1 ..: B columns 128 (role - 1) ..
    const bool is_b = role >= 1, loads = role <= TB / 128;
    const int u = tid & 127, mg = u >> 5, cg = u & 31;
    const bool second = g.ka_split > 0 && ka0 >= g.ka_split;
    const WnRowMap& rm = is_b ? g.b : (second ? g.a1 : g.a);
    const int lcol = is_b ? 128 * (role - 1) + 4 * cg : 4 * cg;   // column inside the tile
    const int col0 = (is_b ? nb0 : ka0) + lcol, ncols = is_b ? g.Nb : g.Ka;
    const int pcol0 = (!is_b && second) ? col0 - g.ka_split : col0;   // column inside the row view
    const bool col_ok = loads && col0 < ncols;
    const bool relu = !is_b && g.relu_a;
    float4 v[8];
    // bf16-stored operand `map` (g.a or g.b named directly: its fields stay scalar), tile origin `org`, PPR pieces per row, loader thread lt
    // win_lo: rows whose index inside the batch entry is below it read as ZERO (never loaded)
    // (Loads are UNCONDITIONAL buffer loads: a piece that lies outside the operand gets an offset beyond the descriptor's range and reads as zero.  Written
    //  as `ok ? *ptr : zero` every load sat in a predicated block of its own with an s_waitcnt vmcnt(0) at its end -- a loader thread's four (eight) loads of
    //  a chunk were four (eight) round trips one after the other, and the weight-gradient products ran at half the rate of a stream: round 6,
    //  profiles/r06_tn_loads.txt.  The descriptor starts at the row split's first row -- wave-uniform --, so a lane carries ONE 32-bit byte offset per load
    //  instead of a 64-bit pointer: these kernels sit at their register cap.)
    constexpr unsigned WN_OOB = 0x80000000u;   // beyond the 2 GB window of wn_rsrc: the load returns zeros
    // element offset of row m of a map (64-bit), and the wave-uniform origin of this workgroup's rows
    auto row_elem = [&](const WnRowMap& map, unsigned q, unsigned rem) -> long long { return (long long)q * map.batch_stride + (map.t0 + (long long)rem) * map.row_stride; };
    const unsigned q_wg = (unsigned)((unsigned long long)m_begin / (unsigned)g.rows_per_batch), rem_wg = (unsigned)m_begin - q_wg * (unsigned)g.rows_per_batch;
    auto fetch16 = [&](const WnRowMap& map, int org, int ncols16, auto pprc, int lt, long long mc, int win_lo = 0, int vo = 0) {   // vo: first register of the set (DEEP)
        constexpr int PPR = decltype(pprc)::value;
        const int row16 = lt / PPR, piece16 = lt % PPR;
        const bool ok16 = org + 8 * piece16 < ncols16;
        const long long e_wg = row_elem(map, q_wg, rem_wg);
        const __amdgpu_buffer_rsrc_t rs = wn_rsrc(reinterpret_cast<const unsigned short*>(map.base) + e_wg);
        const long long m = mc + row16;
        unsigned q = (unsigned)m / (unsigned)g.rows_per_batch, rem = (unsigned)m - q * (unsigned)g.rows_per_batch;
        unsigned off = (unsigned)((row_elem(map, q, rem) - e_wg + org + 8 * piece16) * 2);   // bytes from the descriptor's base
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            const bool okq = ok16 && m + 8 * qq < m_end && (int)rem >= win_lo;
            const wn_v4i got = __builtin_amdgcn_raw_buffer_load_b128(rs, okq ? off : WN_OOB, 0, 0);   // 8 bf16, moved as bits
            v[vo + qq] = make_float4(__int_as_float(got.x), __int_as_float(got.y), __int_as_float(got.z), __int_as_float(got.w));
            rem += 8;
            if (rem >= (unsigned)g.rows_per_batch) {   // the row 8 further down is in a later batch entry
                do { rem -= (unsigned)g.rows_per_batch; ++q; } while (rem >= (unsigned)g.rows_per_batch);
                off = (unsigned)((row_elem(map, q, rem) - e_wg + org + 8 * piece16) * 2);
            } else {
                off += (unsigned)(8 * map.row_stride * 2);
            }
        }
    };
    auto stash16 = [&](unsigned short* imgs, int rs, auto pprc, int lt, int vo = 0) {
        constexpr int PPR = decltype(pprc)::value;
        char* img = reinterpret_cast<char*>(imgs);
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) *reinterpret_cast<float4*>(img + (lt / PPR + 8 * qq) * rs + 16 * (lt % PPR)) = v[vo + qq];
    };
    auto fetch = [&](long long mc, int vo = 0) {
        if (B16 && is_b) { if (loads) fetch16(g.b, nb0, g.Nb, std::integral_constant<int, TB / 8>{}, tid - 128, mc, 0, vo); return; }
        if (A16 && !is_b) {   // (the two tap views of the filter/gate weight gradient: block-uniform choice, the maps' fields stay scalar)
            if (second) fetch16(g.a1, ka0 - g.ka_split, g.Ka - g.ka_split, std::integral_constant<int, T / 8>{}, tid, mc, 0, vo);
            else fetch16(g.a, ka0, g.ka_split > 0 ? g.ka_split : g.Ka, std::integral_constant<int, T / 8>{}, tid, mc, g.a_skip_lo, vo);
            return;
        }
        long long m = mc + mg * 8;
        unsigned q = (unsigned)m / (unsigned)g.rows_per_batch, rem = (unsigned)m - q * (unsigned)g.rows_per_batch;
        const long long e_wg = row_elem(rm, q_wg, rem_wg);
        const __amdgpu_buffer_rsrc_t rs = wn_rsrc(rm.base + e_wg);
        unsigned off = (unsigned)((row_elem(rm, q, rem) - e_wg + pcol0) * 4);
        const int win_lo = (is_b || second) ? 0 : g.a_skip_lo;   // (view `a` only)
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            const bool okr = col_ok && m + rr < m_end && (int)rem >= win_lo;
            const wn_v4i got = __builtin_amdgcn_raw_buffer_load_b128(rs, okr ? off : WN_OOB, 0, 0);
            v[rr] = make_float4(__int_as_float(got.x), __int_as_float(got.y), __int_as_float(got.z), __int_as_float(got.w));
            if (++rem == (unsigned)g.rows_per_batch) {  // next row is in the next batch entry
                rem = 0; ++q;
                off = (unsigned)((row_elem(rm, q, 0) - e_wg + pcol0) * 4);
            } else {
                off += (unsigned)(rm.row_stride * 4);
            }
        }
    };
    auto stash = [&](int buf, int vo = 0) {
        if (!loads) return;
        if (B16 && is_b) { stash16(b_s[buf], RSB, std::integral_constant<int, TB / 8>{}, tid - 128, vo); return; }
        if (A16 && !is_b) { stash16(a_s[buf], RSA, std::integral_constant<int, T / 8>{}, tid, vo); return; }
        unsigned short* dst = (is_b ? b_s[buf] : a_s[buf]) + lcol * LD + mg * 8;
        if (relu) {
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) { v[rr].x = fmaxf(v[rr].x, 0.f); v[rr].y = fmaxf(v[rr].y, 0.f); v[rr].z = fmaxf(v[rr].z, 0.f); v[rr].w = fmaxf(v[rr].w, 0.f); }
        }
        *reinterpret_cast<uint4*>(dst) = make_uint4(wn_pack_bf16(v[0].x, v[1].x), wn_pack_bf16(v[2].x, v[3].x), wn_pack_bf16(v[4].x, v[5].x), wn_pack_bf16(v[6].x, v[7].x));
        *reinterpret_cast<uint4*>(dst + LD) = make_uint4(wn_pack_bf16(v[0].y, v[1].y), wn_pack_bf16(v[2].y, v[3].y), wn_pack_bf16(v[4].y, v[5].y), wn_pack_bf16(v[6].y, v[7].y));
        *reinterpret_cast<uint4*>(dst + 2 * LD) = make_uint4(wn_pack_bf16(v[0].z, v[1].z), wn_pack_bf16(v[2].z, v[3].z), wn_pack_bf16(v[4].z, v[5].z), wn_pack_bf16(v[6].z, v[7].z));
        *reinterpret_cast<uint4*>(dst + 3 * LD) = make_uint4(wn_pack_bf16(v[0].w, v[1].w), wn_pack_bf16(v[2].w, v[3].w), wn_pack_bf16(v[4].w, v[5].w), wn_pack_bf16(v[6].w, v[7].w));
    };
    // bf16-stored operands: byte offset of this lane's first transposing read inside a buffer (ks = 0, j = 0, rows +0..3)
    typedef __attribute__((address_space(3))) wn_s4 lds_s4;
    const int tp = lane & 15, tg = (lane >> 4) & 1, th = lane >> 5;
    const int tra_off = (8 * th + (tp >> 2)) * RSA + (32 * wr + 16 * tg + 4 * (tp & 3)) * 2;
    const int trb_off = (8 * th + (tp >> 2)) * RSB + (128 * wc + 16 * tg + 4 * (tp & 3)) * 2;
    auto products = [&](int buf) {   // the chunk in LDS buffer `buf` against the accumulators
        const unsigned short* ar = a_s[buf] + (A16 ? 0 : (32 * wr + (lane & 31)) * LD + 8 * (lane >> 5));
        const unsigned short* br = b_s[buf] + (B16 ? 0 : (128 * wc + (lane & 31)) * LD + 8 * (lane >> 5));
        const char* at_img = reinterpret_cast<const char*>(a_s[buf]) + tra_off;
        const char* bt_img = reinterpret_cast<const char*>(b_s[buf]) + trb_off;
#pragma unroll
        for (int ks = 0; ks < KC / 16; ++ks) {
            wn_bf16x8 a;
            if constexpr (A16) {
                const char* at = at_img + ks * 16 * RSA;
                const wn_s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(at));
                const wn_s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(at + 4 * RSA));
                a = __builtin_bit_cast(wn_bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            } else {
                a = *reinterpret_cast<const wn_bf16x8*>(ar + 16 * ks);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                wn_bf16x8 b;
                if constexpr (B16) {
                    const char* at = bt_img + ks * 16 * RSB + j * 64;
                    const wn_s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(at));
                    const wn_s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(at + 4 * RSB));
                    b = __builtin_bit_cast(wn_bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                } else {
                    b = *reinterpret_cast<const wn_bf16x8*>(br + 32 * j * LD + 16 * ks);
                }
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
            }
        }
    };
    // DEEP (both operands stored as bf16: a loader thread moves four 16-byte pieces per chunk, half of v[]): TWO chunks in flight per workgroup.  Alone on the
    // chip the one-chunk loop ran the filter/gate weight gradient at 2.5 TB/s -- 3.7 us per 24 KB chunk and workgroup, a load round trip per chunk, where the
    // layer kernels stream 5 TB/s (profiles/r06_train_step_sq_counters.txt) -- and the step is the sum of its kernels' stand-alone times.  Chunk k + 2 is
    // requested before the products of chunk k, chunk k + 1 goes from its registers to LDS behind them; the barrier is the LDS-only one (a __syncthreads
    // would drain the loads in flight).  Loads past the split's last row are predicated off (zeros), so every path issues the same sequence and the
    // compiler's wait counts stay exact.
    constexpr bool DEEP = A16 && B16 && WN_TN_DEEP;
    if constexpr (DEEP) {
        fetch(m_begin, 0);
        stash(0, 0);
        fetch(m_begin + KC, 4);
        wn_lds_barrier();
        for (long long mc = m_begin;;) {
            fetch(mc + 2 * KC, 0);
            products(0);
            stash(1, 4);
            wn_lds_barrier();
            mc += KC;
            if (mc >= m_end) break;
            fetch(mc + 2 * KC, 4);
            products(1);
            stash(0, 0);
            wn_lds_barrier();
            mc += KC;
            if (mc >= m_end) break;
        }
    } else {
        fetch(m_begin);
        stash(0);
        __syncthreads();
        int buf = 0;
        for (long long mc = m_begin; mc < m_end; mc += KC, buf ^= 1) {
            if (mc + KC < m_end) fetch(mc + KC);
            products(buf);
            if (mc + KC < m_end) stash(buf ^ 1);
            __syncthreads();
        }
    }
    const int col = lane & 31;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int ka = ka0 + 32 * wr + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
        if (ka >= g.Ka) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int nb = nb0 + 128 * wc + 32 * j + col;
#if WN_ABL_TN_PLAIN_STORE   // timing ablation (results wrong): what the fp32 atomics of the row splits cost
            if (nb < g.Nb) g.c[g.c_trans ? (size_t)nb * g.ldc + ka : (size_t)ka * g.ldc + nb] = acc[j][i];
#else
            if (nb >= g.Nb) continue;
            if (g.part) g.part[((size_t)split * g.Ka + ka) * g.Nb + nb] = acc[j][i];
            else unsafeAtomicAdd(g.c + (g.c_trans ? (size_t)nb * g.ldc + ka : (size_t)ka * g.ldc + nb), acc[j][i]);
#endif
        }
    }
}

// ---- Weight gradients with both operands stored as bf16 in 256 x 256 tiles, one workgroup per CU (round 6): the filter/gate gradient of a 128 / 128
// layer (ONE tile: both tap views side by side) and the grouped skip gradient (Ka = 512 x Nb = 1280: ten tiles per row split instead of twenty).
//     dWfg^T [2R = 256][2D = 256]  =  [x(t - d) | x(t)]^T . [dF | dG]        over the layer's rows, both operands stored as bf16
// wn_bwd_gemm_tn_bf16<8, true, true> runs this product as two 128 x 256 tiles per row split -- both read all of [dF|dG] (737 MB requested for 491 MB of
// operands per layer at config 5), two workgroups per CU with ONE 24 KB chunk in flight each -- and alone on the chip reaches 2.5 TB/s where the layer
// kernels stream 5 (profiles/r06_tn_loads.txt): it is bound by its bytes in flight.  Here a 512-thread workgroup has a CU to itself and 256 registers a
// lane: the whole 256 x 256 accumulator (wave w: ka strips 32 (w & 3) of BOTH tap views x nb half w >> 2 = 8 MFMA tiles, 128 registers), every operand
// byte requested once, and THREE 32 KB chunks in flight (three register sets of four 16-byte pieces; two LDS stages; LDS-only barriers).
// Loader thread lt: role lt >> 7 -- 0: view a (tap 0, with its zero-pad row window), 1: view a1 (tap 1), 2-3: B -- four pieces per chunk, rows +0 / 8 / 16 / 24.
// LDS images, transposing reads and the epilogue are wn_bwd_gemm_tn_bf16's (both operands row-major [32 rows][256 columns + 32], ds_read_b64_tr_b16).
__global__ __launch_bounds__(512, 2) void wn_bwd_wfg_bf16(WnGemmTnArgs g) {
    constexpr int KC = 32, RS = 256 * 2 + 64, BUF = KC * RS / 2;   // bytes per image row; shorts per stage and operand
    __shared__ __attribute__((aligned(16))) unsigned short a_s[2][BUF];
    __shared__ __attribute__((aligned(16))) unsigned short b_s[2][BUF];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wr = wv & 3, wc = wv >> 2;
    // tiles_ka counts 256-column tiles of A here; the tiles of a row split share an XCD (wn_tile_of)
    unsigned split, tile;
    if (!wn_tile_of(blockIdx.x, (unsigned)(g.tiles_ka * (g.Nb / 256)), (unsigned)g.n_splits, split, tile)) return;
    const int ka0 = (int)(tile % (unsigned)g.tiles_ka) * 256, nb0 = (int)(tile / (unsigned)g.tiles_ka) * 256;
    const long long m_begin = (long long)split * g.rows_per_split;
    long long m_end = m_begin + g.rows_per_split;
    if (m_end > g.M) m_end = g.M;
    if (m_begin >= m_end) return;
    wn_f16v acc[2][4];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[h][j][i] = 0.f;
    // ---- loader
    const int role = __builtin_amdgcn_readfirstlane(tid >> 7);   // (two waves per role: told to the compiler, so that the role's row map and its buffer
    const bool is_b = role >= 2;                                  //  descriptor live in scalar registers -- a lane-dependent map is re-read from the kernel
    WnRowMap map = g.a;                                           //  arguments with vector loads in every chunk, and every wait for those drains the prefetch)
    const bool taps = g.ka_split > 0;                             // two row views of A side by side (Ka = 256: view a, view a1) / one view, 256 of its columns
    if (role == 1 && taps) map = g.a1;
    if (is_b) map = g.b;
    const int win_lo = role == 0 || !taps ? (is_b ? 0 : g.a_skip_lo) : 0;
    const int org = is_b ? nb0 : (taps ? 0 : ka0 + 128 * (role == 1));   // first column of this role's 128 / 256 columns inside its row view
    const int lt = is_b ? tid - 256 : (tid & 127);
    const int ppr = is_b ? 32 : 16;                       // 16-byte pieces per operand row this role covers
    const int row0 = lt / ppr, piece = lt % ppr;
    const int col0 = org + 8 * piece;                     // first column inside the row view
    unsigned short* const img0 = (is_b ? b_s[0] : a_s[0]);
    const int img_off = row0 * RS + (is_b ? 0 : 256 * (role == 1)) + 16 * piece;   // byte offset of this lane's first piece inside a stage (view a1: columns 128..)
    constexpr unsigned OOB = 0x80000000u;
    const unsigned q_wg = (unsigned)((unsigned long long)m_begin / (unsigned)g.rows_per_batch), rem_wg = (unsigned)m_begin - q_wg * (unsigned)g.rows_per_batch;
    const long long e_wg = (long long)q_wg * map.batch_stride + (map.t0 + (long long)rem_wg) * map.row_stride;
    const __amdgpu_buffer_rsrc_t rs = wn_rsrc(reinterpret_cast<const unsigned short*>(map.base) + e_wg);
    float4 v[12];   // three sets of four pieces
    auto fetch = [&](long long mc, int vo) {
        const long long m = mc + row0;
        unsigned q = (unsigned)m / (unsigned)g.rows_per_batch, rem = (unsigned)m - q * (unsigned)g.rows_per_batch;
        unsigned off = (unsigned)(((long long)q * map.batch_stride + (map.t0 + (long long)rem) * map.row_stride - e_wg + col0) * 2);
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            const bool ok = m + 8 * qq < m_end && (int)rem >= win_lo;
            const wn_v4i got = __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? off : OOB, 0, 0);   // 8 bf16, moved as bits; outside the operand: zeros
            v[vo + qq] = make_float4(__int_as_float(got.x), __int_as_float(got.y), __int_as_float(got.z), __int_as_float(got.w));
            rem += 8;
            if (rem >= (unsigned)g.rows_per_batch) {   // the row 8 further down is in a later batch entry
                do { rem -= (unsigned)g.rows_per_batch; ++q; } while (rem >= (unsigned)g.rows_per_batch);
                off = (unsigned)(((long long)q * map.batch_stride + (map.t0 + (long long)rem) * map.row_stride - e_wg + col0) * 2);
            } else {
                off += (unsigned)(8 * map.row_stride * 2);
            }
        }
    };
    auto stash = [&](int buf, int vo) {
        char* img = reinterpret_cast<char*>(img0 + buf * BUF) + img_off;
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) *reinterpret_cast<float4*>(img + 8 * qq * RS) = v[vo + qq];
    };
    // ---- products of the chunk in stage `buf`
    typedef __attribute__((address_space(3))) wn_s4 lds_s4;
    const int tp = lane & 15, tg = (lane >> 4) & 1, th = lane >> 5;
    const int tra_off = (8 * th + (tp >> 2)) * RS + (32 * wr + 16 * tg + 4 * (tp & 3)) * 2;
    const int trb_off = (8 * th + (tp >> 2)) * RS + (128 * wc + 16 * tg + 4 * (tp & 3)) * 2;
    auto products = [&](int buf) {
        const char* at_img = reinterpret_cast<const char*>(a_s[buf]) + tra_off;
        const char* bt_img = reinterpret_cast<const char*>(b_s[buf]) + trb_off;
#pragma unroll
        for (int ks = 0; ks < KC / 16; ++ks) {
            wn_bf16x8 a[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const char* at = at_img + ks * 16 * RS + h * 256;   // tap view h: columns 128 h ..
                const wn_s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(at));
                const wn_s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(at + 4 * RS));
                a[h] = __builtin_bit_cast(wn_bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const char* bt = bt_img + ks * 16 * RS + j * 64;
                const wn_s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(bt));
                const wn_s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(bt + 4 * RS));
                const wn_bf16x8 b = __builtin_bit_cast(wn_bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b, acc[0][j], 0, 0, 0);
                acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b, acc[1][j], 0, 0, 0);
            }
        }
    };
    // ---- three chunks in flight.  Loads past the split's last row are zeros (never out of the descriptor's window), so every path issues the same
    // sequence of memory operations and the compiler's wait counts are exact: chunk k + 1 is waited for with the loads of k + 2 and k + 3 outstanding.
    fetch(m_begin, 0);
    fetch(m_begin + KC, 4);
    fetch(m_begin + 2 * KC, 8);
    stash(0, 0);
    wn_lds_barrier();
    int buf = 0;
    for (long long mc = m_begin;;) {
        fetch(mc + 3 * KC, 0);
        products(buf);
        stash(buf ^ 1, 4);
        wn_lds_barrier();
        buf ^= 1; mc += KC;
        if (mc >= m_end) break;
        fetch(mc + 3 * KC, 4);
        products(buf);
        stash(buf ^ 1, 8);
        wn_lds_barrier();
        buf ^= 1; mc += KC;
        if (mc >= m_end) break;
        fetch(mc + 3 * KC, 8);
        products(buf);
        stash(buf ^ 1, 0);
        wn_lds_barrier();
        buf ^= 1; mc += KC;
        if (mc >= m_end) break;
    }
    // ---- the partial tile: fp32 atomics into the gradient, or (deterministic mode) this split's tile into the workspace
    const int col = lane & 31;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int ka = ka0 + 128 * h + 32 * wr + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int nb = nb0 + 128 * wc + 32 * j + col;
                if (g.part) g.part[((size_t)split * g.Ka + ka) * g.Nb + nb] = acc[h][j][i];
                else unsafeAtomicAdd(g.c + (g.c_trans ? (size_t)nb * g.ldc + ka : (size_t)ka * g.ldc + nb), acc[h][j][i]);
            }
        }
}

// dF = dz * G * (1 - T^2), dG = dz * T * G * (1 - G), written in the packed [F(32) | G(32)] column order of Wfg^T.
// dzg != NULL: the skip path's share of dz -- column block of the per-block product dskip . Wskip^T, [N*out_len][ldg] -- is
// added on the last out_len rows of every batch entry (the rows the skip conv saw).
// PACKED: th holds one dword per element, {bf16 tanh (low half), bf16 sigmoid (high half)} (the bf16 step's saved gates); sg unused.
// Four consecutive channels per thread (D is a multiple of 32: a group of four never straddles the [F(32) | G(32)] packing): 16-byte loads, 8- / 16-byte
// stores.  dz == NULL: no residual share (the LAST layer: its dz is the skip path's alone -- no zero-filled buffer is written and read back for it).
template <bool PACKED>
__global__ __launch_bounds__(256) void wn_bwd_gate(const float* dz, const float* th, const float* sg, float* dfg, long long M, int D,
                                                   const float* dzg, int ldg, int rows, int out_len) {
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= M * D) return;
    const long long m = i / D;
    const int ch = (int)(i - m * D);
    float4 d = dz ? *reinterpret_cast<const float4*>(dz + i) : make_float4(0.f, 0.f, 0.f, 0.f);
    if (dzg) {
        const unsigned n = (unsigned)m / (unsigned)rows, tt = (unsigned)m - n * (unsigned)rows;
        if ((int)tt >= rows - out_len) {
            const long long off = ((long long)n * out_len + ((int)tt - (rows - out_len))) * ldg + ch;
            if (PACKED && WN_DZG_BF16) {   // (the bf16 step stores dzg as bf16: `dzg` points at unsigned short, ldg counts bf16 elements)
                const uint2 e = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(dzg) + off);
                d.x += __uint_as_float(e.x << 16); d.y += __uint_as_float(e.x & 0xffff0000u); d.z += __uint_as_float(e.y << 16); d.w += __uint_as_float(e.y & 0xffff0000u);
            } else {
                const float4 e = *reinterpret_cast<const float4*>(dzg + off);
                d.x += e.x; d.y += e.y; d.z += e.z; d.w += e.w;
            }
        }
    }
    float4 t, s;
    if (PACKED) {
        const uint4 ts = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned*>(th) + i);
        t = make_float4(__uint_as_float(ts.x << 16), __uint_as_float(ts.y << 16), __uint_as_float(ts.z << 16), __uint_as_float(ts.w << 16));
        s = make_float4(__uint_as_float(ts.x & 0xffff0000u), __uint_as_float(ts.y & 0xffff0000u), __uint_as_float(ts.z & 0xffff0000u), __uint_as_float(ts.w & 0xffff0000u));
    } else {
        t = *reinterpret_cast<const float4*>(th + i); s = *reinterpret_cast<const float4*>(sg + i);
    }
    const float4 df = make_float4(d.x * s.x * (1.f - t.x * t.x), d.y * s.y * (1.f - t.y * t.y), d.z * s.z * (1.f - t.z * t.z), d.w * s.w * (1.f - t.w * t.w));
    const float4 dg = make_float4(d.x * t.x * s.x * (1.f - s.x), d.y * t.y * s.y * (1.f - s.y), d.z * t.z * s.z * (1.f - s.z), d.w * t.w * s.w * (1.f - s.w));
    const int nf = 64 * (ch >> 5) + (ch & 31);
    if (PACKED) {   // the bf16 step stores [dF|dG] as bf16
        unsigned short* o = reinterpret_cast<unsigned short*>(dfg) + m * 2 * D + nf;
        *reinterpret_cast<uint2*>(o) = wn_pack_bf16x4(df);
        *reinterpret_cast<uint2*>(o + 32) = wn_pack_bf16x4(dg);
    } else {
        *reinterpret_cast<float4*>(dfg + m * 2 * D + nf) = df;
        *reinterpret_cast<float4*>(dfg + m * 2 * D + nf + 32) = dg;
    }
}

// ---- fused softmax cross-entropy (wavenet_training.py:69-70): one wave per row of 256 logits, four classes per lane.
// row_loss[m] = logsumexp(x) - x[target];  dlogits[m][c] = (softmax(x)[c] - [c == target]) * scale   (scale = 1/M: mean reduction)
__global__ __launch_bounds__(256) void wn_xent_rows(const float* logits, const long long* targets, long long M, float scale, float* row_loss, float* dlogits) {
    const int lane = threadIdx.x & 63;
    const long long m = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const float4 x = reinterpret_cast<const float4*>(logits + m * 256)[lane];
    float mx = fmaxf(fmaxf(x.x, x.y), fmaxf(x.z, x.w));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    const float4 e = make_float4(expf(x.x - mx), expf(x.y - mx), expf(x.z - mx), expf(x.w - mx));
    float sum = (e.x + e.y) + (e.z + e.w);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const long long t = targets[m];
    const bool valid = t >= 0 && t < 256;
    const int tl = (int)(t >> 2), tc = (int)(t & 3);
    float xt = (valid && lane == tl) ? (tc == 0 ? x.x : tc == 1 ? x.y : tc == 2 ? x.z : x.w) : 0.f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) xt += __shfl_xor(xt, o);
    if (lane == 0) row_loss[m] = valid ? (mx + logf(sum)) - xt : __uint_as_float(0x7fc00000u);
    if (dlogits) {
        const float inv = scale / sum;
        float4 d = make_float4(e.x * inv, e.y * inv, e.z * inv, e.w * inv);
        if (valid && lane == tl) {
            if (tc == 0) d.x -= scale; else if (tc == 1) d.y -= scale; else if (tc == 2) d.z -= scale; else d.w -= scale;
        }
        reinterpret_cast<float4*>(dlogits + m * 256)[lane] = d;
    }
}
// loss = scale * sum(row_loss), fp64, fixed order (one workgroup: thread-strided partial sums, then a tree)
__global__ __launch_bounds__(1024) void wn_xent_reduce(const float* row_loss, long long M, double scale, float* loss) {
    __shared__ double part[1024];
    double s = 0.;
    for (long long i = threadIdx.x; i < M; i += 1024) s += (double)row_loss[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) *loss = (float)(part[0] * scale);
}

// out[n] += sum over rows of x[row][n]   (bias gradients); x rows addressed through a row map
// X16: x is stored as bf16 (base points at unsigned short, strides in bf16 elements)
// part != NULL (deterministic mode): block bx stores its sums at part[bx][n]; wn_tn_reduce adds the blocks in order.
template <bool X16>
__global__ void wn_bwd_colsum(WnRowMap x, long long M, int rows_per_batch, int N, float* out, float* part = nullptr) {
    const int n = blockIdx.y * blockDim.x + threadIdx.x;
    const long long m0 = (long long)blockIdx.x * 512;
    if (n >= N) return;
    float s = 0.f;
    for (long long m = m0; m < m0 + 512 && m < M; ++m) {
        if (X16) {
            const unsigned q = (unsigned)m / (unsigned)rows_per_batch, rem = (unsigned)m - q * (unsigned)rows_per_batch;
            s += __uint_as_float((unsigned)(reinterpret_cast<const unsigned short*>(x.base) + (long long)q * x.batch_stride + (x.t0 + (long long)rem) * x.row_stride)[n] << 16);
        } else {
            s += wn_row(x, m, rows_per_batch)[n];
        }
    }
    if (part) part[(size_t)blockIdx.x * N + n] = s;
    else unsafeAtomicAdd(out + n, s);
}


// out[b][c][r] = in[b][r][c]  (in: `rows` x `cols` blocks spaced in_batch_stride floats apart): rebuilds the operand
// layouts the backward GEMMs want from the forward banks after every parameter update
__global__ void wn_transpose_batched(const float* in, long long in_batch_stride, float* out, int rows, int cols) {
    __shared__ float tile[32][33];
    const float* ib = in + (long long)blockIdx.z * in_batch_stride;
    float* ob = out + (long long)blockIdx.z * rows * cols;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: 8 rows per pass
    for (int i = ty; i < 32; i += 8)
        if (r0 + i < rows && c0 + tx < cols) tile[i][tx] = ib[(long long)(r0 + i) * cols + c0 + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
        if (c0 + i < cols && r0 + tx < rows) ob[(long long)(c0 + i) * rows + r0 + tx] = tile[tx][i];
}

// ---- wn_train_pack / wn_train_unpack_grads: the parameters' own tensors (the reference's Conv1d layouts, one allocation each) <-> the flat GEMM layout
// of wn_train_layout, and back for the gradients.  Every piece is a strided transpose
//     flat[off + j * ld + cmap(i)] <-> ref[i * rs + j * cs]      i < rows, j < cols,   cmap(i) = (i / 32) * cm_blk + (i % 32) + cm_off
// (a filter conv's tap: rows = D output channels, cols = R inputs, rs = 2R, cs = 2, ld = 2D, cmap = the [F(32) | G(32)] column packing; a 1x1 conv: a
// plain transpose; a bias: one column).  Up to WN_RELAYOUT_PIECES pieces travel in the kernel arguments of one launch (the tensors' addresses are the
// caller's: nothing to keep in step on the device), 32 x 32 tiles through LDS so that both sides are accessed along their contiguous index.
#define WN_RELAYOUT_PIECES 72
struct WnRelayoutPiece { float* ref; long long off; int rows, cols, rs, cs, ld, cm_blk, cm_off, tile0; };   // tile0: first workgroup of the piece
struct WnRelayoutBatch { WnRelayoutPiece p[WN_RELAYOUT_PIECES]; int n, tiles; };
template <bool UNPACK>
__global__ __launch_bounds__(256) void wn_relayout(WnRelayoutBatch b, float* flat) {
    __shared__ float tile[32][33];
    int lo = 0, hi = b.n - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (b.p[mid].tile0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1; }
    const WnRelayoutPiece& q = b.p[lo];
    const int t = (int)blockIdx.x - q.tile0, tj = (q.cols + 31) / 32;
    const int i0 = (t / tj) * 32, j0 = (t % tj) * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    float* fl = flat + q.off;
    if (!UNPACK) {
        for (int k = ty; k < 32; k += 8)        // ref side: j along the lanes
            if (i0 + k < q.rows && j0 + tx < q.cols) tile[k][tx] = q.ref[(long long)(i0 + k) * q.rs + (long long)(j0 + tx) * q.cs];
        __syncthreads();
        const int i = i0 + tx;
        for (int k = ty; k < 32; k += 8)        // flat side: i along the lanes
            if (i < q.rows && j0 + k < q.cols) fl[(long long)(j0 + k) * q.ld + (i / 32) * q.cm_blk + (i % 32) + q.cm_off] = tile[tx][k];
    } else {
        const int i = i0 + tx;
        for (int k = ty; k < 32; k += 8)
            if (i < q.rows && j0 + k < q.cols) tile[tx][k] = fl[(long long)(j0 + k) * q.ld + (i / 32) * q.cm_blk + (i % 32) + q.cm_off];
        __syncthreads();
        for (int k = ty; k < 32; k += 8)
            if (i0 + k < q.rows && j0 + tx < q.cols) q.ref[(long long)(i0 + k) * q.rs + (long long)(j0 + tx) * q.cs] = tile[k][tx];
    }
}

// Batched priming: copies the newest `count` time steps of a layer's input x (time-major rows of R floats; `x` points at
// time 0 of stream 0, streams `x_batch_stride` floats apart) into the layer's dilation-queue rings -- slot = t mod ML, the
// reference's DilatedQueue layout (wavenet_modules.py:55-57) -- for every stream and every one of the P slice copies.
__global__ void wn_fill_ring(const float* x, long long x_batch_stride, float* ring, int R, int ML, int n_streams, int P, long long n_time,
                             int count) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int r4 = R / 4;
    const long long per_stream = (long long)count * r4;
    if (i >= per_stream * n_streams) return;
    const int s = (int)(i / per_stream);
    const long long rem = i % per_stream;
    const long long t = n_time - count + rem / r4;
    const int q = (int)(rem % r4);
    const float4 v = *reinterpret_cast<const float4*>(x + (long long)s * x_batch_stride + t * R + q * 4);
    for (int c = 0; c < P; ++c)
        *reinterpret_cast<float4*>(ring + (((long long)c * n_streams + s) * ML + (t % ML)) * R + q * 4) = v;
}

#endif  // WN_FORWARD_H
