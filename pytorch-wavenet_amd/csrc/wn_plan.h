// wn_plan.h -- host-side planner and weight packer + the POD structs shared with the kernel.
//
// The generation engine is a SYSTOLIC CHAIN of persistent workgroups (one per CU):
//
//     L0[c] -> L1[c] -> ... -> L(NL-1)[c] -> HEAD[h] -> back to L0 (sampled index feeds start_conv)
//
// Every layer of the reference's stack (wavenet_model.py:131-165) is owned by P workgroups ("layer
// split"), the two end convs (:167-169) by PA workgroups ("head split").  Each workgroup keeps its slice
// of the fp32 weight banks STATIONARY in its CU's LDS for the whole generate_fast() call, owns the
// dilation queue (wavenet_modules.py:42-77) of its layer, and exchanges only activations with its
// neighbours through 8-byte {tag,value} granules in HBM (MI355X guide: "R2, the data IS the flag").
//
// Split of one layer across its P workgroups (c = 0..P-1, Dc = ceil(D/P) gated channels each):
//   filter/gate (wavenet_model.py:147-151): ROW split  -- WG c computes channels [c*Dc, (c+1)*Dc) of
//       tanh(Wf.[x[t-d];x[t]]) * sigmoid(Wg.[...]); it needs the full layer input x[t] (R floats).
//   residual 1x1 (:164-165) and skip 1x1 (:154-162): K split -- WG c multiplies ITS Dc channels of z into
//       partial sums for all R (resp. S) outputs; the consumer adds the P partials in fixed order c=0..P-1
//       (WG 0's partial also carries the "+ x[t]" residual add and the conv bias).
//   The running skip sum travels down the chain in P independent lanes; the head adds the lanes.
// Head WG h: rows [h*Ec,(h+1)*Ec) of end_conv_1 (+bias, ReLU) then the matching K-slice of end_conv_2,
//   publishing partial logits; every L0 workgroup adds the PA partials, applies regulariser/temperature,
//   softmax + inverse-CDF sampling (or argmax) redundantly and bit-identically, and gathers the
//   start_conv column of the sampled class (wavenet_model.py:127 on a one-hot == column gather).
#ifndef WN_PLAN_H
#define WN_PLAN_H

#include <stdint.h>

#define WN_THREADS 256
#define WN_SAMPLER_GROUPS 16
#define WN_LDS_MAX_BYTES (163840 - 1024) /* 160 KiB per CU on gfx950, minus the static LDS __syncthreads_or() uses */

typedef unsigned long long wn_u64;

// One packed matrix-vector product  out[row] = sum_k W[row][k] * x[k],  rows Nr, reduction K.
// 256 threads; T threads cooperate on a row (T a power of two), each owning J float4 of that row per
// pass; a pass covers 256/T rows.  LDS image: float4 W4[(pass*J + j)*256 + tid] -- every ds_read_b128
// of a wave is lane-linear (conflict-free); thread tid holds row pass*(256/T) + tid/T and the k-range
// 4*(j*T + tid%T) .. +3.  x must provide xlen = J*T*4 floats (zero padded).
struct WnMatvec {
    int32_t Nr, K, T, J, passes;
    int32_t off4;  // offset of the image inside the workgroup's LDS blob, in float4 units
    int32_t xlen;  // floats of x consumed (zero padded tail)
    int32_t n4;    // float4 count of the image = passes*J*256
};

struct WnPlan {
    // model (wavenet_model.py:28-39)
    int32_t layers, blocks, NL, R, D, S, E, C, k, has_bias;
    int32_t n_streams;
    // chain geometry
    int32_t P, PA, Dc, Ec, n_wg;
    int32_t HR;  // replicas of the PA head workgroups (wave-specialised kernel: replica j serves the streams s = j mod HR); 1 elsewhere
    WnMatvec fg, res, skip, end1, end2;
    // LDS layout of a layer workgroup (float offsets; blob first)
    int32_t l_bias_fg, l_bias_res, l_bias_skip;  // inside the blob, valid iff has_bias
    int32_t l_xs, l_fgout, l_z, l_xt, l_skin, l_red, l_smp, l_total;
    // LDS layout of a head workgroup
    int32_t h_b1, h_b2, h_sk, h_ev, h_red, h_total;
    int32_t blob_layer_floats, blob_head_floats;  // multiples of 4
    int32_t lds_floats;                           // max(l_total, h_total)
    int32_t red_floats;
    // device tables / buffers
    const float* blobs;       // layer WG w: blobs + w*blob_layer_floats ; head h: blobs + NL*P*blob_layer + h*blob_head
    const float* start_t;     // [C][R] start_conv.weight transposed: column gather = one contiguous row
    const float* start_b;     // [R] or NULL
    const int32_t* dil;       // [NL] dilation of each layer (wavenet_model.py:70-110)
    const int64_t* ring_off;  // [NL] float offset of the layer's queue rings inside `rings`
    const int32_t* wg_map;    // [n_wg] blockIdx -> chain position
    float* rings;             // layer l, copy c, stream s: ring_off[l] + (c*n_streams+s)*ML*R ; ML=(k-1)*d+1
    wn_u64* gx;               // x' partials   [(NL*P)][n_streams][R]
    wn_u64* gs;               // skip lanes    [(NL*P)][n_streams][S]
    wn_u64* gl;               // partial logits[PA][n_streams][C]
    wn_u64* gi;               // sampled class index per stream [n_streams] (multi-stream kernel: samplers -> L0)
    wn_u64* g0;               // variant 3: layer 0's input, the start_conv row of the sampled class [n_streams][R] (samplers -> L0)
    int32_t n_smp;            // dedicated sampler workgroups (0 in the single-stream kernels)
    int32_t start_in_lds;     // v2 single-stream: layer 0 holds start_conv^T in LDS
    uint32_t* status;         // [8] 0: abort code, 1: chain position, 2: eval, 3: stream, 4: where
    uint32_t* xcc_tab;        // [n_wg] XCC id + 1 of every chain position, written by the workgroups at start
    int32_t n_blocks;         // grid size (>= n_wg; blocks mapped to -1 exit at once)
    int32_t allow_plain;      // same-XCD producers may publish with L2-resident (non write-through) stores
    // chain positions: [0, n_lw) layer-role workgroups, then PA * HR head workgroups, then the samplers.  Variant 3: n_lw = NL * P
    // (one layer slice each); variant 4: n_lw stack workgroups of LPW consecutive layers each (wn_kernel_v4.h)
    int32_t n_lw, LPW;
    int64_t head_blob_off;    // float offset of head workgroup 0's weight image inside `blobs` (variants 3 and 4)
};

struct WnRun {
    const int32_t* first;  // [n_streams][n_given]
    int64_t n_given, num_samples, n_eval, t_base;
    float temperature;
    int32_t greedy;
    const float* reg;
    const double* uniforms;
    int32_t* out_idx;
    float* dbg_logits;
    int64_t timeout_ticks;  // wall_clock64 ticks
    long long* prof;        // optional [n_wg][prof_items][4] wall-clock stamps (diagnostics), or NULL
    int32_t prof_items;
    int32_t pad;
    const float* stream_temps;  // optional [n_streams]: per-stream temperature (<= 0: that stream is greedy); NULL = `temperature` for all
    int64_t resident_ticks;     // bound of the start-up residency barrier (wn_resident_barrier), wall_clock64 ticks
};

// ---- host-side planner / packer (plain C++; also parsed, unused, in the device pass) ----
#include <functional>
#include <string>
#include <vector>

static inline int wn_pow2floor(int v) { int p = 1; while (p * 2 <= v) p *= 2; return p; }
static inline int wn_pow2ceil(int v) { int p = 1; while (p < v) p *= 2; return p; }
static inline int wn_cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int wn_round4(int v) { return (v + 3) & ~3; }

static inline WnMatvec wn_make_matvec(int Nr, int K, int off4) {
    WnMatvec m;
    m.Nr = Nr; m.K = K;
    const int K4 = wn_cdiv(K, 4);
    int T = Nr >= WN_THREADS ? 1 : wn_pow2floor(WN_THREADS / Nr);
    const int tmax = wn_pow2ceil(K4);
    if (T > tmax) T = tmax;
    if (T > 64) T = 64;
    m.T = T;
    m.J = wn_cdiv(K4, T);
    m.passes = wn_cdiv(Nr, WN_THREADS / T);
    m.off4 = off4;
    m.xlen = m.J * T * 4;
    m.n4 = m.passes * m.J * WN_THREADS;
    return m;
}

// dst: float image of m (m.n4*4 floats); W(row,k) -> weight or 0 outside the matrix
static inline void wn_pack_matvec(float* dst, const WnMatvec& m, const std::function<float(int, int)>& W) {
    const int rpp = WN_THREADS / m.T;
    for (int p = 0; p < m.passes; ++p)
        for (int j = 0; j < m.J; ++j)
            for (int tid = 0; tid < WN_THREADS; ++tid) {
                const int row = p * rpp + tid / m.T, q = tid % m.T;
                for (int i = 0; i < 4; ++i) {
                    const int kk = (j * m.T + q) * 4 + i;
                    dst[((size_t)(p * m.J + j) * WN_THREADS + tid) * 4 + i] = (row < m.Nr && kk < m.K) ? W(row, kk) : 0.f;
                }
            }
}

// Fills the geometry part of the plan for a given split; returns LDS bytes needed.
static inline int64_t wn_plan_geometry(WnPlan& pl, int P, int PA) {
    pl.P = P; pl.PA = PA; pl.HR = 1;
    pl.Dc = wn_cdiv(pl.D, P);
    pl.Ec = wn_cdiv(pl.E, PA);
    pl.n_wg = pl.NL * P + PA;
    int off4 = 0;
    pl.fg = wn_make_matvec(2 * pl.Dc, pl.k * pl.R, off4); off4 += pl.fg.n4;
    pl.res = wn_make_matvec(pl.R, pl.Dc, off4); off4 += pl.res.n4;
    pl.skip = wn_make_matvec(pl.S, pl.Dc, off4); off4 += pl.skip.n4;
    int o = off4 * 4;
    pl.l_bias_fg = pl.l_bias_res = pl.l_bias_skip = -1;
    if (pl.has_bias) {
        pl.l_bias_fg = o; o += wn_round4(2 * pl.Dc);
        pl.l_bias_res = o; o += wn_round4(pl.R);
        pl.l_bias_skip = o; o += wn_round4(pl.S);
    }
    pl.blob_layer_floats = o;
    int passes = pl.fg.passes;
    if (pl.res.passes > passes) passes = pl.res.passes;
    if (pl.skip.passes > passes) passes = pl.skip.passes;
    // head
    int hoff4 = 0;
    pl.end1 = wn_make_matvec(pl.Ec, pl.S, hoff4); hoff4 += pl.end1.n4;
    pl.end2 = wn_make_matvec(pl.C, pl.Ec, hoff4); hoff4 += pl.end2.n4;
    int ho = hoff4 * 4;
    pl.h_b1 = ho; ho += wn_round4(pl.Ec);
    pl.h_b2 = ho; ho += wn_round4(pl.C);
    pl.blob_head_floats = ho;
    if (pl.end1.passes > passes) passes = pl.end1.passes;
    if (pl.end2.passes > passes) passes = pl.end2.passes;
    pl.red_floats = passes * WN_THREADS;
    // layer scratch
    pl.l_xs = o; o += wn_round4(pl.fg.xlen > pl.k * pl.R ? pl.fg.xlen : pl.k * pl.R);
    pl.l_fgout = o; o += wn_round4(2 * pl.Dc);
    int zlen = pl.res.xlen > pl.skip.xlen ? pl.res.xlen : pl.skip.xlen;
    if (zlen < pl.Dc) zlen = pl.Dc;
    pl.l_z = o; o += wn_round4(zlen);
    pl.l_xt = o; o += wn_round4(pl.R);
    // skin + red double as the sampler scratch of the L0 workgroups (never live at the same time)
    // sampler scratch: lgt,xsm,pp floats (3*C4) + pd doubles (2*C4) + per-group partials (see wn_kernel.h WnSmp)
    const int smp_floats = 5 * wn_round4(pl.C) + 8 * WN_SAMPLER_GROUPS + 16;
    pl.l_skin = o;
    pl.l_smp = o;
    int region = wn_round4(pl.S) + pl.red_floats;
    if (region < smp_floats) region = smp_floats;
    pl.l_red = o + wn_round4(pl.S);
    o += region;
    pl.l_total = o;
    // head scratch
    pl.h_sk = ho; ho += wn_round4(pl.end1.xlen > pl.S ? pl.end1.xlen : pl.S);
    pl.h_ev = ho; ho += wn_round4(pl.end2.xlen > pl.Ec ? pl.end2.xlen : pl.Ec);
    pl.h_red = ho; ho += pl.red_floats;
    pl.h_total = ho;
    pl.lds_floats = pl.l_total > pl.h_total ? pl.l_total : pl.h_total;
    return (int64_t)pl.lds_floats * 4;
}

// Chooses (P, PA): the smallest splits whose LDS image fits one CU, within the CU budget.
// forced_P / forced_PA > 0 override.  Returns empty string on success, else a reason.
static inline std::string wn_plan_choose(WnPlan& pl, int n_cu, int forced_P, int forced_PA) {
    int bestP = -1, bestPA = -1;
    const int p_lo = forced_P > 0 ? forced_P : 1, p_hi = forced_P > 0 ? forced_P : pl.D;
    for (int P = p_lo; P <= p_hi; ++P) {
        if (forced_P <= 0 && P > 1 && wn_cdiv(pl.D, P) == wn_cdiv(pl.D, P - 1)) continue;  // same Dc: no gain
        WnPlan t = pl;
        // layer part only depends on P: probe with a head split that certainly fits
        wn_plan_geometry(t, P, pl.E);
        if ((int64_t)t.l_total * 4 <= WN_LDS_MAX_BYTES) { bestP = P; break; }
    }
    if (bestP < 0) return "no layer split fits the 160 KiB LDS of one CU";
    const int a_lo = forced_PA > 0 ? forced_PA : 1, a_hi = forced_PA > 0 ? forced_PA : pl.E;
    for (int PA = a_lo; PA <= a_hi; ++PA) {
        if (forced_PA <= 0 && PA > 1 && wn_cdiv(pl.E, PA) == wn_cdiv(pl.E, PA - 1)) continue;
        WnPlan t = pl;
        wn_plan_geometry(t, bestP, PA);
        if ((int64_t)t.h_total * 4 <= WN_LDS_MAX_BYTES) { bestPA = PA; break; }
    }
    if (bestPA < 0) return "no head split fits the 160 KiB LDS of one CU";
    const int64_t lds = wn_plan_geometry(pl, bestP, bestPA);
    if (lds > WN_LDS_MAX_BYTES) return "LDS image does not fit";
    if (n_cu > 0) {
        const int per_cu = (int)(WN_LDS_MAX_BYTES / lds) > 0 ? (int)(WN_LDS_MAX_BYTES / lds) : 1;
        const int cap = n_cu * (per_cu > 4 ? 4 : per_cu);
        if (pl.n_wg > cap)
            return "the chain needs " + std::to_string(pl.n_wg) + " co-resident workgroups but the device admits only " +
                   std::to_string(cap);
    }
    return std::string();
}

struct WnHostWeights {  // host pointers, reference layout (see include/wn_abi.h wn_weight_ptrs)
    const float *start_w, *start_b, *filter_w, *filter_b, *gate_w, *gate_b, *res_w, *res_b, *skip_w, *skip_b, *end1_w,
        *end1_b, *end2_w, *end2_b;
};

// Builds all per-workgroup LDS images: [NL*P layer blobs][PA head blobs].
static inline void wn_pack_blobs(const WnPlan& pl, const WnHostWeights& w, std::vector<float>& out) {
    const int R = pl.R, D = pl.D, S = pl.S, E = pl.E, C = pl.C, k = pl.k, Dc = pl.Dc, Ec = pl.Ec;
    out.assign((size_t)pl.NL * pl.P * pl.blob_layer_floats + (size_t)pl.PA * pl.blob_head_floats, 0.f);
    for (int l = 0; l < pl.NL; ++l)
        for (int c = 0; c < pl.P; ++c) {
            float* b = out.data() + ((size_t)l * pl.P + c) * pl.blob_layer_floats;
            const float* fw = w.filter_w + (size_t)l * D * R * k;
            const float* gw = w.gate_w + (size_t)l * D * R * k;
            const float* rw = w.res_w + (size_t)l * R * D;
            const float* sw = w.skip_w + (size_t)l * S * D;
            // rows 0..Dc-1: filter channel c*Dc+row ; rows Dc..2Dc-1: gate.  k index kk = tap*R + ch:
            // tap 0 multiplies x[t-(k-1)d] ... tap k-1 multiplies x[t]  (Appendix A item 1)
            wn_pack_matvec(b + (size_t)pl.fg.off4 * 4, pl.fg, [&](int row, int kk) -> float {
                const bool gate = row >= Dc;
                const int ch_out = c * Dc + (gate ? row - Dc : row);
                if (ch_out >= D) return 0.f;
                const int tap = kk / R, ch = kk % R;
                return (gate ? gw : fw)[((size_t)ch_out * R + ch) * k + tap];
            });
            wn_pack_matvec(b + (size_t)pl.res.off4 * 4, pl.res, [&](int row, int kk) -> float {
                const int ch = c * Dc + kk;
                return ch < D ? rw[(size_t)row * D + ch] : 0.f;
            });
            wn_pack_matvec(b + (size_t)pl.skip.off4 * 4, pl.skip, [&](int row, int kk) -> float {
                const int ch = c * Dc + kk;
                return ch < D ? sw[(size_t)row * D + ch] : 0.f;
            });
            if (pl.has_bias) {
                for (int i = 0; i < Dc; ++i) {
                    const int ch = c * Dc + i;
                    b[pl.l_bias_fg + i] = (ch < D && w.filter_b) ? w.filter_b[(size_t)l * D + ch] : 0.f;
                    b[pl.l_bias_fg + Dc + i] = (ch < D && w.gate_b) ? w.gate_b[(size_t)l * D + ch] : 0.f;
                }
                if (c == 0) {  // biases of the K-split convs ride on lane 0 only
                    for (int i = 0; i < R; ++i) b[pl.l_bias_res + i] = w.res_b ? w.res_b[(size_t)l * R + i] : 0.f;
                    for (int i = 0; i < S; ++i) b[pl.l_bias_skip + i] = w.skip_b ? w.skip_b[(size_t)l * S + i] : 0.f;
                }
            }
        }
    for (int h = 0; h < pl.PA; ++h) {
        float* b = out.data() + (size_t)pl.NL * pl.P * pl.blob_layer_floats + (size_t)h * pl.blob_head_floats;
        wn_pack_matvec(b + (size_t)pl.end1.off4 * 4, pl.end1, [&](int row, int kk) -> float {
            const int e = h * Ec + row;
            return e < E ? w.end1_w[(size_t)e * S + kk] : 0.f;
        });
        wn_pack_matvec(b + (size_t)pl.end2.off4 * 4, pl.end2, [&](int row, int kk) -> float {
            const int e = h * Ec + kk;
            return e < E ? w.end2_w[(size_t)row * E + e] : 0.f;
        });
        for (int i = 0; i < Ec; ++i) b[pl.h_b1 + i] = (h * Ec + i < E) ? w.end1_b[h * Ec + i] : 0.f;
        if (h == 0)
            for (int i = 0; i < C; ++i) b[pl.h_b2 + i] = w.end2_b[i];
    }
}

// ---- forms of the wave-specialised kernel (wn_kernel_v3.h), chosen per job by wn_create
// The throughput form pays once the ring holds more tokens than it has stages (one stream per item then runs into the stages' cycle):
// from n_layers + WN_V3_G2_MARGIN streams -- cfg3 (50 layers): 56 streams (895 k against 887 k samples/s; 48: 783 k against 800 k);
// cfg2 (30 layers): 36 (32 streams: 999 k against 956 k, 64: 1.53 M against 0.96 M); cfg1 (10 layers): 16.
#ifndef WN_V3_G2_MARGIN
#define WN_V3_G2_MARGIN 6
#endif
#define WN_V3_ROUND_STREAMS 128   // streams per round when a job exceeds what one chain holds (wn_handle::rounds)
// bit 0: two streams per pipeline item of a layer workgroup (needs an even stream count); bit 1: two replicas of the head
// workgroups.  `pin`: the WN_V3_MODE environment override ("0".."3"), or NULL.
static inline int wn_v3_mode_for(int n_streams, const char* pin, int n_layers = 50) {
    int mode = (n_streams >= n_layers + WN_V3_G2_MARGIN) ? 3 : 0;
    if (pin && pin[0] >= '0' && pin[0] <= '3' && !pin[1]) mode = pin[0] - '0';
    if (n_streams % 2) mode &= ~1;
    if (n_streams < 2) mode = 0;
    return mode;
}
// Skip-lane slot re-use (wn_kernel_v3.h, WN_V3_SKIP_SLOTS): a form of the two-streams-per-item kernel of shapes that have it (`has_form`: cfg3's), taken
// where the chain is throughput bound -- measured crossover on cfg3 (50 layers): 80 streams -2 %, 96 +1 %, 112 +4 %, 128 +6 %, 150 +2.5 %
// (profiles/r05_skip_lane_slots_by_stream_count.txt).  `pin`: the WN_V3_SLOTS environment override ("0" / "4"), or NULL.  Returns the slot count (0 = off).
static inline int wn_v3_slots_for(int n_streams, int mode, bool has_form, const char* pin, int n_layers = 50) {
    if (!has_form || !(mode & 1)) return 0;
    int slots = n_streams >= 2 * n_layers - 4 ? 4 : 0;
    if (pin && (pin[0] == '0' || pin[0] == '4') && !pin[1]) slots = pin[0] - '0';
    if (slots * 2 > n_streams) slots = 0;
    return slots;
}
// sizes of the rounds a job of n_streams > round_max streams runs in: as few rounds as possible, equal within one or two streams,
// even (two streams per pipeline item) wherever streams are left for it, none above round_max
static inline std::vector<int> wn_v3_round_sizes(int n_streams, int round_max) {
    std::vector<int> out;
    const int K = (n_streams + round_max - 1) / round_max;
    int left = n_streams;
    for (int i = 0; i < K; ++i) {
        int n = (left + (K - i) - 1) / (K - i);
        if (n % 2 && n < left && n < round_max) ++n;
        out.push_back(n);
        left -= n;
    }
    return out;
}

// blockIdx -> chain position such that consecutive chain positions share an XCD (block b is observed
// to land on XCD b % 8; a speed-only assumption -- correctness never depends on it).
static inline void wn_make_wg_map(int n_wg, int n_xcd, std::vector<int32_t>& map) {
    map.assign(n_wg, 0);
    std::vector<int> cnt(n_xcd, 0), pre(n_xcd + 1, 0);
    for (int b = 0; b < n_wg; ++b) cnt[b % n_xcd]++;
    for (int x = 0; x < n_xcd; ++x) pre[x + 1] = pre[x] + cnt[x];
    for (int b = 0; b < n_wg; ++b) map[b] = pre[b % n_xcd] + b / n_xcd;
}

// Layer-aligned placement: every layer's P workgroups (and the head's PA) sit on ONE XCD, so that most hand-offs
// stay inside one XCD's L2.  XCD 0 hosts the head, the samplers, the first layers (sampler -> L0 stays local) AND the last layer:
// the last layer's skip lanes -- S granules from each of its P slices, the widest hand-off of the ring -- reach the head without
// crossing an XCD boundary (round 3; the ring crosses as many boundaries as before, the crossing moved to the narrow x' hand-off
// in front of the last layer).
// map[b] = chain position of block b, or -1 (bystander).  Returns false if the layers do not fit that way.
static inline bool wn_make_wg_map_layers(int NL, int P, int PA, int n_smp, int n_xcd, int cu_per_xcd, std::vector<int32_t>& map,
                                         int* n_blocks) {
    if (PA + n_smp > cu_per_xcd) return false;
    // use as FEW XCDs as hold the chain: every XCD boundary the token crosses costs ~0.4 us more than a local hop
    for (int used = 1; used <= n_xcd; ++used) {
        std::vector<std::vector<int>> S(n_xcd);
        for (int h = 0; h < PA + n_smp; ++h) S[0].push_back(NL * P + h);  // head, then samplers
        const int tail = (used > 1 && NL >= 2 && (cu_per_xcd - (int)S[0].size()) / P >= 2) ? 1 : 0;  // the last layer next to the head
        const int body = NL - tail;
        int next = 0;
        for (int x = 0; x < used; ++x) {
            const int cap = (cu_per_xcd - (int)S[x].size()) / P - (x == 0 ? tail : 0);
            const int want = wn_cdiv(body - next, used - x);
            const int take = want < cap ? want : cap;
            for (int l = next; l < next + take; ++l)
                for (int c = 0; c < P; ++c) S[x].push_back(l * P + c);
            next += take;
        }
        if (next < body) continue;
        for (int l = body; l < NL; ++l)
            for (int c = 0; c < P; ++c) S[0].push_back(l * P + c);
        size_t per = 0;
        for (int x = 0; x < n_xcd; ++x) per = S[x].size() > per ? S[x].size() : per;
        *n_blocks = (int)per * n_xcd;
        map.assign(*n_blocks, -1);
        for (int b = 0; b < *n_blocks; ++b)
            if ((size_t)(b / n_xcd) < S[b % n_xcd].size()) map[b] = S[b % n_xcd][b / n_xcd];
        return true;
    }
    return false;
}

// ---- batched products (wn_forward.h): tile mapping and grids
#if defined(__HIPCC__)
#define WN_HD __host__ __device__
#else
#define WN_HD
#endif
// Workgroup -> tile, XCD-aware.  The weight-gradient products are streams over their row operands, and every tile of one row split
// re-reads the split's rows.  Workgroups are dispatched in id order, round-robin over the 8 XCDs (id % 8), each with its own L2: the
// `nshare` tiles of a sharing group are given ids 8 apart and consecutive in time -- same XCD, same moment -- so the group's rows
// come from HBM once and from that L2 afterwards.  Grids are 1-D: 8 * nshare * ceil(ngroups / 8) workgroups; returns false for the
// padding (group >= ngroups).
static inline WN_HD bool wn_tile_of(unsigned id, unsigned nshare, unsigned ngroups, unsigned& group, unsigned& member) {
    const unsigned xcd = id & 7u, slot = id >> 3;
    member = slot % nshare;
    group = (slot / nshare) * 8u + xcd;
    return group < ngroups;
}
// Grid of a weight-gradient product C[Ka][Nb] += A^T B over M rows with tiles of 128 x tile_nb: the rows are split so that about
// `want` workgroups exist (the resident ones), never below 256 rows per split (every split ends with a tile of atomics), in whole
// rounds of 8 splits where there are that many (wn_tile_of puts a split's tiles on one XCD: 25 splits would load one XCD with four
// splits and the others with three), each a multiple of 32 rows.
struct WnTnGrid { int tiles_ka, tiles_nb, splits; long long rows_per_split; unsigned blocks; };
static inline WnTnGrid wn_tn_grid(long long M, int Ka, int Nb, int tile_nb, int want) {
    WnTnGrid g;
    g.tiles_ka = (Ka + 127) / 128; g.tiles_nb = (Nb + tile_nb - 1) / tile_nb;
    const int tiles = g.tiles_ka * g.tiles_nb;
    long long splits = want / tiles > 1 ? want / tiles : 1;
    const long long most = (M + 255) / 256;
    if (splits > most) splits = most;
    if (splits >= 8) splits -= splits % 8;
    if (splits < 1) splits = 1;
    long long rps = (M + splits - 1) / splits;
    rps = (rps + 31) / 32 * 32;
    splits = (M + rps - 1) / rps;
    g.splits = (int)splits; g.rows_per_split = rps;
    g.blocks = 8u * (unsigned)tiles * (unsigned)((splits + 7) / 8);
    return g;
}

// ---- zero padding of channel shapes (wn_create): the cheapest compiled shape that HOLDS the model.  `rows`: the kernel table
// (residual channels, dilation-channel slice, skip channels, end-channel slice, slices per layer); `usable(R, D, S, E)`: the caller's
// check that the padded configuration can be planned (LDS, CU count).  Cost: workgroups on the token's path first (layers x slices +
// head slices), then the arithmetic.  Returns the row's index or -1; the padded counts in out[4] = {R, D, S, E}.
struct WnShapeRow { int R, DC, S, EC, Pm; };
template <class Usable>
static inline int wn_pad_pick(const WnShapeRow* rows, int n_rows, int R, int D, int S, int E, int n_layers, Usable usable, int out[4]) {
    long long best = -1;
    int pick = -1;
    for (int i = 0; i < n_rows; ++i) {
        const WnShapeRow& e = rows[i];
        const int D2 = e.DC * e.Pm, E2 = (E + e.EC - 1) / e.EC * e.EC;
        if (e.R < R || e.S < S || D2 < D || E2 / e.EC > 16) continue;
        if (!usable(e.R, D2, e.S, E2)) continue;
        const long long cost = ((long long)n_layers * e.Pm + E2 / e.EC) * 100000000ll + (long long)e.R * D2 + (long long)e.S * (D2 + E2);
        if (best < 0 || cost < best) { best = cost; pick = i; out[0] = e.R; out[1] = D2; out[2] = e.S; out[3] = E2; }
    }
    return pick;
}

// ---- variant 4 (wn_kernel_v4.h): streams up to which the stacked chain of n_stack workgroups beats variant 3's longer pipeline
static inline int wn_v4_stream_limit(int n_stack) { const int n = (n_stack + 2) / 2; return n < 1 ? 1 : n; }

// ---- time geometry of WaveNetModel.forward() (wn_forward / wn_train_*; comment: wn_runtime.hip above wn_forward_geometry)
struct WnFwdGeom { std::vector<long long> a, rows, zlo; };
// dil[l] = dilation of layer l.  Returns "" and fills g, or the reason why the reference has no defined result for clips of L samples.
static inline std::string wn_forward_geometry_host(const int32_t* dil, int NL, long long L, long long out_len, WnFwdGeom& g) {
    g.a.assign(NL + 1, 0); g.rows.assign(NL + 1, 0); g.zlo.assign(NL, 0);
    for (int l = 0; l < NL; ++l) {
        const long long d = dil[l], len = L - g.a[l];
        if (len < 1) return "L=" + std::to_string(L) + " leaves layer " + std::to_string(l) + " without input (the reference's conv fails there)";
        const long long pad = (d - len % d) % d, steps = (len + pad) / d;   // per-row length of the dilated layout
        if (steps < 2) return "L=" + std::to_string(L) + " leaves layer " + std::to_string(l) + " (dilation " + std::to_string(d) + ") no output position";
        if (steps == 2 && d > 1)
            return "L=" + std::to_string(L) + " gives layer " + std::to_string(l) + " (dilation " + std::to_string(d) + ") a per-row output length of 1: the reference "
                   "skips the skip path's un-dilation there (wavenet_model.py:155) and its shapes no longer match";
        g.a[l + 1] = g.a[l] - pad + d;
    }
    if (L - g.a[NL] < out_len)
        return "L=" + std::to_string(L) + " yields " + std::to_string(L - g.a[NL]) + " output positions, output_length is " + std::to_string(out_len) +
               " (the reference's view(n * l, c) fails)";
    g.rows[NL] = out_len;
    for (int l = NL - 1; l >= 0; --l) {
        const long long want = g.rows[l + 1] + dil[l], have = L - g.a[l];
        g.rows[l] = want < have ? want : have;
        const long long z = g.a[l] + dil[l] - (L - g.rows[l + 1]);
        g.zlo[l] = z > 0 ? z : 0;
    }
    return std::string();
}
#endif  // WN_PLAN_H
