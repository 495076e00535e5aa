// TEST INFRASTRUCTURE ONLY -- a host-memory TEST DOUBLE of include/wn_abi.h built on the C oracle (oracle/wn_oracle.c).
//
// It exists so that the HOST logic above the C ABI -- mi355_wavenet/engine.py, the generate_fast() facade (segmentation at
// callbacks, RNG consumption, batched-priming hand-over, queue write-back, pickling), stream sharding over torch.distributed --
// is testable in a container without a GPU.  It is NOT a backend of the product: the package only ever loads libwn_mi355.so
// (mi355_wavenet/_abi.py) and raises without it; tests inject this library explicitly (tests/double_lib.py).  It contains no
// kernel code: every evaluation is the oracle's (wno_state_run_f32), so it says nothing about the HIP kernels -- those are
// checked against the oracle on the GPU (tests/test_gpu_*.py).  "Device pointers" are host pointers here.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "../../include/wn_abi.h"

extern "C" {
typedef struct {
    int32_t layers, blocks, dilation_channels, residual_channels, skip_channels, end_channels, classes, kernel_size, bias;
} wno_config;
typedef struct {
    const float *start_w, *start_b, *filter_w, *filter_b, *gate_w, *gate_b, *res_w, *res_b, *skip_w, *skip_b, *end1_w, *end1_b,
        *end2_w, *end2_b;
} wno_weights;
struct wno_state_f32;
wno_state_f32* wno_state_new_f32(const wno_config*);
void wno_state_free_f32(wno_state_f32*);
void wno_state_reset_f32(wno_state_f32*);
void wno_state_queue_f32(const wno_state_f32*, int layer, float* data_out, int32_t* in_pos, int32_t* out_pos);
int wno_state_run_f32(wno_state_f32*, const wno_weights*, const int32_t* first, int64_t n_given, int64_t num_samples, double temperature,
                      const float* regularizer, const double* uniforms, const int32_t* forced, int32_t* out_idx, float* out_logits);
}

static thread_local char g_err[256] = "";
static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

struct wn_handle {
    wn_config cfg;
    wno_config oc;
    std::vector<wno_state_f32*> streams;
    std::vector<std::vector<float>> w;  // the 14 banks, copied
    wno_weights ow;
    bool have_weights;
    long long t_base;
};

extern "C" int wn_abi_version(void) { return WN_ABI_VERSION; }
extern "C" const char* wn_last_error(void) { return g_err; }

extern "C" void wn_destroy(wn_handle* h) {
    if (!h) return;
    for (wno_state_f32* s : h->streams) wno_state_free_f32(s);
    delete h;
}

extern "C" int wn_create(const wn_config* cfg, wn_handle** out) {
    g_err[0] = 0;
    if (!cfg || !out) return fail(WN_E_BADARG, "wn_create: NULL argument");
    *out = nullptr;
    if (cfg->layers < 1 || cfg->blocks < 1 || cfg->dilation_channels < 1 || cfg->residual_channels < 1 || cfg->skip_channels < 1 ||
        cfg->end_channels < 1 || cfg->classes < 2 || cfg->n_streams < 1)
        return fail(WN_E_BADARG, "wn_create: non-positive dimension in wn_config");
    if (cfg->kernel_size < 1) return fail(WN_E_BADARG, "wn_create: kernel_size must be >= 1");
    if (cfg->layers > 24) return fail(WN_E_UNSUPPORTED, "wn_create: layers > 24");
    if (cfg->layer_split < 0 || cfg->head_split < 0 || (cfg->reserved[0] & ~WN_CFG_NO_PADDING) || cfg->reserved[1] || cfg->reserved[2])
        return fail(WN_E_BADARG, "wn_create: negative split / non-zero reserved field");   // (WN_CFG_NO_PADDING is accepted and means nothing here)
    wn_handle* h = new wn_handle();
    h->cfg = *cfg;
    h->oc = wno_config{cfg->layers, cfg->blocks, cfg->dilation_channels, cfg->residual_channels, cfg->skip_channels, cfg->end_channels,
                       cfg->classes, cfg->kernel_size, cfg->bias};
    for (int s = 0; s < cfg->n_streams; ++s) h->streams.push_back(wno_state_new_f32(&h->oc));
    h->have_weights = false;
    h->t_base = 0;
    *out = h;
    return WN_OK;
}

extern "C" int wn_load_weights(wn_handle* h, const wn_weight_ptrs* w) {
    g_err[0] = 0;
    if (!h || !w) return fail(WN_E_BADARG, "wn_load_weights: NULL argument");
    if (!w->start_w || !w->filter_w || !w->gate_w || !w->res_w || !w->skip_w || !w->end1_w || !w->end1_b || !w->end2_w || !w->end2_b)
        return fail(WN_E_BADARG, "wn_load_weights: a mandatory weight pointer is NULL");
    const wn_config& c = h->cfg;
    if (c.bias && (!w->start_b || !w->filter_b || !w->gate_b || !w->res_b || !w->skip_b))
        return fail(WN_E_BADARG, "wn_load_weights: cfg.bias=1 but a stack bias pointer is NULL");
    const size_t NL = (size_t)c.layers * c.blocks, R = c.residual_channels, D = c.dilation_channels, S = c.skip_channels,
                 E = c.end_channels, C = c.classes, k = c.kernel_size;
    const float* src[14] = {w->start_w, w->start_b, w->filter_w, w->filter_b, w->gate_w, w->gate_b, w->res_w,
                            w->res_b,   w->skip_w,  w->skip_b,   w->end1_w,   w->end1_b, w->end2_w, w->end2_b};
    const size_t n[14] = {R * C, R, NL * D * R * k, NL * D, NL * D * R * k, NL * D, NL * R * D, NL * R, NL * S * D, NL * S, E * S, E, C * E, C};
    h->w.assign(14, std::vector<float>());
    const float* dst[14];
    for (int i = 0; i < 14; ++i) {
        const bool stack_bias = i == 1 || i == 3 || i == 5 || i == 7 || i == 9;
        if (src[i] && !(stack_bias && !c.bias)) {
            h->w[i].assign(src[i], src[i] + n[i]);
            dst[i] = h->w[i].data();
        } else {
            dst[i] = nullptr;
        }
    }
    h->ow = wno_weights{dst[0], dst[1], dst[2], dst[3], dst[4], dst[5], dst[6], dst[7], dst[8], dst[9], dst[10], dst[11], dst[12], dst[13]};
    h->have_weights = true;
    return WN_OK;
}

extern "C" int wn_reset(wn_handle* h, void*) {
    g_err[0] = 0;
    if (!h) return fail(WN_E_BADARG, "wn_reset: NULL handle");
    for (wno_state_f32* s : h->streams) wno_state_reset_f32(s);
    h->t_base = 0;
    return WN_OK;
}

extern "C" int wn_generate(wn_handle* h, const wn_generate_args* a) {
    g_err[0] = 0;
    if (!h || !a) return fail(WN_E_BADARG, "wn_generate: NULL argument");
    if (!h->have_weights) return fail(WN_E_STATE, "wn_generate: wn_load_weights has not been called");
    if (a->n_given < 1 || a->num_samples < 0) return fail(WN_E_BADARG, "wn_generate: n_given must be >= 1 and num_samples >= 0");
    if (!a->first_samples) return fail(WN_E_BADARG, "wn_generate: first_samples is NULL");
    if (a->num_samples > 0 && !a->out_idx) return fail(WN_E_BADARG, "wn_generate: out_idx is NULL");
    if (a->flags != 0 || a->reserved != 0) return fail(WN_E_BADARG, "wn_generate: flags/reserved must be 0");
    const int C = h->cfg.classes;
    for (size_t s = 0; s < h->streams.size(); ++s) {
        const float temp = a->stream_temperatures ? a->stream_temperatures[s] : a->temperature;
        const bool greedy = !(temp > 0.f) || a->uniforms == nullptr;
        const int rc = wno_state_run_f32(h->streams[s], &h->ow, a->first_samples + s * (size_t)a->n_given, a->n_given, a->num_samples,
                                         greedy ? 0.0 : (double)temp, a->regularizer, greedy ? nullptr : a->uniforms + s * (size_t)a->num_samples,
                                         nullptr, a->out_idx ? a->out_idx + s * (size_t)a->num_samples : nullptr,
                                         a->dbg_logits ? a->dbg_logits + s * (size_t)a->num_samples * C : nullptr);
        if (rc) return fail(WN_E_BADARG, "wn_generate: oracle refused the job");
    }
    h->t_base += a->n_given - 1 + a->num_samples;
    return WN_OK;
}

extern "C" int wn_wait(wn_handle* h) { return h ? WN_OK : fail(WN_E_BADARG, "wn_wait: NULL handle"); }

extern "C" int wn_prime(wn_handle* h, const int32_t* first_samples, int64_t n_prime, int64_t row_stride, void*) {
    g_err[0] = 0;
    if (!h || !first_samples) return fail(WN_E_BADARG, "wn_prime: NULL argument");
    if (!h->have_weights) return fail(WN_E_STATE, "wn_prime: wn_load_weights has not been called");
    if (n_prime < 0 || row_stride < n_prime) return fail(WN_E_BADARG, "wn_prime: bad n_prime / row_stride");
    if (n_prime == 0) return WN_OK;
    if (h->t_base != 0) return fail(WN_E_STATE, "wn_prime: queues must be freshly reset (queue time is %lld)", h->t_base);
    std::vector<int32_t> row((size_t)n_prime + 1, 0);  // n_prime teacher-forced evaluations = a job with n_given = n_prime + 1, 0 samples
    for (size_t s = 0; s < h->streams.size(); ++s) {
        memcpy(row.data(), first_samples + s * (size_t)row_stride, (size_t)n_prime * sizeof(int32_t));
        if (wno_state_run_f32(h->streams[s], &h->ow, row.data(), n_prime + 1, 0, 0.0, nullptr, nullptr, nullptr, nullptr, nullptr))
            return fail(WN_E_BADARG, "wn_prime: oracle refused the job");
    }
    h->t_base = n_prime;
    return WN_OK;
}

extern "C" int wn_get_info(wn_handle* h, wn_info* out) {
    if (!h || !out) return fail(WN_E_BADARG, "wn_get_info: NULL argument");
    memset(out, 0, sizeof(*out));
    out->abi_version = WN_ABI_VERSION;
    out->n_layers = h->cfg.layers * h->cfg.blocks;
    out->receptive_field = 1 + h->cfg.blocks * (h->cfg.kernel_size - 1) * ((1 << h->cfg.layers) - 1);
    out->evals_done = h->t_base;
    out->kernel_variant = 0;  // no kernel: the oracle
    out->n_chains = 1;
    out->streams_per_item = 1; out->head_replicas = 1; out->n_samplers = 0; out->dev_overrides = 0;
    out->layers_per_workgroup = 1; out->gate_shared = -1; out->gate_waited_ms = 0; out->gate_need_per_xcd = 0;
    out->forward_native = 0; out->workgroups_per_cu = 0; out->resident_timeout_ms = 0; out->skip_lane_slots = 0;
    return WN_OK;
}

extern "C" int wn_export_queue(wn_handle* h, int32_t layer, int32_t stream, float* host_data, int32_t* in_pos, int32_t* out_pos) {
    g_err[0] = 0;
    if (!h || !host_data) return fail(WN_E_BADARG, "wn_export_queue: NULL argument");
    if (layer < 0 || layer >= h->cfg.layers * h->cfg.blocks || stream < 0 || stream >= h->cfg.n_streams)
        return fail(WN_E_BADARG, "wn_export_queue: index out of range");
    wno_state_queue_f32(h->streams[stream], layer, host_data, in_pos, out_pos);
    return WN_OK;
}

// everything that only exists as matrix-core kernels: the double has no stand-in
extern "C" int wn_forward(wn_handle*, const int32_t*, int64_t, int64_t, int64_t, float*, void*) { return fail(WN_E_UNSUPPORTED, "test double: GPU only"); }
extern "C" int wn_set_forward_precision(wn_handle*, int32_t) { return fail(WN_E_UNSUPPORTED, "test double: GPU only"); }
extern "C" int wn_train_get_layout(wn_handle*, wn_train_layout*) { return fail(WN_E_UNSUPPORTED, "test double: GPU only"); }
extern "C" int wn_train_export_params(wn_handle*, float*, void*) { return fail(WN_E_UNSUPPORTED, "test double: GPU only"); }
extern "C" int wn_train_forward(wn_handle*, const float*, const int32_t*, int64_t, int64_t, int64_t, float*, void*) { return fail(WN_E_UNSUPPORTED, "test double: GPU only"); }
extern "C" int wn_train_backward(wn_handle*, const float*, const float*, float*, void*) { return fail(WN_E_UNSUPPORTED, "test double: GPU only"); }
extern "C" int wn_train_loss(wn_handle*, const float*, const int64_t*, int64_t, float*, float*, void*) { return fail(WN_E_UNSUPPORTED, "test double: GPU only"); }
extern "C" int wn_train_pack(wn_handle*, const wn_train_tensors*, float*, void*) { return fail(WN_E_UNSUPPORTED, "test double: GPU only"); }
extern "C" int wn_train_unpack_grads(wn_handle*, const float*, const wn_train_tensors*, void*) { return fail(WN_E_UNSUPPORTED, "test double: GPU only"); }
extern "C" int wn_train_set_deterministic(wn_handle*, int32_t) { return fail(WN_E_UNSUPPORTED, "test double: GPU only"); }
extern "C" int wn_adam_step(const wn_adam_args*) { return fail(WN_E_UNSUPPORTED, "test double: GPU only"); }
extern "C" int wn_profile_next(wn_handle*, int32_t) { return fail(WN_E_UNSUPPORTED, "test double: GPU only"); }
extern "C" int wn_profile_read(wn_handle*, int64_t*, int64_t) { return fail(WN_E_UNSUPPORTED, "test double: GPU only"); }
