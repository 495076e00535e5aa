// wn_optim.h -- the optimiser half of the training step (gfx950, device + host launch code; included by wn_runtime.hip).
//
// Reference: WavenetTrainer.train, /root/reference/wavenet_training.py:72-77 --
//     if self.clip is not None: torch.nn.utils.clip_grad_norm(self.model.parameters(), self.clip)
//     self.optimizer.step()                       # optim.Adam by default (:19-36: lr, weight_decay from the ctor)
// On 205 parameter tensors (config 5) torch runs this as ~130 multi-tensor launches per step.  Here it is a handful: the tensors'
// pointers travel in the kernel arguments, up to WN_OPT_TENSORS per launch (no device-side tables to keep in step with gradients that
// zero_grad(set_to_none=True) re-allocates every step): one pass for the sum of squares of all gradients, one pass that clips and steps.
// The arithmetic is torch.optim.Adam's single-tensor formulas, operation for operation (torch/optim/adam.py, _single_tensor_adam):
//     g      = grad * clip_coef                     clip_coef = min(1, max_norm / (total_norm + 1e-6))   (clip_grad_norm_)
//     g     += weight_decay * p                      (weight_decay != 0)
//     m      = m + (1 - beta1) * (g - m)             _foreach_lerp_(exp_avgs, grads, 1 - beta1)   [beta1 <= 0.5: g - (g - m) * beta1, ATen's other lerp branch]
//     v      = v * beta2;  v = v + ((1 - beta2) * g) * g                     _foreach_mul_, _foreach_addcmul_
//     denom  = sqrt(v) / sqrt(1 - beta2^t) + eps
//     p      = p + (-(lr / (1 - beta1^t))) * (m / denom)                                           _foreach_addcdiv_
// with every scalar formed in double (as Python does) and rounded to fp32 once, and the fp32 roundings pinned per torch kernel.
// The clipped gradient is written back (clip_grad_norm_ works in place; loggers read .grad after the step).
#ifndef WN_OPTIM_H
#define WN_OPTIM_H

#define WN_OPT_TENSORS 48          // tensors per launch: 48 x (4 pointers + 1 size) = 1920 bytes of kernel arguments
#define WN_OPT_CHUNK 4096          // elements per workgroup

struct WnOptBatch {
    float* p[WN_OPT_TENSORS];
    float* g[WN_OPT_TENSORS];
    float* m[WN_OPT_TENSORS];
    float* v[WN_OPT_TENSORS];
    int chunk0[WN_OPT_TENSORS + 1];    // first workgroup of every tensor (prefix sums of ceil(size / WN_OPT_CHUNK))
    long long size[WN_OPT_TENSORS];
    int n;
};

// which tensor a workgroup works on (wave-uniform binary search over <= 48 prefix sums)
static __device__ __forceinline__ int wn_opt_find(const WnOptBatch& b, int wg) {
    int lo = 0, hi = b.n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (b.chunk0[mid] <= wg) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// sum of squares of every gradient element -> one double (atomics of one partial per workgroup)
__global__ __launch_bounds__(256) void wn_opt_sumsq(WnOptBatch b, double* acc) {
    const int t = wn_opt_find(b, (int)blockIdx.x);
    const long long i0 = (long long)((int)blockIdx.x - b.chunk0[t]) * WN_OPT_CHUNK, n = b.size[t];
    const float* g = b.g[t];
    float s = 0.f;
    if ((reinterpret_cast<uintptr_t>(g) & 15u) == 0 && i0 + WN_OPT_CHUNK <= n) {
        const float4* g4 = reinterpret_cast<const float4*>(g + i0);
#pragma unroll
        for (int q = 0; q < WN_OPT_CHUNK / 4 / 256; ++q) {
            const float4 x = g4[q * 256 + threadIdx.x];
            s += (x.x * x.x + x.y * x.y) + (x.z * x.z + x.w * x.w);
        }
    } else {
        for (long long i = i0 + threadIdx.x; i < n && i < i0 + WN_OPT_CHUNK; i += 256) s += g[i] * g[i];
    }
    __shared__ float part[4];
    s = wn_wave_sum(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(acc, (double)((part[0] + part[1]) + (part[2] + part[3])));
}

struct WnAdamScalars {
    float neg_step, sqrt_bc2, one_minus_b1, b2, one_minus_b2, eps, weight_decay, max_norm;   // neg_step = -(lr / (1 - beta1^t)); max_norm <= 0: no clipping
    int lerp_hi;   // the lerp weight 1 - beta1 is >= 0.5 (beta1 <= 0.5): torch's lerp then evaluates b - (b - a) * (1 - w)
};

__global__ __launch_bounds__(256) void wn_opt_adam(WnOptBatch b, WnAdamScalars k, const double* sumsq, float* norm_out) {
    const int t = wn_opt_find(b, (int)blockIdx.x);
    const long long i0 = (long long)((int)blockIdx.x - b.chunk0[t]) * WN_OPT_CHUNK, n = b.size[t];
    float coef = 1.f;
    if (k.max_norm > 0.f) {
        const float total = (float)sqrt(*sumsq);
        const float c = k.max_norm / (total + 1e-6f);
        coef = !(c >= 1.f) ? c : 1.f;   // torch.clamp(c, max=1.0): a NaN norm stays NaN and poisons every gradient, as clip_grad_norm_ does
        if (norm_out && blockIdx.x == 0 && threadIdx.x == 0) *norm_out = total;
    }
    float* p = b.p[t]; float* g = b.g[t]; float* m = b.m[t]; float* v = b.v[t];
    // (each line is one of torch's foreach kernels; the intrinsics pin the roundings: every kernel's result is a rounded fp32, a multiply-add INSIDE one
    //  kernel contracts to an FMA in torch's build as it does here)
    auto one = [&](float pi, float gi, float& mi, float& vi, float& go) -> float {
        gi = __fmul_rn(gi, coef);                                        // _foreach_mul_(grads, clip_coef)
        go = gi;
        if (k.weight_decay != 0.f) gi = __fmaf_rn(k.weight_decay, pi, gi);  // _foreach_add(grads, params, alpha=weight_decay)
        // _foreach_lerp_(exp_avgs, grads, w = 1 - beta1): ATen's lerp is a + w * (b - a) for w < 0.5 and b - (b - a) * (1 - w) from 0.5 up
        mi = k.lerp_hi ? __fmaf_rn(-__fsub_rn(gi, mi), __fsub_rn(1.f, k.one_minus_b1), gi) : __fmaf_rn(k.one_minus_b1, __fsub_rn(gi, mi), mi);
        vi = __fmul_rn(vi, k.b2);                                        // _foreach_mul_(exp_avg_sqs, beta2)
        vi = __fmaf_rn(__fmul_rn(k.one_minus_b2, gi), gi, vi);           // _foreach_addcmul_(exp_avg_sqs, grads, grads, 1 - beta2)
        const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(vi), k.sqrt_bc2), k.eps);   // sqrt, div by sqrt(1 - beta2^t), add eps
        return __fmaf_rn(k.neg_step, __fdiv_rn(mi, denom), pi);          // _foreach_addcdiv_(params, exp_avgs, denom, -step_size)
    };
    const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15u) == 0 && i0 + WN_OPT_CHUNK <= n;
    if (vec) {
#pragma unroll
        for (int q = 0; q < WN_OPT_CHUNK / 4 / 256; ++q) {
            const long long i = i0 + 4 * (q * 256 + (int)threadIdx.x);
            float4 pv = *reinterpret_cast<float4*>(p + i), gv = *reinterpret_cast<float4*>(g + i), mv = *reinterpret_cast<float4*>(m + i), vv = *reinterpret_cast<float4*>(v + i);
            float4 go;
            pv.x = one(pv.x, gv.x, mv.x, vv.x, go.x); pv.y = one(pv.y, gv.y, mv.y, vv.y, go.y);
            pv.z = one(pv.z, gv.z, mv.z, vv.z, go.z); pv.w = one(pv.w, gv.w, mv.w, vv.w, go.w);
            *reinterpret_cast<float4*>(p + i) = pv; *reinterpret_cast<float4*>(m + i) = mv; *reinterpret_cast<float4*>(v + i) = vv;
            if (coef != 1.f) *reinterpret_cast<float4*>(g + i) = go;
        }
    } else {
        for (long long i = i0 + threadIdx.x; i < n && i < i0 + WN_OPT_CHUNK; i += 256) {
            float mi = m[i], vi = v[i], go;
            p[i] = one(p[i], g[i], mi, vi, go);
            m[i] = mi; v[i] = vi;
            if (coef != 1.f) g[i] = go;
        }
    }
}

#endif  // WN_OPTIM_H
