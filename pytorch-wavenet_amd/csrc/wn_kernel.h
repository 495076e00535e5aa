// wn_kernel.h -- the per-timestep work of one persistent workgroup of the generation chain.
//
// The generic kernel (any shape, kernel_size and class count; weights stationary in LDS).  Written in "phase style": a
// workgroup's step is a sequence of phases, each executed by all 256 threads and separated by an LDS-only workgroup barrier;
// no per-thread value lives across a barrier (it goes through LDS).  gfx950 only: one backend, no host build of this file.
//
// Reference lines restated here (paths relative to the reference root):
//   wavenet_model.py:127      start_conv on a one-hot          -> wn_l0_input (column gather)
//   wavenet_model.py:177-184  queue_dilate: push then pop k    -> wn_layer_item "stage" phase
//   wavenet_modules.py:55-72  DilatedQueue enqueue/dequeue     -> ring slot arithmetic in "stage"
//   wavenet_model.py:147-151  filter/gate conv, tanh*sigmoid   -> fg matvec + gating phase
//   wavenet_model.py:154-162  skip 1x1 + accumulate            -> skip matvec (K-split lanes)
//   wavenet_model.py:164-165  residual 1x1 + newest tap        -> res matvec (K-split partials)
//   wavenet_model.py:167-169  relu, end_conv_1, relu, end_conv_2 -> wn_head_item
//   wavenet_model.py:273-294  regulariser, temperature, softmax, np.random.choice / argmax -> wn_sample
#ifndef WN_KERNEL_H
#define WN_KERNEL_H

#include "wn_plan.h"

#include <hip/hip_runtime.h>
#define WN_DEV static __device__ __forceinline__
#define WN_TID_BEGIN ((int)threadIdx.x)
#define WN_TID_STEP WN_THREADS
// the phases exchange data through LDS only: wait for the LDS counter, not for every outstanding vector-memory operation
// (__syncthreads() would also drain in-flight polls and write-through stores)
#define WN_SYNC() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
typedef float4 wn_f4;

#define WN_PHASE for (int tid = WN_TID_BEGIN; tid < WN_THREADS; tid += WN_TID_STEP)
#define WN_UNROLL _Pragma("unroll")

// where-codes reported in status[4] when a hand-off wait gives up
enum { WN_W_LOGITS = 1, WN_W_X = 2, WN_W_SKIN = 3, WN_W_HEAD = 4, WN_W_RESIDENT = 5 };

struct WnCtx {
    const WnPlan* p;
    const WnRun* r;
    float* lds;
    int w;             // chain position of this workgroup
    int fail;          // this thread gave up a hand-off wait
    long long t_start; // wall clock at the start of the current hand-off wait
};

// ------------------------------------------------------------------------------------------------
// granules: one naturally aligned 8-byte {tag = eval+1 (high), fp32 value bits (low)} written by ONE
// write-through (sc1) store and polled with sc1 loads -- visible across XCDs without fences.
WN_DEV wn_u64 wn_pack_granule(uint32_t tag, float v) {
    union { float f; uint32_t u; } c;
    c.f = v;
    return ((wn_u64)tag << 32) | (wn_u64)c.u;
}

WN_DEV void wn_publish(wn_u64* g, uint32_t tag, float v) {
    __hip_atomic_store(g, wn_pack_granule(tag, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// `local` = every consumer of this granule sits on the producer's XCD (checked at run time from the XCC ids): the
// store may then stay in that XCD's L2 -- the coherence point of all its CUs -- instead of writing through to the
// fabric; consumers read it with the same L1-bypassing loads.  Decided per producer, never assumed.
WN_DEV void wn_publish_at(wn_u64* g, uint32_t tag, float v, bool local) {
    if (local) __hip_atomic_store(g, wn_pack_granule(tag, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_store(g, wn_pack_granule(tag, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

WN_DEV void wn_give_up(WnCtx& cx, int where, long long e, int s) {
    cx.fail = 1;
    uint32_t* st = cx.p->status;
    if (atomicCAS(st, 0u, 1u) == 0u) {
        st[1] = (uint32_t)cx.w; st[2] = (uint32_t)e; st[3] = (uint32_t)s; st[4] = (uint32_t)where;
        __threadfence();
    }
}

// ---- Residency barrier.  A job is a chain of persistent workgroups that wait for each other: it only makes progress once ALL of them are
// resident, and plain launches promise nothing of the kind -- CUs held by another kernel (a long torch kernel on another stream, a CU mask)
// leave part of the job in the dispatcher's queue.  Every workgroup therefore checks in first (one atomic add on status[5]) and thread 0
// waits until all n_wg of them have: only then does the workgroup enter the chain, and only from then on do the hand-off timeouts run.  The
// wait has its own bound (r.resident_ticks) and its own report (WN_W_RESIDENT: status[2] = how many had checked in): nothing of the
// job has run at that point -- queues, rings and hand-off words are untouched -- so the host can say exactly that (WN_E_BUSY) instead of a
// hand-off timeout somewhere in the chain ten seconds later.
WN_DEV void wn_resident_barrier(WnCtx& cx) {
    if (threadIdx.x != 0) return;   // (the role's first workgroup barrier holds the other threads back)
    uint32_t* st = cx.p->status;
    const uint32_t want = (uint32_t)cx.p->n_wg;
    // Arrival and give-up exclude each other: a workgroup that runs out of patience POISONS the counter (top bit) with a compare-and-swap on the count
    // it last saw -- if somebody arrived in between the swap fails and it looks again --, and whoever arrives (or looks) after that finds the bit and
    // leaves without touching a queue.  So "the job never started" (WN_E_BUSY: queues unchanged, the call can be repeated) holds for EVERY workgroup,
    // also for the ones the dispatcher only placed after the verdict (round 5 let those pass with seen >= want).
    uint32_t seen = __hip_atomic_fetch_add(st + 5, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    if (seen & 0x80000000u) { cx.fail = 1; return; }
    const long long t0 = (long long)wall_clock64();
    unsigned spins = 0;
    while (seen < want) {
        __builtin_amdgcn_s_sleep(16);
        seen = __hip_atomic_load(st + 5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (seen & 0x80000000u) { cx.fail = 1; return; }
        if ((++spins & 15u) == 0u) {
            if (__hip_atomic_load(st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { cx.fail = 1; return; }   // somebody else gave up
            if ((long long)wall_clock64() - t0 > cx.r->resident_ticks) {
                uint32_t expect = seen;
                if (__hip_atomic_compare_exchange_strong(st + 5, &expect, seen | 0x80000000u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    wn_give_up(cx, WN_W_RESIDENT, (long long)seen, 0);
                    return;
                }
                seen = expect;   // somebody arrived (go on waiting, or everybody is here) or somebody else closed the door
                if (seen & 0x80000000u) { cx.fail = 1; return; }
            }
        }
    }
    cx.t_start = (long long)wall_clock64();
}
// ... and its verdict for the whole workgroup (through one word of the workgroup's LDS, before any role touches it): true = leave
WN_DEV bool wn_not_resident(WnCtx& cx, float* lds) {
    wn_resident_barrier(cx);
    int* word = reinterpret_cast<int*>(lds);
    if (threadIdx.x == 0) *word = cx.fail;
    __syncthreads();
    const int failed = *word;
    __syncthreads();
    cx.fail = 0;   // (known to the compiler again: a workgroup that goes on has not failed)
    cx.t_start = (long long)wall_clock64();
    return failed != 0;
}

// Spin until the granule carries `tag`.  Bounded: gives up after r->timeout_ticks of wall clock or as
// soon as any workgroup has raised the abort word, so the kernel always terminates.
WN_DEV float wn_wait_granule(WnCtx& cx, const wn_u64* g, uint32_t tag, int where, long long e, int s) {
    if (cx.fail) return 0.f;
    unsigned spins = 0;
    for (;;) {
        const wn_u64 v = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((uint32_t)(v >> 32) == tag) return __uint_as_float((uint32_t)v);
        if ((++spins & 63u) == 0u) {
            const uint32_t ab = __hip_atomic_load(cx.p->status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (ab != 0u) { cx.fail = 1; return 0.f; }
            if ((long long)wall_clock64() - cx.t_start > cx.r->timeout_ticks) { wn_give_up(cx, where, e, s); return 0.f; }
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

// unchecked read of a granule (the caller looks at the tag)
WN_DEV wn_u64 wn_peek(const wn_u64* g) {
    return __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
WN_DEV float wn_granule_value(wn_u64 v) {
    union { float f; uint32_t u; } c;
    c.u = (uint32_t)v;
    return c.f;
}

// Sum of the n (<= WN_MAX_FANIN) granules base[j*stride] in the fixed order j = 0..n-1.  All loads are issued together
// and only the ones whose tag is not there yet are waited for one by one: one round trip instead of n.
#define WN_MAX_FANIN 16
WN_DEV float wn_gather_sum(WnCtx& cx, const wn_u64* base, size_t stride, int n, uint32_t tag, int where, long long e, int s) {
    wn_u64 v[WN_MAX_FANIN];
WN_UNROLL
    for (int j = 0; j < WN_MAX_FANIN; ++j) v[j] = j < n ? wn_peek(base + (size_t)j * stride) : 0;
    float sum = 0.f;
WN_UNROLL
    for (int j = 0; j < WN_MAX_FANIN; ++j) {
        if (j < n) sum += ((uint32_t)(v[j] >> 32) == tag) ? wn_granule_value(v[j]) : wn_wait_granule(cx, base + (size_t)j * stride, tag, where, e, s);
    }
    return sum;
}

// true if any thread of the workgroup failed; doubles as a barrier
WN_DEV bool wn_any_failed(WnCtx& cx) {
    int* flag = reinterpret_cast<int*>(cx.lds + cx.p->lds_floats);  // one LDS word behind the layout, zeroed by wn_load_lds
    if (cx.fail) *flag = 1;
    WN_SYNC();
    return *flag != 0;
}

// ------------------------------------------------------------------------------------------------
// Packed matvec from the LDS-resident image (layout: wn_plan.h WnMatvec).  emit(row, sum) is called once
// per output row by one thread.  Two phases + barriers; `red` holds the per-thread partials.
template <class Emit>
WN_DEV void wn_matvec(WnCtx& cx, const WnMatvec& m, const float* xvec, float* red, Emit emit) {
    const wn_f4* W4 = reinterpret_cast<const wn_f4*>(cx.lds) + m.off4;
    const wn_f4* x4 = reinterpret_cast<const wn_f4*>(xvec);
    WN_PHASE {
        const int q = tid & (m.T - 1);
        for (int p = 0; p < m.passes; ++p) {
            const wn_f4* wp = W4 + (size_t)p * m.J * WN_THREADS + tid;
            float acc = 0.f;
            for (int j = 0; j < m.J; ++j) {
                const wn_f4 w = wp[(size_t)j * WN_THREADS];
                const wn_f4 x = x4[j * m.T + q];
                acc += w.x * x.x;
                acc += w.y * x.y;
                acc += w.z * x.z;
                acc += w.w * x.w;
            }
            red[p * WN_THREADS + tid] = acc;
        }
    }
    WN_SYNC();
    WN_PHASE {
        const int rpp = WN_THREADS / m.T;
        for (int row = tid; row < m.Nr; row += WN_THREADS) {
            const float* rr = red + (row / rpp) * WN_THREADS + (row % rpp) * m.T;
            float s = 0.f;
            for (int q = 0; q < m.T; ++q) s += rr[q];
            emit(row, s);
        }
    }
    WN_SYNC();
}

// ------------------------------------------------------------------------------------------------
// sampler scratch carved from the L0 workgroup's [skin|red] LDS region (wn_plan.h l_smp)
struct WnSmp {
    float *lgt, *xsm, *pp;
    double *pd, *pds, *pre, *dscal;
    float *pm, *ps, *fscal;
    int *pa, *pc, *iscal;
};

WN_DEV WnSmp wn_smp_carve(const WnPlan& p, float* lds) {
    WnSmp s;
    const int C4 = (p.C + 3) & ~3, G = WN_SAMPLER_GROUPS;
    float* b = lds + p.l_smp;
    s.lgt = b; s.xsm = b + C4; s.pp = b + 2 * C4;
    s.pd = reinterpret_cast<double*>(b + 3 * C4);
    s.pds = reinterpret_cast<double*>(b + 5 * C4);
    s.pre = s.pds + G;
    s.dscal = s.pre + G;
    float* f = reinterpret_cast<float*>(s.dscal + 2);
    s.pm = f; s.ps = f + G; s.fscal = f + 2 * G;
    int* i = reinterpret_cast<int*>(f + 2 * G + 4);
    s.pa = i; s.pc = i + G; s.iscal = i + 2 * G;
    return s;
}

// Regulariser, temperature, softmax and the draw, exactly in the reference's order
// (wavenet_model.py:280-294; np.random.choice semantics: Appendix A item 10 of SURVEY.md):
//   x -= reg;  T>0: x /= T; p = exp(x-max) * (1/sum) in fp32; cdf = cumsum(double(p)); cdf /= cdf[-1];
//   idx = #{cdf <= u} (searchsorted side='right');   greedy: first index of the maximum.
// Result lands in smp.iscal[0].  lgt[] must hold the summed logits.
WN_DEV void wn_sample(WnCtx& cx, const WnSmp& sm, double u, bool greedy, float temperature) {
    const WnPlan& p = *cx.p;
    const WnRun& r = *cx.r;
    const int C = p.C, G = WN_SAMPLER_GROUPS, chunk = (C + G - 1) / G;
    WN_PHASE {
        for (int i = tid; i < C; i += WN_THREADS) {
            float v = sm.lgt[i];
            if (r.reg) v -= r.reg[i];
            if (!greedy) v = v / temperature;
            sm.xsm[i] = v;
        }
    }
    WN_SYNC();
    WN_PHASE {
        if (tid < G) {
            const int lo = tid * chunk, hi = lo + chunk < C ? lo + chunk : C;
            float m = -INFINITY;
            int am = lo < C ? lo : C - 1;
            for (int i = lo; i < hi; ++i)
                if (sm.xsm[i] > m) { m = sm.xsm[i]; am = i; }
            sm.pm[tid] = m;
            sm.pa[tid] = am;
        }
    }
    WN_SYNC();
    WN_PHASE {
        if (tid == 0) {
            float m = sm.pm[0];
            int am = sm.pa[0];
            for (int g = 1; g < G; ++g)
                if (sm.pm[g] > m) { m = sm.pm[g]; am = sm.pa[g]; }
            sm.fscal[0] = m;
            sm.iscal[0] = am;  // greedy answer
        }
    }
    WN_SYNC();
    if (greedy) return;
    WN_PHASE {
        const float m = sm.fscal[0];
        for (int i = tid; i < C; i += WN_THREADS) sm.pp[i] = expf(sm.xsm[i] - m);
    }
    WN_SYNC();
    WN_PHASE {
        if (tid < G) {
            const int lo = tid * chunk, hi = lo + chunk < C ? lo + chunk : C;
            float s = 0.f;
            for (int i = lo; i < hi; ++i) s += sm.pp[i];
            sm.ps[tid] = s;
        }
    }
    WN_SYNC();
    WN_PHASE {
        if (tid == 0) {
            float s = 0.f;
            for (int g = 0; g < G; ++g) s += sm.ps[g];
            sm.fscal[1] = 1.0f / s;
        }
    }
    WN_SYNC();
    WN_PHASE {
        const float inv = sm.fscal[1];
        for (int i = tid; i < C; i += WN_THREADS) sm.pd[i] = (double)(sm.pp[i] * inv);
    }
    WN_SYNC();
    WN_PHASE {
        if (tid < G) {
            const int lo = tid * chunk, hi = lo + chunk < C ? lo + chunk : C;
            double s = 0.;
            for (int i = lo; i < hi; ++i) s += sm.pd[i];
            sm.pds[tid] = s;
        }
    }
    WN_SYNC();
    WN_PHASE {
        if (tid == 0) {
            double run = 0.;
            for (int g = 0; g < G; ++g) { sm.pre[g] = run; run += sm.pds[g]; }
            sm.dscal[0] = run;
        }
    }
    WN_SYNC();
    WN_PHASE {
        if (tid < G) {
            const int lo = tid * chunk, hi = lo + chunk < C ? lo + chunk : C;
            const double tot = sm.dscal[0];
            double run = sm.pre[tid];
            int cnt = 0;
            for (int i = lo; i < hi; ++i) {
                run += sm.pd[i];
                if (run / tot <= u) ++cnt;
            }
            sm.pc[tid] = cnt;
        }
    }
    WN_SYNC();
    WN_PHASE {
        if (tid == 0) {
            int idx = 0;
            for (int g = 0; g < G; ++g) idx += sm.pc[g];
            if (idx >= C) idx = C - 1;
            sm.iscal[0] = idx;
        }
    }
    WN_SYNC();
}

// ------------------------------------------------------------------------------------------------
// L0 only: obtain the class index that enters the network at evaluation e of stream s and turn it into
// the layer-0 input x (start_conv column).  e == n_eval is the extra, sample-only iteration.
// Returns false if a wait failed.  LDS out: xt[0..R).
WN_DEV bool wn_l0_input(WnCtx& cx, int c, long long e, int s) {
    const WnPlan& p = *cx.p;
    const WnRun& r = *cx.r;
    const WnSmp sm = wn_smp_carve(p, cx.lds);
    float* xt = cx.lds + p.l_xt;
    if (e == 0) {
        WN_PHASE { if (tid == 0) sm.iscal[0] = r.first[(size_t)s * r.n_given]; }
        WN_SYNC();
    } else {
        // partial logits of evaluation e-1 (tag e).  Always waited for -- during priming it is the token
        // that closes the loop (flow control: nobody overwrites a granule slot before it was consumed).
        const uint32_t tag = (uint32_t)e;
        WN_PHASE {
            for (int i = tid; i < p.C; i += WN_THREADS) {
                sm.lgt[i] = wn_gather_sum(cx, p.gl + (size_t)s * p.C + i, (size_t)p.n_streams * p.C, p.PA, tag, WN_W_LOGITS, e, s);
            }
        }
        if (wn_any_failed(cx)) return false;
        if (e < r.n_given) {  // teacher forced (wavenet_model.py:263-264)
            WN_PHASE { if (tid == 0) sm.iscal[0] = r.first[(size_t)s * r.n_given + e]; }
            WN_SYNC();
        } else {
            const long long g = e - r.n_given;  // index of the generated sample
            if (r.dbg_logits && c == 0) {
                WN_PHASE {
                    for (int i = tid; i < p.C; i += WN_THREADS)
                        r.dbg_logits[((size_t)s * r.num_samples + g) * p.C + i] = sm.lgt[i];
                }
            }
            const float temp = r.stream_temps ? r.stream_temps[s] : r.temperature;
            const bool greedy = r.greedy != 0 || !(temp > 0.f);
            const double u = greedy ? 0. : r.uniforms[(size_t)s * r.num_samples + g];
            wn_sample(cx, sm, u, greedy, temp);
            if (c == 0) {
                WN_PHASE { if (tid == 0) r.out_idx[(size_t)s * r.num_samples + g] = sm.iscal[0]; }
            }
        }
    }
    if (e == r.n_eval) { WN_SYNC(); return true; }
    WN_PHASE {
        const int idx = sm.iscal[0];
        for (int i = tid; i < p.R; i += WN_THREADS) xt[i] = p.start_t[(size_t)idx * p.R + i] + (p.start_b ? p.start_b[i] : 0.f);
    }
    WN_SYNC();
    return true;
}

// One (evaluation e, stream s) step of the workgroup that owns slice c of layer l.
WN_DEV bool wn_layer_item(WnCtx& cx, int l, int c, long long e, int s, int tmod) {  // tmod = (t_base + e) mod max_length of this layer's queue
    const WnPlan& p = *cx.p;
    const WnRun& r = *cx.r;
    float* lds = cx.lds;
    float *xs = lds + p.l_xs, *fgout = lds + p.l_fgout, *z = lds + p.l_z, *xt = lds + p.l_xt, *skin = lds + p.l_skin,
          *red = lds + p.l_red;
    const bool prime = e < r.n_given - 1;  // output discarded (wavenet_model.py:260-264): no skip/head work
    const uint32_t tag = (uint32_t)(e + 1);
    const int R = p.R, S = p.S, k = p.k, P = p.P, ns = p.n_streams;
    if (l == 0) {
        if (!wn_l0_input(cx, c, e, s)) return false;
        if (e == r.n_eval) return true;
    } else {
        WN_PHASE {
            for (int i = tid; i < R; i += WN_THREADS) {
                // x = sum of the P partials of layer l-1, fixed order
                xt[i] = wn_gather_sum(cx, p.gx + (((size_t)(l - 1) * P) * ns + s) * R + i, (size_t)ns * R, P, tag, WN_W_X, e, s);
            }
            if (!prime)
                for (int i = tid; i < S; i += WN_THREADS)
                    skin[i] = wn_wait_granule(cx, p.gs + (((size_t)(l - 1) * P + c) * ns + s) * S + i, tag, WN_W_SKIN, e, s);
        }
        if (wn_any_failed(cx)) return false;
    }
    // queue_dilate: push x[t], pop the k taps x[t-(k-1)d] .. x[t]  (wavenet_model.py:177-184)
    {
        const int d = p.dil[l];
        const int ML = (k - 1) * d + 1;  // wavenet_model.py:78
        float* ring = p.rings + p.ring_off[l] + ((size_t)c * ns + s) * (size_t)ML * R;
        WN_PHASE {
            for (int i = tid; i < R; i += WN_THREADS) {
                const float x = xt[i];
                ring[(size_t)tmod * R + i] = x;  // enqueue at in_pos = t mod ML (wavenet_modules.py:55-57)
                xs[(k - 1) * R + i] = x;
                for (int j = 1; j < k; ++j) {  // tap k-1-j is x[t - j*d]; zeros before the stream start.  j*d < ML
                    int pos = tmod - j * d;
                    if (pos < 0) pos += ML;
                    xs[(k - 1 - j) * R + i] = ring[(size_t)pos * R + i];
                }
            }
        }
        WN_SYNC();
    }
    // filter & gate convs on the k taps (wavenet_model.py:147-150)
    wn_matvec(cx, p.fg, xs, red, [&](int row, float v) { fgout[row] = v; });
    WN_PHASE {
        const float* b = p.has_bias ? lds + p.l_bias_fg : nullptr;
        for (int i = tid; i < p.Dc; i += WN_THREADS) {
            const float f = fgout[i] + (b ? b[i] : 0.f);
            const float g = fgout[p.Dc + i] + (b ? b[p.Dc + i] : 0.f);
            z[i] = tanhf(f) * (1.0f / (1.0f + expf(-g)));  // :151
        }
    }
    WN_SYNC();
    // residual 1x1 partial (+ newest tap and bias on lane 0) (wavenet_model.py:164-165); the last layer's
    // residual output is never read by the reference either
    if (l < p.NL - 1) {
        wn_u64* gx = p.gx + ((size_t)cx.w * ns + s) * R;
        const float* b = p.has_bias ? lds + p.l_bias_res : nullptr;
        wn_matvec(cx, p.res, z, red, [&](int row, float v) {
            if (b) v += b[row];
            if (c == 0) v += xt[row];
            wn_publish(gx + row, tag, v);
        });
    }
    // skip 1x1 partial added to this lane's running skip sum (wavenet_model.py:154-162)
    wn_u64* gs = p.gs + ((size_t)cx.w * ns + s) * S;
    if (!prime) {
        const float* b = p.has_bias ? lds + p.l_bias_skip : nullptr;
        wn_matvec(cx, p.skip, z, red, [&](int row, float v) {
            if (b) v += b[row];
            if (l > 0) v += skin[row];
            wn_publish(gs + row, tag, v);
        });
    } else if (l == p.NL - 1) {  // priming: only a token for the head
        WN_PHASE { for (int i = tid; i < S; i += WN_THREADS) wn_publish(gs + i, tag, 0.f); }
        WN_SYNC();
    }
    return true;
}

// One (e, s) step of head workgroup h (wavenet_model.py:167-169).
WN_DEV bool wn_head_item(WnCtx& cx, int h, long long e, int s) {
    const WnPlan& p = *cx.p;
    const WnRun& r = *cx.r;
    float* lds = cx.lds;
    float *sk = lds + p.h_sk, *ev = lds + p.h_ev, *red = lds + p.h_red;
    const bool prime = e < r.n_given - 1;
    const uint32_t tag = (uint32_t)(e + 1);
    const int S = p.S, P = p.P, ns = p.n_streams;
    WN_PHASE {
        for (int i = tid; i < S; i += WN_THREADS) {
            const float sum = wn_gather_sum(cx, p.gs + (((size_t)(p.NL - 1) * P) * ns + s) * S + i, (size_t)ns * S, P, tag, WN_W_HEAD, e, s);
            sk[i] = sum > 0.f ? sum : 0.f;  // relu(skip) :167
        }
    }
    if (wn_any_failed(cx)) return false;
    wn_u64* gl = p.gl + ((size_t)h * ns + s) * p.C;
    if (!prime) {
        const float* b1 = lds + p.h_b1;
        const float* b2 = lds + p.h_b2;
        wn_matvec(cx, p.end1, sk, red, [&](int row, float v) {
            v += b1[row];
            ev[row] = v > 0.f ? v : 0.f;  // relu(end_conv_1) :168
        });
        wn_matvec(cx, p.end2, ev, red, [&](int row, float v) { wn_publish(gl + row, tag, v + b2[row]); });  // :169
    } else {
        WN_PHASE { for (int i = tid; i < p.C; i += WN_THREADS) wn_publish(gl + i, tag, 0.f); }
        WN_SYNC();
    }
    return true;
}

// Copies the workgroup's weight image into LDS and zeroes the scratch behind it.
WN_DEV void wn_load_lds(const WnPlan& p, int w, float* lds) {
    const bool is_layer = w < p.NL * p.P;
    const float* src = is_layer ? p.blobs + (size_t)w * p.blob_layer_floats
                                : p.blobs + (size_t)p.NL * p.P * p.blob_layer_floats + (size_t)(w - p.NL * p.P) * p.blob_head_floats;
    const int n4 = (is_layer ? p.blob_layer_floats : p.blob_head_floats) / 4;
    const int tot4 = (p.lds_floats + 3) / 4 + 1;  // + the fail-flag word behind the layout
    WN_PHASE {
        const wn_f4* s4 = reinterpret_cast<const wn_f4*>(src);
        wn_f4* d4 = reinterpret_cast<wn_f4*>(lds);
        for (int i = tid; i < n4; i += WN_THREADS) d4[i] = s4[i];
        wn_f4 zero;
        zero.x = zero.y = zero.z = zero.w = 0.f;
        for (int i = n4 + tid; i < tot4; i += WN_THREADS) d4[i] = zero;
    }
    WN_SYNC();
}

#endif  // WN_KERNEL_H
