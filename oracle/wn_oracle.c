/* TEST INFRASTRUCTURE ONLY.  CPU oracle for the MI355X WaveNet fast-generation path.
 *
 * Plain-C restatement of /root/reference/wavenet_model.py:125-184,237-315 (+ wavenet_modules.py:42-77)
 * in fp32 (the reference's arithmetic type) and fp64 (to measure the fp32 noise floor).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product path never
 * does.  Pinned against the reference itself and tests/golden/ by tests/test_oracle_pinning.py.
 *
 * Build: see oracle/Makefile  (gcc -O2 -shared -fPIC; -ffp-contract=off so fp32 results do not depend
 * on whether the host compiler fuses multiply-adds).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int32_t layers, blocks, dilation_channels, residual_channels, skip_channels, end_channels, classes,
        kernel_size, bias;
} wno_config;

/* all arrays fp32 in the reference's Conv1d (out, in, k) layout, layers concatenated; *_b NULL if absent */
typedef struct {
    const float *start_w, *start_b;   /* (R, C, 1), (R) */
    const float *filter_w, *filter_b; /* (NL, D, R, k), (NL, D) */
    const float *gate_w, *gate_b;     /* (NL, D, R, k), (NL, D) */
    const float *res_w, *res_b;       /* (NL, R, D, 1), (NL, R) */
    const float *skip_w, *skip_b;     /* (NL, S, D, 1), (NL, S) */
    const float *end1_w, *end1_b;     /* (E, S, 1), (E) */
    const float *end2_w, *end2_b;     /* (C, E, 1), (C) */
} wno_weights;

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)

#define REAL float
#define FN(name) CAT(name, _f32)
#define TANH tanhf
#define EXP expf
#include "wn_oracle_impl.h"
#undef REAL
#undef FN
#undef TANH
#undef EXP

#define REAL double
#define FN(name) CAT(name, _f64)
#define TANH tanh
#define EXP exp
#include "wn_oracle_impl.h"

/* audio_data.py:156-158 with the de-quantisation of wavenet_model.py:296: o = idx/C*2-1 ; mu = C */
void wno_expand(const int32_t *idx, int64_t n, int classes, double *out) {
    for (int64_t i = 0; i < n; ++i) {
        double o = ((double)idx[i] / classes) * 2. - 1;
        double sgn = (o > 0) - (o < 0);
        out[i] = sgn * (exp(fabs(o) * log((double)classes + 1)) - 1) / classes;
    }
}
