/* TEST INFRASTRUCTURE ONLY -- included twice by wn_oracle.c (REAL = float, REAL = double).
 *
 * Plain-C restatement of the reference's fast-generation path.  Every block cites the reference
 * lines it follows (paths relative to /root/reference).  No SIMD, no threads, no tricks: this is
 * the checker, not a product.
 */

/* wavenet_modules.py:42-77  DilatedQueue: ring (R, max_length), in_pos/out_pos */
typedef struct {
    REAL *data;      /* [R][max_length] row-major like the reference tensor */
    int max_length;  /* (k-1)*d + 1           wavenet_model.py:78 */
    int in_pos, out_pos;
} FN(queue);

static void FN(queue_enqueue)(FN(queue) *q, const REAL *col, int R) { /* wavenet_modules.py:55-57 */
    for (int r = 0; r < R; ++r) q->data[(size_t)r * q->max_length + q->in_pos] = col[r];
    q->in_pos = (q->in_pos + 1) % q->max_length;
}

/* wavenet_modules.py:59-72: k taps ending at out_pos, spaced d, oldest first.  The reference builds
 * the wrapped case from two strided slices + cat; position j is (out_pos - (k-1-j)*d) mod max_length. */
static void FN(queue_dequeue)(FN(queue) *q, int k, int d, int R, REAL *taps /* [R][k] */) {
    for (int j = 0; j < k; ++j) {
        int pos = q->out_pos - (k - 1 - j) * d;
        pos %= q->max_length;
        if (pos < 0) pos += q->max_length;
        for (int r = 0; r < R; ++r) taps[(size_t)r * k + j] = q->data[(size_t)r * q->max_length + pos];
    }
    q->out_pos = (q->out_pos + 1) % q->max_length;
}

/* out[o] = b[o] + sum_{i,j} w[o][i][j] * in[i][j]   (nn.Conv1d as cross-correlation on a length-k window) */
static void FN(conv)(const float *w, const float *b, int O, int I, int k, const REAL *in /* [I][k] */, REAL *out) {
    for (int o = 0; o < O; ++o) {
        REAL acc = b ? (REAL)b[o] : (REAL)0;
        const float *wo = w + (size_t)o * I * k;
        for (int i = 0; i < I; ++i)
            for (int j = 0; j < k; ++j) acc += (REAL)wo[(size_t)i * k + j] * in[(size_t)i * k + j];
        out[o] = acc;
    }
}

/* ---- stateful form: the queues outlive a call, like model.dilated_queues between two generate_fast-shaped jobs that do
 * not reset (the facade continues a stream after a progress callback with n_given = 1).  wno_generate below is
 * new + run + free.  Also used by the host-memory ABI test double (tests/double), never by the product. */
typedef struct {
    wno_config c;
    int NL;
    FN(queue) *qs;
} FN(wno_state);

FN(wno_state) *FN(wno_state_new)(const wno_config *c) {
    const int R = c->residual_channels, k = c->kernel_size, NL = c->layers * c->blocks;
    if (k < 1 || NL < 1) return 0;
    FN(wno_state) *st = (FN(wno_state) *)calloc(1, sizeof(*st));
    st->c = *c;
    st->NL = NL;
    st->qs = (FN(queue) *)calloc((size_t)NL, sizeof(*st->qs));
    /* wavenet_model.py:70-110: d_i = 2^(i mod layers); queue max_length (k-1)*d+1; reset() zeroes (:250-251) */
    for (int i = 0; i < NL; ++i) {
        int d = 1 << (i % c->layers);
        st->qs[i].max_length = (k - 1) * d + 1;
        st->qs[i].data = (REAL *)calloc((size_t)R * st->qs[i].max_length, sizeof(REAL));
    }
    return st;
}

void FN(wno_state_free)(FN(wno_state) *st) {
    if (!st) return;
    for (int i = 0; i < st->NL; ++i) free(st->qs[i].data);
    free(st->qs);
    free(st);
}

void FN(wno_state_reset)(FN(wno_state) *st) { /* wavenet_modules.py:74-77 for every layer */
    for (int i = 0; i < st->NL; ++i) {
        memset(st->qs[i].data, 0, sizeof(REAL) * (size_t)st->c.residual_channels * st->qs[i].max_length);
        st->qs[i].in_pos = st->qs[i].out_pos = 0;
    }
}

/* DilatedQueue.data / in_pos / out_pos of one layer (wavenet_modules.py:43-57); data_out is [R][max_length] */
void FN(wno_state_queue)(const FN(wno_state) *st, int layer, float *data_out, int32_t *in_pos, int32_t *out_pos) {
    const FN(queue) *q = &st->qs[layer];
    const size_t n = (size_t)st->c.residual_channels * q->max_length;
    for (size_t i = 0; i < n; ++i) data_out[i] = (float)q->data[i];
    if (in_pos) *in_pos = q->in_pos;
    if (out_pos) *out_pos = q->out_pos;
}

int FN(wno_state_run)(FN(wno_state) *st, const wno_weights *w, const int32_t *first_samples, int64_t n_given,
                      int64_t num_samples, double temperature, const float *regularizer, const double *uniforms,
                      const int32_t *forced, int32_t *out_idx, REAL *out_logits) {
    const wno_config *c = &st->c;
    const int R = c->residual_channels, D = c->dilation_channels, S = c->skip_channels, E = c->end_channels;
    const int C = c->classes, k = c->kernel_size, NL = c->layers * c->blocks;
    if (k < 1 || n_given < 1 || NL < 1) return -1;
    FN(queue) *qs = st->qs;
    REAL *x = (REAL *)malloc(sizeof(REAL) * (size_t)(R > D ? R : D));
    REAL *xn = (REAL *)malloc(sizeof(REAL) * (size_t)R);
    REAL *taps = (REAL *)malloc(sizeof(REAL) * (size_t)R * k);
    REAL *f = (REAL *)malloc(sizeof(REAL) * (size_t)D), *g = (REAL *)malloc(sizeof(REAL) * (size_t)D);
    REAL *z = (REAL *)malloc(sizeof(REAL) * (size_t)D);
    REAL *skip = (REAL *)malloc(sizeof(REAL) * (size_t)S), *s = (REAL *)malloc(sizeof(REAL) * (size_t)S);
    REAL *e = (REAL *)malloc(sizeof(REAL) * (size_t)E), *lg = (REAL *)malloc(sizeof(REAL) * (size_t)C);
    REAL *p = (REAL *)malloc(sizeof(REAL) * (size_t)C);
    const int64_t n_eval = n_given - 1 + num_samples; /* Appendix A item 8 */
    int32_t in_idx = first_samples[0];                /* wavenet_model.py:256-257 */
    for (int64_t ev = 0; ev < n_eval; ++ev) {
        /* start_conv on a strict one-hot == column gather (wavenet_model.py:127) */
        for (int r = 0; r < R; ++r)
            x[r] = (REAL)w->start_w[(size_t)r * C + in_idx] + (w->start_b ? (REAL)w->start_b[r] : (REAL)0);
        for (int o = 0; o < S; ++o) skip[o] = 0;
        for (int i = 0; i < NL; ++i) { /* wavenet_model.py:131-165 */
            int d = 1 << (i % c->layers);
            FN(queue_enqueue)(&qs[i], x, R);         /* :179 push first ... */
            FN(queue_dequeue)(&qs[i], k, d, R, taps); /* :180 ... then pop k taps */
            FN(conv)(w->filter_w + (size_t)i * D * R * k, w->filter_b ? w->filter_b + (size_t)i * D : 0, D, R, k, taps, f);
            FN(conv)(w->gate_w + (size_t)i * D * R * k, w->gate_b ? w->gate_b + (size_t)i * D : 0, D, R, k, taps, g);
            for (int o = 0; o < D; ++o) z[o] = TANH(f[o]) * ((REAL)1 / ((REAL)1 + EXP(-g[o]))); /* :147-151 */
            FN(conv)(w->skip_w + (size_t)i * S * D, w->skip_b ? w->skip_b + (size_t)i * S : 0, S, D, 1, z, s);
            for (int o = 0; o < S; ++o) skip[o] = s[o] + skip[o]; /* :158-162 */
            FN(conv)(w->res_w + (size_t)i * R * D, w->res_b ? w->res_b + (size_t)i * R : 0, R, D, 1, z, xn);
            for (int r = 0; r < R; ++r) x[r] = xn[r] + taps[(size_t)r * k + (k - 1)]; /* :164-165 newest tap */
        }
        if (ev < n_given - 1) { /* priming: output discarded (:260-264) */
            in_idx = first_samples[ev + 1];
            continue;
        }
        for (int o = 0; o < S; ++o) skip[o] = skip[o] > 0 ? skip[o] : 0; /* :167 */
        FN(conv)(w->end1_w, w->end1_b, E, S, 1, skip, e);
        for (int o = 0; o < E; ++o) e[o] = e[o] > 0 ? e[o] : 0; /* :168 */
        FN(conv)(w->end2_w, w->end2_b, C, E, 1, e, lg); /* :169 */
        const int64_t gi = ev - (n_given - 1);
        if (out_logits) memcpy(out_logits + (size_t)gi * C, lg, sizeof(REAL) * (size_t)C);
        for (int o = 0; o < C; ++o) lg[o] -= regularizer ? (REAL)regularizer[o] : (REAL)0; /* :280 */
        int idx = 0;
        if (temperature > 0 && uniforms) {
            /* :284-288  x /= T; softmax (ATen: max, exp(x-max), sum, * 1/sum); np.random.choice(p):
             * float64 cumsum, divide by last, searchsorted(u, 'right') -- Appendix A item 10 */
            REAL mx = lg[0] / (REAL)temperature;
            for (int o = 0; o < C; ++o) { lg[o] = lg[o] / (REAL)temperature; if (lg[o] > mx) mx = lg[o]; }
            REAL sum = 0;
            for (int o = 0; o < C; ++o) { p[o] = EXP(lg[o] - mx); sum += p[o]; }
            REAL inv = (REAL)1 / sum;
            double tot = 0;
            for (int o = 0; o < C; ++o) { p[o] *= inv; tot += (double)p[o]; }
            double run = 0;
            const double u = uniforms[gi];
            idx = 0;
            for (int o = 0; o < C; ++o) { run += (double)p[o]; if (run / tot <= u) idx = o + 1; }
            if (idx >= C) idx = C - 1;
        } else { /* :290-294 greedy: first index of the maximum */
            for (int o = 1; o < C; ++o) if (lg[o] > lg[idx]) idx = o;
        }
        out_idx[gi] = idx;
        in_idx = forced ? forced[gi] : idx; /* :300-302 feedback */
    }
    free(x); free(xn); free(taps); free(f); free(g); free(z); free(skip); free(s); free(e); free(lg); free(p);
    return 0;
}

int FN(wno_generate)(const wno_config *c, const wno_weights *w, const int32_t *first_samples, int64_t n_given,
                     int64_t num_samples, double temperature, const float *regularizer, const double *uniforms,
                     const int32_t *forced, int32_t *out_idx, REAL *out_logits) {
    FN(wno_state) *st = FN(wno_state_new)(c);
    if (!st) return -1;
    const int rc = FN(wno_state_run)(st, w, first_samples, n_given, num_samples, temperature, regularizer, uniforms, forced,
                                     out_idx, out_logits);
    FN(wno_state_free)(st);
    return rc;
}
