// wn_train.inl -- host side of the native training step (included by wn_runtime.hip; GPU build only).
//
// Reference: WavenetTrainer.train (wavenet_training.py:58-107) runs output = model(x); loss = F.cross_entropy(...);
// loss.backward(); optimizer.step().  wn_train_forward / wn_train_backward replace model(x) and the autograd walk through
// WaveNetModel.wavenet (wavenet_model.py:125-171) with fp32 matrix-core GEMMs over all time steps at once:
//
//   forward  (per layer l, rows = the time steps the loss still depends on):
//       [F|G] = [x_l(t-d) | x_l(t)] . Wfg^T + b        T = tanh(F), G = sigmoid(G), z = T*G            (saved: x_l, z, T, G)
//       x_{l+1}(t) = z . Wres^T + bres + x_l(t)          skip += z[last output_length rows] . Wskip^T + bskip
//       logits = relu(relu(skip) . W1^T + b1) . W2^T + b2                                               (saved: skip, e)
//   backward (given dlogits):
//       de = (dlogits . W2) * [e > 0]      dskip = (de . W1) * [skip > 0]       dW2^T = e^T . dlogits   dW1^T = relu(skip)^T . de
//       per layer, last to first, with dx' = dLoss/dx_{l+1}:
//           dz   = dx' . Wres  (+ dskip . Wskip on the skip rows)          dWres^T = z^T . dx'      dWskip^T = z_skiprows^T . dskip
//           dF   = dz * G * (1 - T^2),  dG = dz * T * G * (1 - G)           dWfg^T  = [x_l(t-d) | x_l(t)]^T . [dF|dG]
//           dx_l(t) = dx'(t) + [dF|dG](t) . Wfg(tap 1) + [dF|dG](t+d) . Wfg(tap 0)
//       dstart^T = onehot(indices)^T . dx_0
//   "NN" products reuse wn_fwd_gemm with re-laid-out banks (rebuilt from `params` by wn_transpose_batched on every forward);
//   "TN" products (weight gradients) use wn_bwd_gemm_tn; bias gradients are column sums.


static int wn_train_layout_ws(const wn_handle* h, long long N, long long L, long long out_len, WnTrainLay& t) {
    const WnPlan& pl = h->plan;
    const int R = pl.R, D = pl.D, S = pl.S, E = pl.E, C = pl.C, NL = pl.NL;
    t.N = N; t.L = L; t.out_len = out_len;
    {   // need[l]: trailing positions of layer l's input the loss depends on AND that exist; zlo[l]: rows whose tap is a pad zero (wn_forward_geometry)
        WnFwdGeom geo;
        const int rc = wn_forward_geometry(h, L, out_len, geo, "wn_train_forward");
        if (rc) return rc;
        t.need = geo.rows; t.zlo = geo.zlo;
    }
    t.G = pl.layers < NL ? pl.layers : NL;
    { const char* gb = wn_dev_env("WN_TRAIN_SKIP_BLOCK"); if (gb && atoi(gb) > 0) t.G = atoi(gb) < NL ? atoi(gb) : NL; }   // (A/B runs, with WN_TESTING=1: layers per grouped skip product)
    size_t o = 0;
    auto take = [&](size_t n) { size_t r = o; o += (n + 63) & ~(size_t)63; return r; };
    t.x.resize(NL); t.th.resize(NL); t.sg.resize(NL);
    size_t zmax = 0;
    for (int l = 0; l < NL; ++l) {
        const size_t zl = (size_t)N * t.need[l + 1] * D;
        zmax = zl > zmax ? zl : zmax;
        t.x[l] = take((size_t)N * L * R);
        t.th[l] = take(zl); t.sg[l] = take(zl);
    }
    // z of a skip block's layers lives SIDE BY SIDE in one matrix Z_b: row (n, t) = [z_first(n, t) | z_first+1(n, t) | ...] (cnt * D elements; rows
    // aligned at the clips' ends, as many per clip as the block's first layer has).  z_l is a strided VIEW of it (row stride cnt * D, column block
    // l - first, the layer's own trailing rows) -- every consumer takes row maps --, and the last output_length rows of every clip, all columns, ARE the
    // A operand of the block's grouped skip product and of its weight gradient: rounds 2-4 wrote that operand as a second copy of z on the skip rows
    // (`zg`: 4.45 GB per bf16 config-5 step written by the gate epilogues, and as much workspace).  The rows a deeper layer of the block does not have
    // (at most the block's dilations: ~5 %) are never touched.
    t.nblk = (NL + t.G - 1) / t.G;
    t.zb.resize(t.nblk); t.zb_rows.resize(t.nblk);
    for (int b = 0; b < t.nblk; ++b) {
        const int first = b * t.G, cnt = NL - first < t.G ? NL - first : t.G;
        t.zb_rows[b] = t.need[first + 1];
        t.zb[b] = take((size_t)N * t.zb_rows[b] * cnt * D);
    }
    // The bf16 step keeps a bf16 SHADOW of the residual stream next to the fp32 one.  x_l is a matrix operand four times per step (the
    // two tap views of the filter/gate product and of its weight gradient) and an addend once (the residual); the operand reads convert
    // it to bf16 on their way to LDS anyway.  Written once by the product that produces x_l (WnGemmArgs::c_h), the shadow gives those
    // four reads the same bits at half the bytes: +0.13 GB written, -0.5 GB read per layer at config 5.  The fp32 stream stays what the
    // residual adds run on.
    t.xh.clear();
    if (h->fw_bf16 && h->fwb_ok && R % 128 == 0 && (2 * D) % 256 == 0) {   // (the shapes whose filter/gate products take the 256-column tiles: the forms compiled for a bf16-stored x)
        t.xh.resize(NL);
        for (int l = 0; l < NL; ++l) t.xh[l] = take(((size_t)N * L * R + 1) / 2);
    }
    const size_t Mo = (size_t)N * out_len;
    t.skip = take(Mo * S); t.ev = take(Mo * E); t.dzg = take((size_t)t.nblk * Mo * t.G * D); t.bskip_total = take(S);
    t.res_o = take((size_t)NL * R * D); t.skip_o = take((size_t)NL * S * D); t.w1_o = take((size_t)E * S); t.w2_o = take((size_t)C * E);
    t.fgb0 = take((size_t)NL * 2 * D * R); t.fgb1 = take((size_t)NL * 2 * D * R);
    t.dskip = take(Mo * S); t.de = take(Mo * E); t.dz = take(zmax); t.dfg = take(2 * zmax); t.dfg2 = take(2 * zmax);
    t.dskip_h = (h->fw_bf16 && h->fwb_ok) ? take((Mo * S + 1) / 2) : 0;   // (written by the dskip product next to the fp32 matrix: WnGemmArgs::c_h)
    t.dxa = take((size_t)N * L * R); t.dxb = take((size_t)N * L * R);
    t.colsum_tmp = take(S);
    t.idx = take((size_t)N * L);
    // bf16 operand banks (2 bytes each; sizes in floats): the whole parameter blob as it is (backward products) and the
    // transposed forward banks
    t.bw = take((h->fw_floats + 1) / 2);
    t.bt_fg = take((size_t)NL * 2 * D * 2 * R / 2); t.bt_res = take((size_t)NL * R * D / 2); t.bt_skip = take((size_t)NL * S * D / 2);
    t.bt_w1 = take((size_t)E * S / 2); t.bt_w2 = take((size_t)C * E / 2);
    t.total = o;
    return WN_OK;
}

// Row maps into Z_b (wn_train_layout_ws).  h16: z is stored as bf16 (the maps count bf16 elements then; the base is a bf16 address behind a float pointer).
static WnRowMap wn_z_map(const WnTrainLay& t, float* ws, int NL, int D, int l, bool h16) {       // z_l on its own rows (index 0 = the first of its need[l + 1])
    const int b = l / t.G, first = b * t.G, cnt = NL - first < t.G ? NL - first : t.G, gi = l - first;
    const long long ld = (long long)cnt * D, rows_b = t.zb_rows[b];
    float* base = h16 ? reinterpret_cast<float*>(reinterpret_cast<unsigned short*>(ws + t.zb[b]) + (size_t)gi * D) : ws + t.zb[b] + (size_t)gi * D;
    return WnRowMap{base, rows_b * ld, ld, rows_b - t.need[l + 1]};
}
static WnRowMap wn_zg_map(const WnTrainLay& t, float* ws, int NL, int D, int b) {                 // the block's z on the last output_length rows of every clip, all columns
    const int first = b * t.G, cnt = NL - first < t.G ? NL - first : t.G;
    const long long ld = (long long)cnt * D, rows_b = t.zb_rows[b];
    return WnRowMap{ws + t.zb[b], rows_b * ld, ld, rows_b - t.out_len};
}

extern "C" int wn_train_get_layout(wn_handle* h, wn_train_layout* out) {
    g_err[0] = 0;
    if (!h || !out) return wn_fail(WN_E_BADARG, "wn_train_get_layout: NULL argument");
    if (!h->chains.empty()) return wn_train_get_layout(h->chains[0], out);
    if (!h->have_weights) return wn_fail(WN_E_STATE, "wn_train_get_layout: wn_load_weights has not been called");
    if (!h->fw_ok) return wn_fail(WN_E_UNSUPPORTED, "wn_train: needs kernel_size 2 and channel counts that are multiples of 32");
    if (h->padded) return wn_fail(WN_E_UNSUPPORTED, "wn_train: this handle runs a zero-padded channel shape (its parameter layout is not the caller's)");
    out->total = (int64_t)h->fw_floats;
    out->fg = h->fw_off_fg; out->bfg = h->fw_off_bfg; out->res = h->fw_off_res; out->bres = h->fw_off_bres;
    out->skip = h->fw_off_skip; out->bskip = h->fw_off_bskip; out->bskip_total = h->fw_off_bskip_total;
    out->w1 = h->fw_off_w1; out->b1 = h->fw_off_b1; out->w2 = h->fw_off_w2; out->b2 = h->fw_off_b2;
    out->start_t = h->fw_off_start_t; out->start_b = h->fw_off_start_b;
    return WN_OK;
}

extern "C" int wn_train_set_deterministic(wn_handle* h, int32_t on) {
    g_err[0] = 0;
    if (!h) return wn_fail(WN_E_BADARG, "wn_train_set_deterministic: NULL handle");
    if (!h->chains.empty()) return wn_train_set_deterministic(h->chains[0], on);
    h->deterministic = on != 0;
    return WN_OK;
}

extern "C" int wn_train_export_params(wn_handle* h, float* params, void* hip_stream) {
    g_err[0] = 0;
    if (!h || !params) return wn_fail(WN_E_BADARG, "wn_train_export_params: NULL argument");
    if (!h->chains.empty()) return wn_train_export_params(h->chains[0], params, hip_stream);
    if (!h->have_weights) return wn_fail(WN_E_STATE, "wn_train_export_params: wn_load_weights has not been called");
    if (!h->fw_ok) return wn_fail(WN_E_UNSUPPORTED, "wn_train: needs kernel_size 2 and channel counts that are multiples of 32");
    if (h->padded) return wn_fail(WN_E_UNSUPPORTED, "wn_train: this handle runs a zero-padded channel shape (its parameter layout is not the caller's)");
    { int rc = rt_hip(hipSetDevice(h->cfg.device_id), "hipSetDevice"); if (rc) return rc; }
    return rt_hip(hipMemcpyAsync(params, h->d_fw, h->fw_floats * 4, hipMemcpyDeviceToDevice, (hipStream_t)hip_stream), "hipMemcpyAsync(params)");
}

// wn_train_pack / wn_train_unpack_grads: see wn_relayout (wn_forward.h) for the piece algebra.  A NULL tensor pointer is skipped (unpack: a gradient the
// caller does not want, e.g. the last layer's residual conv, which never reaches the loss).
static int wn_relayout_run(wn_handle* h, const wn_train_tensors* t, float* flat, bool unpack, hipStream_t st, const char* who) {
    const WnPlan& pl = h->plan;
    const int R = pl.R, D = pl.D, S = pl.S, E = pl.E, C = pl.C, NL = pl.NL;
    if (t->n_layers != NL) return wn_fail(WN_E_BADARG, "%s: n_layers = %d, the model has %d", who, t->n_layers, NL);
    if (!t->filter_w || !t->gate_w || !t->res_w || !t->skip_w) return wn_fail(WN_E_BADARG, "%s: a per-layer pointer array is NULL", who);
    if (pl.has_bias && (!t->filter_b || !t->gate_b || !t->res_b || !t->skip_b)) return wn_fail(WN_E_BADARG, "%s: cfg.bias = 1 but a per-layer bias array is NULL", who);
    std::vector<WnRelayoutBatch> batches;
    WnRelayoutBatch b;
    memset(&b, 0, sizeof(b));
    auto flush = [&]() { if (b.n > 0) { batches.push_back(b); memset(&b, 0, sizeof(b)); } };
    bool missing = false;
    auto piece = [&](void* ref, size_t off, int rows, int cols, int rs, int cs, int ld, int cm_blk, int cm_off) {
        if (!ref) { missing = true; return; }
        if (b.n == WN_RELAYOUT_PIECES) flush();
        WnRelayoutPiece& q = b.p[b.n++];
        q.ref = static_cast<float*>(ref); q.off = (long long)off; q.rows = rows; q.cols = cols; q.rs = rs; q.cs = cs; q.ld = ld; q.cm_blk = cm_blk; q.cm_off = cm_off;
        q.tile0 = b.tiles;
        b.tiles += ((rows + 31) / 32) * ((cols + 31) / 32);
    };
    for (int l = 0; l < NL; ++l) {
        for (int tap = 0; tap < 2; ++tap) {   // (D, R, 2) -> rows tap * R .. of [2R][2D], columns [F(32) | G(32)] per 32 channels
            const size_t off = h->fw_off_fg + ((size_t)l * 2 * R + (size_t)tap * R) * 2 * D;
            piece(t->filter_w[l] ? static_cast<float*>(t->filter_w[l]) + tap : nullptr, off, D, R, 2 * R, 2, 2 * D, 64, 0);
            piece(t->gate_w[l] ? static_cast<float*>(t->gate_w[l]) + tap : nullptr, off, D, R, 2 * R, 2, 2 * D, 64, 32);
        }
        piece(t->res_w[l], h->fw_off_res + (size_t)l * D * R, R, D, D, 1, R, 32, 0);       // (R, D, 1) -> [D][R]
        piece(t->skip_w[l], h->fw_off_skip + (size_t)l * D * S, S, D, D, 1, S, 32, 0);     // (S, D, 1) -> [D][S]
        if (pl.has_bias) {
            piece(t->filter_b[l], h->fw_off_bfg + (size_t)l * 2 * D, D, 1, 1, 0, 0, 64, 0);
            piece(t->gate_b[l], h->fw_off_bfg + (size_t)l * 2 * D, D, 1, 1, 0, 0, 64, 32);
            piece(t->res_b[l], h->fw_off_bres + (size_t)l * R, R, 1, 1, 0, 0, 32, 0);
            piece(t->skip_b[l], h->fw_off_bskip + (size_t)l * S, S, 1, 1, 0, 0, 32, 0);
        }
    }
    piece(t->end1_w, h->fw_off_w1, E, S, S, 1, E, 32, 0);        // (E, S, 1) -> [S][E]
    piece(t->end1_b, h->fw_off_b1, E, 1, 1, 0, 0, 32, 0);
    piece(t->end2_w, h->fw_off_w2, C, E, E, 1, C, 32, 0);        // (C, E, 1) -> [E][C]
    piece(t->end2_b, h->fw_off_b2, C, 1, 1, 0, 0, 32, 0);
    piece(t->start_w, h->fw_off_start_t, R, C, C, 1, R, 32, 0);  // (R, C, 1) -> [C][R]
    if (pl.has_bias) piece(t->start_b, h->fw_off_start_b, R, 1, 1, 0, 0, 32, 0);
    flush();
    if (missing && !unpack) return wn_fail(WN_E_BADARG, "%s: a parameter tensor pointer is NULL", who);
    for (const WnRelayoutBatch& bb : batches) {
        if (unpack) hipLaunchKernelGGL(wn_relayout<true>, dim3((unsigned)bb.tiles), dim3(256), 0, st, bb, flat);
        else hipLaunchKernelGGL(wn_relayout<false>, dim3((unsigned)bb.tiles), dim3(256), 0, st, bb, flat);
    }
    return rt_hip(hipGetLastError(), who);
}

static int wn_train_tensors_check(wn_handle* h, const char* who) {
    if (!h->have_weights) return wn_fail(WN_E_STATE, "%s: wn_load_weights has not been called", who);
    if (!h->fw_ok) return wn_fail(WN_E_UNSUPPORTED, "wn_train: needs kernel_size 2 and channel counts that are multiples of 32");
    if (h->padded) return wn_fail(WN_E_UNSUPPORTED, "wn_train: this handle runs a zero-padded channel shape (its parameter layout is not the caller's)");
    return rt_hip(hipSetDevice(h->cfg.device_id), "hipSetDevice");
}

extern "C" int wn_train_pack(wn_handle* h, const wn_train_tensors* tensors, float* params, void* hip_stream) {
    g_err[0] = 0;
    if (!h || !tensors || !params) return wn_fail(WN_E_BADARG, "wn_train_pack: NULL argument");
    if (!h->chains.empty()) return wn_train_pack(h->chains[0], tensors, params, hip_stream);
    { int rc = wn_train_tensors_check(h, "wn_train_pack"); if (rc) return rc; }
    // (sections no tensor maps to -- the bias sections of a model without biases, bskip_total -- read as zero, like the Python pack's torch.zeros)
    int rc = rt_hip(hipMemsetAsync(params, 0, h->fw_floats * 4, (hipStream_t)hip_stream), "hipMemsetAsync(params)");
    return rc ? rc : wn_relayout_run(h, tensors, params, false, (hipStream_t)hip_stream, "wn_train_pack");
}

extern "C" int wn_train_unpack_grads(wn_handle* h, const float* grads, const wn_train_tensors* tensors, void* hip_stream) {
    g_err[0] = 0;
    if (!h || !tensors || !grads) return wn_fail(WN_E_BADARG, "wn_train_unpack_grads: NULL argument");
    if (!h->chains.empty()) return wn_train_unpack_grads(h->chains[0], grads, tensors, hip_stream);
    { int rc = wn_train_tensors_check(h, "wn_train_unpack_grads"); if (rc) return rc; }
    return wn_relayout_run(h, tensors, const_cast<float*>(grads), true, (hipStream_t)hip_stream, "wn_train_unpack_grads");
}

static void wn_launch_cvt_t(hipStream_t st, const float* in, long long in_batch_stride, unsigned short* out, int rows, int cols, int batches) {
    hipLaunchKernelGGL(wn_cvt_bf16_transposed, dim3((unsigned)((cols + 31) / 32), (unsigned)((rows + 31) / 32), (unsigned)batches), dim3(256), 0, st,
                       in, in_batch_stride, out, rows, cols);
}

#ifndef WN_TN_MERGE_TAPS
#define WN_TN_MERGE_TAPS 1
#endif
// Deterministic mode (wn_train_set_deterministic / WN_DETERMINISTIC=1): the row splits of a weight-gradient product and the row blocks of a bias
// gradient store their partial results in a workspace -- one per stream the backward uses: the two run side by side -- and a second kernel adds them
// in order.  Bit-equal gradients from run to run for the price of one write and one read of the partial tiles (at most 67 MB per product).
static thread_local WnDetWs* t_tn_det[2] = {nullptr, nullptr};   // [0]: the caller's stream, [1]: the side stream
static thread_local hipStream_t t_tn_side = nullptr;
struct WnDetScope {   // the launches of one wn_train_forward / wn_train_backward call find the handle's partial-tile workspaces through the thread-locals
    explicit WnDetScope(wn_handle* h) { if (h->deterministic) { t_tn_det[0] = &h->det_ws[0]; t_tn_det[1] = &h->det_ws[1]; } t_tn_side = nullptr; }
    ~WnDetScope() { t_tn_det[0] = t_tn_det[1] = nullptr; t_tn_side = nullptr; }
};
static float* wn_det_part(hipStream_t st, size_t floats) {
    WnDetWs* w = t_tn_det[(t_tn_side && st == t_tn_side) ? 1 : 0];
    if (!w) return nullptr;
    if (w->floats < floats) {
        (void)hipDeviceSynchronize();
        rt_free(w->buf);
        w->buf = (float*)rt_malloc(floats * 4);
        w->floats = w->buf ? floats : 0;
    }
    return w->buf;
}

static void wn_launch_tn(hipStream_t st, WnGemmTnArgs a, bool bf16 = false) {
    // bf16 products with Nb % 256 == 0 take the 128 x 256 tile (A streamed once per 256 columns of B); rows split by wn_tn_grid (wn_plan.h)
    const bool wide16 = bf16 && !a.a_idx && a.a_bf16 && a.b_bf16 && a.Nb % 256 == 0;   // (both operands stored as bf16: the filter/gate weight gradient on the shadow of x -- two tap views, ka_split > 0 --, the skip weight gradient on the shadow of dskip)
    const bool wide = wide16 || (bf16 && !a.a_idx && !a.a_bf16 && a.Nb % 256 == 0);
    // both operands stored as bf16 and 256-column multiples on both sides: 256 x 256 tiles, one workgroup per CU, three chunks in flight (wn_bwd_wfg_bf16,
    // round 6) -- the filter/gate weight gradient of a 128 / 128 layer on the shadow of x (one tile: both tap views) and the grouped skip weight gradient
    // on the shadow of dskip.  WN_NO_TALL_WFG=1 with WN_TESTING=1: the 128 x 256 tiles of rounds 3-5.
    // (Measured, one stream, stand-alone: the filter/gate gradient 145 -> 134 us per layer; the grouped skip gradient 855 -> 892 us in ten 256 x 256 tiles per
    //  row split -- it stays on the 128 x 256 tiles unless WN_TALL_SKIP=1 asks for it.  Step: 54.6 -> 54.2 ms.  profiles/r06_tn_loads.txt.)
    const char* tall_skip = wn_dev_env("WN_TALL_SKIP");
    if (wide16 && !a.relu_a && a.Nb % 256 == 0 && ((a.ka_split == 128 && a.Ka == 256) || (a.ka_split == 0 && a.Ka % 256 == 0 && tall_skip && tall_skip[0] == '1'))) {
        const char* off = wn_dev_env("WN_NO_TALL_WFG");
        if (!(off && off[0] == '1')) {
            const int tiles = (a.Ka / 256) * (a.Nb / 256);
            int want_t = 256;   // workgroups in flight: one per CU
            { const char* wv = wn_dev_env("WN_TALL_WANT"); if (wv && atoi(wv) > 0) want_t = atoi(wv); }
            long long splits = want_t / tiles > 1 ? want_t / tiles : 1;
            if (splits >= 8) splits -= splits % 8;
            long long rps = (a.M + splits - 1) / splits;
            rps = (rps + 31) / 32 * 32;
            splits = (a.M + rps - 1) / rps;
            a.rows_per_split = rps; a.tiles_ka = a.Ka / 256; a.n_splits = (int)splits;
            a.part = wn_det_part(st, (size_t)splits * a.Ka * a.Nb);
            hipLaunchKernelGGL(wn_bwd_wfg_bf16, dim3(8u * (unsigned)tiles * (unsigned)((splits + 7) / 8)), dim3(512), 0, st, a);
            if (a.part)
                hipLaunchKernelGGL(wn_tn_reduce, dim3((unsigned)(((long long)a.Ka * a.Nb / 4 + 255) / 256)), dim3(256), 0, st, a.part, (int)splits, a.Ka, a.Nb, a.c, a.ldc, a.c_trans);
            return;
        }
    }
    // How many workgroups share the rows (each adds its 128 x 128 / 128 x 256 partial tile with fp32 atomics).  The bf16 forms: 512 -- one full round of
    // the resident slots (two 512-thread or three 256-thread workgroups per CU), half the atomic traffic of round 4's 1024 for the 128-column form:
    // 57.6 -> 55.5 ms per config-5 step (384-640: level; 256: 55.6; the 256-column form at 256 / 384 / 768 / 1024: 60.5 / 55.1 / 56.6 / 60.0 against
    // 54.9 at 512; profiles/r05_training_step_byte_cuts.txt).  The fp32 kernel is matrix-pipe bound and wants the rows spread wider: 1024 (512: 179 -> 194 ms).
    int want = (wide || (bf16 && !a.a_idx)) ? 512 : 1024;
    { const char* wv = wn_dev_env(wide ? "WN_TN_WANT_WIDE" : "WN_TN_WANT"); if (wv && atoi(wv) > 0) want = atoi(wv); }   // (A/B runs, with WN_TESTING=1)
    const WnTnGrid tg = wn_tn_grid(a.M, a.Ka, a.Nb, wide ? 256 : 128, want);
    a.rows_per_split = tg.rows_per_split;
    a.tiles_ka = tg.tiles_ka; a.n_splits = tg.splits;
    const dim3 grid(tg.blocks);   // wn_tile_of: the tiles of a row split share an XCD
    a.part = wn_det_part(st, (size_t)tg.splits * a.Ka * a.Nb);   // (NULL outside the deterministic mode: atomics)
    // (b_bf16 / a_bf16: that operand is stored as bf16 -- [dF|dG] in the filter/gate weight gradient, z in the residual and skip ones)
    if (wide16) hipLaunchKernelGGL((wn_bwd_gemm_tn_bf16<8, true, true>), grid, dim3(512), 0, st, a);
    else if (wide && a.b_bf16) hipLaunchKernelGGL((wn_bwd_gemm_tn_bf16<8, false, true>), grid, dim3(512), 0, st, a);
    else if (wide) hipLaunchKernelGGL((wn_bwd_gemm_tn_bf16<8, false, false>), grid, dim3(512), 0, st, a);
    else if (bf16 && !a.a_idx && a.b_bf16) hipLaunchKernelGGL((wn_bwd_gemm_tn_bf16<4, false, true>), grid, dim3(256), 0, st, a);
    else if (bf16 && !a.a_idx && a.a_bf16) hipLaunchKernelGGL((wn_bwd_gemm_tn_bf16<4, true, false>), grid, dim3(256), 0, st, a);
    else if (bf16 && !a.a_idx) hipLaunchKernelGGL((wn_bwd_gemm_tn_bf16<4, false, false>), grid, dim3(256), 0, st, a);  // bf16 matrix operands, fp32 accumulation
    else hipLaunchKernelGGL(wn_bwd_gemm_tn, grid, dim3(256), 0, st, a);
    if (a.part)
        hipLaunchKernelGGL(wn_tn_reduce, dim3((unsigned)(((long long)a.Ka * a.Nb / 4 + 255) / 256)), dim3(256), 0, st, a.part, tg.splits, a.Ka, a.Nb, a.c, a.ldc, a.c_trans);
}

static void wn_launch_colsum(hipStream_t st, const WnRowMap& x, long long M, int rows_per_batch, int N, float* out, bool x16 = false) {
    const unsigned blocks = (unsigned)((M + 511) / 512);
    float* part = (N % 4 == 0) ? wn_det_part(st, (size_t)blocks * N) : nullptr;
    if (x16) hipLaunchKernelGGL(wn_bwd_colsum<true>, dim3(blocks, (unsigned)((N + 63) / 64)), dim3(64), 0, st, x, M, rows_per_batch, N, out, part);
    else hipLaunchKernelGGL(wn_bwd_colsum<false>, dim3(blocks, (unsigned)((N + 63) / 64)), dim3(64), 0, st, x, M, rows_per_batch, N, out, part);
    if (part) hipLaunchKernelGGL(wn_tn_reduce, dim3((unsigned)((N / 4 + 255) / 256)), dim3(256), 0, st, part, (int)blocks, 1, N, out, N, 0);
}

static void wn_launch_transpose(hipStream_t st, const float* in, long long in_batch_stride, float* out, int rows, int cols, int batches) {
    hipLaunchKernelGGL(wn_transpose_batched, dim3((unsigned)((cols + 31) / 32), (unsigned)((rows + 31) / 32), (unsigned)batches), dim3(256), 0, st,
                       in, in_batch_stride, out, rows, cols);
}

// (Round 4 tried the side stream at the LOWEST queue priority -- the chain's workgroups first, the side work in what is left: 63.7 -> 70.5 ms
//  per bf16 config-5 step.  The side work is a third of the step's kernel time; starved, it is left over at the end.
//  profiles/r04_fused_forward_layer.txt.)
extern "C" int wn_train_forward(wn_handle* h, const float* params, const int32_t* indices, int64_t N, int64_t L, int64_t out_len,
                                float* logits, void* hip_stream) {
    g_err[0] = 0;
    if (!h || !params || !indices || !logits) return wn_fail(WN_E_BADARG, "wn_train_forward: NULL argument");
    if (!h->chains.empty()) return wn_train_forward(h->chains[0], params, indices, N, L, out_len, logits, hip_stream);
    if (!h->have_weights) return wn_fail(WN_E_STATE, "wn_train_forward: wn_load_weights has not been called");
    if (N < 1 || out_len < 1) return wn_fail(WN_E_BADARG, "wn_train_forward: N and output_length must be >= 1");
    const WnPlan& pl = h->plan;
    const int R = pl.R, D = pl.D, S = pl.S, E = pl.E, C = pl.C, NL = pl.NL;
    if (!h->fw_ok) return wn_fail(WN_E_UNSUPPORTED, "wn_train_forward: needs kernel_size 2 and channel counts that are multiples of 32");
    if (h->padded) return wn_fail(WN_E_UNSUPPORTED, "wn_train_forward: this handle runs a zero-padded channel shape (its parameter layout is not the caller's)");
    if ((long long)N * L >= 0x7fffffffll) return wn_fail(WN_E_UNSUPPORTED, "wn_train_forward: N*L must stay below 2^31 rows");
    { int rc = rt_hip(hipSetDevice(h->cfg.device_id), "hipSetDevice"); if (rc) return rc; }
    WnTrainLay& t = h->train;
    h->train_valid = false;
    WnDetScope det_scope(h);
    { int rc = wn_train_layout_ws(h, N, L, out_len, t); if (rc) return rc; }
    if (h->tws_floats < t.total) {
        (void)hipDeviceSynchronize();
        rt_free(h->d_tws);
        h->d_tws = nullptr;
        float* p = nullptr;
        if (hipMalloc((void**)&p, t.total * 4) != hipSuccess) { h->tws_floats = 0; return wn_fail(WN_E_NOMEM, "wn_train_forward: workspace of %.1f MB", t.total * 4e-6); }
        h->d_tws = p; h->tws_floats = t.total;
    }
    float* ws = h->d_tws;
    hipStream_t st = (hipStream_t)hip_stream;
    const float* fw = params;
    int rc = rt_hip(hipMemcpyAsync(ws + t.idx, indices, (size_t)N * L * 4, hipMemcpyDeviceToDevice, st), "hipMemcpyAsync(indices)");
    if (rc) return rc;
    // operand layouts of the backward products, rebuilt from the (just updated) parameters
    wn_launch_transpose(st, fw + h->fw_off_res, (long long)D * R, ws + t.res_o, D, R, NL);      // [D][R] -> [R][D]
    for (int b = 0; b < t.nblk; ++b) {  // per block of G layers: [cnt*D][S] -> [S][cnt*D], the operand of the backward's grouped skip product
        const int first = b * t.G, cnt = NL - first < t.G ? NL - first : t.G;
        wn_launch_transpose(st, fw + h->fw_off_skip + (size_t)first * D * S, 0, ws + t.skip_o + (size_t)first * S * D, cnt * D, S, 1);
    }
    wn_launch_transpose(st, fw + h->fw_off_w1, 0, ws + t.w1_o, S, E, 1);                         // [S][E] -> [E][S]
    wn_launch_transpose(st, fw + h->fw_off_w2, 0, ws + t.w2_o, E, C, 1);                         // [E][C] -> [C][E]
    wn_launch_transpose(st, fw + h->fw_off_fg, (long long)2 * R * 2 * D, ws + t.fgb0, R, 2 * D, NL);               // tap 0 rows -> [2D][R]
    wn_launch_transpose(st, fw + h->fw_off_fg + (size_t)R * 2 * D, (long long)2 * R * 2 * D, ws + t.fgb1, R, 2 * D, NL);
    const int G = t.G;
    const bool bf16 = h->fw_bf16 && h->fwb_ok;
    t.bf16 = bf16;
    unsigned short* bw = reinterpret_cast<unsigned short*>(ws + t.bw);
    unsigned short* bt_fg = reinterpret_cast<unsigned short*>(ws + t.bt_fg);
    unsigned short* bt_res = reinterpret_cast<unsigned short*>(ws + t.bt_res);
    unsigned short* bt_skip = reinterpret_cast<unsigned short*>(ws + t.bt_skip);
    unsigned short* bt_w1 = reinterpret_cast<unsigned short*>(ws + t.bt_w1);
    unsigned short* bt_w2 = reinterpret_cast<unsigned short*>(ws + t.bt_w2);
    if (bf16) {
        const long long n = (long long)h->fw_floats;
        hipLaunchKernelGGL(wn_cvt_bf16, dim3((unsigned)((n / 2 + 256) / 256)), dim3(256), 0, st, fw, bw, n);
        wn_launch_cvt_t(st, fw + h->fw_off_fg, (long long)2 * R * 2 * D, bt_fg, 2 * R, 2 * D, NL);        // [2R][2D] -> [2D][2R]
        wn_launch_cvt_t(st, fw + h->fw_off_res, (long long)D * R, bt_res, D, R, NL);                          // [D][R] -> [R][D]
        wn_launch_cvt_t(st, fw + h->fw_off_skip, (long long)G * D * S, bt_skip, G * D, S, NL / G);            // [G*D][S] -> [S][G*D] per block
        wn_launch_cvt_t(st, fw + h->fw_off_w1, 0, bt_w1, S, E, 1);
        wn_launch_cvt_t(st, fw + h->fw_off_w2, 0, bt_w2, E, C, 1);
    }
    if (pl.has_bias) {  // the grouped skip GEMM adds the sum of all layers' skip biases once
        rc = rt_hip(hipMemsetAsync(ws + t.bskip_total, 0, (size_t)S * 4, st), "hipMemsetAsync");
        if (rc) return rc;
        wn_launch_colsum(st, WnRowMap{fw + h->fw_off_bskip, 0, S, 0}, NL, NL, S, ws + t.bskip_total);
    }
    {
        const long long rows = N * L, work = rows * (R / 4);
        hipLaunchKernelGGL(wn_fwd_start, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, st, indices, fw + h->fw_off_start_t,
                           pl.has_bias ? fw + h->fw_off_start_b : nullptr, ws + t.x[0], rows, R,
                           (bf16 && !t.xh.empty()) ? reinterpret_cast<unsigned short*>(ws + t.xh[0]) : (unsigned short*)nullptr);
    }
    float* skip = ws + t.skip; float* ev = ws + t.ev;
    // The grouped skip product of a block (zg . [Wskip of its layers]: 1.2 ms at config 5) hangs off the layer chain -- the next block's
    // layers do not need it, only the head does: it runs on the side stream next to them (see wn_train_backward for the two-stream scheme).
    const char* one_env = wn_dev_env("WN_TRAIN_ONE_STREAM");
    const bool two = !(one_env && one_env[0] == '1');
    if (two && !h->side_stream) {
        rc = rt_hip(hipStreamCreateWithFlags(&h->side_stream, hipStreamNonBlocking), "hipStreamCreateWithFlags");
        if (rc) return rc;
    }
    hipStream_t sd = two ? h->side_stream : st;
    t_tn_side = two ? h->side_stream : nullptr;
    size_t ev_next = 0;
    auto signal = [&](hipStream_t from) -> hipEvent_t {
        if (ev_next == h->events.size()) {
            hipEvent_t e = nullptr;
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
            h->events.push_back(e);
        }
        hipEvent_t e = h->events[ev_next++];
        (void)hipEventRecord(e, from);
        return e;
    };
    auto wait_for = [&](hipStream_t who, hipEvent_t e) { if (two && e) (void)hipStreamWaitEvent(who, e, 0); };
    for (int l = 0; l < NL; ++l) {
        const long long d = h->dil[l], rows = t.need[l + 1], t0 = L - rows;
        const int gi = l % G;
        float* xin = ws + t.x[l];
        const WnRowMap zmap = wn_z_map(t, ws, NL, D, l, bf16);   // z_l: a strided view of the block's Z_b (wn_train_layout_ws)
        WnGemmArgs a;
        memset(&a, 0, sizeof(a));
        const bool shadow = bf16 && !t.xh.empty();   // matrix operand reads of x take its bf16 shadow (wn_train_layout_ws)
        const float* xop = shadow ? ws + t.xh[l] : xin;   // (a bf16 matrix behind a float pointer: the row maps count bf16 elements then)
        a.a0 = WnRowMap{xop, (long long)L * R, R, t0 - d};
        a.a_skip_lo[0] = (int)t.zlo[l];   // (short clips: the reference's left zero padding stands in for x(t - d) on these rows)
        a.a1 = WnRowMap{xop, (long long)L * R, R, t0};
        a.a_bf16 = shadow ? 1 : 0;
        a.k_split = R; a.K = 2 * R; a.bt = fw + h->fw_off_fg + (size_t)l * 2 * R * 2 * D; a.N = 2 * D;
        a.bias = pl.has_bias ? fw + h->fw_off_bfg + (size_t)l * 2 * D : nullptr;
        a.c = zmap;   // (no second copy on the skip rows: the grouped skip product reads those rows of Z_b where they lie)
        a.gate_t = ws + t.th[l]; a.gate_g = ws + t.sg[l];
        a.gate_packed = bf16 ? 1 : 0;  // bf16 step: tanh and sigmoid saved as one {bf16, bf16} dword per element (half the bytes, written once)
        a.c_bf16 = bf16 ? 1 : 0;       //            z (and its copy on the skip rows, zg) stored as bf16
        a.M = N * rows; a.rows_per_batch = (int)rows;
        WnGemmArgs ar;   // the residual product x_{l+1} = z . Wres^T + bias + x_l
        memset(&ar, 0, sizeof(ar));
        if (l < NL - 1) {
            ar.a0 = ar.a1 = zmap;
            ar.k_split = D; ar.K = D; ar.bt = fw + h->fw_off_res + (size_t)l * D * R; ar.N = R;
            ar.bias = pl.has_bias ? fw + h->fw_off_bres + (size_t)l * R : nullptr;
            ar.cin = WnRowMap{xin, (long long)L * R, R, t0};
            ar.c = WnRowMap{ws + t.x[l + 1], (long long)L * R, R, t0};
            if (shadow) ar.c_h = reinterpret_cast<unsigned short*>(ws + t.xh[l + 1]);
            ar.M = N * rows; ar.rows_per_batch = (int)rows; ar.a_bf16 = bf16 ? 1 : 0;
        }
        // (bf16 step, the 128/128 shape: both products of the layer in one launch, z handed over in LDS -- wn_fwd_layer_bf16)
        const bool fused = bf16 && l < NL - 1 && wn_launch_layer(st, a, bt_fg + (size_t)l * 2 * D * 2 * R, ar, bt_res + (size_t)l * R * D);
        if (!fused) {
            wn_launch_nn(st, WN_EPI_GATE, a, bf16 ? bt_fg + (size_t)l * 2 * D * 2 * R : nullptr);
            if (l < NL - 1) wn_launch_nn(st, WN_EPI_PLAIN, ar, bf16 ? bt_res + (size_t)l * R * D : nullptr);
        }
        if (gi == G - 1 || l == NL - 1) {
            const int first = l - gi, cnt = gi + 1;
            memset(&a, 0, sizeof(a));
            a.a0 = a.a1 = wn_zg_map(t, ws, NL, D, first / G);
            a.k_split = cnt * D; a.K = cnt * D; a.bt = fw + h->fw_off_skip + (size_t)first * D * S; a.N = S;
            a.bias = (pl.has_bias && first == 0) ? ws + t.bskip_total : nullptr;
            if (first > 0) a.cin = WnRowMap{skip, out_len * S, S, 0};
            a.c = WnRowMap{skip, out_len * S, S, 0};
            a.M = N * out_len; a.rows_per_batch = (int)out_len; a.a_bf16 = bf16 ? 1 : 0;
            wait_for(sd, signal(st));   // the block's Z_b is complete (every gate product of the block ran on the caller's stream)
            wn_launch_nn(sd, WN_EPI_PLAIN, a, bf16 ? bt_skip + (size_t)(first / G) * S * G * D : nullptr);
        }
    }
    wait_for(st, signal(sd));   // skip is complete
    {
        WnGemmArgs a;
        memset(&a, 0, sizeof(a));
        a.a0 = a.a1 = WnRowMap{skip, out_len * S, S, 0};
        a.k_split = S; a.K = S; a.bt = fw + h->fw_off_w1; a.N = E; a.bias = fw + h->fw_off_b1;
        a.c = WnRowMap{ev, out_len * E, E, 0};
        a.M = N * out_len; a.rows_per_batch = (int)out_len; a.relu_a = 1; a.relu_c = 1;
        wn_launch_nn(st, WN_EPI_PLAIN, a, bf16 ? bt_w1 : nullptr);
        memset(&a, 0, sizeof(a));
        a.a0 = a.a1 = WnRowMap{ev, out_len * E, E, 0};
        a.k_split = E; a.K = E; a.bt = fw + h->fw_off_w2; a.N = C; a.bias = fw + h->fw_off_b2;
        a.c = WnRowMap{logits, out_len * C, C, 0};
        a.M = N * out_len; a.rows_per_batch = (int)out_len;
        wn_launch_nn(st, WN_EPI_PLAIN, a, bf16 ? bt_w2 : nullptr);
    }
    rc = rt_hip(hipGetLastError(), "wn_train_forward launches");
    if (rc) return rc;
    h->train_valid = true;
    return WN_OK;
}

extern "C" int wn_train_backward(wn_handle* h, const float* params, const float* dlogits, float* grads, void* hip_stream) {
    g_err[0] = 0;
    if (!h || !params || !dlogits || !grads) return wn_fail(WN_E_BADARG, "wn_train_backward: NULL argument");
    if (!h->chains.empty()) return wn_train_backward(h->chains[0], params, dlogits, grads, hip_stream);
    if (!h->train_valid) return wn_fail(WN_E_STATE, "wn_train_backward: no wn_train_forward to differentiate");
    { int rc = rt_hip(hipSetDevice(h->cfg.device_id), "hipSetDevice"); if (rc) return rc; }
    const WnPlan& pl = h->plan;
    const int R = pl.R, D = pl.D, S = pl.S, E = pl.E, C = pl.C, NL = pl.NL;
    const WnTrainLay& t = h->train;
    WnDetScope det_scope(h);
    const long long N = t.N, L = t.L, out_len = t.out_len, Mo = N * out_len;
    float* ws = h->d_tws;
    hipStream_t st = (hipStream_t)hip_stream;
    (void)params;  // the operand layouts of this step's parameters were rebuilt by wn_train_forward
    int rc = rt_hip(hipMemsetAsync(grads, 0, h->fw_floats * 4, st), "hipMemsetAsync(grads)");
    if (rc) return rc;
    float* dskip = ws + t.dskip; float* de = ws + t.de;
    // Two streams.  The activation-gradient chain (dz -> [dF|dG] -> dx, layer after layer) is strictly sequential; the weight-gradient
    // products only hang off it (dWres needs dx', dWfg needs [dF|dG], the grouped dWskip needs dskip) and nobody waits for them before
    // the optimizer.  They run on a side stream, ordered by events: every product of the chain then has a second, independent kernel
    // next to it that fills its tail (a 128-row-tile product of 3250 workgroups leaves the last of its four waves of workgroups 17 % full)
    // and uses the memory system while the other one computes.  [dF|dG] is double buffered (layer l writes buffer l & 1): the side
    // stream may still read layer l's while the chain writes layer l-1's; the chain waits for the side stream only where it would
    // overwrite something the side stream has not read yet (dx ping-pong buffer: dWres of layer l before dx of layer l-1; [dF|dG]
    // buffer: dWfg of layer l before the gate derivative of layer l-2).  WN_TRAIN_ONE_STREAM=1 (with WN_TESTING=1) keeps everything on
    // the caller's stream (A/B runs).
    const char* one_env = wn_dev_env("WN_TRAIN_ONE_STREAM");
    const bool two = !(one_env && one_env[0] == '1');
    if (two && !h->side_stream) {
        rc = rt_hip(hipStreamCreateWithFlags(&h->side_stream, hipStreamNonBlocking), "hipStreamCreateWithFlags");
        if (rc) return rc;
    }
    hipStream_t sd = two ? h->side_stream : st;
    t_tn_side = two ? h->side_stream : nullptr;
    size_t ev_next = 0;
    auto signal = [&](hipStream_t from) -> hipEvent_t {   // an event recorded on `from` now
        if (ev_next == h->events.size()) {
            hipEvent_t e = nullptr;
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
            h->events.push_back(e);
        }
        hipEvent_t e = h->events[ev_next++];
        (void)hipEventRecord(e, from);
        return e;
    };
    auto wait_for = [&](hipStream_t who, hipEvent_t e) { if (two && e) (void)hipStreamWaitEvent(who, e, 0); };
    std::vector<hipEvent_t> res_read(NL, nullptr), fg_read(NL, nullptr);   // side stream: dWres / dWfg of layer l have read their operands
    // bf16 mode: the "NN" products take their weight operand ([N][K] bf16) straight from the bf16 copy of the parameter blob
    const unsigned short* bw = t.bf16 ? reinterpret_cast<const unsigned short*>(ws + t.bw) : nullptr;
    const float* skip = ws + t.skip; const float* ev = ws + t.ev;
    WnGemmArgs a;
    WnGemmTnArgs g;
    // ---- head.  Only de and dskip are on the way to the layers; the head's own weight and bias gradients (two products over the rows, three column sums:
    // 0.7 ms at config 5) hang off them like every other weight gradient and run on the side stream, behind the first dzg product the chain waits for
    // (round 4 ran them on the caller's stream in front of dskip).
    auto head_weight_grads = [&](hipStream_t s2) -> int {
        WnGemmTnArgs gh;
        memset(&gh, 0, sizeof(gh));   // dW2^T [E][C] = e^T . dlogits
        gh.a = WnRowMap{ev, out_len * E, E, 0}; gh.b = WnRowMap{dlogits, out_len * C, C, 0};
        gh.Ka = E; gh.Nb = C; gh.c = grads + h->fw_off_w2; gh.ldc = C; gh.M = Mo; gh.rows_per_batch = (int)out_len;
        wn_launch_tn(s2, gh, t.bf16);
        wn_launch_colsum(s2, WnRowMap{dlogits, out_len * C, C, 0}, Mo, (int)out_len, C, grads + h->fw_off_b2);
        memset(&gh, 0, sizeof(gh));   // dW1^T [S][E] = relu(skip)^T . de
        gh.a = WnRowMap{skip, out_len * S, S, 0}; gh.b = WnRowMap{de, out_len * E, E, 0}; gh.relu_a = 1;
        gh.Ka = S; gh.Nb = E; gh.c = grads + h->fw_off_w1; gh.ldc = E; gh.M = Mo; gh.rows_per_batch = (int)out_len;
        wn_launch_tn(s2, gh, t.bf16);
        wn_launch_colsum(s2, WnRowMap{de, out_len * E, E, 0}, Mo, (int)out_len, E, grads + h->fw_off_b1);
        if (pl.has_bias) {          // every layer's skip bias sees the same gradient
            int rc2 = rt_hip(hipMemsetAsync(ws + t.colsum_tmp, 0, (size_t)S * 4, s2), "hipMemsetAsync");
            if (rc2) return rc2;
            wn_launch_colsum(s2, WnRowMap{dskip, out_len * S, S, 0}, Mo, (int)out_len, S, ws + t.colsum_tmp);
            for (int l = 0; l < NL; ++l) {
                rc2 = rt_hip(hipMemcpyAsync(grads + h->fw_off_bskip + (size_t)l * S, ws + t.colsum_tmp, (size_t)S * 4, hipMemcpyDeviceToDevice, s2), "hipMemcpyAsync(dbskip)");
                if (rc2) return rc2;
            }
        }
        return 0;
    };
    memset(&a, 0, sizeof(a));   // de = (dlogits . W2) * [e > 0]
    a.a0 = a.a1 = WnRowMap{dlogits, out_len * C, C, 0};
    a.k_split = C; a.K = C; a.bt = ws + t.w2_o; a.N = E;
    a.c = WnRowMap{de, out_len * E, E, 0}; a.mask = ev;
    a.M = Mo; a.rows_per_batch = (int)out_len;
    wn_launch_nn(st, WN_EPI_PLAIN, a, bw ? bw + h->fw_off_w2 : nullptr);
    memset(&a, 0, sizeof(a));   // dskip = (de . W1) * [skip > 0]
    a.a0 = a.a1 = WnRowMap{de, out_len * E, E, 0};
    a.k_split = E; a.K = E; a.bt = ws + t.w1_o; a.N = S;
    a.c = WnRowMap{dskip, out_len * S, S, 0}; a.mask = skip;
    // bf16 step: dskip is a matrix operand twice per skip block (the dzg product's A, the skip weight gradient's) and an fp32 column sum once (the
    // skip biases): the product also writes the bits those operand reads would round it to (as x has its shadow), they take half the bytes
    unsigned short* dskip_h = t.bf16 ? reinterpret_cast<unsigned short*>(ws + t.dskip_h) : nullptr;
    { const char* off = wn_dev_env("WN_NO_DSKIP_SHADOW"); if (off && off[0] == '1') dskip_h = nullptr; }   // (A/B runs, with WN_TESTING=1)
    a.c_h = dskip_h;
    a.M = Mo; a.rows_per_batch = (int)out_len;
    wn_launch_nn(st, WN_EPI_PLAIN, a, bw ? bw + h->fw_off_w1 : nullptr);
    wait_for(sd, signal(st));   // grads cleared, de and dskip complete: the side stream may start
    // The skip path's gradients, one block of G layers at a time (as in the forward):
    //   dzg_b [Mo][cnt*D] = dskip . [Wskip of the block's layers]      dWskip^T of the block [cnt*D][S] = zg^T . dskip
    // so dskip (0.7 GB at config 5) is read twice per block instead of twice per layer; the gate step of a layer adds its column block of
    // dzg_b to dz on the skip rows.  Both products only need dskip: all blocks are enqueued NOW on the side stream, last block first --
    // the chain waits 1.2 ms for the last block's dzg (as it always did) and finds the others ready (one dzg buffer per block).
    std::vector<hipEvent_t> dzg_ready(t.nblk, nullptr);
    // bf16 step: dzg is STORED as bf16 (WN_DZG_BF16, wn_forward.h) -- element (block b, row, column) of the buffer, as a float pointer the row maps carry
    const bool dzg16 = t.bf16 && WN_DZG_BF16;
    auto dzg_at = [&](int b, size_t col) -> float* {
        const size_t e = (size_t)b * ((size_t)Mo * t.G * D) + col;
        return dzg16 ? reinterpret_cast<float*>(reinterpret_cast<unsigned short*>(ws + t.dzg) + e) : ws + t.dzg + e;
    };
    // dWskip^T of block b.  Nobody waits for it before the optimizer, so it can run on either stream.  Round 6 (profiles/r06_train_timeline.txt): the
    // caller's stream sits idle ~190 us in front of every layer of the chain, waiting for the side stream to have read its double-buffered [dF|dG] / dx
    // buffers (side stream: 30 ms of products per backward, chain: 22 ms) -- but SOME kernel runs 97 % of the time, and moving products to the caller's
    // stream (WN_TRAIN_SKIP_MAIN=1: the five dWskip behind each block of the chain; WN_TRAIN_RES_MAIN=k: every k-th layer's dWres; both with
    // WN_TESTING=1) bought nothing (54.6 -> 54.5-57.4 ms): the step is bound by the bytes its kernels move, not by the queue they wait in.  Default: all on
    // the side stream, as in rounds 4-5.
    const char* skm_env = wn_dev_env("WN_TRAIN_SKIP_MAIN");
    const bool skip_main = two && skm_env && skm_env[0] == '1';   // (measured: no gain -- profiles/r06_train_timeline.txt; default off)
    const char* rsm_env = wn_dev_env("WN_TRAIN_RES_MAIN");
    const int res_main = two && rsm_env ? atoi(rsm_env) : 0;
    auto skip_weight_grads = [&](hipStream_t s2, int b) {
        const int first = b * t.G, cnt = NL - first < t.G ? NL - first : t.G;
        WnGemmTnArgs gs;
        memset(&gs, 0, sizeof(gs));
        gs.a = wn_zg_map(t, ws, NL, D, b); gs.b = WnRowMap{dskip, out_len * S, S, 0};   // (the block's z on the skip rows, where the forward left it)
        gs.Ka = cnt * D; gs.Nb = S; gs.c = grads + h->fw_off_skip + (size_t)first * D * S; gs.ldc = S; gs.M = Mo; gs.rows_per_batch = (int)out_len;
        if (t.bf16) {   // zg is stored as bf16: it goes in as B (operands swapped, C written transposed -- same [cnt*D][S] gradient)
            WnRowMap zmap = gs.a; gs.a = gs.b; gs.b = zmap; gs.Ka = S; gs.Nb = cnt * D; gs.b_bf16 = 1; gs.c_trans = 1;
            if (dskip_h && (cnt * D) % 256 == 0) { gs.a = WnRowMap{reinterpret_cast<const float*>(dskip_h), out_len * S, S, 0}; gs.a_bf16 = 1; }   // (the 256-column tile has the form with both operands stored as bf16)
        }
        wn_launch_tn(s2, gs, t.bf16);
    };
    for (int b = t.nblk - 1; b >= 0; --b) {
        const int first = b * t.G, cnt = NL - first < t.G ? NL - first : t.G;
        float* dzg_b = dzg_at(b, 0);
        memset(&a, 0, sizeof(a));
        if (dskip_h) { a.a0 = a.a1 = WnRowMap{reinterpret_cast<const float*>(dskip_h), out_len * S, S, 0}; a.a_bf16 = 1; }   // (the same bits, half the bytes)
        else a.a0 = a.a1 = WnRowMap{dskip, out_len * S, S, 0};
        a.k_split = S; a.K = S; a.bt = ws + t.skip_o + (size_t)first * S * D; a.N = cnt * D;
        a.c = WnRowMap{dzg_b, out_len * (long long)cnt * D, (long long)cnt * D, 0}; a.c_bf16 = dzg16 ? 1 : 0;
        a.M = Mo; a.rows_per_batch = (int)out_len;
        wn_launch_nn(sd, WN_EPI_PLAIN, a, bw ? bw + h->fw_off_skip + (size_t)first * D * S : nullptr);
        if (two) dzg_ready[b] = signal(sd);
        if (b == t.nblk - 1) { rc = head_weight_grads(sd); if (rc) return rc; }   // (behind the one product the chain is waiting for)
        if (!skip_main) skip_weight_grads(sd, b);
    }
    // ---- layers, last to first
    float* dxn = ws + t.dxa;  // dLoss/dx_{l+1}
    float* dxc = ws + t.dxb;  // dLoss/dx_l
    // The arguments of layer k's gate-derivative product dz = dx' . Wres -> [dF | dG] (+ the layer's share of dzg on the skip rows): the
    // gate derivative is the product's epilogue (WN_EPI_GATE_BWD), dz is never written (round 2: dz to HBM, then a streaming gate kernel).
    auto gate_bwd_args = [&](int k, const float* dx_in, WnGemmArgs& o) {
        const long long rows_k = t.need[k + 1], t0_k = L - rows_k;
        const int gi_k = k % t.G, first_k = k - gi_k, cnt_k = NL - first_k < t.G ? NL - first_k : t.G;
        memset(&o, 0, sizeof(o));
        o.a0 = o.a1 = WnRowMap{dx_in, L * (long long)R, R, t0_k};
        o.k_split = R; o.K = R; o.bt = ws + t.res_o + (size_t)k * R * D; o.N = D;
        o.c = WnRowMap{ws + ((k & 1) ? t.dfg2 : t.dfg), rows_k * 2 * D, 2 * D, 0}; o.c_bf16 = t.bf16 ? 1 : 0;   // bf16 step: [dF|dG] is STORED as bf16 (it only ever feeds bf16 matrix operands)
        o.c2 = WnRowMap{dzg_at(k / t.G, (size_t)gi_k * D), out_len * (long long)cnt_k * D, (long long)cnt_k * D, 0};   // (bf16 step: bf16 elements, WN_DZG_BF16)
        o.c2_first_row = (int)(rows_k - out_len);
        o.gate_t = ws + t.th[k]; o.gate_g = ws + t.sg[k]; o.gate_packed = t.bf16 ? 1 : 0;
        o.M = N * rows_k; o.rows_per_batch = (int)rows_k;
    };
    // what the chain has to wait for before it may write layer k's [dF|dG]: the buffer's previous reader (dWfg of layer k + 2, side stream)
    // and -- entering a block from above -- the block's dzg (side stream, enqueued before the loop)
    auto gate_bwd_waits = [&](int k) {
        if (k + 2 < NL) wait_for(st, fg_read[k + 2]);
        const int gi_k = k % t.G, first_k = k - gi_k, cnt_k = NL - first_k < t.G ? NL - first_k : t.G;
        if (gi_k == cnt_k - 1) wait_for(st, dzg_ready[k / t.G]);
    };
    bool have_dfg = false;   // [dF|dG] of the layer at hand came out of the previous iteration's fused launch (wn_bwd_layer_bf16)
    for (int l = NL - 1; l >= 0; --l) {
        const long long d = h->dil[l], rows = t.need[l + 1], t0 = L - rows, M = N * rows;
        const float* xin = ws + t.x[l];
        const bool has_res = l < NL - 1;
        float* dfg = ws + ((l & 1) ? t.dfg2 : t.dfg);
        const int gi = l % t.G, first = l - gi, cnt = NL - first < t.G ? NL - first : t.G;
        if (!have_dfg) gate_bwd_waits(l);
        if (has_res) {
            if (!have_dfg) {
                gate_bwd_args(l, dxn, a);
                wn_launch_nn(st, WN_EPI_GATE_BWD, a, bw ? bw + h->fw_off_res + (size_t)l * D * R : nullptr);
            }
            memset(&g, 0, sizeof(g));   // dWres^T [D][R] = z^T . dx'
            g.a = wn_z_map(t, ws, NL, D, l, t.bf16); g.b = WnRowMap{dxn, L * (long long)R, R, t0};
            g.Ka = D; g.Nb = R; g.c = grads + h->fw_off_res + (size_t)l * D * R; g.ldc = R; g.M = M; g.rows_per_batch = (int)rows;
            g.a_bf16 = t.bf16 ? 1 : 0;   // z is stored as bf16 in the bf16 step (here as A: ~1000 row splits, see wn_bwd_gemm_tn_bf16)
            const bool res_here = res_main > 0 && l % res_main == 0;   // (this layer's dWres on the caller's stream: in order behind dx', no event)
            hipStream_t sr = res_here ? st : sd;
            wn_launch_tn(sr, g, t.bf16);   // (dx' is complete: the side stream waited for the dx product of layer l + 1, below)
            if (pl.has_bias) wn_launch_colsum(sr, WnRowMap{dxn, L * (long long)R, R, t0}, M, (int)rows, R, grads + h->fw_off_bres + (size_t)l * R);
            if (two && !res_here) res_read[l] = signal(sd);
        } else {   // the last layer has no residual output: dz is its share of dzg alone
            const long long work = M * D / 4;   // (four channels per thread; no residual share: dz = NULL, nothing zero-filled)
            const float* no_dz = nullptr;
            if (t.bf16)
                hipLaunchKernelGGL(wn_bwd_gate<true>, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, st, no_dz, ws + t.th[l], ws + t.sg[l], dfg, M, D,
                                   dzg_at(l / t.G, (size_t)gi * D), cnt * D, (int)rows, (int)out_len);
            else
                hipLaunchKernelGGL(wn_bwd_gate<false>, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, st, no_dz, ws + t.th[l], ws + t.sg[l], dfg, M, D,
                                   dzg_at(l / t.G, (size_t)gi * D), cnt * D, (int)rows, (int)out_len);
        }
        wait_for(sd, signal(st));   // [dF|dG] of this layer is complete
        // dWfg^T [2R][2D]: rows 0..R-1 = x_l(t - d)^T . dfg (tap 0), rows R.. = x_l(t)^T . dfg (tap 1) -- one launch, the taps are two
        // row views of A (ka_split): the workgroups of the two taps run side by side and read the same rows of dfg.
        memset(&g, 0, sizeof(g));
        const bool shadow = t.bf16 && !t.xh.empty();   // (the bf16 shadow of x: both operands of this product are stored as bf16 then)
        const float* xop = shadow ? ws + t.xh[l] : xin;
        g.a = WnRowMap{xop, L * (long long)R, R, t0 - d}; g.a1 = WnRowMap{xop, L * (long long)R, R, t0}; g.ka_split = R;
        g.a_bf16 = shadow ? 1 : 0;
        g.b = WnRowMap{dfg, rows * 2 * D, 2 * D, 0}; g.b_bf16 = t.bf16 ? 1 : 0;
        g.Ka = 2 * R; g.Nb = 2 * D; g.c = grads + h->fw_off_fg + (size_t)l * 2 * R * 2 * D; g.ldc = 2 * D;
        g.M = M; g.rows_per_batch = (int)rows;
        g.a_skip_lo = (int)t.zlo[l];   // tap 0 on the rows where the forward read a pad zero: no contribution
#if WN_TN_MERGE_TAPS
        if (R % 128 == 0) {
            wn_launch_tn(sd, g, t.bf16);
        } else
#endif
        for (int tap = 0; tap < 2; ++tap) {
            WnGemmTnArgs g1 = g;
            g1.a = tap ? g.a1 : g.a; g1.ka_split = 0; g1.Ka = R; g1.c = g.c + (size_t)tap * R * 2 * D;
            if (tap) g1.a_skip_lo = 0;
            wn_launch_tn(sd, g1, t.bf16);
        }
        if (pl.has_bias) wn_launch_colsum(sd, WnRowMap{dfg, rows * 2 * D, 2 * D, 0}, M, (int)rows, 2 * D, grads + h->fw_off_bfg + (size_t)l * 2 * D, t.bf16);
        if (two) fg_read[l] = signal(sd);
        // dx_l on ITS rows [lo, L) (the last need[l] time steps; lo = t0 - sh with sh = d, or less where the clip is so short that layer
        // l's input starts later than t0 - d) in ONE product over two row-shifted views of dfg:
        //     dx_l(t) = dx'(t) [t >= t0]  +  dfg(t) . Wfg(tap 1) [t >= t0]  +  dfg(t + d) . Wfg(tap 0) [t < L - d]
        // K = 4D: columns 0..2D-1 take dfg(t) against tap 1's rows, columns 2D.. take dfg(t + d) against tap 0's.  The views' row
        // windows (a_skip_*) read as zero where a shift runs off dfg, so nothing is cleared first and dx_l is written exactly once
        // (round 2: a memset of dx and two read-modify-write products per layer).  (Positions before lo do not exist: what the forward
        // read there were the reference's pad zeros, which have no gradient.)
        const long long rows_l = t.need[l], sh = rows_l - rows;   // 0 <= sh <= d
        if (l + 1 < NL) wait_for(st, res_read[l + 1]);   // (dx_l goes into the buffer dWres of layer l + 1 read dx_{l+2} from)
        memset(&a, 0, sizeof(a));
        a.a0 = WnRowMap{dfg, rows * 2 * D, 2 * D, -sh};      // dfg(t):     row (t - t0) = rem - sh, valid from rem = sh
        a.a1 = WnRowMap{dfg, rows * 2 * D, 2 * D, d - sh};   // dfg(t + d): row rem + d - sh, valid while rem < rows_l - d
        a.a_skip_lo[0] = (int)sh; a.a_skip_hi[1] = (int)d; a.a_bf16 = t.bf16 ? 1 : 0;
        a.k_split = 2 * D; a.K = 4 * D; a.bt = ws + t.fgb1 + (size_t)l * 2 * D * R; a.bt1 = ws + t.fgb0 + (size_t)l * 2 * D * R; a.N = R;
        if (has_res) { a.cin = WnRowMap{dxn, L * (long long)R, R, t0 - sh}; a.cin_skip_lo = (int)sh; }
        a.c = WnRowMap{dxc, L * (long long)R, R, t0 - sh};
        a.M = N * rows_l; a.rows_per_batch = (int)rows_l;
        {
            const unsigned short* w = bw ? bw + h->fw_off_fg + (size_t)l * 2 * R * 2 * D : nullptr;  // native [2R][2D]: rows 0..R-1 tap 0, R.. tap 1
            // bf16 step, the 128 / 128 shape: this product and layer l - 1's gate-derivative product (same rows, dx_l as its A operand) are
            // ONE launch -- dx_l is handed over in LDS (wn_bwd_layer_bf16).  Layer l - 1's waits move in front of it.
            have_dfg = false;
            if (t.bf16 && l >= 1 && w) {
                WnGemmArgs ag;
                gate_bwd_args(l - 1, dxc, ag);
                if (ag.M == a.M && ag.a0.t0 == a.c.t0 && R == 128 && D == 128 && wn_fused_layer_enabled()) {
                    gate_bwd_waits(l - 1);
                    have_dfg = wn_launch_bwd_layer(st, a, w + (size_t)R * 2 * D, w, 2 * D, ag, bw + h->fw_off_res + (size_t)(l - 1) * D * R);
                }
            }
            if (!have_dfg) wn_launch_nn(st, WN_EPI_PLAIN, a, w ? w + (size_t)R * 2 * D : nullptr, w, 2 * D);
        }
        wait_for(sd, signal(st));   // dx_l is complete: dWres of layer l - 1 may read it (fused: [dF|dG] of layer l - 1 as well)
        if (skip_main && gi == 0) skip_weight_grads(st, l / t.G);   // (the chain has left block l / G: its skip weight gradient behind it, on this stream)
        float* tmp = dxn; dxn = dxc; dxc = tmp;
    }
    // ---- start_conv: dstart^T [C][R] = onehot(indices)^T . dx_0 over the rows dx_0 exists on (the last need[0] time steps) -- in FRONT of the join: it
    // only needs dx_0, and runs next to what the side stream still has to do (the first layer's weight gradients)
    {
        const long long r0 = t.need[0], f0 = L - r0;
        memset(&g, 0, sizeof(g));
        g.a = WnRowMap{nullptr, L, 1, f0}; g.a_idx = reinterpret_cast<const int32_t*>(ws + t.idx);
        g.b = WnRowMap{dxn, L * (long long)R, R, f0};
        g.Ka = C; g.Nb = R; g.c = grads + h->fw_off_start_t; g.ldc = R; g.M = N * r0; g.rows_per_batch = (int)r0;
        wn_launch_tn(st, g, t.bf16);
        if (pl.has_bias) wn_launch_colsum(st, WnRowMap{dxn, L * (long long)R, R, f0}, N * r0, (int)r0, R, grads + h->fw_off_start_b);
    }
    wait_for(st, signal(sd));   // join: every weight gradient is complete before the caller's stream goes on
    return rt_hip(hipGetLastError(), "wn_train_backward launches");
}

extern "C" int wn_train_loss(wn_handle* h, const float* logits, const int64_t* targets, int64_t M, float* loss, float* dlogits, void* hip_stream) {
    g_err[0] = 0;
    if (!h || !logits || !targets || !loss) return wn_fail(WN_E_BADARG, "wn_train_loss: NULL argument");
    if (!h->chains.empty()) return wn_train_loss(h->chains[0], logits, targets, M, loss, dlogits, hip_stream);
    if (M < 1) return wn_fail(WN_E_BADARG, "wn_train_loss: M must be >= 1");
    if (h->plan.C != 256) return wn_fail(WN_E_UNSUPPORTED, "wn_train_loss: classes = %d (the fused loss is written for 256)", h->plan.C);
    { int rc = rt_hip(hipSetDevice(h->cfg.device_id), "hipSetDevice"); if (rc) return rc; }
    if (h->xent_rows < (size_t)M) {
        (void)hipDeviceSynchronize();
        rt_free(h->d_xent);
        h->d_xent = (float*)rt_malloc((size_t)M * 4);
        h->xent_rows = h->d_xent ? (size_t)M : 0;
        if (!h->d_xent) return wn_fail(WN_E_NOMEM, "wn_train_loss: %lld row losses", (long long)M);
    }
    hipStream_t st = (hipStream_t)hip_stream;
    hipLaunchKernelGGL(wn_xent_rows, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, logits, reinterpret_cast<const long long*>(targets), (long long)M,
                       (float)(1.0 / (double)M), h->d_xent, dlogits);
    hipLaunchKernelGGL(wn_xent_reduce, dim3(1), dim3(1024), 0, st, h->d_xent, (long long)M, 1.0 / (double)M, loss);
    return rt_hip(hipGetLastError(), "wn_train_loss launches");
}
