"""TEST INFRASTRUCTURE (oracle/): the reference's training step -- model(x) -> F.cross_entropy -> backward
(/root/reference/wavenet_model.py:125-196, wavenet_training.py:64-72) -- restated in torch with its matrix operands ROUNDED TO
BF16 exactly where the product's bf16 step rounds them (pytorch-wavenet_amd/csrc/wn_train.inl), everything else in float32 /
float64 like the reference.  Only tests/ and tests/golden/make_golden.py import this; the product never does.

Why it exists: the bf16 training step is the one the bench leads with for BASELINE configs[4], and until round 5 its only check was
against the product's own fp32 step on an 8-layer model.  This restatement gives it an oracle at depth: with ``round_operands=False``
it is pinned to the REAL reference (tests/golden/make_golden.py --v5 asserts logits and every parameter gradient against the imported
reference on the 50-layer cfg3 stack before it writes anything); with ``round_operands=True`` the same graph carries the product's
rounding points and nothing else:

  * every matrix product takes both operands rounded to bf16 (round-to-nearest-even), accumulates exactly (float64 here; the kernels
    accumulate in fp32 on the matrix cores -- the difference is 1e-7 of scale, the roundings are 4e-3 of an element) -- forward products AND
    the backward's: dA = rb(dY) . rb(W), dW = rb(A)^T . rb(dY);
  * z = tanh(F) * sigmoid(G) is STORED as bf16 (consumers see the rounded value; the gradient passes straight through);
  * tanh(F) and sigmoid(G) are saved for the backward as bf16: the gate derivative dF = dz * G * (1 - T^2), dG = dz * T * G * (1 - G) is
    evaluated on the rounded pair, and [dF | dG] is stored as bf16 (the bias gradient of the filter / gate convs sums the stored values);
  * the residual stream x, the skip sum, the pre-activations, biases, the loss and every bias gradient stay fp32; start_conv's gradient
    is an exact gather-sum (one-hot rows);
  * full-length clips only (L >= receptive_field + output_length - 1: no returned position sees the reference's pad zeros).

Time-major formulation (rows = time steps): layer l reads x_l on its last rows_l steps, x_l(t - d) and x_l(t) are two row windows of the
same matrix (the reference's dilate() copies are a re-indexing of exactly these rows, tests/test_plan_host.py pins the geometry).
"""
import numpy as np
import torch


def rb(t):
    """round to bf16 and back (round-to-nearest-even: torch's conversion, the matrix cores' v_cvt_pk_bf16_f32)"""
    return t.to(torch.bfloat16).to(t.dtype)


class _MM(torch.autograd.Function):
    """Y = A . W^T with both operands rounded to bf16 when `rnd`, exact (float64) accumulation, float32 result."""

    @staticmethod
    def forward(ctx, A, W, rnd):
        a, w = (rb(A), rb(W)) if rnd else (A, W)
        ctx.save_for_backward(a, w)
        ctx.rnd = rnd
        return (a.double() @ w.double().t()).float()

    @staticmethod
    def backward(ctx, dY):
        a, w = ctx.saved_tensors
        dy = rb(dY) if ctx.rnd else dY
        dA = (dy.double() @ w.double()).float()
        dW = (dy.double().t() @ a.double()).float()
        return dA, dW, None


class _Gate(torch.autograd.Function):
    """z = tanh(F) * sigmoid(G); bf16 step: z stored as bf16, the pair saved as bf16, [dF | dG] stored as bf16."""

    @staticmethod
    def forward(ctx, F, G, rnd):
        T, S = torch.tanh(F), torch.sigmoid(G)
        z = T * S
        if rnd:
            z, T, S = rb(z), rb(T), rb(S)
        ctx.save_for_backward(T, S)
        ctx.rnd = rnd
        return z

    @staticmethod
    def backward(ctx, dz):
        T, S = ctx.saved_tensors
        dF = dz * S * (1.0 - T * T)
        dG = dz * T * S * (1.0 - S)
        if ctx.rnd:
            dF, dG = rb(dF), rb(dG)
        return dF, dG, None


def step(cfg, weights, ids, target, output_length, round_operands=True):
    """One forward -> cross_entropy -> backward of the reference's model on class indices ids (N, L) and targets (N * output_length,).

    weights: name -> array in the reference's Conv1d layout (mi355_wavenet.synth.init_weights).  Returns (logits (N * output_length,
    classes) float32 numpy, loss float, gradients: name -> float32 numpy in the same layout)."""
    rnd = bool(round_operands)
    NL = cfg["layers"] * cfg["blocks"]
    k = cfg.get("kernel_size", 2)
    assert k == 2, "the matrix-core step is written for kernel_size 2"
    bias = bool(cfg.get("bias", False))
    P = {n: torch.tensor(np.asarray(v, dtype=np.float32), requires_grad=True) for n, v in weights.items()}
    ids = torch.as_tensor(np.asarray(ids), dtype=torch.long)
    N, L = ids.shape
    rf = 1 + cfg["blocks"] * (2 ** cfg["layers"] - 1)
    assert L >= rf + output_length - 1, "full-length clips only"
    # start_conv on a one-hot input = a row gather of start_conv^T (exact)              wavenet_model.py:127
    x = P["start_conv.weight"][:, :, 0].t()[ids]                                        # (N, L, R)
    if bias:
        x = x + P["start_conv.bias"]
    skip = None
    for l in range(NL):
        d = 2 ** (l % cfg["layers"])
        wf, wg = P["filter_convs.%d.weight" % l], P["gate_convs.%d.weight" % l]         # (D, R, 2): tap 0 multiplies x(t - d), tap 1 x(t)
        a0, a1 = x[:, :-d, :], x[:, d:, :]
        rows = a1.shape[1]
        A = torch.cat([a0, a1], dim=2).reshape(N * rows, -1)                             # [x(t - d) | x(t)]
        Wf = torch.cat([wf[:, :, 0], wf[:, :, 1]], dim=1)
        Wg = torch.cat([wg[:, :, 0], wg[:, :, 1]], dim=1)
        Fp, Gp = _MM.apply(A, Wf, rnd), _MM.apply(A, Wg, rnd)                            # wavenet_model.py:147-151
        if bias:
            Fp, Gp = Fp + P["filter_convs.%d.bias" % l], Gp + P["gate_convs.%d.bias" % l]
        z = _Gate.apply(Fp, Gp, rnd)                                                     # (N * rows, D)
        zs = z.reshape(N, rows, -1)[:, -output_length:, :].reshape(N * output_length, -1)
        s = _MM.apply(zs, P["skip_convs.%d.weight" % l][:, :, 0], rnd)                   # :154-162 (only the returned positions matter)
        if bias:
            s = s + P["skip_convs.%d.bias" % l]
        skip = s if skip is None else skip + s
        if l < NL - 1:
            xn = _MM.apply(z, P["residual_convs.%d.weight" % l][:, :, 0], rnd)           # :164-165
            if bias:
                xn = xn + P["residual_convs.%d.bias" % l]
            x = xn.reshape(N, rows, -1) + a1
    e = torch.relu(_MM.apply(torch.relu(skip), P["end_conv_1.weight"][:, :, 0], rnd) + P["end_conv_1.bias"])   # :167-169
    logits = _MM.apply(e, P["end_conv_2.weight"][:, :, 0], rnd) + P["end_conv_2.bias"]
    loss = torch.nn.functional.cross_entropy(logits, torch.as_tensor(np.asarray(target), dtype=torch.long))     # wavenet_training.py:69
    loss.backward()
    grads = {n: (p.grad.numpy() if p.grad is not None else np.zeros(tuple(p.shape), dtype=np.float32)) for n, p in P.items()}
    return logits.detach().numpy(), float(loss.detach()), grads
