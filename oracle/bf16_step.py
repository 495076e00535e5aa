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
  * the skip convs' share of dz -- dzg = rb(dskip) . rb(Wskip), one product per block of layers in the kernels -- is STORED as bf16 (round 5,
    WN_DZG_BF16 in csrc/wn_forward.h); the gate derivative adds it to the residual conv's share (fp32, never stored) in fp32;
  * the residual stream x, the skip sum, the pre-activations, biases, the loss and every bias gradient stay fp32; start_conv's gradient
    is an exact gather-sum (one-hot rows);
  * full-length clips only (L >= receptive_field + output_length - 1: no returned position sees the reference's pad zeros).

WHAT THIS ORACLE CAN AND CANNOT PIN (measured in round 5, profiles/r05_bf16_step_oracle.txt).  A rounding to bf16 is a discontinuity: two
evaluations of this very model that differ only in the ORDER of their fp32 accumulations (1e-7 of a value) round a few elements in 10^5 to
the other neighbour -- a difference of a whole bf16 ulp (0.4 %) of that element.  Downstream of such a flip the two evaluations differ by
1e-4, which flips a percent of the next layer's roundings, and after four or five layers their rounding errors are statistically
INDEPENDENT: the difference between the two evaluations is as large as the difference of either from the unrounded reference.  On a
one-layer model the product reproduces this restatement EXACTLY (every logit to 1e-6, barring a single flip); on the 50-layer cfg3 stack
three accumulation orders of this restatement differ from each other by 0.047 of a 3.6 logit scale, and each from the fp32 reference by
0.055-0.061 -- the product by 0.058.  So: shallow models pin the rounding POINTS (which operand is rounded where: any mistake there shows
as a systematic difference at depth one); deep models can only pin the rounding NOISE LEVEL -- the product's deviation from the reference's
fp32 result must not exceed what this restatement's own evaluation orders show.  A bound of "1e-3 of scale" at 50 layers is not a
property bf16 arithmetic has.

Time-major formulation (rows = time steps): layer l reads x_l on its last rows_l steps, x_l(t - d) and x_l(t) are two row windows of the
same matrix (the reference's dilate() copies are a re-indexing of exactly these rows, tests/test_plan_host.py pins the geometry).
"""
import numpy as np
import torch


def rb(t):
    """round to bf16 and back (round-to-nearest-even: torch's conversion, the matrix cores' v_cvt_pk_bf16_f32)"""
    return t.to(torch.bfloat16).to(t.dtype)


# How the products ACCUMULATE is not part of the rounding model -- and at depth it matters: see step().  "exact": float64 accumulation (the
# fixture's order); "f32": float32 accumulation in the CPU GEMM's order; "f32perm": float32 accumulation over a permuted K axis.
ACCUMULATE = "exact"


def _dot(a, b):   # a (M, K) . b (K, N)
    if ACCUMULATE == "exact":
        return (a.double() @ b.double()).float()
    if ACCUMULATE == "f32perm":
        idx = torch.randperm(a.shape[1], generator=torch.Generator().manual_seed(a.shape[1] + 7))
        return a[:, idx] @ b[idx, :]
    return a @ b


class _MM(torch.autograd.Function):
    """Y = A . W^T with both operands rounded to bf16 when `rnd`; float32 result; accumulation as ACCUMULATE says (default: exact).
    store_da: the activation gradient dA this product hands back is STORED as bf16 (the skip convs: their share of dz, `dzg`)."""

    @staticmethod
    def forward(ctx, A, W, rnd, store_da=False):
        a, w = (rb(A), rb(W)) if rnd else (A, W)
        ctx.save_for_backward(a, w)
        ctx.rnd = rnd
        ctx.store_da = bool(store_da) and rnd
        return _dot(a, w.t())

    @staticmethod
    def backward(ctx, dY):
        a, w = ctx.saved_tensors
        dy = rb(dY) if ctx.rnd else dY
        dA = _dot(dy, w)
        return (rb(dA) if ctx.store_da else dA), _dot(dy.t(), a), None, None


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
        s = _MM.apply(zs, P["skip_convs.%d.weight" % l][:, :, 0], rnd, True)             # :154-162 (only the returned positions matter; dzg stored as bf16)
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
