"""Native training step: model(x) and loss.backward() of the dilated-conv stack as fp32 matrix-core GEMMs.

Replaces, for one-hot inputs on an MI355X, the autograd graph the reference builds in WaveNetModel.forward
(wavenet_model.py:125-196) when WavenetTrainer.train (wavenet_training.py:58-107) calls ``output = self.model(x)`` and
``loss.backward()``.  The parameters stay ordinary ``nn.Parameter``s in the reference's Conv1d layouts (so optimisers,
``clip_grad_norm`` and ``torch.save(model)`` keep working): every step they are packed into the flat GEMM layout of
``wn_train_layout`` (include/wn_abi.h) by the engine itself (wn_train_pack: the tensors' addresses in, a handful of launches -- round 5
did it with ~150 torch stack / view / copy launches per step), and the flat gradient that wn_train_backward returns goes back into ONE
buffer in the parameters' own layouts (wn_train_unpack_grads), of which every parameter's gradient is a view.  ``pack`` / ``unpack``
below are the same maps in torch ops: the tests' second opinion on the native ones, not on the step's path.  The loss (F.cross_entropy on the returned logits, wavenet_training.py:69-70) is one fused pass
over the logits as well (wn_train_loss: value and gradient together), behind ``cross_entropy`` below.
"""
import ctypes

import torch

from . import _abi


class StackRunner:
    """Owns one engine handle (for the plan / layout / workspace) and runs the packed forward / backward."""

    def __init__(self, engine, model_shape=None):
        """model_shape: (R, D, S, E) of the MODEL when the engine was created for a zero-padded channel shape (wavenet_model.py:
        _native_train_forward pads channel counts that are not multiples of 32 up to multiples of 64): the parameters then travel through
        zero-filled tensors of the engine's shape (pad_tensors) and only their own block of the gradients comes back (crop_grads)."""
        self.eng = engine
        lay = _abi.wn_train_layout()
        engine.lib.check(engine.lib.dll.wn_train_get_layout(engine._h, ctypes.byref(lay)))
        self.total = int(lay.total)
        self.off = {n: int(getattr(lay, n)) for n in _abi.TRAIN_SECTIONS}
        c = engine.cfg
        self.NL = c["layers"] * c["blocks"]
        self.R, self.D, self.S, self.E, self.C = (c["residual_channels"], c["dilation_channels"], c["skip_channels"],
                                                  c["end_channels"], c["classes"])
        self.bias = bool(c.get("bias", False))
        self.ticket = 0
        self.device = engine.mem.device
        self._flat = None        # the packed parameters of the step in flight (one buffer per model: stable addresses)
        self._gflat = None       # the flat gradient wn_train_backward writes
        self._ptr_key = None     # the parameter tensors the cached pointer tables below were built for
        self._ptr_tabs = None
        self.padded = model_shape is not None and tuple(model_shape) != (self.R, self.D, self.S, self.E)
        self._pad_bufs = {}      # (key, index) -> the zero-filled tensor of the engine's shape a parameter is copied into every step

    # ---- zero padding of the channel shape (training on channel counts that are not multiples of 32) -----------------------------
    # Padded channels carry zeros through the whole step: zero start_conv rows and zero residual rows keep the extra residual channels at 0,
    # zero filter / gate rows give tanh(0) * sigmoid(0) = 0, extra skip / end channels are relu(0) = 0 against zero weights -- and no gradient
    # reaches a real weight through them, nor a padded weight at all (its activation or its incoming gradient is zero).  The logits and the real
    # parameters' gradients are those of the model's own shape; the tests compare them with torch autograd.
    def padded_shape(self, key):
        R, D, S, E, C = self.R, self.D, self.S, self.E, self.C
        return {"start_w": (R, C, 1), "start_b": (R,), "filter_w": (D, R, 2), "gate_w": (D, R, 2), "filter_b": (D,), "gate_b": (D,),
                "res_w": (R, D, 1), "res_b": (R,), "skip_w": (S, D, 1), "skip_b": (S,), "end1_w": (E, S, 1), "end1_b": (E,),
                "end2_w": (C, E, 1), "end2_b": (C,)}[key]

    def pad_tensors(self, by_key):
        """{key: [parameter tensors]} -> the same parameters inside zero-filled tensors of the engine's channel shape (kept from step to step:
        only a parameter's own block is ever written)."""
        out = {}
        for key, ts in by_key.items():
            shape = self.padded_shape(key)
            row = []
            for i, t in enumerate(ts):
                buf = self._pad_bufs.get((key, i))
                if buf is None or buf.device != t.device:
                    buf = torch.zeros(shape, dtype=torch.float32, device=t.device)
                    self._pad_bufs[(key, i)] = buf
                buf[tuple(slice(0, n) for n in t.shape)].copy_(t)
                row.append(buf)
            out[key] = row
        return out

    @staticmethod
    def crop_grads(grads, real):
        """Gradients of the padded tensors -> the parameters' own blocks (contiguous copies; None stays None)."""
        return {key: [None if g is None else g[tuple(slice(0, n) for n in t.shape)].contiguous() for g, t in zip(gs, real[key])] for key, gs in grads.items()}

    # ---- layout conversion (reference Conv1d layouts <-> wn_train_layout) ------------------------------------------
    def sizes(self):
        NL, R, D, S, E, C = self.NL, self.R, self.D, self.S, self.E, self.C
        return {"fg": NL * 2 * R * 2 * D, "bfg": NL * 2 * D, "res": NL * D * R, "bres": NL * R, "skip": NL * D * S,
                "bskip": NL * S, "bskip_total": S, "w1": S * E, "b1": E, "w2": E * C, "b2": C, "start_t": C * R, "start_b": R}

    def pack(self, p):
        """p: dict with stacked tensors start_w (R,C,1), filter_w/gate_w (NL,D,R,2), res_w (NL,R,D,1), skip_w (NL,S,D,1),
        end1_w (E,S,1), end2_w (C,E,1), end1_b, end2_b and, with bias, start_b, filter_b, gate_b (NL,D), res_b, skip_b."""
        NL, R, D, S, E, C = self.NL, self.R, self.D, self.S, self.E, self.C
        flat = torch.zeros(self.total, dtype=torch.float32, device=self.device)
        sz, off = self.sizes(), self.off

        def put(name, t):
            flat[off[name]:off[name] + sz[name]] = t.reshape(-1)

        fg = torch.stack([p["filter_w"], p["gate_w"]], dim=1)                       # (NL, gate, D, R, tap)
        fg = fg.reshape(NL, 2, D // 32, 32, R, 2).permute(0, 5, 4, 2, 1, 3)         # (NL, tap, R, grp, gate, c32)
        put("fg", fg)
        put("res", p["res_w"].reshape(NL, R, D).transpose(1, 2))
        put("skip", p["skip_w"].reshape(NL, S, D).transpose(1, 2))
        put("w1", p["end1_w"].reshape(E, S).t())
        put("b1", p["end1_b"])
        put("w2", p["end2_w"].reshape(C, E).t())
        put("b2", p["end2_b"])
        put("start_t", p["start_w"].reshape(R, C).t())
        if self.bias:
            bfg = torch.stack([p["filter_b"], p["gate_b"]], dim=1).reshape(NL, 2, D // 32, 32).permute(0, 2, 1, 3)
            put("bfg", bfg)
            put("bres", p["res_b"])
            put("bskip", p["skip_b"])
            put("start_b", p["start_b"])
        return flat

    def unpack(self, flat):
        """Inverse of pack() (used for gradients): flat -> dict of tensors in the reference layouts."""
        NL, R, D, S, E, C = self.NL, self.R, self.D, self.S, self.E, self.C
        sz, off = self.sizes(), self.off

        def get(name):
            return flat[off[name]:off[name] + sz[name]]

        fg = get("fg").reshape(NL, 2, R, D // 32, 2, 32).permute(0, 4, 3, 5, 2, 1).reshape(NL, 2, D, R, 2)  # (NL, gate, D, R, tap)
        out = {"filter_w": fg[:, 0].contiguous(), "gate_w": fg[:, 1].contiguous(),
               "res_w": get("res").reshape(NL, D, R).transpose(1, 2).reshape(NL, R, D, 1).contiguous(),
               "skip_w": get("skip").reshape(NL, D, S).transpose(1, 2).reshape(NL, S, D, 1).contiguous(),
               "end1_w": get("w1").reshape(S, E).t().reshape(E, S, 1).contiguous(), "end1_b": get("b1").clone(),
               "end2_w": get("w2").reshape(E, C).t().reshape(C, E, 1).contiguous(), "end2_b": get("b2").clone(),
               "start_w": get("start_t").reshape(C, R).t().reshape(R, C, 1).contiguous()}
        if self.bias:
            bfg = get("bfg").reshape(NL, D // 32, 2, 32).permute(0, 2, 1, 3).reshape(NL, 2, D)
            out.update({"filter_b": bfg[:, 0].contiguous(), "gate_b": bfg[:, 1].contiguous(), "res_b": get("bres").reshape(NL, R).clone(),
                        "skip_b": get("bskip").reshape(NL, S).clone(), "start_b": get("start_b").clone()})
        return out

    # ---- native layout conversion: the tensors themselves in, a view per gradient out ---------------------------------------
    @staticmethod
    def _tensor_table(by_key, NL):
        """wn_train_tensors over {key: [tensors or None]} (PARAM_ORDER keys); returns (struct, keep-alive list)."""
        t = _abi.wn_train_tensors()
        t.n_layers, t.reserved = NL, 0
        keep = []
        names = {"filter_w": "filter_w", "gate_w": "gate_w", "res_w": "res_w", "skip_w": "skip_w", "filter_b": "filter_b", "gate_b": "gate_b",
                 "res_b": "res_b", "skip_b": "skip_b"}
        for key, field in names.items():
            ts = by_key.get(key)
            if ts is None:
                setattr(t, field, None)
                continue
            arr = (ctypes.c_void_p * NL)(*[(x.data_ptr() if x is not None else None) for x in ts])
            keep.append(arr)
            setattr(t, field, ctypes.cast(arr, ctypes.c_void_p))
        for key in ("start_w", "start_b", "end1_w", "end1_b", "end2_w", "end2_b"):
            ts = by_key.get(key)
            setattr(t, key, ts[0].data_ptr() if ts is not None and ts[0] is not None else None)
        return t, keep

    def pack_native(self, by_key):
        """by_key: {key: list of the parameter tensors} -> the flat GEMM layout (one launch per 72 pieces: 5 at config 5)."""
        tensors = [x for k in PARAM_ORDER for x in (by_key.get(k) or [])]
        for x in tensors:
            if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()):
                raise TypeError("the native training step takes contiguous fp32 parameters on the MI355X (got %s %s on %s)" % (x.dtype, tuple(x.shape), x.device))
        key = tuple(x.data_ptr() for x in tensors)
        if key != self._ptr_key:
            self._ptr_tabs = self._tensor_table(by_key, self.NL)
            self._ptr_key = key
        if self._flat is None:
            self._flat = torch.empty(self.total, dtype=torch.float32, device=self.device)
        e = self.eng
        e.lib.check(e.lib.dll.wn_train_pack(e._h, ctypes.byref(self._ptr_tabs[0]), self._flat.data_ptr(), e.mem.stream()))
        return self._flat

    def unpack_native(self, gflat, by_key):
        """The flat gradient -> {key: [gradient tensors]}: views of ONE new buffer in the parameters' own layouts (the last layer's residual conv gets
        None: it never reaches the loss, and upstream its .grad stays None)."""
        shapes = {k: [tuple(x.shape) for x in ts] for k, ts in by_key.items()}
        total = sum(int(torch.Size(sh).numel()) for shs in shapes.values() for sh in shs)
        buf = torch.empty(total, dtype=torch.float32, device=self.device)
        out, pos = {}, 0
        for k in PARAM_ORDER:
            if k not in shapes:
                continue
            views = []
            for i, sh in enumerate(shapes[k]):
                n = int(torch.Size(sh).numel())
                views.append(None if (k in ("res_w", "res_b") and i == self.NL - 1) else buf[pos:pos + n].view(sh))
                pos += n
            out[k] = views
        t, keep = self._tensor_table(out, self.NL)
        e = self.eng
        e.lib.check(e.lib.dll.wn_train_unpack_grads(e._h, gflat.data_ptr(), ctypes.byref(t), e.mem.stream()))
        del keep
        return out

    def set_deterministic(self, on):
        """Bit-reproducible weight / bias gradients (wn_train_set_deterministic: partial tiles + an ordered reduction instead of fp32 atomics)."""
        e = self.eng
        e.lib.check(e.lib.dll.wn_train_set_deterministic(e._h, 1 if on else 0))

    # ---- the two launches ----------------------------------------------------------------------------------------------
    def forward(self, flat, idx, output_length):
        idx = idx.to(self.device, torch.int32).contiguous()
        n, l = idx.shape
        logits = torch.empty(n * output_length, self.C, dtype=torch.float32, device=self.device)
        e = self.eng
        e.lib.check(e.lib.dll.wn_train_forward(e._h, flat.data_ptr(), idx.data_ptr(), n, l, int(output_length), logits.data_ptr(),
                                               e.mem.stream()))
        self.ticket += 1
        return logits

    def backward(self, flat, dlogits):
        dlogits = dlogits.to(torch.float32).contiguous()
        if self._gflat is None:
            self._gflat = torch.empty(self.total, dtype=torch.float32, device=self.device)
        grads = self._gflat
        e = self.eng
        e.lib.check(e.lib.dll.wn_train_backward(e._h, flat.data_ptr(), dlogits.data_ptr(), grads.data_ptr(), e.mem.stream()))
        return grads

    def export_params(self):
        flat = torch.empty(self.total, dtype=torch.float32, device=self.device)
        e = self.eng
        e.lib.check(e.lib.dll.wn_train_export_params(e._h, flat.data_ptr(), e.mem.stream()))
        return flat


PARAM_ORDER = ("start_w", "filter_w", "gate_w", "res_w", "skip_w", "end1_w", "end1_b", "end2_w", "end2_b",
               "start_b", "filter_b", "gate_b", "res_b", "skip_b")


SINGLE_KEYS = ("start_w", "start_b", "end1_w", "end1_b", "end2_w", "end2_b")


class StackFunction(torch.autograd.Function):
    """logits = stack(indices; parameters).  Inputs after ``output_length``: the model's parameters as flat lists, per layer,
    in the order given by ``names`` (a tuple of (key, count) pairs); gradients come back in the same order."""

    @staticmethod
    def forward(ctx, runner, idx, output_length, names, *tensors):
        by_key, pos = {}, 0
        for key, count in names:
            by_key[key] = list(tensors[pos:pos + count])
            pos += count
        real = {k: [x.detach() for x in v] for k, v in by_key.items()}
        packed = runner.pad_tensors(real) if runner.padded else real   # (a zero-padded channel shape: see StackRunner.pad_tensors)
        flat = runner.pack_native(packed)
        logits = runner.forward(flat, idx, output_length)
        ctx.runner, ctx.flat, ctx.names, ctx.ticket = runner, flat, names, runner.ticket
        ctx.shapes = packed                                  # (shapes only: detached aliases / the runner's own padded tensors, no copies)
        ctx.real = real if runner.padded else None
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        r = ctx.runner
        if ctx.ticket != r.ticket:
            raise RuntimeError("native WaveNet backward: another forward ran on this model since the one being differentiated "
                               "(the saved activations live in one workspace per model)")
        g = r.unpack_native(r.backward(ctx.flat, dlogits), ctx.shapes)
        if ctx.real is not None:
            g = r.crop_grads(g, ctx.real)
        out = []
        for key, count in ctx.names:
            out.extend(g[key])
        return (None, None, None, None, *out)


class XentFunction(torch.autograd.Function):
    """loss = F.cross_entropy(logits, target) on the engine: one pass over the logits yields the mean loss and dLoss/dlogits
    (torch: log_softmax, nll_loss and their two backward kernels)."""

    @staticmethod
    def forward(ctx, runner, logits, target):
        logits = logits.contiguous()
        target = target.to(device=logits.device, dtype=torch.int64).contiguous()
        m = logits.size(0)
        loss = torch.empty((), dtype=torch.float32, device=logits.device)
        dl = torch.empty_like(logits) if ctx.needs_input_grad[1] else None
        e = runner.eng
        e.lib.check(e.lib.dll.wn_train_loss(e._h, logits.data_ptr(), target.data_ptr(), m, loss.data_ptr(),
                                            dl.data_ptr() if dl is not None else None, e.mem.stream()))
        ctx.dl = dl
        return loss

    @staticmethod
    def backward(ctx, g):
        dl, ctx.dl = ctx.dl, None
        return None, (dl.mul_(g) if dl is not None else None), None


def cross_entropy(runner, logits, target):
    """Drop-in for ``F.cross_entropy(logits, target)`` (mean reduction, class-index targets) on fp32 CUDA logits of 256 classes."""
    return XentFunction.apply(runner, logits, target.reshape(-1))
