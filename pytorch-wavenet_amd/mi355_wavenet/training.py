"""Native training step: model(x) and loss.backward() of the dilated-conv stack as fp32 matrix-core GEMMs.

Replaces, for one-hot inputs on an MI355X, the autograd graph the reference builds in WaveNetModel.forward
(wavenet_model.py:125-196) when WavenetTrainer.train (wavenet_training.py:58-107) calls ``output = self.model(x)`` and
``loss.backward()``.  The parameters stay ordinary ``nn.Parameter``s in the reference's Conv1d layouts (so optimisers,
``clip_grad_norm`` and ``torch.save(model)`` keep working): every step they are packed into the flat GEMM layout of
``wn_train_layout`` (include/wn_abi.h) with a handful of torch view ops, and the flat gradient that wn_train_backward
returns is unpacked the same way.  The loss (F.cross_entropy on the returned logits, wavenet_training.py:69-70) is one fused pass
over the logits as well (wn_train_loss: value and gradient together), behind ``cross_entropy`` below.
"""
import ctypes

import torch

from . import _abi


class StackRunner:
    """Owns one engine handle (for the plan / layout / workspace) and runs the packed forward / backward."""

    def __init__(self, engine):
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
        grads = torch.empty(self.total, dtype=torch.float32, device=self.device)
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


class StackFunction(torch.autograd.Function):
    """logits = stack(indices; parameters).  Inputs after ``output_length``: the model's parameters as flat lists, per layer,
    in the order given by ``names`` (a tuple of (key, count) pairs); gradients come back in the same order."""

    @staticmethod
    def forward(ctx, runner, idx, output_length, names, *tensors):
        stacked, pos = {}, 0
        for key, count in names:
            ts = tensors[pos:pos + count]
            pos += count
            stacked[key] = ts[0] if count == 1 and key in ("start_w", "start_b", "end1_w", "end1_b", "end2_w", "end2_b") else torch.stack(ts)
        flat = runner.pack(stacked)
        logits = runner.forward(flat, idx, output_length)
        ctx.runner, ctx.flat, ctx.names, ctx.ticket = runner, flat, names, runner.ticket
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        r = ctx.runner
        if ctx.ticket != r.ticket:
            raise RuntimeError("native WaveNet backward: another forward ran on this model since the one being differentiated "
                               "(the saved activations live in one workspace per model)")
        g = r.unpack(r.backward(ctx.flat, dlogits))
        out = []
        for key, count in ctx.names:
            if count == 1 and key in ("start_w", "start_b", "end1_w", "end1_b", "end2_w", "end2_b"):
                out.append(g[key])
            else:
                rows = list(g[key].unbind(0))
                if key in ("res_w", "res_b"):
                    rows[-1] = None  # the last layer's residual conv never reaches the loss (also upstream: its .grad stays None)
                out.extend(rows)
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
