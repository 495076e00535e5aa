"""``FusedAdam``: torch.optim.Adam's step -- and, fused in front of it, ``torch.nn.utils.clip_grad_norm_`` -- as the engine's native
optimiser kernels (C ABI ``wn_adam_step``, csrc/wn_optim.h): the optimiser half of the reference's training step
(/root/reference/wavenet_training.py:72-77: clip_grad_norm, optimizer.step(); ``optim.Adam`` is the trainer's default, :19-36).

A torch.optim.Optimizer subclass with Adam's constructor and state layout (``state[p] = {"step", "exp_avg", "exp_avg_sq"}``, so
``state_dict()`` / ``load_state_dict()`` interchange with torch.optim.Adam), stepping ALL parameters of a group in a handful of launches
(two per 48 tensors) instead of torch's ~130 multi-tensor launches on config 5's 205 parameters.  The update formulas are torch's, operation
for operation; results agree with torch.optim.Adam to rounding (tests/test_gpu_training.py).  fp32 CUDA parameters only (anything else
raises: no silent fallback); ``amsgrad`` / ``maximize`` / ``capturable`` are not offered.
"""
import ctypes

import torch

from . import _abi


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, max_grad_norm=None):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, max_grad_norm=max_grad_norm))
        self._lib = None
        self._scratch = {}
        self.last_total_norm = None   # device scalar tensor: the gradients' total 2-norm of the last clipped step

    @torch.no_grad()
    def step(self, closure=None, max_grad_norm=None):
        """One Adam step; ``max_grad_norm`` (or the group's) > 0 clips the gradients of ALL groups' parameters TOGETHER to that total 2-norm
        first -- ``torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm)`` fused into the step."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if self._lib is None:
            self._lib = _abi.load_product_library()
        jobs = []   # one per group that has gradients: (group, args builder inputs)
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            dev = ps[0].device
            for p in ps:
                if not (p.is_cuda and p.dtype == torch.float32 and p.grad.dtype == torch.float32 and p.device == dev and p.is_contiguous() and not p.grad.is_sparse):
                    raise TypeError("FusedAdam steps contiguous fp32 parameters of ONE MI355X (got %s %s on %s)" % (p.dtype, tuple(p.shape), p.device))
            steps = set()
            for p in ps:
                st = self.state[p]
                if not st:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                steps.add(int(st["step"]))
            if len(steps) != 1:
                raise RuntimeError("FusedAdam: the parameters of a group must have taken the same number of steps (got %r)" % sorted(steps))
            grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in ps]
            clip = max_grad_norm if max_grad_norm is not None else group.get("max_grad_norm")
            jobs.append((group, ps, grads, dev, steps.pop(), float(clip) if clip else 0.0))
        if not jobs:
            return loss
        if len({j[3] for j in jobs}) != 1:
            raise TypeError("FusedAdam steps the parameters of ONE MI355X (the groups live on %s)" % sorted(str(j[3]) for j in jobs))
        clips = {j[5] for j in jobs}
        if len(jobs) > 1 and len(clips) != 1:
            raise ValueError("FusedAdam: the groups are clipped TOGETHER (clip_grad_norm_ over all parameters): one max_grad_norm for all of them, got %r" % sorted(clips))
        dev = jobs[0][3]
        key = (dev.index or 0)
        if key not in self._scratch:
            self._scratch[key] = (torch.zeros(1, dtype=torch.float64, device=dev), torch.zeros((), dtype=torch.float32, device=dev))
        acc, norm = self._scratch[key]
        stream = torch.cuda.current_stream(dev).cuda_stream

        def call(job, flags):
            group, ps, grads, _, step, clip = job
            n = len(ps)
            arr = lambda vals: (ctypes.c_void_p * n)(*vals)   # noqa: E731
            sizes = (ctypes.c_int64 * n)(*[p.numel() for p in ps])
            tabs = [arr([p.data_ptr() for p in ps]), arr([g.data_ptr() for g in grads]),
                    arr([self.state[p]["exp_avg"].data_ptr() for p in ps]), arr([self.state[p]["exp_avg_sq"].data_ptr() for p in ps])]
            b1, b2 = group["betas"]
            a = _abi.wn_adam_args(n, dev.index or 0, ctypes.cast(sizes, ctypes.c_void_p), ctypes.cast(tabs[0], ctypes.c_void_p),
                                  ctypes.cast(tabs[1], ctypes.c_void_p), ctypes.cast(tabs[2], ctypes.c_void_p), ctypes.cast(tabs[3], ctypes.c_void_p),
                                  float(group["lr"]), float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]),
                                  clip, step, norm.data_ptr(), acc.data_ptr(), stream, flags)
            self._lib.check(self._lib.dll.wn_adam_step(ctypes.byref(a)))

        clip = jobs[0][5]
        if len(jobs) > 1 and clip:
            # the total norm is the norm over ALL groups' gradients: one norm pass per group into the same accumulator first, then every group
            # steps on that total (round 5 took a norm per group: wrong with a second group, e.g. biases without weight decay -- ADVICE r05)
            for k, job in enumerate(jobs):
                call(job, _abi.WN_ADAM_NORM_ONLY | (_abi.WN_ADAM_NORM_KEEP if k else 0))
            for job in jobs:
                call(job, _abi.WN_ADAM_NORM_GIVEN)
        else:
            for job in jobs:
                call(job, 0)
        for _, ps, grads, _, _, c in jobs:
            for p, g in zip(ps, grads):
                if g is not p.grad and c:
                    p.grad.copy_(g)   # (a non-contiguous .grad was stepped through a copy: the clipped values go back)
        if clip:
            self.last_total_norm = norm
        return loss
