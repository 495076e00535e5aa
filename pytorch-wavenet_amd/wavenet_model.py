"""Drop-in for the reference's ``wavenet_model`` module: ``WaveNetModel``, ``load_latest_model_from``,
``load_to_cpu`` -- same constructor, parameter names, attributes and method signatures
(/root/reference/wavenet_model.py:8-346), so ``from wavenet_model import *`` callers, reference
``state_dict``s and pickled snapshots keep working.

What is different underneath:
  * ``generate_fast()`` does not run the Python per-sample loop.  It hands the whole job to the MI355X
    engine (mi355_wavenet.engine -> C ABI include/wn_abi.h -> HIP kernels csrc/wn_kernel_v3.h / wn_kernel.h): one persistent
    kernel launch per call (per progress interval when a callback is given).  There is NO CPU fallback: without
    a gfx950 GPU and the built library it raises.
  * the global numpy RNG is consumed exactly as the reference does (one ``random_sample`` per generated
    sample when ``temperature > 0``: ``np.random.choice``, wavenet_model.py:288) -- the draws are made on the
    host and shipped to the kernel, which performs the same float64 inverse-CDF lookup.
  * extension: ``first_samples`` may be 2-D ``(streams, n_given)``; the result is then ``(streams, num_samples)``.
  * the priming window (``first_samples`` of length n > 64) is evaluated as one batched matrix-core pass (wn_prime).
  * ``forward()`` on a CUDA one-hot batch runs natively on the matrix cores (wn_forward; wn_train_forward / wn_train_backward
    behind a torch.autograd.Function when gradients are wanted) -- including clips shorter than receptive_field +
    output_length - 1, where the reference left-pads the layers' activations with zeros (wavenet_modules.py:24-27: per-layer row
    windows in the kernels).  The reference's algorithm with torch ops remains for: CPU tensors, inputs that are not one-hot,
    gradients w.r.t. the input, kernel_size != 2, a class count that is not a multiple of 32 under autograd (channel counts that are not
    multiples of 32 run natively, zero-padded: with and, since round 6, under autograd), and the input lengths for which the reference itself has no defined result (its error, or its shapes, are reproduced).
"""
import os
import os.path
import time

from wavenet_modules import *  # noqa: F401,F403  (the reference re-exports these names, wavenet_model.py:4)
from audio_data import *  # noqa: F401,F403       (and these, :5)

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


class WaveNetModel(nn.Module):
    """
    A Complete Wavenet Model (constructor arguments as in the reference, wavenet_model.py:28-39)

    Args:
        layers (Int):               Number of layers in each block
        blocks (Int):               Number of wavenet blocks of this model
        dilation_channels (Int):    Number of channels for the dilated convolution
        residual_channels (Int):    Number of channels for the residual connection
        skip_channels (Int):        Number of channels for the skip connections
        end_channels (Int):         Number of channels of the penultimate 1x1 convolution
        classes (Int):              Number of possible values each sample can have
        output_length (Int):        Number of samples that are generated for each input
        kernel_size (Int):          Size of the dilation kernel
        dtype:                      Parameter type of this model (legacy tensor type object or torch.dtype)
        bias (Bool):                Whether the stack convolutions carry a bias
    """

    def __init__(self, layers=10, blocks=4, dilation_channels=32, residual_channels=32, skip_channels=256,
                 end_channels=256, classes=256, output_length=32, kernel_size=2, dtype=torch.FloatTensor, bias=False):
        super(WaveNetModel, self).__init__()
        self.layers = layers
        self.blocks = blocks
        self.dilation_channels = dilation_channels
        self.residual_channels = residual_channels
        self.skip_channels = skip_channels
        self.end_channels = end_channels
        self.classes = classes
        self.kernel_size = kernel_size
        self.dtype = dtype
        self.bias = bias

        self.dilations = []        # (dilation, previous dilation) per layer  (wavenet_model.py:75)
        self.dilated_queues = []   # plain list on the module, not in state_dict (:56-57, :78-81)
        self.filter_convs = nn.ModuleList()
        self.gate_convs = nn.ModuleList()
        self.residual_convs = nn.ModuleList()
        self.skip_convs = nn.ModuleList()
        self.start_conv = nn.Conv1d(classes, residual_channels, kernel_size=1, bias=bias)

        receptive_field = 1
        previous = 1
        for _ in range(blocks):
            span = kernel_size - 1
            d = 1
            for _ in range(layers):
                self.dilations.append((d, previous))
                self.dilated_queues.append(DilatedQueue(max_length=(kernel_size - 1) * d + 1,
                                                        num_channels=residual_channels, dilation=d, dtype=dtype))
                self.filter_convs.append(nn.Conv1d(residual_channels, dilation_channels, kernel_size, bias=bias))
                self.gate_convs.append(nn.Conv1d(residual_channels, dilation_channels, kernel_size, bias=bias))
                self.residual_convs.append(nn.Conv1d(dilation_channels, residual_channels, 1, bias=bias))
                self.skip_convs.append(nn.Conv1d(dilation_channels, skip_channels, 1, bias=bias))
                receptive_field += span
                span *= 2
                previous = d
                d *= 2
        self.end_conv_1 = nn.Conv1d(skip_channels, end_channels, 1, bias=True)
        self.end_conv_2 = nn.Conv1d(end_channels, classes, 1, bias=True)
        self.output_length = output_length
        self.receptive_field = receptive_field
        self._wn_engine = None
        self._wn_engine_key = None
        self._wn_forward_calls = 0
        self._wn_train_runner = None
        self._wn_train_calls = 0
        # Extension: operand precision of the native matrix-core forward / backward GEMMs: "fp32" (default: equals the
        # reference's fp32 graph to rounding) or "bf16" (bf16 operands, fp32 accumulation and fp32 residual stream;
        # needs channel counts that are multiples of 64, otherwise fp32 is used)
        self.matrix_precision = "fp32"
        # Extension: True = the native backward's weight / bias gradients are bit-reproducible from run to run (ordered reduction of the row splits'
        # partial tiles instead of fp32 atomics: C ABI wn_train_set_deterministic); None = the library's default (off, or WN_DETERMINISTIC=1)
        self.deterministic_gradients = None

    # ------------------------------------------------------------------ training path (torch ops)
    def wavenet(self, input, dilation_func):
        """The residual stack (wavenet_model.py:125-171) with a pluggable dilation function."""
        x = self.start_conv(input)
        skip = None
        for i in range(self.blocks * self.layers):
            dilation, init_dilation = self.dilations[i]
            residual = dilation_func(x, dilation, init_dilation, i)
            x = torch.tanh(self.filter_convs[i](residual)) * torch.sigmoid(self.gate_convs[i](residual))
            s = x
            if x.size(2) != 1:
                s = dilate(x, 1, init_dilation=dilation)
            s = self.skip_convs[i](s)
            # right-aligned crop-add (the reference gets skip=0 on the first layer through a bare except)
            skip = s if skip is None else s + skip[:, :, -s.size(2):]
            x = self.residual_convs[i](x)
            x = x + residual[:, :, (self.kernel_size - 1):]
        x = F.relu(skip)
        x = F.relu(self.end_conv_1(x))
        return self.end_conv_2(x)

    def wavenet_dilate(self, input, dilation, init_dilation, i):
        return dilate(input, dilation, init_dilation)

    def queue_dilate(self, input, dilation, init_dilation, i):
        queue = self.dilated_queues[i]
        queue.enqueue(input.data[0])
        return queue.dequeue(num_deq=self.kernel_size, dilation=dilation).unsqueeze(0)

    def _native_supported(self):
        """Shapes the matrix-core kernels cover (wn_forward / wn_train_*): kernel_size 2, channel counts multiples of 32."""
        return self.kernel_size == 2 and not any(c % 32 for c in (self.residual_channels, self.dilation_channels, self.skip_channels,
                                                                  self.end_channels, self.classes))

    def _native_trainable(self):
        """Shapes the native training step covers: kernel_size 2 and a class count that is a multiple of 32 -- channel counts that are not multiples
        of 32 are zero-padded up to multiples of 64 for it (round 6: mi355_wavenet/training.py StackRunner.pad_tensors; same logits, same gradients)."""
        return self.kernel_size == 2 and self.classes % 32 == 0

    def _padded_train_config(self):
        """(config, model shape) of the training engine: the model's own when its channel counts are multiples of 32, else padded to multiples of 64
        (so that the bf16 step's kernels apply as well)."""
        cfg = self._config()
        if self._native_supported():
            return cfg, None
        shape = (self.residual_channels, self.dilation_channels, self.skip_channels, self.end_channels)
        up = lambda c: (c + 63) // 64 * 64  # noqa: E731
        cfg.update(residual_channels=up(shape[0]), dilation_channels=up(shape[1]), skip_channels=up(shape[2]), end_channels=up(shape[3]))
        return cfg, shape

    def _fallback(self, reason, warn=True):
        """forward() on a CUDA tensor is about to run the reference's algorithm in torch ops (MIOpen conv1d + autograd) instead of the
        native kernels: counted per reason (wn_stats()) and said out loud ONCE per reason -- the dual path is never silent."""
        stats = self.__dict__.setdefault("_wn_fallbacks", {})
        stats[reason] = stats.get(reason, 0) + 1
        if warn and stats[reason] == 1:
            import warnings
            warnings.warn("WaveNetModel.forward(): torch path instead of the MI355X matrix-core kernels: %s "
                          "(said once per reason; model.wn_stats() counts every call)" % reason, RuntimeWarning, stacklevel=4)
        return None

    def wn_stats(self):
        """Extension: which path forward() / model(x) calls of this module took so far: {'native_forward', 'native_train_forward',
        'torch_fallbacks': {reason: calls}} -- CPU tensors are not fallbacks (the reference's own path), everything else on a CUDA
        tensor that did not reach wn_forward / wn_train_* is."""
        return {"native_forward": int(getattr(self, "_wn_forward_calls", 0)), "native_train_forward": int(getattr(self, "_wn_train_calls", 0)),
                "torch_fallbacks": dict(self.__dict__.get("_wn_fallbacks", {}))}

    def _native_forward(self, input):
        """Matrix-core forward (C ABI wn_forward, or wn_train_forward + wn_train_backward behind a torch.autograd.Function
        when gradients are wanted) when it applies: CUDA input that is exactly one-hot, shapes the GEMM kernels support.  Returns
        None otherwise -- the caller then runs the torch path, and on a CUDA tensor that is counted and warned about (_fallback).
        Short clips (the reference's zero-padding regime) are served natively; the engine answers WN_E_UNSUPPORTED only for lengths
        at which the reference has no defined result, and the torch path then reproduces what the reference does there.  A library
        that is not built is NOT a reason to fall back: on a CUDA tensor that raises (the product must not run silently without its
        kernels)."""
        if not input.is_cuda:
            return None   # the reference's own path: nothing to report
        if input.dim() != 3 or input.size(1) != self.classes:
            return self._fallback("input is not (N, classes, L)")
        if self.kernel_size != 2:
            return self._fallback("kernel_size %d (the matrix-core kernels are written for 2)" % self.kernel_size)
        want_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        # Without autograd the engine decides: a channel shape that is not a multiple of 32 may still run natively, zero-padded into a
        # compiled shape (include/wn_abi.h: wn_create); with autograd the handle keeps the model's own shape and needs the multiples.
        no_native = getattr(self, "_wn_forward_unsupported", None) == self._forward_shape_key(input.device)
        if want_grad and not self._native_trainable():
            return self._fallback("a class count that is not a multiple of 32 under autograd")
        if not want_grad and no_native and not self._native_supported():
            return self._fallback("channel counts that are not multiples of 32 and fit no compiled shape")
        if torch.is_grad_enabled() and input.requires_grad:
            return self._fallback("a gradient with respect to the one-hot input is wanted")
        if want_grad and input.dtype != torch.float32:
            return self._fallback("autograd on a %s input" % str(input.dtype).replace("torch.", ""))
        if want_grad and os.environ.get("WN_TORCH_BACKWARD") == "1":
            return self._fallback("WN_TORCH_BACKWARD=1")
        vals, idx = input.max(dim=1)
        if not bool(((vals == 1) & (input.sum(dim=1) == 1)).all()):
            return self._fallback("the input is not a one-hot batch (start_conv is a real contraction)")
        from mi355_wavenet import _abi
        try:
            if want_grad:
                return self._native_train_forward(idx)
            eng = self._forward_engine()
            if not eng.info()["forward_native"]:
                # THIS channel shape has no native forward on this device (not zero-padded into a compiled shape): do not ask again.
                # Keyed on what the engine REPORTS (wn_info.forward_native), not on the wording of an error.
                self._wn_forward_unsupported = self._forward_shape_key(input.device)
                return self._fallback("channel counts that are not multiples of 32 and fit no compiled shape")
            self._apply_precision(eng)
            out = eng.forward_indices(idx, self.output_length)
        except _abi.WnError as e:
            if e.code == _abi.WN_E_UNSUPPORTED:
                # (refusals of ONE call -- N*L >= 2^31 rows, a clip length the reference has no result for -- say nothing about the next call;
                #  the torch path reproduces what the reference does there: its error, or its shapes)
                return self._fallback("the engine refused this call: %s" % e, warn=False)
            raise
        self._wn_forward_calls = getattr(self, "_wn_forward_calls", 0) + 1
        return out.to(input.dtype)

    def _forward_shape_key(self, device):
        return (str(device), self.residual_channels, self.dilation_channels, self.skip_channels, self.end_channels, self.classes)

    def _forward_engine(self):
        """The engine forward() runs on: the generation engine when it holds the current parameters -- whatever its stream count, so that a
        validation forward between two generate_fast() calls leaves it (and its deferred queues) alone --, else a one-stream engine."""
        eng = self._wn_engine
        if eng is not None and self._wn_engine_key is not None:
            plist = list(self.parameters())
            dev = plist[0].device
            index = dev.index if dev.type == "cuda" and dev.index is not None else int(os.environ.get("WN_DEVICE", "0"))
            if self._wn_engine_key[1:] == (index, tuple((v.data_ptr(), v._version) for v in plist)):
                return eng
        return self._engine(1)

    def _apply_precision(self, eng):
        want = getattr(self, "matrix_precision", "fp32") == "bf16"
        c = eng.cfg   # (the ENGINE's channel shape: the training engine of a model with odd channel counts is zero-padded to multiples of 64)
        if want and any(c[k] % 64 for k in ("residual_channels", "dilation_channels", "skip_channels", "end_channels")):
            want = False
        eng.set_forward_precision(want)

    def _native_train_forward(self, idx):
        """model(x) with a native backward: see mi355_wavenet/training.py."""
        from mi355_wavenet import engine, training
        runner = getattr(self, "_wn_train_runner", None)
        dev = next(self.parameters()).device
        if runner is None or runner.device != dev:
            cfg, shape = self._padded_train_config()
            weights = dict(self.state_dict())
            if shape is not None:   # (the handle is created on zero-padded weights of ITS shape; the step's parameters are passed per call)
                probe = training.StackRunner.__new__(training.StackRunner)
                probe.R, probe.D, probe.S, probe.E, probe.C = (cfg["residual_channels"], cfg["dilation_channels"], cfg["skip_channels"],
                                                               cfg["end_channels"], cfg["classes"])
                kinds = {"start_conv": "start", "end_conv_1": "end1", "end_conv_2": "end2", "filter_convs": "filter", "gate_convs": "gate",
                         "residual_convs": "res", "skip_convs": "skip"}
                padded = {}
                for name, t in weights.items():
                    parts = name.split(".")
                    key = kinds[parts[0]] + ("_w" if parts[-1] == "weight" else "_b")
                    buf = torch.zeros(probe.padded_shape(key), dtype=t.dtype, device=t.device)
                    buf[tuple(slice(0, n) for n in t.shape)].copy_(t)
                    padded[name] = buf
                weights = padded
            eng = engine.Engine(cfg, weights, n_streams=1, device_index=dev.index or 0, pad_channels=False)
            runner = training.StackRunner(eng, model_shape=shape)  # the handle only provides plan, layout and workspace: parameters are passed per call
            self._wn_train_runner = runner
        names, tensors = [], []

        def add(key, ts):
            names.append((key, len(ts)))
            tensors.extend(ts)

        add("start_w", [self.start_conv.weight])
        add("filter_w", [m.weight for m in self.filter_convs])
        add("gate_w", [m.weight for m in self.gate_convs])
        add("res_w", [m.weight for m in self.residual_convs])
        add("skip_w", [m.weight for m in self.skip_convs])
        add("end1_w", [self.end_conv_1.weight]); add("end1_b", [self.end_conv_1.bias])
        add("end2_w", [self.end_conv_2.weight]); add("end2_b", [self.end_conv_2.bias])
        if self.start_conv.bias is not None:
            add("start_b", [self.start_conv.bias])
            add("filter_b", [m.bias for m in self.filter_convs])
            add("gate_b", [m.bias for m in self.gate_convs])
            add("res_b", [m.bias for m in self.residual_convs])
            add("skip_b", [m.bias for m in self.skip_convs])
        self._wn_train_calls = getattr(self, "_wn_train_calls", 0) + 1
        self._apply_precision(runner.eng)
        det = getattr(self, "deterministic_gradients", None)
        if det is not None:
            runner.set_deterministic(bool(det))
        return training.StackFunction.apply(runner, idx, self.output_length, tuple(names), *tensors)

    def _checked_indices(self, indices, check, training=False):
        idx = torch.as_tensor(indices)
        if idx.dim() != 2:
            raise ValueError("indices must be (N, L) class indices")
        if training and self._native_trainable():
            pass   # (the training engine pads odd channel counts itself)
        elif not self._native_supported():
            raise ValueError("the index-based forward needs kernel_size 2 and channel counts that are multiples of 32 "
                             "(residual %d, dilation %d, skip %d, end %d, classes %d)" % (
                                 self.residual_channels, self.dilation_channels, self.skip_channels, self.end_channels, self.classes))
        if idx.size(1) < 2:
            raise ValueError("items of %d sample(s): forward() needs at least two" % idx.size(1))
        if check and idx.numel() and (int(idx.min()) < 0 or int(idx.max()) >= self.classes):
            raise ValueError("class indices outside [0, %d)" % self.classes)  # they address rows of start_conv^T on the GPU
        return idx

    def forward_indices(self, indices, check=True):
        """Extension: forward() on class indices (N, L) instead of a one-hot (N, classes, L) tensor -- what the
        dataset holds before audio_data.py:119-121 inflates it 256x.  Inference only (matrix-core path, no autograd).
        ``check=False`` skips the range check of the indices (one device sync) when the producer guarantees it."""
        idx = self._checked_indices(indices, check)
        eng = self._forward_engine()
        self._apply_precision(eng)
        out = self._index_call(lambda: eng.forward_indices(idx, self.output_length))
        self._wn_forward_calls = getattr(self, "_wn_forward_calls", 0) + 1
        return out

    @staticmethod
    def _index_call(fn):
        """The index-based extensions have no torch path to fall back to: a clip length the engine refuses (one the reference has no
        defined result for either, include/wn_abi.h: wn_forward) is the caller's ValueError, with the engine's reason."""
        from mi355_wavenet import _abi
        try:
            return fn()
        except _abi.WnError as e:
            if e.code == _abi.WN_E_UNSUPPORTED:
                raise ValueError(str(e)) from e
            raise

    def train_forward_indices(self, indices, check=True):
        """Extension: the differentiable forward() on class indices (N, L) -- the training-time sibling of forward_indices:
        logits (N*output_length, classes) whose backward runs natively (see _native_train_forward).  MI355X only."""
        idx = self._checked_indices(indices, check, training=True)
        return self._index_call(lambda: self._native_train_forward(idx))

    def forward(self, input):
        """(N, classes, L) one-hot -> (N*output_length, classes) logits (wavenet_model.py:186-196)."""
        native = self._native_forward(input)
        if native is not None:
            return native
        x = self.wavenet(input, dilation_func=self.wavenet_dilate)
        n, c, _ = x.size()
        l = self.output_length
        x = x[:, :, -l:].transpose(1, 2).contiguous()
        return x.view(n * l, c)

    def generate(self, num_samples, first_samples=None, temperature=1.):
        raise NotImplementedError("WaveNetModel.generate() is dead code upstream (undefined self.scope, "
                                  "wavenet_model.py:209); use generate_fast()")

    # ------------------------------------------------------------------ generation path (HIP engine)
    def _config(self):
        return dict(layers=self.layers, blocks=self.blocks, dilation_channels=self.dilation_channels,
                    residual_channels=self.residual_channels, skip_channels=self.skip_channels,
                    end_channels=self.end_channels, classes=self.classes, kernel_size=self.kernel_size,
                    bias=self.start_conv.bias is not None)

    def _engine(self, n_streams):
        """The MI355X engine holding this module's current parameters (rebuilt when they change)."""
        from mi355_wavenet import engine
        plist = list(self.parameters())
        dev = plist[0].device
        index = dev.index if dev.type == "cuda" and dev.index is not None else int(os.environ.get("WN_DEVICE", "0"))
        key = (n_streams, index, tuple((v.data_ptr(), v._version) for v in plist))  # (every state_dict entry of this module is a parameter)
        if self._wn_engine is None or self._wn_engine_key != key:
            params = list(self.state_dict().items())
            if self._wn_engine is not None and self._wn_engine_key[:2] == key[:2]:
                self._wn_engine.load_weights(dict(params))
            else:
                if self._wn_engine is not None:
                    self._flush_queues()  # their loaders read the engine that is about to go
                    self._wn_engine.close()
                self._wn_engine = engine.Engine(self._config(), dict(params), n_streams=n_streams, device_index=index)
            self._wn_engine_key = key
        return self._wn_engine

    def generate_fast(self, num_samples, first_samples=None, temperature=1., regularize=0.,
                      progress_callback=None, progress_interval=100):
        """Same contract as wavenet_model.py:237-315; returns float64 ndarray (num_samples,)."""
        self.eval()
        if first_samples is None:
            first_samples = torch.LongTensor(1).zero_() + (self.classes // 2)
        first = torch.as_tensor(first_samples).detach().cpu().numpy().astype(np.int64)
        batched = first.ndim == 2
        first = first.reshape(first.shape[0], -1) if batched else first.reshape(1, -1)
        n_streams, num_given = first.shape
        total_samples = num_given + num_samples
        for queue in self.dilated_queues:  # :250-251
            queue.reset()
        eng = self._engine(n_streams)
        eng.reset()
        sampled = temperature > 0

        # The job is evaluations ev = 0 .. n_eval-1 (num_given-1 priming + num_samples generating).
        #   priming (:259-269): all n_prime teacher-forced evaluations run as ONE batched pass over the given window
        #     (C ABI wn_prime: the stack as matrix-core GEMMs over all positions, queues filled from the result), then the
        #     callbacks the reference would have fired at i % interval == 0 are delivered in order; shapes the batched
        #     pass does not cover take the per-sample chain (WN_E_UNSUPPORTED), cut at the callbacks like below.
        #   generating (:276-311): one persistent launch per piece, cut where the reference would have called back
        #     ((i + num_given) % interval == 0, :308-311) or printed its timing line (after generating step 99, :304-306).
        n_prime = num_given - 1
        n_eval = n_prime + num_samples
        a = 0
        primed_upto = 0  # priming evaluations whose callbacks were already delivered (batched priming)
        if n_prime >= eng.PRIME_BATCH_MIN and eng.prime_host(first[:, :n_prime]):
            a = primed_upto = n_prime
            if progress_callback is not None:
                for i in range(0, n_prime, progress_interval):
                    progress_callback(i, total_samples)
        self._wn_last_prime_batched = a > 0
        cuts = {n_eval}
        if n_prime > a:
            cuts.add(n_prime)  # the generating loop (and the reference's stopwatch, :275) starts here
        if num_samples >= 100:
            cuts.add(n_prime + 100)
        if progress_callback is not None:
            cuts.update(i + 1 for i in range(a, n_prime) if i % progress_interval == 0)
            cuts.update(n_prime + i + 1 for i in range(num_samples) if (i + num_given) % progress_interval == 0)
        def callback_due(ev):  # the reference calls back after evaluation ev (:266-269, :308-311)
            if progress_callback is None or ev < 0:
                return False
            if ev < n_prime:
                return ev >= primed_upto and ev % progress_interval == 0
            return (ev - n_prime + num_given) % progress_interval == 0

        def n_generated(a0, b0):  # samples the piece [a0, b0) of evaluations generates
            return (b0 - a0) - max(0, min(b0, n_prime) - a0)

        pieces = []
        last = None
        tic = time.time()
        ends = sorted(c for c in cuts if c > a)
        drawn = None  # the NEXT piece's uniforms, drawn and uploaded while the current piece's kernel ran
        for k, b in enumerate(sorted(cuts)):
            if b > a:
                if a == n_prime:
                    tic = time.time()  # :275
                seg_prime = max(0, min(b, n_prime) - a)
                n_new = (b - a) - seg_prime
                head = first[:, a:a + seg_prime + 1] if a < num_given else last
                # one uniform per generated sample from the GLOBAL numpy RNG, drawn right before the piece runs -- or, when nothing
                # of the caller's runs between two pieces (no callback due at the cut: e.g. the cut behind generating step 99 that only
                # exists for the reference's timing print), while the PREVIOUS piece's kernel runs: same draws, same order
                u = drawn if drawn is not None else (np.random.random_sample((n_streams, n_new)) if (sampled and n_new > 0) else None)
                drawn = None
                nxt = [c for c in ends if c > b]
                overlap = None
                if sampled and nxt and not callback_due(b - 1) and n_generated(b, nxt[0]) > 0:
                    def overlap(n_next=n_generated(b, nxt[0])):
                        nonlocal drawn
                        drawn = eng.upload_uniforms(np.random.random_sample((n_streams, n_next)))
                out = eng.generate(n_new, head, temperature=temperature, regularize=regularize, uniforms=u, reset=False, while_running=overlap)
                if n_new > 0:
                    pieces.append(out)
                    last = out[:, -1:].astype(np.int64)
                a = b
            ev = b - 1  # the evaluation that just finished
            if ev < 0:
                continue
            if ev >= n_prime and ev - n_prime + 1 == 100:
                toc = time.time()
                print("one generating step does take approximately " + str((toc - tic) * 0.01) + " seconds)")
            if callback_due(ev):  # (priming: not a second time after batched priming)
                progress_callback(ev if ev < n_prime else ev - n_prime + num_given, total_samples)
        idx = np.concatenate(pieces, axis=1) if pieces else np.zeros((n_streams, 0), dtype=np.int32)
        self._defer_queues(eng)
        self.train()
        mu_gen = self._expand_indices(idx)  # :296, :314
        return mu_gen if batched else mu_gen[0]

    def _expand_indices(self, idx):
        """Class indices -> float64 audio: ``o = idx / classes * 2 - 1`` (wavenet_model.py:296) then ``mu_law_expansion(o)``
        (:314, audio_data.py:156-158).  Both are elementwise, so the 256 possible results are computed once with the very
        same numpy expressions and looked up (10 ms -> 0.3 ms per 128 k samples); the table is checked once against the
        elementwise evaluation on a probe that puts every class at shifted array positions."""
        classes = self.classes
        tab = getattr(self, "_wn_expansion", None)
        if tab is None or tab[0] != classes:
            every = np.arange(classes, dtype=np.int64)
            table = mu_law_expansion((every / classes) * 2. - 1, classes)
            probe = np.concatenate([every[3:], every[::-1], every[:5]])
            ok = bool(np.array_equal(table[probe], mu_law_expansion((probe / classes) * 2. - 1, classes)))
            tab = self._wn_expansion = (classes, table if ok else None)
        idx = np.asarray(idx).astype(np.int64)
        if tab[1] is None:
            return mu_law_expansion((idx / classes) * 2. - 1, classes)
        return tab[1][idx]

    def _defer_queues(self, eng, stream=0):
        """The reference leaves model.dilated_queues in their final state (wavenet_model.py:177-184).  Here that state is on
        the GPU: every queue gets a loader that downloads its ring on first access (wn_export_queue); nothing is copied
        for callers that never look."""
        for layer, queue in enumerate(self.dilated_queues):
            queue._defer(lambda layer=layer: eng.export_queue(layer, stream))

    def _flush_queues(self):
        for queue in self.dilated_queues:
            if getattr(queue, "_lazy", None) is not None:
                queue._sync()

    def generate_fast_streams(self, num_samples, temperatures, first_samples=None, regularize=0.):
        """Extension: one generate_fast() per entry of ``temperatures`` -- the reference's generate_audio loop
        (wavenet_training.py:115-124) -- as parallel streams of ONE persistent-kernel job.  Returns float64
        (len(temperatures), num_samples), row k equal to ``generate_fast(num_samples, first_samples, temperatures[k])``
        of sequential calls: sampled rows consume the global numpy RNG in the order of ``temperatures`` (one draw per
        sample), greedy rows (temperature <= 0, wavenet_model.py:290-294) consume none."""
        self.eval()
        temps = [float(t) for t in temperatures]
        if first_samples is None:
            first_samples = torch.LongTensor(1).zero_() + (self.classes // 2)
        first = torch.as_tensor(first_samples).detach().cpu().numpy().astype(np.int64).reshape(1, -1)
        first = np.repeat(first, len(temps), axis=0)
        for queue in self.dilated_queues:
            queue.reset()
        eng = self._engine(len(temps))
        uniforms = np.zeros((len(temps), num_samples), dtype=np.float64)
        for k, t in enumerate(temps):
            if t > 0:
                uniforms[k] = np.random.random_sample(num_samples)
        idx = eng.generate(num_samples, first, temperature=np.asarray(temps, dtype=np.float32), regularize=regularize,
                           uniforms=uniforms if any(t > 0 for t in temps) else None)
        self._defer_queues(eng, stream=len(temps) - 1)  # sequential calls would leave the LAST temperature's queues
        self.train()
        return self._expand_indices(idx)

    # ------------------------------------------------------------------ bookkeeping
    def parameter_count(self):
        return sum(int(np.prod(list(p.size()))) for p in self.parameters())

    def cpu(self, type=torch.FloatTensor):
        self.dtype = type
        for q in self.dilated_queues:
            q.dtype = self.dtype
        return super().cpu()

    def __getstate__(self):  # the engine handle is not picklable and is rebuilt on demand
        self._flush_queues()
        state = self.__dict__.copy()
        state["_wn_engine"] = None
        state["_wn_engine_key"] = None
        state["_wn_train_runner"] = None
        state.pop("_wn_expansion", None)
        state.pop("_wn_forward_unsupported", None)   # (an answer about THIS device's library: not part of the model)
        return state

    def __setstate__(self, state):
        """Also accepts snapshots pickled by the REFERENCE's class (torch.save(model), wavenet_training.py:84-88 -- its only
        checkpoint format): their __dict__ has no end_channels / bias / engine fields (wavenet_model.py:42-56)."""
        super().__setstate__(state)   # nn.Module back-fills the hook / buffer attributes that a snapshot of an older torch lacks
        d = self.__dict__
        d.setdefault("_wn_engine", None)
        d.setdefault("_wn_engine_key", None)
        d.setdefault("_wn_forward_calls", 0)
        d.setdefault("_wn_train_runner", None)
        d.setdefault("_wn_train_calls", 0)
        d.setdefault("matrix_precision", "fp32")
        d.setdefault("deterministic_gradients", None)
        if "end_channels" not in d:
            d["end_channels"] = self.end_conv_1.out_channels
        if "bias" not in d:
            d["bias"] = self.start_conv.bias is not None


def load_latest_model_from(location, use_cuda=True):
    """Newest file (by ctime) in ``location`` -> model (wavenet_model.py:330-340)."""
    files = [location + "/" + f for f in os.listdir(location)]
    newest_file = max(files, key=os.path.getctime)
    print("load model " + newest_file)
    if use_cuda:
        model = torch.load(newest_file, weights_only=False)
    else:
        model = load_to_cpu(newest_file)
    return model


def load_to_cpu(path):
    model = torch.load(path, map_location=lambda storage, loc: storage, weights_only=False)
    model.cpu()
    return model
