"""Synthetic random-weight WaveNet parameter sets (there is no network for checkpoints).

Weight names and Conv1d shapes (out, in, k) are the reference's (wavenet_model.py:59-119) so
a dict produced here can be fed to ``WaveNetModel.load_state_dict`` of either implementation,
to the C-ABI ``wn_load_weights`` and to the oracle.

Init follows SURVEY.md section 8(c) "synthetic-weight caveat": PyTorch's default init with
``bias=False`` gives logits dominated by ``end_conv_2.bias`` (a weak parity test), so parity
and bench runs use a seeded N(0, gain/sqrt(fan_in)) init with gain 1.0 and N(0, 0.1) biases.
``numpy.random.RandomState`` (MT19937 + legacy normal) is bit-stable across numpy versions, so
the weights never have to be stored in fixtures.
"""
from collections import OrderedDict

import numpy as np

# BASELINE.json configs: name -> ctor kwargs
CONFIGS = {
    "cfg1": dict(layers=5, blocks=2, dilation_channels=32, residual_channels=32, skip_channels=256,
                 end_channels=256, classes=256, kernel_size=2, bias=False),
    "cfg2": dict(layers=10, blocks=3, dilation_channels=64, residual_channels=64, skip_channels=256,
                 end_channels=256, classes=256, kernel_size=2, bias=False),
    "cfg3": dict(layers=10, blocks=5, dilation_channels=128, residual_channels=128, skip_channels=512,
                 end_channels=256, classes=256, kernel_size=2, bias=False),
    # the only chaconne config present in the reference tree (train_script.py:17-25)
    "chaconne": dict(layers=10, blocks=3, dilation_channels=32, residual_channels=32, skip_channels=1024,
                     end_channels=512, classes=256, kernel_size=2, bias=True),
    # tiny shapes for fixtures / smoke
    "tiny": dict(layers=3, blocks=2, dilation_channels=16, residual_channels=16, skip_channels=32,
                 end_channels=32, classes=256, kernel_size=2, bias=False),
    "tiny_bias": dict(layers=3, blocks=2, dilation_channels=8, residual_channels=12, skip_channels=20,
                      end_channels=24, classes=256, kernel_size=2, bias=True),
}


def full_config(**kw):
    cfg = dict(layers=10, blocks=4, dilation_channels=32, residual_channels=32, skip_channels=256,
               end_channels=256, classes=256, kernel_size=2, bias=False)  # wavenet_model.py:28-39 defaults
    cfg.update(kw)
    return cfg


def dilation_list(cfg):
    """Per-layer dilation d_i = 2**(i mod layers) (wavenet_model.py:70-110)."""
    return [2 ** i for _ in range(cfg["blocks"]) for i in range(cfg["layers"])]


def receptive_field(cfg):
    """wavenet_model.py:53,106-107: 1 + blocks * (k-1) * (2**layers - 1)."""
    return 1 + cfg["blocks"] * (cfg["kernel_size"] - 1) * (2 ** cfg["layers"] - 1)


def param_shapes(cfg):
    """OrderedDict name -> shape, in the reference's ``state_dict`` naming."""
    c = full_config(**cfg)
    R, D, S, E, C, k = (c["residual_channels"], c["dilation_channels"], c["skip_channels"],
                        c["end_channels"], c["classes"], c["kernel_size"])
    shapes = OrderedDict()
    nl = c["layers"] * c["blocks"]
    # registration order of the reference ctor: ModuleLists first, then start_conv, end convs
    for i in range(nl):
        shapes["filter_convs.%d.weight" % i] = (D, R, k)
        if c["bias"]:
            shapes["filter_convs.%d.bias" % i] = (D,)
    for i in range(nl):
        shapes["gate_convs.%d.weight" % i] = (D, R, k)
        if c["bias"]:
            shapes["gate_convs.%d.bias" % i] = (D,)
    for i in range(nl):
        shapes["residual_convs.%d.weight" % i] = (R, D, 1)
        if c["bias"]:
            shapes["residual_convs.%d.bias" % i] = (R,)
    for i in range(nl):
        shapes["skip_convs.%d.weight" % i] = (S, D, 1)
        if c["bias"]:
            shapes["skip_convs.%d.bias" % i] = (S,)
    shapes["start_conv.weight"] = (R, C, 1)
    if c["bias"]:
        shapes["start_conv.bias"] = (R,)
    shapes["end_conv_1.weight"] = (E, S, 1)
    shapes["end_conv_1.bias"] = (E,)
    shapes["end_conv_2.weight"] = (C, E, 1)
    shapes["end_conv_2.bias"] = (C,)
    return shapes


def init_weights(cfg, seed=0, gain=1.0, bias_std=0.1):
    """Seeded gain/sqrt(fan_in) normal init; returns OrderedDict name -> float32 ndarray."""
    rng = np.random.RandomState(seed)
    out = OrderedDict()
    for name, shape in sorted(param_shapes(cfg).items()):  # sorted: independent of dict order
        if name.endswith(".weight"):
            fan_in = shape[1] * shape[2]
            w = rng.standard_normal(shape) * (gain / np.sqrt(fan_in))
        else:
            w = rng.standard_normal(shape) * bias_std
        out[name] = np.ascontiguousarray(w, dtype=np.float32)
    return OrderedDict((k, out[k]) for k in param_shapes(cfg))


def parameter_count(cfg):
    return int(sum(int(np.prod(s)) for s in param_shapes(cfg).values()))


def algorithmic_bytes_per_step(cfg, n_streams=1):
    """W_touched + n_streams * (Q_traffic + 8).

    W_touched = every stack/end weight and bias + ONE column of start_conv (one-hot input);
    Q_traffic = NL * 2 * R * 4 B (one column read + one column written per layer per stream).
    """
    c = full_config(**cfg)
    shapes = param_shapes(c)
    w = 0
    for name, shape in shapes.items():
        n = int(np.prod(shape))
        if name == "start_conv.weight":
            n = shape[0]
        w += n
    nl = c["layers"] * c["blocks"]
    q = nl * 2 * c["residual_channels"] * 4
    return w * 4 + n_streams * (q + 8)


def flops_per_step(cfg):
    """2*(NL*(2*D*R*k + R*D + S*D) + S*E + E*C)."""
    c = full_config(**cfg)
    R, D, S, E, C, k = (c["residual_channels"], c["dilation_channels"], c["skip_channels"],
                        c["end_channels"], c["classes"], c["kernel_size"])
    nl = c["layers"] * c["blocks"]
    return 2 * (nl * (2 * D * R * k + R * D + S * D) + S * E + E * C)
