"""dev tool (GPU box): bf16 forward of shallow models (1, 2, 3, 4, 8 layers at cfg2's and cfg3's widths) -> gpurun_out/r05_bf16_depth.npz"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("pytorch-wavenet_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from mi355_wavenet import engine, synth  # noqa: E402

out = {}
for wname, ch, sk, en in (("w64", 64, 256, 256), ("w128", 128, 512, 256)):
    for layers, blocks in ((1, 1), (2, 1), (3, 1), (4, 1), (4, 2)):
        cfg = dict(layers=layers, blocks=blocks, dilation_channels=ch, residual_channels=ch, skip_channels=sk, end_channels=en, classes=256, kernel_size=2, bias=False)
        W = synth.init_weights(cfg, seed=5)
        rf = 1 + blocks * (2 ** layers - 1)
        ids = np.random.RandomState(3).randint(0, 256, (2, rf + 7 + 8 * 2 ** layers))
        eng = engine.Engine(cfg, W, n_streams=1, pad_channels=False)
        for prec in (0, 1):
            eng.set_forward_precision(bool(prec))
            y = eng.forward_indices(ids, 8).cpu().numpy()
            out["%s_%dx%d_%s" % (wname, layers, blocks, "bf16" if prec else "fp32")] = y
        eng.close()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "r05_bf16_depth.npz"), **out)
print("ok", len(out))
