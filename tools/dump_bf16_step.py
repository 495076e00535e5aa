"""dev tool (GPU box): the product's bf16 training step on the golden_v5 cases -> gpurun_out/r05_bf16_product.npz (logits, loss, gradient digests,
and the fp32 step's logits), for comparing rounding models of oracle/bf16_step.py offline."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("pytorch-wavenet_amd", "oracle", "tests", os.path.join("tests", "golden")):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import digest as dg  # noqa: E402
import wavenet_model  # noqa: E402
from mi355_wavenet import synth  # noqa: E402

B64 = dict(layers=4, blocks=2, dilation_channels=64, residual_channels=64, skip_channels=128, end_channels=128, classes=256, kernel_size=2, bias=True)
CASES = {"cfg3": "cfg3", "cfg2": "cfg2", "b64": B64}
g5 = np.load(os.path.join(ROOT, "tests", "golden", "golden_v5.npz"))
out = {}
for case, c in CASES.items():
    cfg = synth.CONFIGS[c] if isinstance(c, str) else c
    wseed, N, out_len, L = [int(v) for v in g5["bf16_%s_meta" % case]]
    for prec in ("fp32", "bf16"):
        m = wavenet_model.WaveNetModel(output_length=out_len, **cfg)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.init_weights(cfg, seed=wseed).items()})
        m = m.cuda()
        m.matrix_precision = prec
        ids = torch.from_numpy(g5["bf16_%s_ids" % case].astype(np.int64))
        x = torch.zeros(N, 256, ids.shape[1]).scatter_(1, ids.view(N, 1, -1), 1.0).cuda()
        target = torch.from_numpy(g5["bf16_%s_target" % case].astype(np.int64)).cuda()
        y = m(x)
        loss = torch.nn.functional.cross_entropy(y, target)
        loss.backward()
        out["%s_%s_out" % (case, prec)] = y.detach().cpu().numpy()
        out["%s_%s_loss" % (case, prec)] = np.array([float(loss.detach())])
        for k, p in m.named_parameters():
            gr = p.grad.cpu().numpy() if p.grad is not None else np.zeros(tuple(p.shape), np.float32)
            out["%s_%s_d_%s" % (case, prec, k)] = dg.tensor_digest(k, gr)
        print(case, prec, float(loss.detach()), m.wn_stats(), m._wn_train_runner.eng.info()["forward_native"])
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "r05_bf16_product.npz"), **out)
