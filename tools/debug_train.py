import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "pytorch-wavenet_amd"))
import torch
import test_gpu_training as T
bias = len(sys.argv) > 1 and sys.argv[1] == "bias"
m = T._model(bias)
x, target = T._batch(m, 3, 0)
out_t, loss_t, g_t = T._step(m, x, target, True)
out_n, loss_n, g_n = T._step(m, x, target, False)
print("logits", float((out_n - out_t).abs().max()), loss_t, loss_n)
for k in g_t:
    if g_t[k] is None:
        print(k, "None", g_n[k] is None); continue
    s = float(g_t[k].abs().max()); e = float((g_n[k] - g_t[k]).abs().max())
    print("%-28s scale %.3e err %.3e rel %.2e |native| %.3e" % (k, s, e, e / (s + 1e-30), float(g_n[k].abs().max())))
