"""dev tool: at config 5's clip length, N = 2 -- native fp32 gradients vs the real reference's digests (golden_v6) vs the facade's torch path on the GPU,
per tensor and per digest component."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-wavenet_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import digest as dg
import test_gpu_train_cfg5 as T
z = np.load(os.path.join(ROOT, "tests", "golden", "golden_v6.npz")); v6 = {k: z[k] for k in z.files}
ids, target, out_len = T._inputs(v6, 2)
m = T._model(out_len)
m.deterministic_gradients = True
logits_n, loss_n, g_n = T._native_step(m, ids, target)
x = torch.zeros(2, 256, T.L).scatter_(1, ids.view(2, 1, T.L), 1.0).cuda()
os.environ["WN_TORCH_BACKWARD"] = "1"
import warnings
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    m.zero_grad(set_to_none=True)
    out_t = m(x); loss_t = torch.nn.functional.cross_entropy(out_t, target.cuda()); loss_t.backward(); torch.cuda.synchronize()
g_t = {k: (p.grad.detach().cpu().numpy() if p.grad is not None else np.zeros(tuple(p.shape), np.float32)) for k, p in m.named_parameters()}
ref = T._ref_digests(v6, "n2")
dn, dt = dg.digest(g_n), dg.digest(g_t)
def comps(r, g):
    return [abs(g[0]-r[0])/r[0], abs(g[1]-r[1])/r[1], float(np.abs(g[2:6]-r[2:6]).max())/r[1], float(np.abs(g[6:]-r[6:]).max())/r[0]]
rows = []
for k, r in ref.items():
    if r[0] > 0:
        cn, ct = comps(r, dn[k]), comps(r, dt[k])
        e = float(np.abs(g_n[k]-g_t[k]).max())/float(np.abs(g_t[k]).max())
        rows.append((max(cn), k, cn, ct, e))
rows.sort(reverse=True)
print("loss native %.7f torch %.7f ref %.7f" % (loss_n, float(loss_t.detach()), float(v6["cfg5_n2_loss"][0])))
print("worst 12 tensors by native-vs-reference digest deviation: [max, norm, proj, elems]; torch-gpu vs ref; native vs torch elementwise/max")
for r in rows[:12]:
    print("%-26s native %s  torch %s  n-vs-t %.2e" % (r[1], ["%.1e" % c for c in r[2]], ["%.1e" % c for c in r[3]], r[4]))
print("best 5:")
for r in rows[-5:]:
    print("%-26s native %s  torch %s  n-vs-t %.2e" % (r[1], ["%.1e" % c for c in r[2]], ["%.1e" % c for c in r[3]], r[4]))
k = rows[0][1]
print(k, "ref elems", ref[k][6:14], "\n native", dn[k][6:14], "\n torch ", dt[k][6:14])
