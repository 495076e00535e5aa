"""dev tool: native fp32 gradients against the facade's torch path on the CPU in fp32 AND fp64, one clip, growing output_length: does the deviation
jump (a tiling bug) or grow smoothly / sit at the fp32 reference's own distance from the fp64 truth (conditioning)?"""
import copy, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-wavenet_amd"))
import wavenet_model
from mi355_wavenet import synth
cfg = synth.CONFIGS["cfg3"]
W = synth.init_weights(cfg, seed=41)
for out_len in [int(v) for v in sys.argv[1:]] or [32, 127, 129, 600, 3000]:
    m = wavenet_model.WaveNetModel(output_length=out_len, **cfg)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()})
    L = m.receptive_field + out_len - 1
    rs = np.random.RandomState(3)
    ids = torch.from_numpy(rs.randint(0, 256, (1, L))); target = torch.from_numpy(rs.randint(0, 256, (out_len,)))
    x = torch.zeros(1, 256, L).scatter_(1, ids.view(1, 1, L), 1.0)
    res = {}
    for name in ("cpu32", "cpu64", "native"):
        mm = copy.deepcopy(m)
        if name == "cpu64":
            mm = mm.double(); xx = x.double()
        elif name == "native":
            mm = mm.cuda(); mm.deterministic_gradients = True; xx = x.cuda()
        else:
            xx = x
        t0 = time.time()
        out = mm(xx); loss = torch.nn.functional.cross_entropy(out, target.to(out.device)); loss.backward()
        res[name] = {k: p.grad.detach().double().cpu().numpy() for k, p in mm.named_parameters() if p.grad is not None}
        res[name]["_loss"] = float(loss.detach())
    def dev(a, b):
        worst = (0, None)
        for k in res[b]:
            if k.startswith("_"): continue
            s = np.abs(res[b][k]).max()
            e = np.abs(res[a][k] - res[b][k]).max() / s
            if e > worst[0]: worst = (e, k)
        return worst
    print("out_len %5d  loss %.6f | native vs cpu64 %.2e (%s) | cpu32 vs cpu64 %.2e (%s) | native vs cpu32 %.2e" % (
        out_len, res["native"]["_loss"], *dev("native", "cpu64"), *dev("cpu32", "cpu64"), dev("native", "cpu32")[0]), flush=True)
