import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "pytorch-wavenet_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
from mi355_wavenet import _abi, engine, synth
if os.environ.get("WN_DEV_LIB"):
    _abi.PRODUCT_LIB = os.path.abspath(os.environ["WN_DEV_LIB"])
import c_oracle
from parity_common import make_case
MINI3 = dict(synth.CONFIGS["cfg3"], layers=3, blocks=2)
ns = 6
cfg, W, first, uniforms = make_case(MINI3, 82, ns, 1, 64)
first[:] = first[0]; uniforms[:] = uniforms[0]      # identical streams: every stream must produce the same logits
eng = engine.Engine(cfg, W, n_streams=ns)
for n1 in (1, 2, 3, 5, 23):
    a = eng.generate(n1, first, temperature=1.0, uniforms=uniforms[:, :n1])
    b, lb = eng.generate(8, a[:, -1:], temperature=1.0, uniforms=uniforms[:, n1:n1 + 8], reset=False, want_logits=True)
    o, ol = c_oracle.generate(cfg, W, n1 + 8, first[0], 1.0, 0.0, uniforms[0])
    devs = [float(np.abs(lb[s][0] - ol[n1]).max()) for s in range(ns)]
    devs1 = [float(np.abs(lb[s][1] - ol[n1 + 1]).max()) for s in range(ns)]
    print("launch 1 = %2d evaluations: first-evaluation logit dev per stream %s ; second evaluation %s" % (n1, ["%.1e" % d for d in devs], ["%.1e" % d for d in devs1]))
eng.close()
