"""dev tool (GPU box): what happens to a cfg3 job launched while another process's unbooked cfg3 job holds the chip"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("pytorch-wavenet_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
os.environ["WN_TESTING"] = "1"; os.environ["WN_NO_DEVICE_GATE"] = "1"
import numpy as np, torch
from parity_common import make_case
from mi355_wavenet import engine, _abi
HOG = r'''
import os, sys, time
import numpy as np, torch
from parity_common import make_case
from mi355_wavenet import engine
cfg, W, first, uniforms = make_case("cfg3", 58, 2, 20, 8)
hog = engine.Engine(cfg, W, n_streams=2)
n = int(os.environ.get("HOG_N", "5000"))
hfirst = hog.mem.upload(np.full((2, 1), 128, dtype=np.int32)); hout = hog.mem.empty((2, n), np.int32)
hog.reset(); torch.cuda.synchronize()
t0 = time.time()
hog.launch(hfirst, 1, n, 0.0, None, None, hout, None, timeout_ms=60000)
open(sys.argv[1], "w").close()
hog.wait()
print("HOG DONE %.3f s" % (time.time() - t0), hog.info()["kernel_variant"], hog.info()["n_workgroups"], flush=True)
'''
open("/tmp/hog.py", "w").write(HOG)
env = dict(os.environ); env["WN_KERNEL"] = "generic"; env["PYTHONPATH"] = os.pathsep.join([os.path.join(ROOT, p) for p in ("pytorch-wavenet_amd", "oracle", "tests")])
N = 300
cfg, W, first, uniforms = make_case("cfg3", 58, 2, 20, N)
job = engine.Engine(cfg, W, n_streams=2)
job.generate(8, first, temperature=1.0, uniforms=uniforms[:, :8])
for resident_ms in ("150", "60000"):
    os.environ["WN_RESIDENT_TIMEOUT_MS"] = resident_ms
    if os.path.exists("/tmp/hog_ready"): os.remove("/tmp/hog_ready")
    p = subprocess.Popen([sys.executable, "/tmp/hog.py", "/tmp/hog_ready"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    while not os.path.exists("/tmp/hog_ready"): time.sleep(0.0005)
    t0 = time.time()
    try:
        job.generate(N, first, temperature=1.0, uniforms=uniforms, timeout_ms=300, batched_prime=False)
        res = "ok"
    except _abi.WnError as e:
        res = "ERR %s" % e
    dt = time.time() - t0
    so, se = p.communicate()
    print("resident_ms", resident_ms, "-> job took %.3f s:" % dt, res[:200], "|", so.strip(), se.strip()[-200:], flush=True)
    print("   info", {k: v for k, v in job.info().items() if k in ("resident_timeout_ms", "evals_done", "workgroups_per_cu", "n_workgroups")})
