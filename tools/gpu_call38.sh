#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
( time timeout 1200 python -m pytest tests/test_gpu_training.py -m gpu -q -x 2>&1 | tail -8 ) 2>&1 | tee gpurun_out/r02_pytest_training_loss.log
python - <<'PY' 2>&1 | grep -v amdgpu | tee gpurun_out/r02_train_loss_timing.txt
import sys, time, torch
sys.path.insert(0, "/root/repo/pytorch-wavenet_amd"); sys.path.insert(0, "/root/repo")
import torch.nn.functional as F
import wavenet_model
from mi355_wavenet import training
m = wavenet_model.WaveNetModel(layers=3, blocks=2, dilation_channels=32, residual_channels=32, skip_channels=64, end_channels=64, classes=256, output_length=16, kernel_size=2, bias=False).cuda()
L = m.receptive_field + m.output_length - 1
idx = torch.randint(0, 256, (2, L))
x = torch.zeros(2, 256, L).scatter_(1, idx.unsqueeze(1), 1.0).cuda()
m(x)
runner = m._wn_train_runner
M = 32 * 10885
logits = torch.randn(M, 256, device="cuda").requires_grad_(True)
tgt = torch.randint(0, 256, (M,), device="cuda")
for name, fn in (("torch F.cross_entropy + backward", lambda: F.cross_entropy(logits, tgt)), ("wn_train_loss (value + gradient)", lambda: training.cross_entropy(runner, logits, tgt))):
    for _ in range(2):
        logits.grad = None; fn().backward()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        logits.grad = None; fn().backward()
    torch.cuda.synchronize()
    print("%-36s M = %d rows x 256: %.3f ms" % (name, M, (time.perf_counter() - t0) * 100))
PY
