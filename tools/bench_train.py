"""Times one training step (model(x); cross_entropy; backward; Adam step) at BASELINE config 5's shape: layers=10 blocks=5
dil/res=128 skip=512, N one-second 16 kHz mu-law clips, through the facade -- native matrix-core forward + backward
(wn_train_forward / wn_train_backward) next to the facade's torch path (MIOpen conv1d + autograd) on the same GPU.

    python tools/bench_train.py [N] [L] [--no-torch] [--reps=K]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-wavenet_amd"))
import torch  # noqa: E402

import wavenet_model  # noqa: E402

if os.environ.get("WN_DEV_LIB"):  # A/B runs: another build of the library
    from mi355_wavenet import _abi
    _abi.PRODUCT_LIB = os.path.abspath(os.environ["WN_DEV_LIB"])


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    N = int(args[0]) if len(args) > 0 else 32
    L = int(args[1]) if len(args) > 1 else 16000
    torch.manual_seed(0)
    m = wavenet_model.WaveNetModel(layers=10, blocks=5, dilation_channels=128, residual_channels=128, skip_channels=512,
                                   end_channels=256, classes=256, output_length=1, kernel_size=2, bias=False).cuda()
    m.output_length = L - m.receptive_field + 1
    out_len = m.output_length
    g = torch.Generator().manual_seed(1)
    idx = torch.randint(0, 256, (N, L), generator=g).cuda()
    x = torch.zeros(N, 256, L, device="cuda").scatter_(1, idx.unsqueeze(1), 1.0)
    target = torch.randint(0, 256, (N * out_len,), generator=g).cuda()
    from mi355_wavenet.optim import FusedAdam   # (what WavenetTrainer uses on the GPU and bench.py's training legs time; --torch-adam: torch's foreach kernels)
    opt = torch.optim.Adam(m.parameters(), lr=1e-4) if "--torch-adam" in sys.argv else FusedAdam(m.parameters(), lr=1e-4)
    R = D = 128; S = 512; E = 256; C = 256
    need, fwd = out_len, 0
    for d in reversed([2 ** (i % 10) for i in range(50)]):
        fwd += 2 * N * need * (2 * R * 2 * D + D * R) + 2 * N * out_len * D * S
        need += d
    fwd += 2 * N * out_len * (S * E + E * C)

    def step():
        opt.zero_grad(set_to_none=True)
        if "--facade-loss" in sys.argv:   # model(x) on a one-hot input and torch's cross_entropy (the reference's own lines; the one-hot scatter and torch's loss kernels are in the profile then)
            loss = torch.nn.functional.cross_entropy(m(x), target)
        else:                             # what WavenetTrainer.train_step and bench.py's training legs run: class indices in, the engine's fused loss
            from mi355_wavenet import training
            logits = m.train_forward_indices(idx)
            loss = training.cross_entropy(m._wn_train_runner, logits, target)
        loss.backward()
        opt.step()
        return loss

    def timed(label, reps):
        for _ in range(2):
            loss = step()
        torch.cuda.synchronize()
        if "--enqueue-time" in sys.argv:   # host time to ENQUEUE one step (every call returns before the GPU work is done): what a hipGraph could save at most
            t0 = time.perf_counter()
            loss = step()
            host_ms = (time.perf_counter() - t0) * 1e3
            torch.cuda.synchronize()
            print("%s: host time to enqueue one step %.2f ms" % (label, host_ms))
        t0 = time.perf_counter()
        for _ in range(reps):
            loss = step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        print("%s: %.1f ms / step (loss %.4f); forward GEMM work %.2f TFLOP, step ~3x -> %.1f TFLOP/s; %.0f clip-seconds/s" % (
            label, ms, float(loss.detach()), fwd / 1e12, 3 * fwd / ms / 1e9, N * L / 16000 / (ms * 1e-3)))
        return ms

    reps = ([int(v.split("=")[1]) for v in sys.argv if v.startswith("--reps=")] or [3])[0]
    a = timed("native fp32 matrix-core step", reps) if "--only-bf16" not in sys.argv else 0.0
    if "--only-fp32" not in sys.argv:
        m.matrix_precision = "bf16"
        timed("native step, bf16 operands for forward + activation gradients", reps)
        m.matrix_precision = "fp32"
    peak = torch.cuda.max_memory_allocated() / 2**30
    print("torch-allocated peak %.1f GiB (the native workspace is allocated by the library, not by torch)" % peak)
    if "--no-torch" not in sys.argv:
        os.environ["WN_TORCH_BACKWARD"] = "1"
        try:
            b = timed("torch path (MIOpen conv1d + autograd)", 2)
            print("speed-up %.2fx" % (b / a))
        except Exception as e:  # noqa: BLE001
            print("torch path failed:", type(e).__name__, str(e)[:200])


if __name__ == "__main__":
    main()
