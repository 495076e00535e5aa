"""dev tool: what the vendor GEMM (torch.mm -> hipBLASLt / rocBLAS) does on the three grouped skip products of the config-5 training step
(M = 32 x 10885 skip rows, block of 10 layers): forward skip (K = 1280, N = 512), dzg (K = 512, N = 1280), skip weight gradient (contraction over the rows).
The step's own kernels take 0.85-0.97 ms each stand-alone (profiles/r05_training_step_byte_cuts.txt)."""
import time
import torch

M = 32 * 10885
dev = "cuda"
def bench(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

for dt in (torch.bfloat16,):
    zg = torch.randn(M, 1280, device=dev, dtype=dt)
    wsk = torch.randn(512, 1280, device=dev, dtype=dt)      # [S][G*D]
    dskip = torch.randn(M, 512, device=dev, dtype=dt)
    skip32 = torch.zeros(M, 512, device=dev, dtype=torch.float32)
    out = torch.empty(M, 512, device=dev, dtype=dt)
    print("forward skip  zg[M,1280] . W^T -> [M,512]  : %.3f ms (bf16 out)" % bench(lambda: torch.mm(zg, wsk.t(), out=out)))
    o2 = torch.empty(M, 1280, device=dev, dtype=dt)
    print("dzg           dskip[M,512] . W -> [M,1280]  : %.3f ms (bf16 out)" % bench(lambda: torch.mm(dskip, wsk, out=o2)))
    o3 = torch.empty(512, 1280, device=dev, dtype=dt)
    print("dWskip        dskip^T[512,M] . zg[M,1280]   : %.3f ms" % bench(lambda: torch.mm(dskip.t(), zg, out=o3)))
    # with an fp32 accumulate of the output (the forward product adds into the fp32 skip sum): addmm in fp32 is not a bf16 GEMM; time the pure streams instead
    print("stream: read + write an fp32 [M,512] matrix   : %.3f ms" % bench(lambda: skip32.add_(1.0)))
