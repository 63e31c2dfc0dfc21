"""Which accumulation order does THIS host's torch CPU nn.Linear (MKL sgemm behind addmm) follow at K = 256?  The k-ascending fma chain
(csrc/reference_order*.hip) is what Intel hosts compute at every K and the GPU pool's EPYC hosts at K <= 128; at K = 256 the EPYC hosts
differ (round 5).  Candidates: K cut into blocks of kb, each block a k-ascending fma chain from zero, the block results added in order."""
import platform
import subprocess

import torch
import torch.nn.functional as F


def blocked(A, W, b, kb, bias_first=False):
    K = A.shape[1]
    tot = None
    for a0 in range(0, K, kb):
        acc = torch.zeros(A.shape[0], W.shape[0], dtype=torch.float64)
        if tot is None and bias_first:
            acc = acc + b.double()
        for k in range(a0, min(a0 + kb, K)):
            acc = (acc + A[:, k:k + 1].double() * W[:, k].double().unsqueeze(0)).float().double()
        tot = acc if tot is None else (tot + acc).float().double()
    return tot.float() if bias_first else (tot + b.double()).float()


print(torch.__version__, platform.processor(), torch.backends.cpu.get_cpu_capability(), "threads", torch.get_num_threads())
print(subprocess.run('lscpu | grep -E "Model name" | cut -c1-200', shell=True, capture_output=True, text=True).stdout.strip())
g = torch.Generator().manual_seed(0)
default_threads = torch.get_num_threads()
for M, K, N in ((64, 256, 64), (1000, 256, 256), (1000, 192, 192), (1000, 384, 128), (1000, 512, 64)):
    A, W, b = 10 * torch.randn(M, K, generator=g), 0.3 * torch.randn(N, K, generator=g), torch.randn(N, generator=g)
    ys = {}
    for threads in (default_threads, 1):
        torch.set_num_threads(threads)
        ys[threads] = F.linear(A, W, b)
    torch.set_num_threads(8)   # (the emulation below is thousands of tiny ops)
    cands = {kb: blocked(A, W, b, kb) for kb in (K, 32, 64, 96, 128, 160, 192, 224, 256, 384) if kb <= K}
    for threads, y in ys.items():
        print(f"threads={threads} M={M} K={K} N={N} mismatch vs blocked chains:", " ".join(f"kb={kb}: {(y != c).float().mean().item():.3f}" for kb, c in cands.items()), flush=True)
