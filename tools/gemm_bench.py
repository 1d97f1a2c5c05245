#!/usr/bin/env python3
"""cm_gemm_h3 at the BASELINE config-4 shapes (13824 tokens, embed 256, mlp 256): forward (NT), data gradient (NN), weight
gradient (TN, split K) -- time per launch, algorithmic TFLOP/s."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from climate_amd import ops  # noqa: E402
from conv_microbench import timeit  # noqa: E402

M = 64 * 216
for name, n, k in (("in_proj", 768, 256), ("out_proj / linear1 / linear2", 256, 256), ("encoder.2 (im2col)", 256, 1152)):
    x = torch.randn(M, k, device="cuda"); w = torch.randn(n, k, device="cuda") * 0.05; dy = torch.randn(M, n, device="cuda")
    b = torch.randn(n, device="cuda")
    fl = 2.0 * M * n * k
    y = torch.empty(M, n, device="cuda")
    t = timeit(lambda: ops.gemm(x, w, M, n, k, bias=b, out=y), 20)
    print(f"{name:30s} fwd  [M={M}, N={n}, K={k}]   {t:7.1f} us  {fl / t / 1e6:6.1f} TF")
    dx = torch.empty(M, k, device="cuda")
    t = timeit(lambda: ops.gemm(dy, w, M, k, n, trans_b=True, out=dx), 20)
    print(f"{'':30s} dgrad[M={M}, N={k}, K={n}]   {t:7.1f} us  {fl / t / 1e6:6.1f} TF")
    for ks in (8, 16, 32, 54):
        dw = torch.zeros(n, k, device="cuda")
        t = timeit(lambda: ops.gemm(dy, x, n, k, M, trans_a=True, trans_b=True, out=dw, ksplit=ks), 20)
        print(f"{'':30s} wgrad[M={n}, N={k}, K={M}] ksplit {ks:2d} {t:7.1f} us  {fl / t / 1e6:6.1f} TF")
