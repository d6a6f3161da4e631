"""Run the gate+up GEMM a few times under each tile variant (driver for rocprofv3 --pmc passes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bagel_amd import _lib as K  # noqa: E402

variants = [int(v) for v in sys.argv[1:]] or [0, 6]
M, N, Kd = 12288, 37888, 3584
A = torch.randn((M, Kd), device="cuda:0", dtype=torch.bfloat16)
W = torch.randn((N, Kd), device="cuda:0", dtype=torch.bfloat16) * 0.02
C = torch.empty((M, N // 2), device="cuda:0", dtype=torch.bfloat16)
for v in variants:
    K.lib().bagel_set_gemm_variant(v)
    for _ in range(4):
        K.gemm(A, W, C, epilogue=K.EPI_SWIGLU)
torch.cuda.synchronize()
