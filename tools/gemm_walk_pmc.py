"""Driver for rocprofv3 --pmc FETCH_SIZE passes over the gate+up and down GEMMs of one library build (GEMM_LIB names a
build of tools/build_variant.py, default the library in the tree): fabric-side read bytes per launch as a function of
the tile walk (python tools/build_variant.py walk4 -DGEMM_GROUP_M=4, ... walk16 -DGEMM_GROUP_M=16)."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bagel_amd import _lib as K  # noqa: E402

name = os.environ.get("GEMM_LIB")
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "_ab", f"lib{name}.so")) if name else K.lib()
dev = "cuda:0"
for M, N, Kd, epi in ((12288, 37888, 3584, K.EPI_SWIGLU), (12288, 3584, 18944, K.EPI_RESIDUAL)):
    A = torch.randn((M, Kd), device=dev, dtype=torch.bfloat16)
    W = torch.randn((N, Kd), device=dev, dtype=torch.bfloat16) * 0.02
    C = torch.empty((M, N // 2 if epi == K.EPI_SWIGLU else N), device=dev, dtype=torch.bfloat16)
    R = torch.randn_like(C) if epi == K.EPI_RESIDUAL else None
    for _ in range(4):
        rc = lib.bagel_gemm_bf16(K._p(A), K.L(A.stride(0)), K._p(W), K.L(W.stride(0)), K._p(C), K.L(C.stride(0)), K.I(M), K.I(N),
                                 K.I(Kd), K._p(None), K._p(R), K.L(R.stride(0) if R is not None else 0), K.I(epi), K.F(1.0),
                                 K._stream())
        assert rc == 0
torch.cuda.synchronize()
