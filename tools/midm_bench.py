"""bagel_gemm_bf16 at row counts between the weight-streaming kernels (<= 64 rows) and the full tiles: time per layer
shape against the time the weights alone need at 6 TB/s.

    python tools/midm_bench.py [M ...]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bagel_amd import _lib as K  # noqa: E402

USE_WS = os.environ.get("USE_WS", "1") != "0"   # 0: bagel_gemm_bf16 (no workspace, never splits K)
SHAPES = [("qkv", 4608, 3584, K.EPI_BIAS), ("o_proj", 3584, 3584, K.EPI_RESIDUAL), ("gate_up", 37888, 3584, K.EPI_SWIGLU),
          ("down", 3584, 18944, K.EPI_RESIDUAL)]


def main():
    dev = "cuda:0"
    if os.environ.get("STAGES"):    # development: LDS stages of the 128 x 128 kernel
        K.lib().bagel_set_gemm_stages(int(os.environ["STAGES"]))
    if os.environ.get("VARIANT"):   # development: force a tile variant of bagel_gemm_bf16 (disables the split-K form)
        K.lib().bagel_set_gemm_variant(int(os.environ["VARIANT"]))
    for M in [int(v) for v in sys.argv[1:]] or [96, 128, 256, 512, 1024, 2048]:
        tot = 0.0
        line = []
        for name, N, Kd, epi in SHAPES:
            A = torch.randn((M, Kd), device=dev, dtype=torch.bfloat16)
            copies = max(2, int(600e6 // (N * Kd * 2)) + 1)
            Ws = [torch.randn((N, Kd), device=dev, dtype=torch.bfloat16) * 0.02 for _ in range(copies)]
            C = torch.empty((M, N // 2 if epi == K.EPI_SWIGLU else N), device=dev, dtype=torch.bfloat16)
            R = torch.randn_like(C) if epi == K.EPI_RESIDUAL else None
            b = torch.randn((N,), device=dev, dtype=torch.bfloat16) if epi == K.EPI_BIAS else None
            ws = torch.empty((max(K.gemm_workspace_bytes(M, N, Kd, epi), 16),), device=dev, dtype=torch.uint8) if USE_WS else None
            for W in Ws:
                K.gemm(A, W, C, bias=b, residual=R, epilogue=epi, workspace=ws)
            ts = []
            for it in range(10):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                K.gemm(A, Ws[it % copies], C, bias=b, residual=R, epilogue=epi, workspace=ws)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            ts.sort()
            med = ts[len(ts) // 2] * 1e3
            tot += med
            floor = max(N * Kd * 2 / 6e12, 2.0 * M * N * Kd / 1.4e15) * 1e6
            line.append(f"{name} {med:7.1f} us (floor {floor:6.1f})")
            del Ws
        print(f"M={M:5d}: " + "  ".join(line) + f"  | layer {tot:7.1f} us", flush=True)


if __name__ == "__main__":
    main()
