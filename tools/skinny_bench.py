"""Micro-benchmark of bagel_skinny_gemm_bf16 (weight streaming, HBM-bound) on the understanding expert's shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bagel_amd import _lib as K  # noqa: E402

SHAPES = [("qkv", 4608, 3584, K.SK_BIAS), ("o_proj", 3584, 3584, K.SK_RESIDUAL), ("gate_up", 37888, 3584, K.SK_SWIGLU),
          ("down", 3584, 18944, K.SK_RESIDUAL), ("lm_head", 152064, 3584, K.SK_BIAS)]


def main():
    dev = "cuda:0"
    forms = [int(v) for v in os.environ.get("FORMS", "0,1,2").split(",")]
    # 0: split-K MFMA + reduce launch, 1: one-launch v_dot2c form (a wave owns whole rows), 2: one-launch MFMA,
    # 3: one-launch MFMA with the weights brought in by LDS-DMA
    for form in forms:
      gemv = form == 1
      K.lib().bagel_set_stream_mfma(1 if form >= 2 else 0)
      K.lib().bagel_set_stream_dma(1 if form == 3 else 0)
      print(f"--- form {form}")
      for M in [int(v) for v in sys.argv[1:]] or [6, 1]:
          total_t = total_b = 0.0
          for name, N, Kd, epi in SHAPES:
              X = torch.randn((M, Kd), device=dev, dtype=torch.bfloat16)
              # several copies of the weight so that successive launches do not hit the 256 MB Infinity Cache
              copies = max(2, int(600e6 // (N * Kd * 2)) + 1)
              Ws = [torch.randn((N, Kd), device=dev, dtype=torch.bfloat16) * 0.02 for _ in range(copies)]
              ncol = N // 2 if epi == K.SK_SWIGLU else N
              C = torch.empty((M, ncol), device=dev, dtype=torch.bfloat16)
              R = torch.randn((M, ncol), device=dev, dtype=torch.bfloat16) if epi == K.SK_RESIDUAL else None
              b = torch.randn((N,), device=dev, dtype=torch.bfloat16) if epi == K.SK_BIAS else None
              ws = torch.empty((16 * M * N,), device=dev, dtype=torch.float32)
              for W in Ws:
                  K.skinny_gemm(X, W, C, ws, bias=b, residual=R, epilogue=epi, gemv=bool(gemv))
              if form == 3:
                  K.lib().bagel_set_stream_dma(0)
                  C2 = torch.empty_like(C)
                  K.skinny_gemm(X, Ws[-1], C2, ws, bias=b, residual=R, epilogue=epi)
                  K.lib().bagel_set_stream_dma(1)
                  torch.cuda.synchronize()
                  assert torch.equal(C, C2), (name, M, float((C.float() - C2.float()).abs().max()))
              times = []
              for it in range(12):
                  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                  e0.record()
                  K.skinny_gemm(X, Ws[it % copies], C, ws, bias=b, residual=R, epilogue=epi, gemv=bool(gemv))
                  e1.record()
                  torch.cuda.synchronize()
                  times.append(e0.elapsed_time(e1))
              times.sort()
              med = times[len(times) // 2]
              nbytes = N * Kd * 2
              if name != "lm_head":
                  total_t += med
                  total_b += nbytes
              print(f"M={M:2d} {name:8s} N={N:6d} K={Kd:5d}: median {med * 1e3:7.1f} us  {nbytes / med / 1e6:6.0f} GB/s", flush=True)
              del Ws
          print(f"M={M:2d} one layer (4 projections): {total_t * 1e3:.1f} us  {total_b / total_t / 1e6:.0f} GB/s", flush=True)


if __name__ == "__main__":
    main()
