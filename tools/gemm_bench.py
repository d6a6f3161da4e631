"""Micro-benchmark of bagel_gemm_bf16 tile variants on the generation expert's shapes (random operands).

    python tools/gemm_bench.py [variant ...]      # variants are values of BAGEL_GEMM_VARIANT

Variants are interleaved round-robin in one process and timed with HIP events on the launch stream.
GEMM_AB_LIB=<path to another build of the library> adds that build's default variant as "alt" to every round.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bagel_amd import _lib as K  # noqa: E402

SHAPES = [  # name, M, N, K, epilogue
    ("gate_up", 12288, 37888, 3584, K.EPI_SWIGLU),
    ("down", 12288, 3584, 18944, K.EPI_RESIDUAL),
    ("qkv", 12288, 4608, 3584, K.EPI_BIAS),
    ("o_proj", 12288, 3584, 3584, K.EPI_RESIDUAL),
    ("gate_up_1unit", 4096, 37888, 3584, K.EPI_SWIGLU),
    ("down_1unit", 4096, 3584, 18944, K.EPI_RESIDUAL),
    ("qkv_1unit", 4096, 4608, 3584, K.EPI_BIAS),
    ("o_proj_1unit", 4096, 3584, 3584, K.EPI_RESIDUAL),
]


def main():
    variants = [int(v) for v in sys.argv[1:]] or [-1]
    dev = "cuda:0"
    lib = K.lib()
    rounds = int(os.environ.get("ROUNDS", "5"))
    alt_lib = None
    if os.environ.get("GEMM_AB_LIB"):
        import ctypes
        alt_lib = ctypes.CDLL(os.path.abspath(os.environ["GEMM_AB_LIB"]))
        variants = variants + ["alt"]

    def run(v, A, W, C, b, R, epi):
        if v == "alt":
            M, Kd = A.shape
            rc = alt_lib.bagel_gemm_bf16(K._p(A), K.L(A.stride(0)), K._p(W), K.L(W.stride(0)), K._p(C), K.L(C.stride(0)),
                                         K.I(M), K.I(W.shape[0]), K.I(Kd), K._p(b), K._p(R),
                                         K.L(R.stride(0) if R is not None else 0), K.I(epi), K.F(1.0), K._stream())
            assert rc == 0
        else:
            lib.bagel_set_gemm_variant(v)
            K.gemm(A, W, C, bias=b, residual=R, epilogue=epi)
    for name, M, N, Kd, epi in SHAPES:
        A = torch.randn((M, Kd), device=dev, dtype=torch.bfloat16)
        W = torch.randn((N, Kd), device=dev, dtype=torch.bfloat16) * 0.02
        C = torch.empty((M, N // 2 if epi == K.EPI_SWIGLU else N), device=dev, dtype=torch.bfloat16)
        R = torch.randn_like(C) if epi == K.EPI_RESIDUAL else None
        b = torch.randn((N,), device=dev, dtype=torch.bfloat16) if epi == K.EPI_BIAS else None
        flops = 2.0 * M * N * Kd
        times = {v: [] for v in variants}
        for r in range(rounds + 1):
            for v in variants:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                try:
                    for _ in range(3):
                        run(v, A, W, C, b, R, epi)
                except RuntimeError:      # this tile shape does not implement the epilogue
                    times.pop(v, None)
                    continue
                e1.record()
                torch.cuda.synchronize()
                if r > 0 and v in times:
                    times[v].append(e0.elapsed_time(e1) / 3)
        if alt_lib is not None:       # same bits from both builds?
            run(variants[0], A, W, C, b, R, epi)
            ref = C.float().clone()
            run("alt", A, W, C, b, R, epi)
            print(f"{name:14s} max |alt - lib| = {(C.float() - ref).abs().max().item():.3e}")
        for v in [v for v in variants if times.get(v)]:
            t = sorted(times[v])
            med = t[len(t) // 2]
            print(f"{name:14s} variant {v}: median {med:.3f} ms  {flops / med / 1e9:.0f} TFLOP/s   min {t[0]:.3f} ms {flops / t[0] / 1e9:.0f} TFLOP/s",
                  flush=True)
        del A, W, C, R, b
    lib.bagel_set_gemm_variant(-1)


if __name__ == "__main__":
    main()
