"""GEMM diagnostics on the generation expert's shapes (random operands), one process:
  * hipBLASLt through torch.mm as a known-good reference on the same device and data;
  * the clock the chip holds inside the pipelined kernel (diagnostic build with s_memtime / s_memrealtime stamps:
    python tools/build_variant.py clock -DBAGEL_DIAG_CLOCK);
  * the same kernel with every workgroup loading tile (0, 0), i.e. no traffic beyond L2
    (python tools/build_variant.py sametile -DBAGEL_DIAG_SAME_TILE): how much of the time / clock the
    L2-miss traffic costs.
"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bagel_amd import _lib as K  # noqa: E402

dev = "cuda:0"
SHAPES = [("gate_up", 12288, 37888, 3584, K.EPI_SWIGLU), ("down", 12288, 3584, 18944, K.EPI_RESIDUAL),
          ("qkv", 12288, 4608, 3584, K.EPI_BIAS), ("o_proj", 12288, 3584, 3584, K.EPI_RESIDUAL)]


def load(name):
    path = os.path.join(ROOT, "tools", "_ab", f"lib{name}.so")
    return ctypes.CDLL(path) if os.path.exists(path) else None


def call(lib, A, W, C, b, R, epi):
    M, Kd = A.shape
    rc = lib.bagel_gemm_bf16(K._p(A), K.L(A.stride(0)), K._p(W), K.L(W.stride(0)), K._p(C), K.L(C.stride(0)), K.I(M),
                             K.I(W.shape[0]), K.I(Kd), K._p(b), K._p(R), K.L(R.stride(0) if R is not None else 0), K.I(epi),
                             K.F(1.0), K._stream())
    assert rc == 0


def timed(fn, reps=3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    libs = {"lib": K.lib(), "clock": load("clock"), "sametile": load("sametile")}
    libs.update({n: load(n) for n in sys.argv[1:]})
    rounds = int(os.environ.get("ROUNDS", "5"))
    for name, M, N, Kd, epi in SHAPES:
        A = torch.randn((M, Kd), device=dev, dtype=torch.bfloat16)
        W = torch.randn((N, Kd), device=dev, dtype=torch.bfloat16) * 0.02
        if os.environ.get("OPERANDS") == "zeros":    # how much of the time is the power the data's switching costs
            A.zero_()
            W.zero_()
        C = torch.empty((M, N // 2 if epi == K.EPI_SWIGLU else N), device=dev, dtype=torch.bfloat16)
        R = torch.randn_like(C) if epi == K.EPI_RESIDUAL else None
        b = torch.randn((N,), device=dev, dtype=torch.bfloat16) if epi == K.EPI_BIAS else None
        Cfull = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
        flops = 2.0 * M * N * Kd
        arms = {n: (lambda l=l: call(l, A, W, C, b, R, epi)) for n, l in libs.items() if l is not None}
        arms["hipblaslt(torch.mm)"] = lambda: torch.mm(A, W.t(), out=Cfull)
        times = {n: [] for n in arms}
        for r in range(rounds + 1):
            for n, fn in arms.items():
                t = timed(fn)
                if r:
                    times[n].append(t)
        for n, t in times.items():
            t.sort()
            print(f"{name:8s} {n:22s} median {t[len(t) // 2]:.3f} ms {flops / t[len(t) // 2] / 1e9:6.0f} TFLOP/s   min {t[0]:.3f} ms", flush=True)
        if libs.get("clock") is not None:
            # after >= 2 s of back-to-back launches, stamp one launch
            lib = libs["clock"]
            for _ in range(int(2000 / max(times["clock"][0], 0.05))):
                call(lib, A, W, C, b, R, epi)
            dbg = torch.zeros(2 * 256, device=dev, dtype=torch.int64)
            lib.bagel_set_gemm_diag(ctypes.c_void_p(dbg.data_ptr()))
            call(lib, A, W, C, b, R, epi)
            torch.cuda.synchronize()
            lib.bagel_set_gemm_diag(ctypes.c_void_p(0))
            d = dbg.view(256, 2).double()
            ok = d[:, 1] > 0
            mhz = (d[ok, 0] / d[ok, 1] * 100.0)
            print(f"{name:8s} in-kernel clock: median {float(mhz.median()):.0f} MHz (min {float(mhz.min()):.0f}, max {float(mhz.max()):.0f}); "
                  f"workgroup life {float(d[ok, 1].median()) / 100.0:.0f} us", flush=True)
        del A, W, C, R, b, Cfull


if __name__ == "__main__":
    main()
