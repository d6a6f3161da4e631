"""Random shapes through bagel_skinny_gemm_bf16 (every form) and bagel_gemm_ws_bf16 against an fp32 statement.

    python tools/fuzz_linear.py [cases [first_case]]
"""
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bagel_amd import _lib as K  # noqa: E402

DEV = "cuda:0"
BF16 = torch.bfloat16


def close(got, ref, what, extra=None):
    """extra: an additional per-element allowance (SwiGLU: one bf16 ulp of the gate, times the up value)."""
    g, r = got.float().cpu(), ref.float().cpu()
    assert torch.isfinite(g).all(), what
    tol = 2.0 ** -6 * r.abs().clamp(min=1.0) + 1e-2
    if extra is not None:
        tol = tol + extra.float().cpu()
    bad = (g - r).abs() > tol
    if bad.any():
        i = int(((g - r).abs() - tol).argmax())
        raise AssertionError((what, "max diff", float((g - r).abs().max()), "elements over", int(bad.sum()),
                              "worst: got", float(g.flatten()[i]), "want", float(r.flatten()[i]), "allowed", float(tol.flatten()[i])))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0    # earlier cases only advance the shape generator
    rnd = random.Random(7)
    lib = K.lib()
    for case in range(n):
        which = rnd.choice(["skinny", "skinny", "ws"])
        if which == "skinny":
            M = rnd.randint(1, 64)
            Kd = 128 * rnd.randint(1, 40)
            epi = rnd.choice([K.SK_BIAS, K.SK_RESIDUAL, K.SK_SWIGLU, K.SK_BIAS_SILU])
            N = (32 if epi == K.SK_SWIGLU else 16) * rnd.randint(1, 300)
            form = rnd.choice(["dma", "regs", "splitk", "gemv"])
        else:
            M = rnd.randint(65, 1100)
            Kd = 64 * rnd.randint(1, 120)
            epi = rnd.choice([K.EPI_BIAS, K.EPI_RESIDUAL, K.EPI_BIAS_RESIDUAL, K.EPI_BIAS_GELU, K.EPI_SWIGLU])
            N = 32 * rnd.randint(1, 300) if epi == K.EPI_SWIGLU else 4 * rnd.randint(2, 1200)
            form = "ws"
        pad_x, pad_c = rnd.choice([0, 8, 64]), rnd.choice([0, 8, 32])
        if case < first:
            continue
        g = torch.Generator(device=DEV).manual_seed(case)
        Xs = torch.randn((M, Kd + pad_x), generator=g, device=DEV).to(BF16)
        X = Xs[:, :Kd]
        W = (torch.randn((N, Kd), generator=g, device=DEV) * 0.05).to(BF16)
        ncol = N // 2 if epi == (K.SK_SWIGLU if which == "skinny" else K.EPI_SWIGLU) else N
        Cs = torch.full((M, ncol + pad_c), float("nan"), device=DEV, dtype=BF16)
        C = Cs[:, :ncol]
        b = torch.randn((N,), generator=g, device=DEV).to(BF16)
        R = torch.randn((M, ncol), generator=g, device=DEV).to(BF16)
        acc = X.float() @ W.float().t()
        extra = None
        what = f"case {case}: {which} {form} M={M} N={N} K={Kd} epi={epi} pad_x={pad_x} pad_c={pad_c}"
        if which == "skinny":
            lib.bagel_set_stream_mfma(0 if form == "splitk" else 1)
            lib.bagel_set_stream_dma(0 if form == "regs" else 1)
            ws = torch.empty((K.skinny_workspace_bytes(M, N),), dtype=torch.uint8, device=DEV)
            gemv = form == "gemv"
            if epi == K.SK_BIAS:
                K.skinny_gemm(X, W, C, ws, bias=b, epilogue=epi, gemv=gemv)
                ref = (acc + b.float()).to(BF16)
            elif epi == K.SK_RESIDUAL:
                K.skinny_gemm(X, W, C, ws, residual=R, epilogue=epi, gemv=gemv)
                ref = (acc.to(BF16).float() + R.float()).to(BF16)
                extra = 2.0 ** -7 * acc.abs()    # the product is rounded before the add: its ulp survives a cancelling residual
            elif epi == K.SK_BIAS_SILU:
                K.skinny_gemm(X, W, C, ws, bias=b, epilogue=epi, gemv=gemv)
                ref = torch.nn.functional.silu((acc + b.float()).to(BF16).float()).to(BF16)
            else:
                K.skinny_gemm(X, W, C, ws, epilogue=epi, gemv=gemv)
                a3 = acc.view(M, N // 32, 2, 16)
                gate, up = a3[:, :, 0].reshape(M, ncol).to(BF16).float(), a3[:, :, 1].reshape(M, ncol).to(BF16).float()
                ref = (torch.nn.functional.silu(gate).to(BF16).float() * up).to(BF16)
                extra = 2.0 ** -7 * gate.abs().clamp(min=1.0) * up.abs()
            lib.bagel_set_stream_mfma(1)
            lib.bagel_set_stream_dma(1)
        else:
            ws = torch.empty((max(K.gemm_workspace_bytes(M, N, Kd, epi), 16),), dtype=torch.uint8, device=DEV)
            if epi == K.EPI_BIAS:
                K.gemm(X, W, C, bias=b, epilogue=epi, workspace=ws)
                ref = (acc + b.float()).to(BF16)
            elif epi == K.EPI_RESIDUAL:
                K.gemm(X, W, C, residual=R, epilogue=epi, workspace=ws)
                ref = (acc.to(BF16).float() + R.float()).to(BF16)
                extra = 2.0 ** -7 * acc.abs()
            elif epi == K.EPI_BIAS_RESIDUAL:
                K.gemm(X, W, C, bias=b, residual=R, epilogue=epi, workspace=ws)
                ref = ((acc + b.float()).to(BF16).float() + R.float()).to(BF16)
                extra = 2.0 ** -7 * (acc + b.float()).abs()
            elif epi == K.EPI_SWIGLU:
                K.gemm(X, W, C, epilogue=epi, workspace=ws)
                a3 = acc.view(M, N // 32, 2, 16)
                gate, up = a3[:, :, 0].reshape(M, ncol).to(BF16).float(), a3[:, :, 1].reshape(M, ncol).to(BF16).float()
                ref = (torch.nn.functional.silu(gate).to(BF16).float() * up).to(BF16)
                extra = 2.0 ** -7 * gate.abs().clamp(min=1.0) * up.abs()
            else:
                K.gemm(X, W, C, bias=b, epilogue=epi, workspace=ws)
                ref = torch.nn.functional.gelu((acc + b.float()).to(BF16).float(), approximate="tanh").to(BF16)
        torch.cuda.synchronize()
        close(C, ref, what, extra)
        if pad_c:
            assert torch.isnan(Cs[:, ncol:].float()).all(), what + ": wrote past the output columns"
        if case % 20 == 0:
            print(what, "ok", flush=True)
    print(f"{n} cases ok")


if __name__ == "__main__":
    main()
