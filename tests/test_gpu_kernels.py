"""Kernel-level parity: each C-ABI entry point against the CPU checker's restatement of the same op.

bf16 outputs are compared after both sides round at the same points, so the tolerance is a couple of bf16
ulps at the largest magnitude plus fp32 accumulation-order noise.
"""
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import bagel_oracle as O  # noqa: E402

BF16 = torch.bfloat16
DEV = "cuda:0"


def _K():
    from bagel_amd import _lib
    return _lib


def _rand(shape, std=1.0, seed=0, dtype=BF16):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * std).to(dtype)


def _close(got, ref, ulps=2.0, what="", mag=0.0):
    """``mag``: largest magnitude of an intermediate that is rounded to bf16 on the way (a product rounded before a
    residual is added can be larger than any output, and its ulp survives the add)."""
    got, ref = got.float().cpu(), ref.float().cpu()
    # a bf16 ulp at the largest magnitude (spacing 2^(e-7) in [2^e, 2^(e+1))): two roundings that both fall on
    # ties move a value by two of them (seen: 8.65625 -> 8.625 / 8.6875, then 11.03125 / 11.09375 -> 11.0 / 11.125)
    scale = max(float(ref.abs().max()), float(mag), 1e-30)
    tol = ulps * 2.0 ** (math.floor(math.log2(scale)) - 7) + 1e-6
    err = float((got - ref).abs().max())
    assert err <= tol, f"{what}: max-abs {err} > {tol}"
    # and almost all elements must agree to one ulp of their own magnitude
    rel = (got - ref).abs() <= (2.0 ** -7 * ref.abs() + 1e-3 * float(ref.abs().max()))
    assert rel.float().mean() > 0.999, f"{what}: only {rel.float().mean():.4f} of elements within 1 ulp"


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 512, 256), (130, 384, 192), (4096, 4608, 3584),
                                   (64, 64, 3584), (1000, 3584, 64), (4098, 64, 256)])
def test_gemm_bias(M, N, K):
    k = _K()
    A, W, b = _rand((M, K), 1.0, 1), _rand((N, K), 0.05, 2), _rand((N,), 0.5, 3)
    C = torch.full((M, N), float("nan"), dtype=BF16, device=DEV)
    k.gemm(A.to(DEV), W.to(DEV), C, bias=b.to(DEV))
    ref = (A.float() @ W.float().t() + b.float()).to(BF16)
    _close(C, ref, what=f"gemm bias {M}x{N}x{K}")
    C2 = torch.full((M, N), float("nan"), dtype=BF16, device=DEV)
    k.gemm(A.to(DEV), W.to(DEV), C2)
    _close(C2, (A.float() @ W.float().t()).to(BF16), what="gemm nobias")


def test_gemm_asymmetric_layout():
    """A = identity against an asymmetric W: catches a transposed or permuted C write."""
    k = _K()
    n = 256
    A = torch.eye(n, dtype=BF16)
    W = (torch.arange(n * n).reshape(n, n) % 251).to(BF16) / 8
    C = torch.empty((n, n), dtype=BF16, device=DEV)
    k.gemm(A.to(DEV), W.to(DEV), C)
    assert torch.equal(C.cpu(), W.t().contiguous())


@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (300, 3584, 512), (4096, 3584, 3584)])
def test_gemm_residual_inplace(M, N, K):
    k = _K()
    A, W, R = _rand((M, K), 1.0, 4), _rand((N, K), 0.05, 5), _rand((M, N), 1.0, 6)
    X = R.to(DEV).clone()
    k.gemm(A.to(DEV), W.to(DEV), X, residual=X, epilogue=k.EPI_RESIDUAL)
    ref = ((A.float() @ W.float().t()).to(BF16).float() + R.float()).to(BF16)
    _close(X, ref, what="gemm residual")


@pytest.mark.parametrize("M,I,K", [(128, 512, 256), (200, 2304, 1792), (4096, 18944, 3584)])
def test_gemm_swiglu(M, I, K):
    k = _K()
    from bagel_amd.engine import _interleave_gate_up
    A, G, U = _rand((M, K), 1.0, 7), _rand((I, K), 0.03, 8), _rand((I, K), 0.03, 9)
    Wgu = _interleave_gate_up(G, U).to(DEV)
    C = torch.empty((M, I), dtype=BF16, device=DEV)
    k.gemm(A.to(DEV), Wgu, C, epilogue=k.EPI_SWIGLU)
    g = (A.float() @ G.float().t()).to(BF16)
    u = (A.float() @ U.float().t()).to(BF16)
    ref = torch.nn.functional.silu(g) * u
    _close(C, ref, ulps=3.0, what="gemm swiglu")


@pytest.mark.parametrize("N,Kd,epi", [(3584, 3584, "residual"), (4608, 3584, "bias"), (3584, 18944, "residual"),
                                      (1152, 4352, "bias_residual"), (4304, 1152, "bias_gelu"), (528, 320, "bias")])
def test_gemm_split_k_mid_rows(N, Kd, epi):
    """bagel_gemm_ws_bf16 at 17 ... 512 rows: the split-K form against the one-pass kernel and the fp32 statement, and a
    row's bits do not depend on how many rows share the launch."""
    k = _K()
    g = torch.Generator(device=DEV).manual_seed(23)
    Mmax = 512
    X = torch.randn((Mmax, Kd), generator=g, device=DEV).to(BF16)
    W = (torch.randn((N, Kd), generator=g, device=DEV) * 0.03).to(BF16)
    b = torch.randn((N,), generator=g, device=DEV).to(BF16) if "bias" in epi else None
    R = torch.randn((Mmax, N), generator=g, device=DEV).to(BF16) if "residual" in epi else None
    code = {"bias": k.EPI_BIAS, "residual": k.EPI_RESIDUAL, "bias_residual": k.EPI_BIAS_RESIDUAL, "bias_gelu": k.EPI_BIAS_GELU}[epi]
    need = k.gemm_workspace_bytes(Mmax, N, Kd, code)
    ws = torch.empty((max(need, 16),), dtype=torch.uint8, device=DEV)
    full = torch.full((Mmax, N), float("nan"), dtype=BF16, device=DEV)
    k.gemm(X, W, full, bias=b, residual=R, epilogue=code, workspace=ws)
    plain = torch.empty((Mmax, N), dtype=BF16, device=DEV)
    k.gemm(X, W, plain, bias=b, residual=R, epilogue=code)
    if need:
        assert not torch.equal(full, plain) or Kd <= 512   # another summation order (the split was taken)
    acc = X.float() @ W.float().t()
    if epi == "bias":
        ref = (acc + b.float()).to(BF16)
    elif epi == "residual":
        ref = (acc.to(BF16).float() + R.float()).to(BF16)
    elif epi == "bias_residual":
        ref = ((acc + b.float()).to(BF16).float() + R.float()).to(BF16)
    else:
        ref = torch.nn.functional.gelu((acc + b.float()).to(BF16).float(), approximate="tanh").to(BF16)
    _close(full, ref, ulps=3.0 if epi == "bias_gelu" else 2.0, what=f"split-K gemm {epi}")
    for rows in (17, 33, 65, 100, 128, 300):
        C = torch.empty((rows, N), dtype=BF16, device=DEV)
        k.gemm(X[:rows], W, C, bias=b, residual=None if R is None else R[:rows], epilogue=code, workspace=ws)
        assert torch.equal(C, full[:rows]), f"{rows} rows differ from the same rows of a {Mmax}-row launch"
    # in place (the residual IS the output) and without a workspace large enough (falls back to the one-pass kernel)
    if epi == "residual":
        Xr = R[:200].clone()
        k.gemm(X[:200], W, Xr, residual=Xr, epilogue=code, workspace=ws)
        assert torch.equal(Xr, full[:200])
    small = torch.empty((64,), dtype=torch.uint8, device=DEV)
    C = torch.empty((200, N), dtype=BF16, device=DEV)
    k.gemm(X[:200], W, C, bias=b, residual=None if R is None else R[:200], epilogue=code, workspace=small)
    assert torch.equal(C, plain[:200])
    # the second row class, 513 ... 1024 rows: split only on long reductions, again the same for every row count
    if Kd >= 8192:
        X2 = torch.randn((1024, Kd), generator=g, device=DEV).to(BF16)
        R2 = torch.randn((1024, N), generator=g, device=DEV).to(BF16)
        need2 = k.gemm_workspace_bytes(1024, N, Kd, code)
        assert need2 > 0
        ws2 = torch.empty((need2,), dtype=torch.uint8, device=DEV)
        full2 = torch.empty((1024, N), dtype=BF16, device=DEV)
        k.gemm(X2, W, full2, residual=R2, epilogue=code, workspace=ws2)
        _close(full2, ((X2.float() @ W.float().t()).to(BF16).float() + R2.float()).to(BF16), what="split-K gemm, 1024 rows")
        C = torch.empty((700, N), dtype=BF16, device=DEV)
        k.gemm(X2[:700], W, C, residual=R2[:700], epilogue=code, workspace=ws2)
        assert torch.equal(C, full2[:700])
    assert k.gemm_workspace_bytes(1025, N, Kd, code) == 0 and k.gemm_workspace_bytes(16, N, Kd, code) == 0


@pytest.mark.parametrize("M,N,Kd", [(40, 3584, 3584), (200, 3584, 18944), (700, 3584, 18944), (300, 1024, 256), (2000, 3584, 3584)])
def test_gemm_residual_rmsnorm_matches_two_step(M, N, Kd):
    """bagel_gemm_ws_residual_rmsnorm_bf16 against bagel_gemm_ws_bf16 + bagel_rmsnorm_bf16: the same C bit for bit (the
    partial tiles are added in the same order), the normalised rows within an ulp (the row's sum of squares is added
    in another order)."""
    k = _K()
    g = torch.Generator(device=DEV).manual_seed(31)
    X = torch.randn((M, Kd), generator=g, device=DEV).to(BF16)
    W = (torch.randn((N, Kd), generator=g, device=DEV) * 0.03).to(BF16)
    R = torch.randn((M, N), generator=g, device=DEV).to(BF16)
    nw = (torch.randn((N,), generator=g, device=DEV) * 0.1 + 1).to(BF16)
    ws = torch.empty((max(k.gemm_workspace_bytes(M, N, Kd, k.EPI_RESIDUAL), 16),), dtype=torch.uint8, device=DEV)
    C1, Xn1 = torch.empty((M, N), dtype=BF16, device=DEV), torch.empty((M, N), dtype=BF16, device=DEV)
    k.gemm(X, W, C1, residual=R, epilogue=k.EPI_RESIDUAL, workspace=ws)
    k.rmsnorm(C1, Xn1, nw, nw, M, 1e-6)
    C2 = R.clone()                          # in place: the residual is the output
    Xn2 = torch.full((M, N), float("nan"), dtype=BF16, device=DEV)
    k.gemm_residual_rmsnorm(X, W, C2, C2, nw, Xn2, 1e-6, ws)
    assert torch.equal(C1, C2)
    _close(Xn2, Xn1, ulps=1.0, what="fused reduce + residual + RMSNorm")


def test_gemm_random_shapes():
    """bagel_gemm_bf16 with the launcher's own tile choice over random row counts (ragged last tiles in both directions),
    widths, depths and every epilogue, against an fp32 product on the device."""
    import random
    k = _K()
    rnd = random.Random(5)
    for case in range(int(os.environ.get("BAGEL_FUZZ_CASES", 30))):      # a longer soak: BAGEL_FUZZ_CASES=500
        if case and case % 100 == 0:
            print(f"  random GEMM shapes: {case}", flush=True)
        M = rnd.choice([1, 7, 130, 255, 257, 1000, 4098, 6001, 12294])
        Kd = 64 * rnd.randint(1, 40)
        epi = rnd.choice(["bias", "residual", "swiglu", "bias_residual", "f32", "bias_gelu"])
        N = 32 * rnd.randint(1, 150) if epi == "swiglu" else 4 * rnd.randint(1, 1300)
        g = torch.Generator(device=DEV).manual_seed(200 + case)
        X = torch.randn((M, Kd), generator=g, device=DEV).to(BF16)
        W = (torch.randn((N, Kd), generator=g, device=DEV) * 0.05).to(BF16)
        b = torch.randn((N,), generator=g, device=DEV).to(BF16)
        ncol = N // 2 if epi == "swiglu" else N
        R = torch.randn((M, ncol), generator=g, device=DEV).to(BF16)
        acc = X.float() @ W.float().t()
        what = f"case {case}: M={M} N={N} K={Kd} {epi}"
        mag = 0.0
        if epi == "f32":
            C = torch.full((M, N), float("nan"), dtype=torch.float32, device=DEV)
            k.gemm(X, W, C, epilogue=k.EPI_F32_SCALED, alpha=0.25)
            assert torch.allclose(C, 0.25 * acc, rtol=1e-4, atol=1e-3), what
            continue
        C = torch.full((M, ncol), float("nan"), dtype=BF16, device=DEV)
        if epi == "bias":
            k.gemm(X, W, C, bias=b)
            ref = (acc + b.float()).to(BF16)
        elif epi == "residual":
            k.gemm(X, W, C, residual=R, epilogue=k.EPI_RESIDUAL)
            ref = (acc.to(BF16).float() + R.float()).to(BF16)
            mag = float(acc.abs().max())
        elif epi == "bias_residual":
            k.gemm(X, W, C, bias=b, residual=R, epilogue=k.EPI_BIAS_RESIDUAL)
            ref = ((acc + b.float()).to(BF16).float() + R.float()).to(BF16)
            mag = float((acc + b.float()).abs().max())
        elif epi == "bias_gelu":
            k.gemm(X, W, C, bias=b, epilogue=k.EPI_BIAS_GELU)
            ref = torch.nn.functional.gelu((acc + b.float()).to(BF16).float(), approximate="tanh").to(BF16)
        else:
            k.gemm(X, W, C, epilogue=k.EPI_SWIGLU)
            a3 = acc.view(M, N // 32, 2, 16)
            gate, up = a3[:, :, 0].reshape(M, ncol).to(BF16).float(), a3[:, :, 1].reshape(M, ncol).to(BF16).float()
            ref = (torch.nn.functional.silu(gate).to(BF16).float() * up).to(BF16)
        _close(C, ref, ulps=4.0 if epi in ("swiglu", "bias_gelu") else 2.0, what=what, mag=mag)


def test_linear_entry_points_random_shapes(monkeypatch):
    """Random row counts, widths, depths, epilogues, row strides and forms through bagel_skinny_gemm_bf16 /
    bagel_skinny_gemv_bf16 / bagel_gemm_ws_bf16 against the fp32 statement; columns past the output stay untouched."""
    import sys
    from tools import fuzz_linear
    monkeypatch.setattr(sys, "argv", ["fuzz_linear.py", "150"])
    fuzz_linear.main()


@pytest.fixture(params=["one_launch", "one_launch_registers", "split_k"])
def skinny_form(request):
    """The forms of bagel_skinny_gemm_bf16: the one-launch MFMA kernel (default; it brings the weights in by LDS-DMA
    where its rings fit), the same with the weights loaded to registers at every row count, and the split-K + reduce
    fallback."""
    lib = _K().lib()
    lib.bagel_set_stream_mfma(0 if request.param == "split_k" else 1)
    lib.bagel_set_stream_dma(0 if request.param == "one_launch_registers" else 1)
    yield request.param
    lib.bagel_set_stream_mfma(1)
    lib.bagel_set_stream_dma(1)


@pytest.mark.parametrize("M", [1, 2, 6, 16, 17, 48, 64])
@pytest.mark.parametrize("N,K", [(512, 256), (4608, 3584), (3584, 18944)])
def test_skinny_gemm(M, N, K, skinny_form):
    k = _K()
    X, W, b = _rand((M, K), 1.0, 10), _rand((N, K), 0.03, 11), _rand((N,), 0.5, 12)
    ws = torch.empty(k.skinny_workspace_bytes(M, N), dtype=torch.uint8, device=DEV)
    C = torch.full((M, N), float("nan"), dtype=BF16, device=DEV)
    k.skinny_gemm(X.to(DEV), W.to(DEV), C, ws, bias=b.to(DEV))
    _close(C, (X.float() @ W.float().t() + b.float()).to(BF16), what="skinny bias")
    R = _rand((M, N), 1.0, 13)
    Xr = R.to(DEV).clone()
    k.skinny_gemm(X.to(DEV), W.to(DEV), Xr, ws, residual=Xr, epilogue=k.SK_RESIDUAL)
    _close(Xr, ((X.float() @ W.float().t()).to(BF16).float() + R.float()).to(BF16), what="skinny residual")
    C3 = torch.empty((M, N), dtype=BF16, device=DEV)
    k.skinny_gemm(X.to(DEV), W.to(DEV), C3, ws, bias=b.to(DEV), epilogue=k.SK_BIAS_SILU)
    _close(C3, torch.nn.functional.silu((X.float() @ W.float().t() + b.float()).to(BF16)), ulps=3.0, what="skinny silu")


@pytest.mark.parametrize("N,Kd,epi", [(3584, 3584, "bias"), (3584, 18944, "residual"), (4608, 1792, "swiglu"),
                                      (152064, 512, "bias")])
def test_skinny_gemm_row_count_invariance(N, Kd, epi):
    """bagel_skinny_gemm_bf16: a row's result does not depend on how many rows share the launch (the packed CFG branches
    and the packed batch rely on it) - across the kernel's row-group variants and its wide form for the vocabulary."""
    k = _K()
    g = torch.Generator(device=DEV).manual_seed(17)
    X = torch.randn((64, Kd), generator=g, device=DEV).to(BF16)
    W = (torch.randn((N, Kd), generator=g, device=DEV) * 0.03).to(BF16)
    b = torch.randn((N,), generator=g, device=DEV).to(BF16) if epi == "bias" else None
    ncol = N // 2 if epi == "swiglu" else N
    R = torch.randn((64, ncol), generator=g, device=DEV).to(BF16) if epi == "residual" else None
    code = {"bias": k.SK_BIAS, "residual": k.SK_RESIDUAL, "swiglu": k.SK_SWIGLU}[epi]
    ws = torch.empty(k.skinny_workspace_bytes(64, N), dtype=torch.uint8, device=DEV)
    full = torch.empty((64, ncol), dtype=BF16, device=DEV)
    k.skinny_gemm(X, W, full, ws, bias=b, residual=R, epilogue=code)
    ref = X.float() @ W.float().t()
    if epi == "bias":
        _close(full, (ref + b.float()).to(BF16), what="row-count invariance reference")
    for rows in (1, 2, 6, 9, 16, 17, 33, 48):
        C = torch.empty((rows, ncol), dtype=BF16, device=DEV)
        k.skinny_gemm(X[:rows], W, C, ws, bias=b, residual=None if R is None else R[:rows], epilogue=code)
        assert torch.equal(C, full[:rows]), f"{rows} rows differ from the same rows of a 64-row launch"
    # the weights come in by LDS-DMA (SwiGLU above 16 rows and the widest projections aside): the register form, too
    k.lib().bagel_set_stream_dma(0)
    try:
        for rows in (3, 16, 24, 40, 64):
            C = torch.empty((rows, ncol), dtype=BF16, device=DEV)
            k.skinny_gemm(X[:rows], W, C, ws, bias=b, residual=None if R is None else R[:rows], epilogue=code)
            assert torch.equal(C, full[:rows]), f"{rows} rows (register form) differ from a 64-row launch"
    finally:
        k.lib().bagel_set_stream_dma(1)


@pytest.mark.parametrize("M", [6, 20, 64])
def test_skinny_swiglu(M, skinny_form):
    k = _K()
    from bagel_amd.engine import _interleave_gate_up
    I, Kd = 2304, 1792
    X, G, U = _rand((M, Kd), 1.0, 14), _rand((I, Kd), 0.03, 15), _rand((I, Kd), 0.03, 16)
    ws = torch.empty(k.skinny_workspace_bytes(M, 2 * I), dtype=torch.uint8, device=DEV)
    C = torch.empty((M, I), dtype=BF16, device=DEV)
    k.skinny_gemm(X.to(DEV), _interleave_gate_up(G, U).to(DEV), C, ws, epilogue=k.SK_SWIGLU)
    ref = torch.nn.functional.silu((X.float() @ G.float().t()).to(BF16)) * (X.float() @ U.float().t()).to(BF16)
    _close(C, ref, ulps=3.0, what="skinny swiglu")


@pytest.mark.parametrize("M", [1, 2, 3, 4, 6])
@pytest.mark.parametrize("N,K", [(512, 512), (4608, 3584), (3584, 18944), (1008, 1024)])
def test_skinny_gemv_shapes(M, N, K):
    """Shapes with K a multiple of 512: at one to four rows these take the one-launch form (a wave owns whole weight
    rows), above that the split-K form; both must agree with the same fp32 statement."""
    k = _K()
    X, W, b = _rand((M, K), 1.0, 40), _rand((N, K), 0.03, 41), _rand((N,), 0.5, 42)
    ws = torch.empty(k.skinny_workspace_bytes(M, N), dtype=torch.uint8, device=DEV)
    C = torch.full((M, N), float("nan"), dtype=BF16, device=DEV)
    k.skinny_gemm(X.to(DEV), W.to(DEV), C, ws, bias=b.to(DEV), gemv=True)
    _close(C, (X.float() @ W.float().t() + b.float()).to(BF16), what="gemv bias")
    R = _rand((M, N), 1.0, 43)
    Xr = R.to(DEV).clone()
    k.skinny_gemm(X.to(DEV), W.to(DEV), Xr, ws, residual=Xr, epilogue=k.SK_RESIDUAL, gemv=True)
    _close(Xr, ((X.float() @ W.float().t()).to(BF16).float() + R.float()).to(BF16), what="gemv residual")


@pytest.mark.parametrize("M", [1, 2, 4, 6])
@pytest.mark.parametrize("epi", ["bias", "swiglu", "bias_silu"])
def test_skinny_gemm_rmsnorm_in(M, epi):
    """RMSNorm applied inside the weight-streaming kernel against the two-launch statement of the same thing
    (bagel_rmsnorm_bf16, then bagel_skinny_gemm_bf16): the normalised rows round to bf16 at the same two points,
    so only the fp32 summation order differs."""
    k = _K()
    from bagel_amd.engine import _interleave_gate_up
    Kd, N = 3584, 4608
    X = _rand((M, Kd), 3.0, 50).to(DEV)
    nw = (_rand((Kd,), 0.1, 51) + 1).to(BF16).to(DEV)
    if epi == "swiglu":
        W = _interleave_gate_up(_rand((N // 2, Kd), 0.03, 52), _rand((N // 2, Kd), 0.03, 53)).to(DEV)
        code, b, ncol = k.SK_SWIGLU, None, N // 2
    else:
        W, b, ncol = _rand((N, Kd), 0.03, 52).to(DEV), _rand((N,), 0.5, 54).to(DEV), N
        code = k.SK_BIAS if epi == "bias" else k.SK_BIAS_SILU
    ws = torch.empty(k.skinny_workspace_bytes(M, N), dtype=torch.uint8, device=DEV)
    Xn = torch.empty((M, Kd), dtype=BF16, device=DEV)
    ref = torch.empty((M, ncol), dtype=BF16, device=DEV)
    k.rmsnorm(X, Xn, nw, nw, M, 1e-6)
    k.skinny_gemm(Xn, W, ref, ws, bias=b, epilogue=code)
    got = torch.full((M, ncol), float("nan"), dtype=BF16, device=DEV)
    scratch = torch.empty((M, Kd), dtype=BF16, device=DEV)
    k.skinny_gemm_rmsnorm_in(X, nw, 1e-6, W, got, ws, Xn=scratch, bias=b, epilogue=code)
    _close(got, ref, ulps=3.0 if epi != "bias" else 2.0, what=f"rmsnorm-in {epi} M={M}")
    # and against the checker's own RMSNorm + fp32 matmul
    xn = O.rms_norm(X.cpu(), nw.cpu(), 1e-6).float()
    if epi == "bias":
        _close(got, (xn @ W.cpu().float().t() + b.cpu().float()).to(BF16), what="rmsnorm-in vs checker")


@pytest.mark.parametrize("rows,H,split", [(5, 256, 3), (4098, 3584, 4096), (130, 896, 0)])
def test_rmsnorm(rows, H, split):
    k = _K()
    x, w1, w2 = _rand((rows, H), 2.0, 20), _rand((H,), 0.1, 21) + 1, _rand((H,), 0.1, 22) + 1
    w1, w2 = w1.to(BF16), w2.to(BF16)
    out = torch.empty((rows, H), dtype=BF16, device=DEV)
    k.rmsnorm(x.to(DEV), out, w1.to(DEV), w2.to(DEV), split, 1e-6)
    ref = torch.cat([O.rms_norm(x[:split], w1, 1e-6), O.rms_norm(x[split:], w2, 1e-6)])
    _close(out, ref, what="rmsnorm")


@pytest.mark.parametrize("und_rounding", [False, True])
def test_qk_norm_rope_kv(und_rounding):
    k = _K()
    M, Hq, Hkv, D, split, S = 70, 4, 2, 128, 64, 192
    qkv = _rand((M, (Hq + 2 * Hkv) * D), 1.5, 30)
    qn = [(_rand((D,), 0.1, 31 + i) + 1).to(BF16) for i in range(2)]
    kn = [(_rand((D,), 0.1, 33 + i) + 1).to(BF16) for i in range(2)]
    pos = torch.tensor([3, 1000, 77])
    cos, sin = O.rotary_cos_sin(pos, D, 1e6, BF16)
    row_cs = (torch.arange(M) % 3).to(torch.int32)
    row_slot = torch.randperm(S, generator=torch.Generator().manual_seed(5))[:M].to(torch.int32)
    Q = torch.empty((M, Hq, D), dtype=BF16, device=DEV)
    Kws = torch.zeros((Hkv, S, D), dtype=BF16, device=DEV)
    Vt = torch.zeros((Hkv, D, S), dtype=BF16, device=DEV)
    k.qk_norm_rope_kv(qkv.to(DEV), M, Hq, Hkv, D, qn[0].to(DEV), qn[1].to(DEV), kn[0].to(DEV), kn[1].to(DEV), split,
                      cos.to(DEV), sin.to(DEV), row_cs.to(DEV), row_slot.to(DEV), Q, Kws, Vt, S, 1e-6, und_rounding)
    q = qkv[:, :Hq * D].view(M, Hq, D)
    kk = qkv[:, Hq * D:(Hq + Hkv) * D].view(M, Hkv, D)
    v = qkv[:, (Hq + Hkv) * D:].view(M, Hkv, D)
    if not und_rounding:
        q, kk = q.float(), kk.float()
    qn_ = torch.cat([O.rms_norm(q[:split], qn[0], 1e-6), O.rms_norm(q[split:], qn[1], 1e-6)])
    kn_ = torch.cat([O.rms_norm(kk[:split], kn[0], 1e-6), O.rms_norm(kk[split:], kn[1], 1e-6)])
    qe, ke = O.apply_rope(qn_, kn_, cos[row_cs.long()], sin[row_cs.long()])
    _close(Q, qe.to(BF16), what="q")
    rs = row_slot.long()
    _close(Kws.cpu()[:, rs, :].permute(1, 0, 2), ke.to(BF16), what="k scatter")
    assert torch.equal(Vt.cpu()[:, :, rs].permute(2, 0, 1), v), "v transposed scatter"
    untouched = torch.ones(S, dtype=torch.bool)
    untouched[rs] = False
    assert float(Kws.cpu()[:, untouched].abs().max()) == 0.0 and float(Vt.cpu()[:, :, untouched].abs().max()) == 0.0


def _attention_case(q_lens, ctx_lens, Hq, Hkv, seed, spike=False, causal=False, D=128, form=-1, mfma16=0):
    """Build Q and a KV workspace with per-sample [context | queries] ranges and the tile table.
    ``form``: -1 lets the launcher choose by launch size, 0 / 1 force the kernel's throughput form (single LDS
    buffers, three workgroups per CU) / latency form (double buffers).  ``mfma16``: 1 runs the head_dim-128 kernel
    built on the 16 x 16 x 32 matrix instruction (a measured alternative, not the default)."""
    k = _K()
    g = torch.Generator().manual_seed(seed)
    M = sum(q_lens)
    kv_lens = [a + b for a, b in zip(q_lens, ctx_lens)]
    starts, s = [], 0
    for n in kv_lens:
        starts.append(s)
        s += (n + 63) // 64 * 64
    S = s
    Q = torch.randn(M, Hq, D, generator=g).to(BF16) * 2
    Kf = torch.randn(Hkv, S, D, generator=g).to(BF16) * 2
    Vf = torch.randn(Hkv, S, D, generator=g).to(BF16)
    if spike:  # one key much closer to one query: forces a late jump of the running maximum
        Kf[:, starts[0] + kv_lens[0] - 3] = Q[5, 0] * 3
    tiles, r0 = [], 0
    for n, st, kl in zip(q_lens, starts, kv_lens):
        for q0 in range(0, n, 128):
            tiles.append((r0 + q0, min(128, n - q0), st, kl, (kl - n + q0) if causal else -1, 0, 0, 0))
        r0 += n
    Od = torch.full((M, Hq, D), float("nan"), dtype=BF16, device=DEV)
    k.lib().bagel_set_attn_deep(form)
    k.lib().bagel_set_attn_mfma16(mfma16)
    try:
        k.varlen_attention(Q.to(DEV), Kf.to(DEV), Vf.permute(0, 2, 1).contiguous().to(DEV), Od,
                           torch.tensor(tiles, dtype=torch.int32).to(DEV), len(tiles), Hq, Hkv, D, S)
    finally:
        k.lib().bagel_set_attn_deep(-1)
        k.lib().bagel_set_attn_mfma16(0)
    kk = torch.cat([Kf[:, st:st + kl].permute(1, 0, 2) for st, kl in zip(starts, kv_lens)])
    vv = torch.cat([Vf[:, st:st + kl].permute(1, 0, 2) for st, kl in zip(starts, kv_lens)])
    ref = O.varlen_attention(Q, kk, vv, q_lens, kv_lens, causal=causal)
    return Od, ref


@pytest.mark.parametrize("q_lens,ctx_lens,Hq,Hkv", [
    ([128], [0], 2, 1), ([2], [62], 2, 1), ([66, 130], [7, 300], 4, 2), ([258], [34], 14, 2), ([1026], [4100], 7, 1)])
@pytest.mark.parametrize("form", [0, 1])
def test_varlen_attention(q_lens, ctx_lens, Hq, Hkv, form):
    got, ref = _attention_case(q_lens, ctx_lens, Hq, Hkv, seed=40, form=form)
    assert not torch.isnan(got.float()).any()
    _close(got, ref, ulps=2.0, what="attention")


def test_varlen_attention_random_cases():
    """Random sample counts, query and context lengths (ragged tiles, one-row tiles, contexts shorter and longer than a
    key tile), head counts, head widths, masks and kernel forms against the checker's attention."""
    import random
    rnd = random.Random(11)
    for case in range(int(os.environ.get("BAGEL_FUZZ_CASES", 24))):
        if case and case % 100 == 0:
            print(f"  random attention cases: {case}", flush=True)
        n = rnd.randint(1, 4)
        q_lens = [rnd.choice([1, 2, 31, 64, 65, 127, 128, 129, 200, 300]) for _ in range(n)]
        ctx_lens = [rnd.choice([0, 1, 7, 63, 64, 65, 130, 500]) for _ in range(n)]
        Hkv = rnd.choice([1, 2])
        Hq = Hkv * rnd.choice([1, 2, 7])
        D = rnd.choice([64, 128])
        causal = rnd.random() < 0.4
        form = rnd.choice([0, 1])
        got, ref = _attention_case(q_lens, ctx_lens, Hq, Hkv, seed=100 + case, causal=causal, D=D, form=form)
        what = f"case {case}: q {q_lens} ctx {ctx_lens} Hq {Hq} Hkv {Hkv} D {D} causal {causal} form {form}"
        assert not torch.isnan(got.float()).any(), what
        _close(got, ref, ulps=2.0, what=what)


@pytest.mark.parametrize("q_lens,ctx_lens,Hq,Hkv", [([34], [0], 2, 1), ([5, 200], [0, 77], 4, 2), ([300], [130], 14, 2)])
@pytest.mark.parametrize("form", [0, 1])
def test_varlen_attention_causal(q_lens, ctx_lens, Hq, Hkv, form):
    got, ref = _attention_case(q_lens, ctx_lens, Hq, Hkv, seed=42, causal=True, form=form)
    assert not torch.isnan(got.float()).any()
    _close(got, ref, ulps=2.0, what="causal attention")


@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("q_lens,ctx_lens,Hq,Hkv", [([258], [18], 14, 2), ([2, 70], [0, 129], 4, 2)])
@pytest.mark.parametrize("form", [0, 1])
def test_varlen_attention_head_dim_64(q_lens, ctx_lens, Hq, Hkv, causal, form):
    got, ref = _attention_case(q_lens, ctx_lens, Hq, Hkv, seed=43, causal=causal, D=64, form=form)
    assert not torch.isnan(got.float()).any()
    _close(got, ref, ulps=2.0, what="attention D=64")


@pytest.mark.parametrize("form", [0, 1])
def test_varlen_attention_running_max_jump(form):
    got, ref = _attention_case([130], [500], 2, 1, seed=41, spike=True, form=form)
    _close(got, ref, ulps=2.0, what="attention spike")


@pytest.mark.parametrize("kv_lens,Hq,Hkv,D", [([10], 14, 2, 128), ([700], 28, 4, 128), ([65, 2500, 129], 14, 2, 128),
                                               ([9035], 28, 4, 128), ([64, 321], 14, 2, 64), ([300, 1], 2, 1, 128),
                                               ([17000], 14, 2, 128),
                                               ([400 + 37 * i for i in range(16)], 28, 4, 128),
                                               ([30 + 61 * i for i in range(40)], 28, 4, 128)])
@pytest.mark.parametrize("und_rounding", [True, False])
def test_decode_attention_matches_three_launch_path(kv_lens, Hq, Hkv, D, und_rounding):
    """One new token per sample: the one-launch kernel against bagel_qk_norm_rope_kv_bf16 + bagel_varlen_attention_bf16
    on the same inputs.  The key / value caches must come out bit-identical (same arithmetic, same slots); the
    outputs are held to the bar of the tiled kernel's own tests, against exact fp32 attention over those caches.
    Two launches in a row on one workspace: the tickets must be back at zero after the first."""
    k = _K()
    B = len(kv_lens)
    qkv_dim = (Hq + 2 * Hkv) * D
    kv_start, s = [], 0
    for n in kv_lens:
        kv_start.append(s)
        s += (n + 63) // 64 * 64
    S = s
    g = torch.Generator().manual_seed(61)
    Kc = (torch.randn(Hkv, S, D, generator=g) * 1.0).to(BF16)
    Vc = (torch.randn(Hkv, D, S, generator=g) * 1.0).to(BF16)
    for b, n in enumerate(kv_lens):                 # padding and the new token's own slot: zero values, garbage keys
        Vc[:, :, kv_start[b] + n - 1:kv_start[b] + (n + 63) // 64 * 64] = 0
    qkv = _rand((B, qkv_dim), 1.5, 62).to(DEV)
    qn, kn = (_rand((D,), 0.1, 63) + 1).to(BF16).to(DEV), (_rand((D,), 0.1, 64) + 1).to(BF16).to(DEV)
    pos = torch.tensor([n + 5 for n in kv_lens])
    cos, sin = O.rotary_cos_sin(pos, D, 1e6, BF16)
    row_cs = torch.arange(B, dtype=torch.int32, device=DEV)
    row_slot = torch.tensor([kv_start[b] + kv_lens[b] - 1 for b in range(B)], dtype=torch.int32, device=DEV)
    tiles = torch.tensor([(b, 1, kv_start[b], kv_lens[b], -1, 0, 0, 0) for b in range(B)], dtype=torch.int32, device=DEV)
    # three launches
    K1, V1 = Kc.to(DEV).clone(), Vc.to(DEV).clone()
    Q = torch.empty((B, Hq, D), dtype=BF16, device=DEV)
    O1 = torch.empty((B, Hq * D), dtype=BF16, device=DEV)
    k.qk_norm_rope_kv(qkv, B, Hq, Hkv, D, qn, qn, kn, kn, B, cos.to(DEV), sin.to(DEV), row_cs, row_slot, Q, K1, V1, S, 1e-6,
                      und_rounding)
    k.varlen_attention(Q, K1, V1, O1, tiles, B, Hq, Hkv, D, S)
    rep = Hq // Hkv
    exact = torch.empty((B, Hq, D))
    for b, n in enumerate(kv_lens):
        kk = K1[:, kv_start[b]:kv_start[b] + n].float().cpu().repeat_interleave(rep, dim=0)        # [Hq, n, D]
        vv = V1[:, :, kv_start[b]:kv_start[b] + n].float().cpu().repeat_interleave(rep, dim=0)     # [Hq, D, n]
        pr = torch.softmax(torch.einsum("hd,hnd->hn", Q[b].float().cpu(), kk) / math.sqrt(D), dim=-1)
        exact[b] = torch.einsum("hn,hdn->hd", pr, vv)
    exact = exact.reshape(B, Hq * D).to(BF16)

    def near_exact(got, what):
        # the bound of the tiled kernel's tests on the largest error; the share of elements within one ulp is taken
        # at 0.99 because the smallest case has under two thousand outputs
        err = (got.float().cpu() - exact.float()).abs()
        top = float(exact.float().abs().max())
        assert float(err.max()) <= 2.0 * 2.0 ** -8 * top + 1e-6, f"{what}: max-abs {float(err.max())}"
        assert (err <= 2.0 ** -7 * exact.float().abs() + 1e-3 * top).float().mean() > 0.99, what

    near_exact(O1, "tiled attention on one-row tiles")
    # one launch, twice
    assert k.decode_attention_supported(B, Hq, Hkv, D) == (Hq // Hkv in (2, 7))
    ws = torch.zeros(k.decode_attention_workspace_bytes(B, Hq, Hkv, D), dtype=torch.uint8, device=DEV)
    for _ in range(2):
        K2, V2 = Kc.to(DEV).clone(), Vc.to(DEV).clone()
        O2 = torch.full((B, Hq * D), float("nan"), dtype=BF16, device=DEV)
        k.decode_attention(qkv, Hq, Hkv, D, qn, kn, cos.to(DEV), sin.to(DEV), row_cs, row_slot, tiles, K2, V2, S, O2, 1e-6,
                           und_rounding, ws)
        assert torch.equal(K2, K1) and torch.equal(V2, V1), "key / value caches differ"
        assert int(ws[:1024].view(torch.int32).abs().sum()) == 0, "tickets not returned to zero"
        near_exact(O2, "decode attention")
    # a caller's bound on the key count that is too small (and one that is generous) changes the chunking, never the keys
    for bound in (64, 4 * max(kv_lens)):
        K2, V2 = Kc.to(DEV).clone(), Vc.to(DEV).clone()
        O2 = torch.full((B, Hq * D), float("nan"), dtype=BF16, device=DEV)
        k.decode_attention(qkv, Hq, Hkv, D, qn, kn, cos.to(DEV), sin.to(DEV), row_cs, row_slot, tiles, K2, V2, S, O2, 1e-6,
                           und_rounding, ws, max_kv_len=bound)
        assert torch.equal(K2, K1) and torch.equal(V2, V1) and int(ws[:1024].view(torch.int32).abs().sum()) == 0
        near_exact(O2, f"decode attention, key bound {bound}")


def test_varlen_attention_forms_agree():
    """Both forms walk the same tiles in the same order: identical bits."""
    a, _ = _attention_case([66, 130], [7, 300], 4, 2, seed=44, form=0)
    b, _ = _attention_case([66, 130], [7, 300], 4, 2, seed=44, form=1)
    assert torch.equal(a, b)


@pytest.mark.parametrize("q_lens,ctx_lens,Hq,Hkv,causal", [
    ([128], [0], 2, 1, False), ([2], [62], 2, 1, False), ([66, 130], [7, 300], 4, 2, False), ([258], [34], 14, 2, False),
    ([1026], [4100], 7, 1, False), ([5, 200], [0, 77], 4, 2, True), ([300], [130], 14, 2, True)])
def test_varlen_attention_mfma16_variant(q_lens, ctx_lens, Hq, Hkv, causal):
    """The 16 x 16 x 32 build of the kernel (selected only by its development knob): same bar as the default, and
    its two forms agree bit for bit."""
    a, ref = _attention_case(q_lens, ctx_lens, Hq, Hkv, seed=45, causal=causal, form=0, mfma16=1)
    b, _ = _attention_case(q_lens, ctx_lens, Hq, Hkv, seed=45, causal=causal, form=1, mfma16=1)
    assert not torch.isnan(a.float()).any()
    _close(a, ref, ulps=2.0, what="attention, 16x16x32 form")
    assert torch.equal(a, b)


@pytest.mark.parametrize("renorm,use_img,rmin", [("global", True, 0.0), ("global", False, 0.0), ("channel", True, 0.0),
                                                 ("text_channel", True, 0.3), ("text_channel", False, 0.0)])
def test_cfg_euler(renorm, use_img, rmin):
    k = _K()
    rows, P = 1000, 64
    vc, vx, vi = _rand((rows, P), 0.8, 50), _rand((rows, P), 0.8, 51), _rand((rows, P), 0.8, 52)
    vx = (vc.float() + 0.05 * vx.float()).to(BF16)
    x = _rand((rows, P), 1.0, 53, dtype=torch.float32)
    dt = torch.tensor(0.0123456789)
    s_img = 1.5 if use_img else 1.0
    ref_v = O.cfg_combine(vc, vx, vi if use_img else None, 4.0, s_img, rmin, renorm)
    ref_x = x - ref_v * dt
    xd = x.to(DEV).clone()
    xb = torch.empty((rows, P), dtype=BF16, device=DEV)
    vo = torch.empty((rows, P), dtype=BF16, device=DEV)
    ws = torch.empty(k.cfg_workspace_bytes(rows, P), dtype=torch.uint8, device=DEV)
    k.cfg_euler_update(vc.to(DEV), vx.to(DEV), vi.to(DEV) if use_img else None, 4.0, s_img, renorm, rmin, float(dt),
                       xd, xb, vo, ws)
    mism = (vo.cpu() != ref_v).float().mean()
    assert mism < 1e-3, f"{renorm}: {mism:.5f} of combined-velocity elements differ from the torch bf16 graph"
    assert float((xd.cpu() - ref_x).abs().max()) < 1e-3
    assert torch.equal(xb.cpu(), xd.cpu().to(BF16))


def test_cfg_euler_no_cfg_and_errors():
    k = _K()
    rows, P = 64, 64
    vc = _rand((rows, P), 0.8, 60)
    x = _rand((rows, P), 1.0, 61, dtype=torch.float32)
    xd = x.to(DEV).clone()
    xb = torch.empty((rows, P), dtype=BF16, device=DEV)
    vo = torch.empty((rows, P), dtype=BF16, device=DEV)
    ws = torch.empty(k.cfg_workspace_bytes(rows, P), dtype=torch.uint8, device=DEV)
    k.cfg_euler_update(vc.to(DEV), None, None, 1.0, 1.0, "global", 0.0, 0.25, xd, xb, vo, ws)
    assert torch.equal(vo.cpu(), vc)
    assert torch.equal(xd.cpu(), x - (vc.float() * 0.25).to(BF16).float())
    with pytest.raises(NotImplementedError):
        k.cfg_euler_update(vc.to(DEV), vc.to(DEV), None, 4.0, 1.0, "local", 0.0, 0.25, xd, xb, vo, ws)
    with pytest.raises(RuntimeError):
        k.cfg_euler_update(vc.to(DEV), None, None, 4.0, 1.0, "global", 0.0, 0.25, xd, xb, vo, ws)


def test_timestep_embedding_and_embed_latent():
    k = _K()
    from bagel_amd.config import tiny_dims
    from bagel_amd.weights import make_llm_state_dict
    from bagel_amd.engine import MoTEngine, MoTWeights
    d = tiny_dims(1)
    sd = make_llm_state_dict(d, 0)
    eng = MoTEngine(d, MoTWeights.from_state_dict(d, sd, DEV), DEV)
    ts, _ = O.flow_timesteps(12, 3.0)
    table = eng.timestep_table(ts)
    ref = O.time_embedder(sd, ts, d.timestep_freq_dim)
    _close(table, ref, what="t_emb")
    sinus = torch.empty((ts.numel(), d.timestep_freq_dim), dtype=BF16, device=DEV)
    k.timestep_sinusoid(ts.to(DEV), sinus)
    _close(sinus, O.timestep_embedding(ts, d.timestep_freq_dim).to(BF16), ulps=1.0, what="sinusoid")


def test_error_reporting():
    k = _K()
    A = torch.zeros((8, 32), dtype=BF16, device=DEV)
    with pytest.raises(RuntimeError, match="multiple of 64"):
        k.gemm(A, A, torch.zeros((8, 8), dtype=BF16, device=DEV))


@pytest.mark.parametrize("variant", [0, 3, 4, 5, 6, 8])
@pytest.mark.parametrize("M,N,K,epi", [(700, 512, 256, "bias"), (512, 768, 1024, "residual"), (300, 1024, 192, "swiglu"),
                                       (4096, 4608, 3584, "bias"), (256, 256, 64, "bias"), (260, 256, 128, "residual"),
                                       (1024, 3584, 512, "residual"), (520, 228, 64, "bias")])
def test_gemm_tile_variants(variant, M, N, K, epi):
    """Every tile variant (128x128, 256x256 pipelined in both wave layouts, 256x224, 256x288) must give the
    same answers, ragged M and N included."""
    k = _K()
    lib = k.lib()
    if epi == "swiglu" and variant in (4, 5):
        with pytest.raises(RuntimeError, match="no SwiGLU epilogue"):
            lib.bagel_set_gemm_variant(variant)
            try:
                k.gemm(torch.zeros((M, K), dtype=BF16, device=DEV), torch.zeros((N, K), dtype=BF16, device=DEV),
                       torch.zeros((M, N // 2), dtype=BF16, device=DEV), epilogue=k.EPI_SWIGLU)
            finally:
                lib.bagel_set_gemm_variant(-1)
        return
    A, W = _rand((M, K), 1.0, 70), _rand((N, K), 0.05, 71)
    acc = A.float() @ W.float().t()
    lib.bagel_set_gemm_variant(variant)
    try:
        if epi == "bias":
            b = _rand((N,), 0.5, 72)
            C = torch.full((M, N), float("nan"), dtype=BF16, device=DEV)
            k.gemm(A.to(DEV), W.to(DEV), C, bias=b.to(DEV))
            ref = (acc + b.float()).to(BF16)
        elif epi == "residual":
            R = _rand((M, N), 1.0, 73)
            C = R.to(DEV).clone()
            k.gemm(A.to(DEV), W.to(DEV), C, residual=C, epilogue=k.EPI_RESIDUAL)
            ref = (acc.to(BF16).float() + R.float()).to(BF16)
        else:
            C = torch.full((M, N // 2), float("nan"), dtype=BF16, device=DEV)
            k.gemm(A.to(DEV), W.to(DEV), C, epilogue=k.EPI_SWIGLU)
            a3 = acc.view(M, N // 32, 2, 16)
            ref = (torch.nn.functional.silu(a3[:, :, 0].to(BF16)) * a3[:, :, 1].to(BF16)).reshape(M, N // 2)
    finally:
        lib.bagel_set_gemm_variant(-1)
    _close(C, ref, ulps=3.0, what=f"variant {variant} {epi}")


@pytest.mark.parametrize("M,N,Kd", [(200, 1152, 4352), (1000, 3584, 3584), (96, 64, 3584), (4098, 3584, 64)])
def test_gemm_lds_stages_are_invisible(M, N, Kd):
    """The 128 x 128 tile kernel with two and with three LDS stages walks the same K steps in the same order."""
    k = _K()
    X, W, b = _rand((M, Kd), 1.0, 70).to(DEV), _rand((N, Kd), 0.03, 71).to(DEV), _rand((N,), 0.5, 72).to(DEV)
    out = {}
    try:
        k.lib().bagel_set_gemm_variant(0)
        for stages in (2, 3):
            k.lib().bagel_set_gemm_stages(stages)
            C = torch.full((M, N), float("nan"), dtype=BF16, device=DEV)
            k.gemm(X, W, C, bias=b)
            out[stages] = C
    finally:
        k.lib().bagel_set_gemm_stages(0)
        k.lib().bagel_set_gemm_variant(-1)
    assert torch.equal(out[2], out[3])
    _close(out[3], (X.float() @ W.float().t() + b.float()).to(BF16), what="three-stage tile kernel")


@pytest.mark.parametrize("variant", [3, 4, 5, 6])
@pytest.mark.parametrize("M,N,K,epi", [(700, 512, 256, "bias"), (300, 1024, 320, "swiglu"), (1030, 3584, 8192, "residual")])
def test_gemm_dma_split_is_invisible(variant, M, N, K, epi):
    """The pipelined tiles stage their operands either an eighth per wave or all on the four leading waves
    (chosen by K); both must give the same bits, ragged M and N included."""
    k = _K()
    lib = k.lib()
    if epi == "swiglu" and variant in (4, 5):
        pytest.skip("this tile shape has no SwiGLU epilogue")
    A, W = _rand((M, K), 1.0, 80).to(DEV), _rand((N, K), 0.05, 81).to(DEV)
    b = _rand((N,), 0.5, 82).to(DEV)
    R = _rand((M, N), 1.0, 83).to(DEV)
    outs = []
    lib.bagel_set_gemm_variant(variant)
    try:
        for lead in (0, 1):
            lib.bagel_set_gemm_lead(lead)
            if epi == "bias":
                C = torch.full((M, N), float("nan"), dtype=BF16, device=DEV)
                k.gemm(A, W, C, bias=b)
            elif epi == "residual":
                C = R.clone()
                k.gemm(A, W, C, residual=C, epilogue=k.EPI_RESIDUAL)
            else:
                C = torch.full((M, N // 2), float("nan"), dtype=BF16, device=DEV)
                k.gemm(A, W, C, epilogue=k.EPI_SWIGLU)
            outs.append(C.clone())
    finally:
        lib.bagel_set_gemm_variant(-1)
        lib.bagel_set_gemm_lead(-1)
    assert not torch.isnan(outs[0].float()).any()
    assert torch.equal(outs[0], outs[1])
