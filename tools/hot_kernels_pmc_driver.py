"""Launch each hot kernel of one 3-branch flow step at BAGEL-7B dims a few times on random operands
(driver for rocprofv3 --pmc passes; shapes: M = 3 x 4096 latent rows, 34-token context)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bagel_amd import _lib as K  # noqa: E402

dev = "cuda:0"
BF = torch.bfloat16
M, H, I, Hq, Hkv, D = 12288, 3584, 18944, 28, 4, 128
R = 3


def rn(*shape, s=1.0):
    return torch.randn(shape, device=dev, dtype=BF) * s


X = rn(M, H)
Xn = torch.empty_like(X)
w_norm = rn(H, s=0.1) + 1
for _ in range(R):
    K.rmsnorm(X, Xn, w_norm, None, M, 1e-6)

Wqkv, bqkv = rn(4608, H, s=0.02), rn(4608, s=0.02)
QKV = torch.empty((M, 4608), device=dev, dtype=BF)
for _ in range(R):
    K.gemm(Xn, Wqkv, QKV, bias=bqkv)

units, nv, ctx = 3, 4096, 34
tiles, row_slot, s = [], [], 0
for u in range(units):
    kv_len = ctx + nv
    for q0 in range(0, nv, 128):
        tiles.append((u * nv + q0, 128, s, kv_len, -1, 0, 0, 0))
    row_slot += list(range(s + ctx, s + ctx + nv))
    s += (kv_len + 63) // 64 * 64
S = s
cos, sin = rn(units, D), rn(units, D)
row_cs = torch.arange(units, device=dev, dtype=torch.int32).repeat_interleave(nv)
row_slot = torch.tensor(row_slot, device=dev, dtype=torch.int32)
Q = torch.empty((M, Hq * D), device=dev, dtype=BF)
Kw = rn(Hkv, S, D)
Vt = rn(Hkv, D, S)
qn, kn = rn(D, s=0.1) + 1, rn(D, s=0.1) + 1
for _ in range(R):
    K.qk_norm_rope_kv(QKV, M, Hq, Hkv, D, qn, None, kn, None, M, cos, sin, row_cs, row_slot, Q, Kw, Vt, S, 1e-6)
O = torch.empty_like(Q)
tt = torch.tensor(tiles, dtype=torch.int32, device=dev)
for _ in range(R):
    K.varlen_attention(Q, Kw, Vt, O, tt, len(tiles), Hq, Hkv, D, S)

Wo = rn(H, H, s=0.02)
for _ in range(R):
    K.gemm(O, Wo, X, residual=X, epilogue=K.EPI_RESIDUAL)
Wgu = rn(2 * I, H, s=0.02)
Hm = torch.empty((M, I), device=dev, dtype=BF)
for _ in range(R):
    K.gemm(Xn, Wgu, Hm, epilogue=K.EPI_SWIGLU)
Wd = rn(H, I, s=0.02)
for _ in range(R):
    K.gemm(Hm, Wd, X, residual=X, epilogue=K.EPI_RESIDUAL)
torch.cuda.synchronize()
print("done")

if os.environ.get("PMC_SKINNY") == "1":
    # the understanding expert's weight-streaming projections of the same step (6 text rows) and the timestep MLP
    print("skinny section", flush=True)
    Mt = 6
    ws = torch.empty((16 * 64 * 2 * I,), device=dev, dtype=torch.float32)
    xs = rn(Mt, H)
    for name, W_, out_cols, epi in (("qkv", Wqkv, 4608, K.SK_BIAS), ("gate_up", Wgu, I, K.SK_SWIGLU)):
        out = torch.empty((Mt, out_cols), device=dev, dtype=BF)
        for _ in range(R):
            K.skinny_gemm(xs, W_, out, ws, bias=bqkv if epi == K.SK_BIAS else None, epilogue=epi)
        torch.cuda.synchronize()
        print("skinny", name, "ok", flush=True)
    hs = rn(Mt, I)
    for _ in range(R):
        K.skinny_gemm(hs, Wd, xs, ws, residual=xs, epilogue=K.SK_RESIDUAL)
    torch.cuda.synchronize()
    print("skinny down ok", flush=True)
