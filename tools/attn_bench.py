"""Micro-benchmark of bagel_varlen_attention_bf16 on the generation shapes (random Q/K/V)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bagel_amd import _lib as K  # noqa: E402


def case(name, units, nv, ctx, Hq=28, Hkv=4, D=128, iters=10):
    dev = "cuda:0"
    M = units * (nv + 2)
    Mv = units * nv
    tiles, s = [], 0
    for u in range(units):
        kv_len = ctx + nv + 2
        for q0 in range(0, nv, 128):
            tiles.append((u * nv + q0, min(128, nv - q0), s, kv_len, -1, 0, 0, 0))
        tiles.append((Mv + 2 * u, 2, s, kv_len, -1, 0, 0, 0))
        s += (kv_len + 63) // 64 * 64
    S = s
    Q = torch.randn((M, Hq, D), device=dev, dtype=torch.bfloat16) * 2
    Kw = torch.randn((Hkv, S, D), device=dev, dtype=torch.bfloat16) * 2
    Vt = torch.randn((Hkv, D, S), device=dev, dtype=torch.bfloat16)
    O = torch.empty_like(Q)
    t = torch.tensor(tiles, dtype=torch.int32, device=dev)
    flops = 4.0 * Hq * D * units * (nv + 2) * (ctx + nv + 2)
    # A/B in one process: "lib" is the library in the tree; ATTN_AB_LIB names another build of the same entry
    # point (e.g. the previous commit's kernel compiled to tools/_ab/), run round by round beside it.
    variants = ["lib"]
    if os.environ.get("ATTN_FORMS"):   # also time the kernel's two forms explicitly
        variants += ["throughput", "latency"]
    # ATTN_AB_LIBS=a.so,b.so (or ATTN_AB_LIB=one.so): other builds of the entry point, run round by round beside "lib"
    alts = [a for a in (os.environ.get("ATTN_AB_LIBS") or os.environ.get("ATTN_AB_LIB") or "").split(",") if a]
    alt_libs = {}
    if alts:
        import ctypes
        for a in alts:
            nm = os.path.basename(a).replace("libattn_", "").replace(".so", "")
            alt_libs[nm] = ctypes.CDLL(os.path.abspath(a))
            variants.append(nm)

    def run(v):
        if v in ("lib", "throughput", "latency"):
            K.lib().bagel_set_attn_deep({"lib": -1, "throughput": 0, "latency": 1}[v])
            K.varlen_attention(Q, Kw, Vt, O, t, len(tiles), Hq, Hkv, D, S)
        else:
            rc = alt_libs[v].bagel_varlen_attention_bf16(K._p(Q), K._p(Kw), K._p(Vt), K._p(O), K._p(t), K.I(len(tiles)), K.I(Hq),
                                                     K.I(Hkv), K.I(D), K.L(S), K._stream())
            assert rc == 0

    times = {v: [] for v in variants}
    ref = None
    for v in variants:
        for _ in range(2):
            run(v)
        if ref is None:
            ref = O.float().clone()
        else:
            print(f"   variant {v} max |diff| vs variant {variants[0]}: {(O.float() - ref).abs().max().item():.3e}")
    for _ in range(iters):          # variants interleaved round by round in one process
        for v in variants:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run(v)
            e1.record()
            torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1))
    for v in variants:
        ts = sorted(times[v])
        med = ts[len(ts) // 2]
        print(f"{name:24s} {v:12s} median {med:.3f} ms  {flops / med / 1e9:.0f} TFLOP/s  (min {ts[0]:.3f} ms)", flush=True)


if __name__ == "__main__":
    case("t2i 3 units ctx34", 3, 4096, 34)
    case("t2i 1 unit ctx34", 1, 4096, 34)
    case("edit 3 units ctx9034", 3, 4096, 9034)
    case("edit 24 units ctx9034", 24, 4096, 9034, iters=3)
