"""Micro-benchmark of the decode step's attention: bagel_decode_attention_bf16 (one launch) against
bagel_qk_norm_rope_kv_bf16 + bagel_varlen_attention_bf16 (three launches), BAGEL-7B head layout, one sample.
Each timed launch works on another layer's cache (28 of them) so that keys and values come from HBM as in the model.

    python tools/decode_attn_bench.py [kv_len ...]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bagel_amd import _lib as K  # noqa: E402

Hq, Hkv, D, L = 28, 4, 128, 28
BF16 = torch.bfloat16


def main():
    dev = "cuda:0"
    flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
    for kv_len in [int(v) for v in sys.argv[1:]] or [500, 700, 2000, 9000]:
        S = (kv_len + 63) // 64 * 64
        Kws = torch.randn((L, Hkv, S, D), device=dev, dtype=BF16)
        Vt = torch.randn((L, Hkv, D, S), device=dev, dtype=BF16)
        Vt[..., kv_len - 1:] = 0
        qkv = torch.randn((1, (Hq + 2 * Hkv) * D), device=dev, dtype=BF16)
        qn = torch.ones(D, device=dev, dtype=BF16)
        cos = torch.ones((1, D), device=dev, dtype=BF16)
        sin = torch.zeros((1, D), device=dev, dtype=BF16)
        row_cs = torch.zeros(1, dtype=torch.int32, device=dev)
        row_slot = torch.tensor([kv_len - 1], dtype=torch.int32, device=dev)
        tiles = torch.tensor([(0, 1, 0, kv_len, -1, 0, 0, 0)], dtype=torch.int32, device=dev)
        Q = torch.empty((1, Hq, D), device=dev, dtype=BF16)
        O = torch.empty((1, Hq * D), device=dev, dtype=BF16)
        ws = torch.zeros(K.decode_attention_workspace_bytes(1, Hq, Hkv, D), dtype=torch.uint8, device=dev)

        def three(l):
            K.qk_norm_rope_kv(qkv, 1, Hq, Hkv, D, qn, qn, qn, qn, 1, cos, sin, row_cs, row_slot, Q, Kws[l], Vt[l], S, 1e-6, True)
            K.varlen_attention(Q, Kws[l], Vt[l], O, tiles, 1, Hq, Hkv, D, S)

        def one(l):
            K.decode_attention(qkv, Hq, Hkv, D, qn, qn, cos, sin, row_cs, row_slot, tiles, Kws[l], Vt[l], S, O, 1e-6, True, ws)

        def timed(fn):
            for l in range(L):
                fn(l)
            flush.zero_()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for l in range(L):
                fn(l)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / L * 1e3

        line = f"kv_len {kv_len:5d}: three launches {timed(three):6.1f} us"
        for mc in (256, 512, 1024, 4096):
            K.lib().bagel_set_decode_min_chunk(mc)
            line += f" | one launch, min chunk {mc}: {timed(one):6.1f} us"
        K.lib().bagel_set_decode_min_chunk(256)
        print(line, flush=True)


if __name__ == "__main__":
    main()
