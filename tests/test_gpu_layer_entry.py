"""bagel_mot_layer_forward / bagel_mot_forward (one C call per layer / per stack) against the same layer
assembled from the per-operation entry points, bit for bit, through the C-ABI.

Reference semantics: Qwen2MoTDecoderLayer.forward_inference (qwen2_navit.py:771-845)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF16 = torch.bfloat16


def _weights(H, I, Hq, Hkv, D, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    rn = lambda shape, std, mean=0.0: torch.empty(shape, device=DEV, dtype=BF16).normal_(mean, std, generator=g)  # noqa: E731
    qkv = (Hq + 2 * Hkv) * D
    return [dict(wqkv=rn((qkv, H), 0.05), bqkv=rn((qkv,), 0.05), wo=rn((H, Hq * D), 0.05), wgu=rn((2 * I, H), 0.05),
                 wd=rn((H, I), 0.05), ln_in=rn((H,), 0.1, 1.0), ln_post=rn((H,), 0.1, 1.0), q_norm=rn((D,), 0.1, 1.0),
                 k_norm=rn((D,), 0.1, 1.0)) for _ in range(2)]          # [gen, und]


def _layer_struct(K, w):
    layer = K.MotLayer()
    for expert, ww in ((layer.gen, w[0]), (layer.und, w[1])):
        for k, t in ww.items():
            setattr(expert, k, t.data_ptr())
    return layer


def _setup(Mv, Mt, H, I, Hq, Hkv, D, ctx, seed):
    """One sample: ctx context keys, then the Mv + Mt new tokens (all at one RoPE position, non-causal)."""
    g = torch.Generator(device=DEV).manual_seed(seed + 100)
    M = Mv + Mt
    S = (ctx + M + 63) // 64 * 64
    X = torch.randn((M, H), device=DEV, generator=g).to(BF16)
    Kw = torch.zeros((Hkv, S, D), device=DEV, dtype=BF16)
    Vt = torch.zeros((Hkv, D, S), device=DEV, dtype=BF16)
    Kw[:, :ctx] = torch.randn((Hkv, ctx, D), device=DEV, generator=g).to(BF16)
    Vt[:, :, :ctx] = (torch.randn((Hkv, D, ctx), device=DEV, generator=g) * 0.5).to(BF16)
    cos = torch.randn((1, D), device=DEV, generator=g).to(BF16)
    sin = torch.randn((1, D), device=DEV, generator=g).to(BF16)
    row_cs = torch.zeros(M, dtype=torch.int32, device=DEV)
    row_slot = torch.arange(ctx, ctx + M, dtype=torch.int32, device=DEV)
    tiles = [(q0, min(128, Mv - q0), 0, ctx + M, -1, 0, 0, 0) for q0 in range(0, Mv, 128)]
    tiles += [(Mv + q0, min(128, Mt - q0), 0, ctx + M, -1, 0, 0, 0) for q0 in range(0, Mt, 128)]
    return X, Kw, Vt, cos, sin, row_cs, row_slot, torch.tensor(tiles, dtype=torch.int32, device=DEV), len(tiles), S


def _per_op_layer(K, w, X, Kw, Vt, cos, sin, row_cs, row_slot, tiles, n_tiles, S, Mv, Mt, H, I, Hq, Hkv, D, skw, next_norms):
    """The layer from the per-operation entry points (what the host loop used to issue)."""
    g, u = w
    M = Mv + Mt
    qkv = (Hq + 2 * Hkv) * D
    eps = 1e-6
    buf = lambda n: torch.empty((M, n), device=DEV, dtype=BF16)  # noqa: E731
    Xn, QKV, Q, O, Hm = buf(H), buf(qkv), buf(Hq * D), buf(Hq * D), buf(I)
    first = g if Mv else u
    split = Mv if Mv else M
    fused = Mv == 0 and Mt <= 16
    tile = Mv == 0 and Mt > 16      # no generation rows: the tile kernels from 17 rows on, norms in the reduce launches

    def und(x, wt, out, **kw):
        if not tile:
            # beside generation rows the understanding rows keep the weight-streaming kernel, 64 rows per launch
            # (bits independent of the row count)
            ep = {None: K.SK_BIAS, "residual": K.SK_RESIDUAL, "swiglu": K.SK_SWIGLU}[kw.pop("kind", None)]
            res = kw.pop("residual", None)
            for r0 in range(0, x.shape[0], 64):
                r1 = min(r0 + 64, x.shape[0])
                extra = dict(kw, residual=res[r0:r1]) if res is not None else kw
                K.skinny_gemm(x[r0:r1], wt, out[r0:r1], skw, epilogue=ep, **extra)
            return
        ep = {None: K.EPI_BIAS, "residual": K.EPI_RESIDUAL, "swiglu": K.EPI_SWIGLU}[kw.pop("kind", None)]
        return K.gemm(x, wt, out, epilogue=ep, workspace=skw, **kw)

    K.rmsnorm(X, Xn, first["ln_in"], u["ln_in"], split, eps)
    if Mv:
        K.gemm(Xn, g["wqkv"], QKV, bias=g["bqkv"], M=Mv)
    if Mt:
        und(Xn[Mv:], u["wqkv"], QKV[Mv:], bias=u["bqkv"])
    K.qk_norm_rope_kv(QKV, M, Hq, Hkv, D, first["q_norm"], u["q_norm"], first["k_norm"], u["k_norm"], split, cos, sin, row_cs,
                      row_slot, Q, Kw, Vt, S, eps)
    K.varlen_attention(Q, Kw, Vt, O, tiles, n_tiles, Hq, Hkv, D, S)
    if Mv:
        K.gemm(O, g["wo"], X, residual=X, epilogue=K.EPI_RESIDUAL, M=Mv)
    if fused:
        K.skinny_gemm_residual_rmsnorm(O, u["wo"], X, X, u["ln_post"], Xn, eps, skw)
    elif tile:
        K.gemm_residual_rmsnorm(O, u["wo"], X, X, u["ln_post"], Xn, eps, skw)
    else:
        if Mt:
            und(O[Mv:], u["wo"], X[Mv:], residual=X[Mv:], kind="residual")
        K.rmsnorm(X, Xn, first["ln_post"], u["ln_post"], split, eps)
    if Mv:
        K.gemm(Xn, g["wgu"], Hm, epilogue=K.EPI_SWIGLU, M=Mv)
    if Mt:
        und(Xn[Mv:], u["wgu"], Hm[Mv:], kind="swiglu")
    if Mv:
        K.gemm(Hm, g["wd"], X, residual=X, epilogue=K.EPI_RESIDUAL, M=Mv)
    if fused and next_norms is not None:
        K.skinny_gemm_residual_rmsnorm(Hm, u["wd"], X, X, next_norms[1], Xn, eps, skw)
    elif tile and next_norms is not None:
        K.gemm_residual_rmsnorm(Hm, u["wd"], X, X, next_norms[1], Xn, eps, skw)
    else:
        if Mt:
            und(Hm[Mv:], u["wd"], X[Mv:], residual=X[Mv:], kind="residual")
        if next_norms is not None:
            K.rmsnorm(X, Xn, next_norms[0] if Mv else next_norms[1], next_norms[1], split, eps)
    return X, Xn


@pytest.mark.parametrize("Mv,Mt,ctx", [(300, 6, 37), (0, 5, 90), (0, 150, 0), (0, 40, 20), (256, 0, 10), (130, 70, 3)])
def test_layer_entry_matches_per_op_sequence(Mv, Mt, ctx):
    from bagel_amd import _lib as K
    H, I, Hq, Hkv, D = 256, 512, 2, 1, 128
    w = _weights(H, I, Hq, Hkv, D, seed=Mv + Mt)
    nxt = (w[0]["ln_post"], w[1]["ln_in"])          # any two norm weights serve as "the norm that follows"
    X0, Kw0, Vt0, cos, sin, row_cs, row_slot, tiles, n_tiles, S = _setup(Mv, Mt, H, I, Hq, Hkv, D, ctx, seed=7)
    skw = torch.empty(16 * 64 * 2 * I, device=DEV, dtype=torch.float32)
    # reference: per-operation calls
    Xa, Kwa, Vta = X0.clone(), Kw0.clone(), Vt0.clone()
    Xa, Xna = _per_op_layer(K, w, Xa, Kwa, Vta, cos, sin, row_cs, row_slot, tiles, n_tiles, S, Mv, Mt, H, I, Hq, Hkv, D, skw, nxt)
    # one call
    M = Mv + Mt
    qkv = (Hq + 2 * Hkv) * D
    Xb, Kwb, Vtb = X0.clone(), Kw0.clone(), Vt0.clone()
    buf = lambda n: torch.empty((M, n), device=DEV, dtype=BF16)  # noqa: E731
    Xnb, QKV, Q, O, Hm = buf(H), buf(qkv), buf(Hq * D), buf(Hq * D), buf(I)
    rows = K.mot_rows(Mv, Mt, Xb, Xnb, QKV, Q, O, Hm, cos, sin, row_cs, row_slot, tiles, n_tiles, S, skw)
    dims = K.MotDims(H, I, Hq, Hkv, D, 1e-6)
    K.mot_layer_forward(dims, _layer_struct(K, w), rows, Kwb, Vtb, next_norm_gen=nxt[0], next_norm_und=nxt[1])
    torch.cuda.synchronize()
    assert torch.isfinite(Xb.float()).all()
    assert torch.equal(Xa, Xb) and torch.equal(Xna, Xnb)
    assert torch.equal(Kwa, Kwb) and torch.equal(Vta, Vtb)


def test_layer_entry_flags_and_errors():
    from bagel_amd import _lib as K
    H, I, Hq, Hkv, D = 256, 512, 2, 1, 128
    Mv, Mt, ctx = 64, 2, 5
    w = _weights(H, I, Hq, Hkv, D, seed=3)
    X0, Kw0, Vt0, cos, sin, row_cs, row_slot, tiles, n_tiles, S = _setup(Mv, Mt, H, I, Hq, Hkv, D, ctx, seed=9)
    M = Mv + Mt
    qkv = (Hq + 2 * Hkv) * D
    skw = torch.empty(16 * 64 * 2 * I, device=DEV, dtype=torch.float32)
    buf = lambda n: torch.empty((M, n), device=DEV, dtype=BF16)  # noqa: E731
    X, Kw, Vt = X0.clone(), Kw0.clone(), Vt0.clone()
    Xn, QKV, Q, O, Hm = buf(H), buf(qkv), buf(Hq * D), buf(Hq * D), buf(I)
    rows = K.mot_rows(Mv, Mt, X, Xn, QKV, Q, O, Hm, cos, sin, row_cs, row_slot, tiles, n_tiles, S, skw)
    dims = K.MotDims(H, I, Hq, Hkv, D, 1e-6)
    layer = _layer_struct(K, w)
    # KV_ONLY: keys/values written, residual stream untouched
    K.mot_layer_forward(dims, layer, rows, Kw, Vt, flags=1)
    torch.cuda.synchronize()
    assert torch.equal(X, X0) and not torch.equal(Kw, Kw0)
    assert torch.equal(Kw[:, :ctx], Kw0[:, :ctx])                     # the context slots are only read
    # empty forward is refused with a message
    bad = K.mot_rows(0, 0, X, Xn, QKV, Q, O, Hm, cos, sin, row_cs, row_slot, tiles, n_tiles, S, skw)
    with pytest.raises(RuntimeError, match="rows_gen=0 rows_und=0"):
        K.mot_layer_forward(dims, layer, bad, Kw, Vt)
