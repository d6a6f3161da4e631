"""Host side of the generation forward: fused weight layout, KV workspace, launch sequence.

One ``forward_gen`` call runs every local (sample, CFG-branch) unit of a flow step as ONE packed
sequence through the 28 MoT layers: the latent tokens of all units form one contiguous row block that
goes through the generation expert's MFMA GEMMs, the two text tokens of every unit form a second block
that goes through the understanding expert's weight-streaming path.  Attention is non-causal and every
query token of a sample shares one RoPE position (bagel.py:1185), so the physical row order inside a
sample is free; only sample membership matters, and that is carried by the attention tile table.

All arithmetic happens in libbagel_hip.so; torch allocates buffers and builds small index tables.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import torch

from . import _lib as K
from .config import BagelDims

BF16 = torch.bfloat16


def _interleave_gate_up(gate: torch.Tensor, up: torch.Tensor) -> torch.Tensor:
    """[16 gate rows | 16 up rows] per 16 features: the layout BAGEL_EPI_SWIGLU expects."""
    I, H = gate.shape
    return torch.stack([gate.view(I // 16, 16, H), up.view(I // 16, 16, H)], dim=1).reshape(2 * I, H).contiguous()


@dataclass
class LayerWeights:
    wqkv: List[torch.Tensor]      # [gen, und]  [(Hq+2Hkv)*D, H]
    bqkv: List[torch.Tensor]
    wo: List[torch.Tensor]        # [H, Hq*D]
    wgu: List[torch.Tensor]       # [2I, H] interleaved
    wd: List[torch.Tensor]        # [H, I]
    ln_in: List[torch.Tensor]
    ln_post: List[torch.Tensor]
    qn: List[torch.Tensor]
    kn: List[torch.Tensor]


class MoTWeights:
    """Device-resident weights in the fused layouts the kernels read.  Index 0 = generation expert
    (``*_moe_gen`` parameters), 1 = understanding expert."""

    def __init__(self, dims: BagelDims, device):
        self.dims = dims
        self.device = device
        self.layers: List[LayerWeights] = []
        self.embed_tokens = None
        self.norm = [None, None]
        self.t0_w = self.t0_b = self.t2_w = self.t2_b = None
        self.v2l_w = self.v2l_b = self.l2v_w = self.l2v_b = None
        self.pos_embed = None
        self.lm_head = None         # [vocab, H]; only text decoding needs it

    def layer_table(self):
        """The layers as the C array ``bagel_mot_layer[num_layers]`` that bagel_mot_forward walks (device addresses of
        the tensors held by ``self.layers``; built once)."""
        if getattr(self, "_layer_table", None) is None or len(self._layer_table) != len(self.layers):
            table = (K.MotLayer * len(self.layers))()
            for dst, lw in zip(table, self.layers):
                for e, expert in ((0, dst.gen), (1, dst.und)):
                    for name, t in (("wqkv", lw.wqkv[e]), ("bqkv", lw.bqkv[e]), ("wo", lw.wo[e]), ("wgu", lw.wgu[e]),
                                    ("wd", lw.wd[e]), ("ln_in", lw.ln_in[e]), ("ln_post", lw.ln_post[e]),
                                    ("q_norm", lw.qn[e]), ("k_norm", lw.kn[e])):
                        assert t.is_contiguous()
                        setattr(expert, name, t.data_ptr())
            self._layer_table = table
        return self._layer_table

    @classmethod
    def from_state_dict(cls, dims: BagelDims, sd: Dict[str, torch.Tensor], device) -> "MoTWeights":
        """``sd`` uses the reference's parameter names (the layout of ema.safetensors)."""
        self = cls(dims, device)

        def g(name):
            return sd[name].to(device=device, dtype=BF16)

        for i in range(dims.num_layers):
            p = f"language_model.model.layers.{i}."
            a = p + "self_attn."
            lw = LayerWeights([], [], [], [], [], [], [], [], [])
            for suf in ("_moe_gen", ""):
                lw.wqkv.append(torch.cat([g(a + f"q_proj{suf}.weight"), g(a + f"k_proj{suf}.weight"),
                                          g(a + f"v_proj{suf}.weight")], dim=0).contiguous())
                lw.bqkv.append(torch.cat([g(a + f"q_proj{suf}.bias"), g(a + f"k_proj{suf}.bias"),
                                          g(a + f"v_proj{suf}.bias")], dim=0).contiguous())
                lw.wo.append(g(a + f"o_proj{suf}.weight").contiguous())
                lw.wgu.append(_interleave_gate_up(g(p + f"mlp{suf}.gate_proj.weight"), g(p + f"mlp{suf}.up_proj.weight")))
                lw.wd.append(g(p + f"mlp{suf}.down_proj.weight").contiguous())
                lw.ln_in.append(g(p + f"input_layernorm{suf}.weight"))
                lw.ln_post.append(g(p + f"post_attention_layernorm{suf}.weight"))
                lw.qn.append(g(a + f"q_norm{suf}.weight"))
                lw.kn.append(g(a + f"k_norm{suf}.weight"))
            self.layers.append(lw)
        self.embed_tokens = g("language_model.model.embed_tokens.weight")
        self.norm = [g("language_model.model.norm_moe_gen.weight"), g("language_model.model.norm.weight")]
        self.t0_w, self.t0_b = g("time_embedder.mlp.0.weight"), g("time_embedder.mlp.0.bias")
        self.t2_w, self.t2_b = g("time_embedder.mlp.2.weight"), g("time_embedder.mlp.2.bias")
        self.v2l_w, self.v2l_b = g("vae2llm.weight"), g("vae2llm.bias")
        self.l2v_w, self.l2v_b = g("llm2vae.weight"), g("llm2vae.bias")
        self.pos_embed = g("latent_pos_embed.pos_embed")
        if "language_model.lm_head.weight" in sd:
            self.lm_head = g("language_model.lm_head.weight")
        return self

    @classmethod
    def random(cls, dims: BagelDims, device, seed: int = 0, vocab_rows: Optional[int] = None) -> "MoTWeights":
        """Random-init weights generated directly on the device in the fused layouts (benchmarks only:
        no checkpoint exists offline).  Same distributions as bagel_amd.weights.make_llm_state_dict."""
        from .weights import get_2d_sincos_pos_embed
        self = cls(dims, device)
        gen = torch.Generator(device=device)
        gen.manual_seed(seed)
        H, I = dims.hidden_size, dims.intermediate_size
        qkv = dims.q_dim + 2 * dims.kv_dim

        def rn(shape, std, mean=0.0):
            t = torch.empty(shape, device=device, dtype=BF16)
            t.normal_(mean, std, generator=gen)
            return t

        for _ in range(dims.num_layers):
            lw = LayerWeights([], [], [], [], [], [], [], [], [])
            for _e in range(2):
                lw.wqkv.append(rn((qkv, H), 0.02))
                lw.bqkv.append(rn((qkv,), 0.02))
                lw.wo.append(rn((H, dims.q_dim), 0.02))
                lw.wgu.append(rn((2 * I, H), 0.02))
                lw.wd.append(rn((H, I), 0.02))
                lw.ln_in.append(rn((H,), 0.1, 1.0))
                lw.ln_post.append(rn((H,), 0.1, 1.0))
                lw.qn.append(rn((dims.head_dim,), 0.1, 1.0))
                lw.kn.append(rn((dims.head_dim,), 0.1, 1.0))
            self.layers.append(lw)
        self.embed_tokens = rn((vocab_rows or dims.vocab_size, H), 0.02)
        self.norm = [rn((H,), 0.1, 1.0), rn((H,), 0.1, 1.0)]
        F = dims.timestep_freq_dim
        self.t0_w, self.t0_b = rn((H, F), 0.02), rn((H,), 0.02)
        self.t2_w, self.t2_b = rn((H, H), 0.02), rn((H,), 0.02)
        P = dims.patch_latent_dim
        self.v2l_w, self.v2l_b = rn((H, P), 0.02), rn((H,), 0.02)
        self.l2v_w, self.l2v_b = rn((P, H), 0.8 / H ** 0.5), rn((P,), 0.02)
        pe = get_2d_sincos_pos_embed(H, dims.max_latent_size)
        self.pos_embed = torch.from_numpy(pe).float().to(BF16).to(device)
        return self


@dataclass
class TaylorUpdate:
    """One CFG branch's TaylorSeer cache update after a full forward: the branch's latent rows of the plan that ran
    (``row0`` .. ``row0 + rows``) feed its factor cache."""
    row0: int
    rows: int
    factors: torch.Tensor       # [7, rows * H] bf16 view (row stride = the whole cache's), this branch's rows
    n_factors: int              # factors the cache holds after this update
    distance: int = 1           # steps (of this branch) since its previous full step


class TaylorSchedule:
    """Step types and factor counts of one CFG branch: cal_type / force_scheduler / derivative_approximation of
    modeling/cache_utils/taylorseer.py:11-29, 61-117 with cache_init's constants (:123-153: fresh_threshold 3,
    max_order 6, first_enhance 5, taylor_cache on)."""
    FRESH, MAX_ORDER, FIRST = 3, 6, 5

    def __init__(self):
        self.step = 0
        self.activated = [0]
        self.counter = 0
        self.n_factors = 0
        self.kind = "full"

    def advance(self):
        """Call once per forward of the branch, before it: decides the step type (cal_type)."""
        first = self.step < self.FIRST
        if first or self.counter == self.FRESH - 1:
            self.kind = "full"
            self.counter = 0
            self.activated.append(self.step)
            if self.step == 0:
                self.n_factors = 0                       # taylor_cache_init
            # derivative_approximation: one more order per full step once step > first_enhance - 2
            self.n_factors = min(self.n_factors + 1, self.MAX_ORDER + 1) if (self.n_factors and self.step > self.FIRST - 2) else 1
        else:
            self.counter += 1
            self.kind = "Taylor"
        return self.kind

    @property
    def distance(self):
        return self.activated[-1] - self.activated[-2]

    @property
    def x(self):
        return self.step - self.activated[-1]

    def done(self):
        self.step += 1


@dataclass
class GenUnit:
    """One (sample, CFG branch) pair of a flow step."""
    sample: int                 # index into the packed batch
    branch: int                 # 0 cond, 1 text-uncond, 2 image-uncond
    ctx_len: int                # context tokens already in the KV cache
    rope_pos: int               # RoPE position shared by all query tokens of the sample
    ctx_k: Optional[List[torch.Tensor]] = None   # per layer [ctx_len, Hkv, D] (device, bf16) or None
    ctx_v: Optional[List[torch.Tensor]] = None


@dataclass
class GenPlan:
    units: List[GenUnit]
    nv: List[int]               # latent tokens per unit
    Mv: int
    Mt: int
    S: int
    kv_start: List[int]
    vae_row0: List[int]
    row_cs: torch.Tensor
    row_slot: torch.Tensor
    src_row: torch.Tensor
    tiles: torch.Tensor
    n_tiles: int
    cos: torch.Tensor
    sin: torch.Tensor

    @property
    def M(self):
        return self.Mv + self.Mt


def rope_table(positions: Sequence[int], head_dim: int, theta: float, device):
    """cos/sin rows for the given positions, rounded to bf16 (Qwen2RotaryEmbedding.forward,
    modeling_qwen2.py:130-150; positions are constant over the flow steps so this runs once per image)."""
    pos = torch.tensor(list(positions), dtype=torch.long)
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.int64).float() / head_dim))
    freqs = (inv_freq[None, :, None].float() @ pos[None, None, :].float()).transpose(1, 2)[0]
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(BF16).to(device), emb.sin().to(BF16).to(device)


class MoTEngine:
    def __init__(self, dims: BagelDims, weights: MoTWeights, device):
        assert dims.head_dim in (128, 64), "the attention kernel is built for head_dim 128 and 64"
        self.dims = dims
        self.w = weights
        self.device = device
        self._buf: Dict[str, torch.Tensor] = {}
        self._cdims = K.MotDims(dims.hidden_size, dims.intermediate_size, dims.num_heads, dims.num_kv_heads, dims.head_dim,
                                dims.rms_norm_eps)
        self.kws = None
        self.vtws = None

    # ---- buffers --------------------------------------------------------------------------
    def _get(self, name: str, shape, dtype=BF16, zero=False) -> torch.Tensor:
        n = 1
        for s in shape:
            n *= s
        cur = self._buf.get(name)
        if cur is None or cur.numel() < n or cur.dtype != dtype:
            cur = (torch.zeros if zero else torch.empty)(max(n, 1), device=self.device, dtype=dtype)
            self._buf[name] = cur
        return cur[:n].view(*shape)

    # ---- planning -------------------------------------------------------------------------
    def build_plan(self, units: List[GenUnit], sample_nv: Sequence[int], sample_row0: Sequence[int],
                   kv_start: Optional[Sequence[int]] = None, S: Optional[int] = None) -> GenPlan:
        """Row layout: latent rows of unit 0, unit 1, ... then two text rows per unit.
        ``sample_row0[b]`` is the first row of sample b in the packed latent x_t.  ``kv_start`` / ``S``: the units' key
        slots inside a workspace laid out by another plan (a plan over a subset of that plan's units)."""
        d = self.dims
        nv = [sample_nv[u.sample] for u in units]
        Mv, Mt = sum(nv), 2 * len(units)
        own_slots = kv_start is None
        kv_start = [] if own_slots else list(kv_start)
        vae_row0 = []
        s = r = 0
        for u, n in zip(units, nv):
            if own_slots:
                kv_start.append(s)
            vae_row0.append(r)
            s += (u.ctx_len + n + 2 + 63) // 64 * 64
            r += n
        S = s if S is None else S
        row_cs = torch.empty(Mv + Mt, dtype=torch.int32)
        row_slot = torch.empty(Mv + Mt, dtype=torch.int32)
        src_row = torch.empty(Mv, dtype=torch.int32)
        tiles = []
        for i, (u, n) in enumerate(zip(units, nv)):
            r0, ks = vae_row0[i], kv_start[i]
            row_cs[r0:r0 + n] = i
            row_slot[r0:r0 + n] = torch.arange(ks + u.ctx_len, ks + u.ctx_len + n, dtype=torch.int32)
            src_row[r0:r0 + n] = torch.arange(sample_row0[u.sample], sample_row0[u.sample] + n, dtype=torch.int32)
            t0 = Mv + 2 * i
            row_cs[t0:t0 + 2] = i
            row_slot[t0] = ks + u.ctx_len + n
            row_slot[t0 + 1] = ks + u.ctx_len + n + 1
            kv_len = u.ctx_len + n + 2
            for q0 in range(0, n, 128):
                tiles.append((r0 + q0, min(128, n - q0), ks, kv_len, -1, 0, 0, 0))
            tiles.append((t0, 2, ks, kv_len, -1, 0, 0, 0))
        cos, sin = rope_table([u.rope_pos for u in units], d.head_dim, d.rope_theta, self.device)
        dev = self.device
        return GenPlan(units, nv, Mv, Mt, S, kv_start, vae_row0, row_cs.to(dev), row_slot.to(dev), src_row.to(dev),
                       torch.tensor(tiles, dtype=torch.int32).to(dev), len(tiles), cos, sin)

    @staticmethod
    def _zero_value_padding(vtws: torch.Tensor, ranges):
        """The attention kernel masks the scores of key slots past a sample's length, but their probabilities
        (exact zeros) still multiply whatever the value slots hold, and 0 x NaN is NaN: the value slots between a
        sample's last key and the end of its 64-slot padding must be finite.  Only those are cleared (a few
        columns per sample), not the workspace (17 GB for a batch of long editing contexts).  Key slots need
        nothing: masked scores are replaced, not multiplied."""
        for a, b in ranges:
            if b > a:
                vtws[..., a:b].zero_()

    def load_context(self, plan: GenPlan):
        """Allocate the KV workspace and copy every unit's context keys/values into it (once per image).
        K [L][Hkv][S][D] row-major, Vt [L][Hkv][D][S] transposed."""
        d = self.dims
        L, Hkv, D, S = d.num_layers, d.num_kv_heads, d.head_dim, plan.S
        self.kws = self._get("kws", (L, Hkv, S, D), zero=True)
        self.vtws = self._get("vtws", (L, Hkv, D, S), zero=True)
        ends = plan.kv_start[1:] + [S]
        self._zero_value_padding(self.vtws, [(ks + u.ctx_len + n + 2, e) for u, n, ks, e in
                                             zip(plan.units, plan.nv, plan.kv_start, ends)])
        for u, ks in zip(plan.units, plan.kv_start):
            if u.ctx_len == 0:
                continue
            for l in range(L):
                self.kws[l, :, ks:ks + u.ctx_len, :].copy_(u.ctx_k[l].permute(1, 0, 2))
                self.vtws[l, :, :, ks:ks + u.ctx_len].copy_(u.ctx_v[l].permute(1, 2, 0))

    # ---- the forward ----------------------------------------------------------------------
    def taylor_velocity(self, factors: torch.Tensor, rows: int, n_factors: int, x: int) -> torch.Tensor:
        """A CFG branch's velocities on one of its skipped ("Taylor") steps: the Taylor expansion of its cached
        last-layer output (every layer of the reference returns its own expansion and ignores its input,
        qwen2_navit.py:838-843, so only the last layer's reaches the head), then final norm + llm2vae.  No embedding,
        no layers, no KV."""
        d, w = self.dims, self.w
        X = self._get("tX", (rows, d.hidden_size))
        Xn = self._get("tXn", (rows, d.hidden_size))
        V = self._get("tV", (rows, d.patch_latent_dim))
        K.taylor_eval(factors, rows, d.hidden_size, n_factors, x, X)
        K.rmsnorm(X, Xn, w.norm[0], w.norm[1], rows, d.rms_norm_eps, rows=rows)
        K.gemm(Xn, w.l2v_w, V, bias=w.l2v_b, M=rows)
        return V

    def forward_gen(self, plan: GenPlan, x_bf16: torch.Tensor, temb: torch.Tensor, text_emb: torch.Tensor,
                    taylor_updates: Optional[List["TaylorUpdate"]] = None):
        """x_bf16 [rows of the packed latent, P], temb [H], text_emb [Mt, H] -> velocities [Mv, P] (bf16),
        rows in unit order.  Mirrors Bagel._forward_flow's embedding + forward_inference(mode="gen") +
        llm2vae for all units at once (bagel.py:1389-1426, qwen2_navit.py:1041-1115).

        ``taylor_updates`` (TaylorSeer): after the layers, the last layer's output of each listed row range updates
        that CFG branch's factor cache (derivative_approximation, modeling/cache_utils/taylorseer.py:11-29)."""
        d, w = self.dims, self.w
        H, I, Hq, Hkv, D = d.hidden_size, d.intermediate_size, d.num_heads, d.num_kv_heads, d.head_dim
        Mv, Mt, M = plan.Mv, plan.Mt, plan.M
        qkv_dim = d.q_dim + 2 * d.kv_dim
        X = self._get("X", (M, H))
        Xn = self._get("Xn", (M, H))
        QKV = self._get("QKV", (M, qkv_dim))
        Q = self._get("Q", (M, d.q_dim))
        O = self._get("O", (M, d.q_dim))
        Hm = self._get("Hm", (M, I))
        lin = self._get("lin", (x_bf16.shape[0], H))
        Vout = self._get("Vout", (Mv, d.patch_latent_dim))
        skw = self._get("skinny_ws", (16 * 64 * max(2 * I, qkv_dim),), dtype=torch.float32)
        eps = d.rms_norm_eps

        # embedding: vae2llm(x_t) + t_emb + pos_embed on latent rows, embed_tokens on text rows
        K.gemm(x_bf16, w.v2l_w, lin, bias=w.v2l_b)
        K.embed_latent(lin, temb, w.pos_embed, self._pos_id, plan.src_row, X, Mv)
        K.gather_rows(text_emb, None, X[Mv:], Mt)

        # all layers + the final norm of the latent rows in one C call (the text rows' outputs are never read,
        # bagel.py:1426)
        rows = K.mot_rows(Mv, Mt, X, Xn, QKV, Q, O, Hm, plan.cos, plan.sin, plan.row_cs, plan.row_slot, plan.tiles,
                          plan.n_tiles, plan.S, skw)
        K.mot_forward(self._cdims, w.layer_table(), len(w.layers), rows, self.kws, self.vtws, final_norm_gen=w.norm[0],
                      final_norm_und=w.norm[1], final_norm_rows=Mv)

        for tu in taylor_updates or ():
            K.taylor_update(X[tu.row0:tu.row0 + tu.rows], tu.factors, tu.n_factors, tu.distance)
        K.gemm(Xn, w.l2v_w, Vout, bias=w.l2v_b, M=Mv)
        return Vout

    def extract_gen_kv(self, plan: GenPlan):
        """After forward_gen: per layer the merged keys/values ``[sum(ctx + n + 2), Hkv, D]`` of the plan's units
        in the reference's order [context | start_of_image | latent tokens | end_of_image] per sample
        (the workspace keeps the two text rows after the latent rows)."""
        L = self.dims.num_layers
        order = []
        for u, n, ks in zip(plan.units, plan.nv, plan.kv_start):
            c = u.ctx_len
            order += list(range(ks, ks + c)) + [ks + c + n] + list(range(ks + c, ks + c + n)) + [ks + c + n + 1]
        idx = torch.tensor(order, dtype=torch.long, device=self.device)
        keys = [self.kws[l].index_select(1, idx).permute(1, 0, 2).contiguous() for l in range(L)]
        values = [self.vtws[l].index_select(2, idx).permute(2, 0, 1).contiguous() for l in range(L)]
        return keys, values

    # ---- understanding-mode prefill (context KV build) ------------------------------------------
    def prefill_und(self, emb: torch.Tensor, lens: Sequence[int], pos0: Sequence[int], ctx_lens: Sequence[int],
                    ctx_k: Sequence[Optional[List[torch.Tensor]]], ctx_v: Sequence[Optional[List[torch.Tensor]]],
                    is_causal: bool = True, same_position: bool = False):
        """Qwen2Model.forward_inference(mode="und", update_past_key_values=True) for packed text tokens
        (qwen2_navit.py:1041-1115 via bagel.py:844-874): returns per layer the merged keys and values
        ``[sum(ctx + new), Hkv, D]`` in sample order, i.e. the tensors NaiveCache holds afterwards.
        Sample b contributes ``lens[b]`` new tokens at RoPE positions pos0[b].. after ``ctx_lens[b]``
        context tokens (``same_position``: all of a sample's tokens share position pos0[b], as the tokens of
        one image do, bagel.py:917).  The last layer's attention and MLP are skipped: only keys/values are kept."""
        d, w = self.dims, self.w
        H, I, Hq, Hkv, D = d.hidden_size, d.intermediate_size, d.num_heads, d.num_kv_heads, d.head_dim
        M = int(sum(lens))
        assert emb.shape[0] == M
        kv_start, row0 = [], []
        s = r = 0
        for n, c in zip(lens, ctx_lens):
            kv_start.append(s)
            row0.append(r)
            s += (c + n + 63) // 64 * 64
            r += n
        S = s
        L = d.num_layers
        kws = self._get("kws_und", (L, Hkv, S, D), zero=True)
        vtws = self._get("vtws_und", (L, Hkv, D, S), zero=True)
        self._zero_value_padding(vtws, [(ks + c + n, e) for ks, n, c, e in zip(kv_start, lens, ctx_lens, kv_start[1:] + [S])])
        positions, row_slot, tiles = [], [], []
        for b, (n, c) in enumerate(zip(lens, ctx_lens)):
            ks = kv_start[b]
            if c > 0:
                for l in range(L):
                    kws[l, :, ks:ks + c, :].copy_(ctx_k[b][l].permute(1, 0, 2))
                    vtws[l, :, :, ks:ks + c].copy_(ctx_v[b][l].permute(1, 2, 0))
            positions += [pos0[b]] * n if same_position else list(range(pos0[b], pos0[b] + n))
            row_slot += list(range(ks + c, ks + c + n))
            for q0 in range(0, n, 128):
                tiles.append((row0[b] + q0, min(128, n - q0), ks, c + n, (c + q0) if is_causal else -1, 0, 0, 0))
        cos, sin = rope_table(positions, D, d.rope_theta, self.device)
        row_cs = torch.arange(M, dtype=torch.int32, device=self.device)
        row_slot = torch.tensor(row_slot, dtype=torch.int32, device=self.device)
        tiles_t = torch.tensor(tiles, dtype=torch.int32, device=self.device)
        qkv_dim = d.q_dim + 2 * d.kv_dim
        X = self._get("uX", (M, H))
        X.copy_(emb)
        Xn = self._get("uXn", (M, H))
        QKV = self._get("uQKV", (M, qkv_dim))
        Q = self._get("uQ", (M, d.q_dim))
        O = self._get("uO", (M, d.q_dim))
        Hm = self._get("uHm", (M, I))
        eps = d.rms_norm_eps
        skw = self._get("skinny_ws", (16 * 64 * max(2 * I, qkv_dim),), dtype=torch.float32)
        rows = K.mot_rows(0, M, X, Xn, QKV, Q, O, Hm, cos, sin, row_cs, row_slot, tiles_t, len(tiles), S, skw,
                          und_rounding=True)
        # the last layer's attention and MLP are skipped: only its keys/values are kept
        K.mot_forward(self._cdims, w.layer_table(), L, rows, kws, vtws, kv_only_last=True)
        keys, values = [], []
        for l in range(L):
            keys.append(torch.cat([kws[l, :, ks:ks + c + n, :].permute(1, 0, 2) for ks, n, c in zip(kv_start, lens, ctx_lens)]).contiguous())
            values.append(torch.cat([vtws[l, :, :, ks:ks + c + n].permute(2, 0, 1) for ks, n, c in zip(kv_start, lens, ctx_lens)]).contiguous())
        return keys, values

    # ---- general packed forward (the language model's forward_inference) ------------------------
    def forward_sequence(self, seq: torch.Tensor, lens: Sequence[int], positions: Sequence[int], ctx_lens: Sequence[int],
                         ctx_k, ctx_v, is_causal: bool, gen_rows: Optional[Sequence[int]] = None,
                         need_kv: bool = True):
        """Qwen2Model.forward_inference (qwen2_navit.py:1041-1115) for an already embedded packed sequence.

        ``seq`` [M, H]: sample b owns ``lens[b]`` consecutive rows that attend its ``ctx_lens[b]`` context tokens
        plus (causally or not) the sample's own rows; ``positions`` is the RoPE position of every row.
        ``gen_rows`` (mode="gen"): rows that go through the generation expert; the others use the understanding
        expert.  ``None`` = mode="und".  Returns (final-normed hidden states [M, H] in the given row order,
        merged keys, merged values) with keys/values as NaiveCache holds them ([sum(ctx + len), Hkv, D] per
        layer) or None when ``need_kv`` is false.

        Rows are processed sorted by expert (generation rows first) so each expert sees one contiguous row
        block; key slots keep the sequence order.  A causal mask needs the sequence order inside a tile, so it
        is only offered when all rows use one expert."""
        d, w = self.dims, self.w
        H, I, Hq, Hkv, D, L = d.hidden_size, d.intermediate_size, d.num_heads, d.num_kv_heads, d.head_dim, d.num_layers
        M = int(sum(lens))
        assert seq.shape[0] == M and len(positions) == M
        dev = self.device
        gen = sorted(int(r) for r in gen_rows) if gen_rows is not None else []
        gen_set = set(gen)
        und = [r for r in range(M) if r not in gen_set]
        Mv = len(gen)
        if is_causal and Mv not in (0, M):
            raise NotImplementedError("a causal mask over rows of both experts is not supported")
        perm = gen + und                      # sorted row -> original row
        inv = [0] * M
        for i, r in enumerate(perm):
            inv[r] = i
        kv_start, row0 = [], []
        s = r = 0
        for n, c in zip(lens, ctx_lens):
            kv_start.append(s)
            row0.append(r)
            s += (c + n + 63) // 64 * 64
            r += n
        S = s
        kws = self._get("kws_seq", (L, Hkv, S, D), zero=True)
        vtws = self._get("vtws_seq", (L, Hkv, D, S), zero=True)
        self._zero_value_padding(vtws, [(ks + c + n, e) for ks, n, c, e in zip(kv_start, lens, ctx_lens, kv_start[1:] + [S])])
        slot_of = [0] * M
        tiles = []
        for b, (n, c) in enumerate(zip(lens, ctx_lens)):
            ks = kv_start[b]
            if c > 0:
                for l in range(L):
                    kws[l, :, ks:ks + c, :].copy_(ctx_k[b][l].permute(1, 0, 2))
                    vtws[l, :, :, ks:ks + c].copy_(ctx_v[b][l].permute(1, 2, 0))
            rows = list(range(row0[b], row0[b] + n))
            for j, rr in enumerate(rows):
                slot_of[rr] = ks + c + j
            # the sample's rows of each expert are contiguous after the sort
            for block in ([inv[rr] for rr in rows if rr in gen_set], [inv[rr] for rr in rows if rr not in gen_set]):
                for q0 in range(0, len(block), 128):
                    first = block[q0]
                    causal_base = (c + (perm[first] - row0[b])) if is_causal else -1
                    tiles.append((first, min(128, len(block) - q0), ks, c + n, causal_base, 0, 0, 0))
        cos, sin = rope_table([positions[r_] for r_ in perm], D, d.rope_theta, dev)
        row_cs = torch.arange(M, dtype=torch.int32, device=dev)
        row_slot = torch.tensor([slot_of[r_] for r_ in perm], dtype=torch.int32, device=dev)
        tiles_t = torch.tensor(tiles, dtype=torch.int32, device=dev)
        perm_t = torch.tensor(perm, dtype=torch.int32, device=dev)
        qkv_dim = d.q_dim + 2 * d.kv_dim
        X = self._get("sX", (M, H))
        K.gather_rows(seq.to(device=dev, dtype=BF16).contiguous(), perm_t, X, M)
        Xn = self._get("sXn", (M, H))
        QKV = self._get("sQKV", (M, qkv_dim))
        Q = self._get("sQ", (M, d.q_dim))
        O = self._get("sO", (M, d.q_dim))
        Hm = self._get("sHm", (M, I))
        eps = d.rms_norm_eps
        Mt = M - Mv

        skw = self._get("skinny_ws", (16 * 64 * max(2 * I, qkv_dim),), dtype=torch.float32)
        # mode="und" rounds the normalised q/k and the RoPE products to bf16 (qwen2_navit.py:526-530), the mixed graph
        # keeps them in fp32 until the final cast (:549-568)
        rows = K.mot_rows(Mv, Mt, X, Xn, QKV, Q, O, Hm, cos, sin, row_cs, row_slot, tiles_t, len(tiles), S, skw,
                          und_rounding=gen_rows is None)
        K.mot_forward(self._cdims, w.layer_table(), L, rows, kws, vtws, final_norm_gen=w.norm[0], final_norm_und=w.norm[1])
        hidden = torch.empty((M, H), device=dev, dtype=BF16)
        K.gather_rows(Xn, torch.tensor(inv, dtype=torch.int32, device=dev), hidden, M)
        keys = values = None
        if need_kv:
            keys = [torch.cat([kws[l, :, ks:ks + c + n, :].permute(1, 0, 2) for ks, n, c in zip(kv_start, lens, ctx_lens)]).contiguous()
                    for l in range(L)]
            values = [torch.cat([vtws[l, :, :, ks:ks + c + n].permute(2, 0, 1) for ks, n, c in zip(kv_start, lens, ctx_lens)]).contiguous()
                      for l in range(L)]
        return hidden, keys, values

    # ---- autoregressive text decoding ---------------------------------------------------------
    def decode_tokens(self, start_tokens: Sequence[int], pos0: Sequence[int], ctx_lens: Sequence[int],
                      ctx_k, ctx_v, max_length: int, end_token_id: Optional[int] = None, sample_fn=None,
                      logits_trace: Optional[list] = None) -> torch.Tensor:
        """Bagel.generate_text (bagel.py:1526-1596): one token per sample per step through the understanding
        expert, causal over [context | generated so far].

        The reference re-merges the whole KV cache and rebuilds its index lists on the host for every token
        (qwen2_navit.py:574-581, bagel.py:1551-1587).  Here the KV workspace is sized once for
        context + max_length, each step's keys/values are written in place at slot context + step, every
        index table of every step is built once up front, and the chosen token never leaves the device
        between steps (argmax kernel -> embedding gather); the host only looks at the tokens every 16 steps
        to honour ``end_token_id`` (batch-1 semantics of bagel.py:1592).
        Returns the fed tokens ``[steps, B]`` (int64), starting with ``start_tokens``.
        """
        d, w = self.dims, self.w
        assert w.lm_head is not None, "text decoding needs language_model.lm_head.weight"
        H, I, Hq, Hkv, D, L = d.hidden_size, d.intermediate_size, d.num_heads, d.num_kv_heads, d.head_dim, d.num_layers
        B = len(start_tokens)
        dev = self.device
        kv_start, s = [], 0
        for c in ctx_lens:
            kv_start.append(s)
            s += (c + max_length + 63) // 64 * 64
        S = s
        kws = self._get("kws_dec", (L, Hkv, S, D), zero=True)
        vtws = self._get("vtws_dec", (L, Hkv, D, S), zero=True)
        # slots of tokens not generated yet are masked like padding until their step writes them
        self._zero_value_padding(vtws, [(ks + c, e) for ks, c, e in zip(kv_start, ctx_lens, kv_start[1:] + [S])])
        for b, c in enumerate(ctx_lens):
            if c > 0:
                for l in range(L):
                    kws[l, :, kv_start[b]:kv_start[b] + c, :].copy_(ctx_k[b][l].permute(1, 0, 2))
                    vtws[l, :, :, kv_start[b]:kv_start[b] + c].copy_(ctx_v[b][l].permute(1, 2, 0))
        # per-step tables for all steps at once: RoPE rows, key slots, attention tiles (one query row each)
        positions = [pos0[b] + t for t in range(max_length) for b in range(B)]
        cos, sin = rope_table(positions, D, d.rope_theta, dev)
        row_cs = torch.arange(max_length * B, dtype=torch.int32, device=dev).view(max_length, B)
        row_slot = torch.tensor([[kv_start[b] + ctx_lens[b] + t for b in range(B)] for t in range(max_length)],
                                dtype=torch.int32, device=dev)
        tiles = torch.tensor([[(b, 1, kv_start[b], ctx_lens[b] + t + 1, -1, 0, 0, 0) for b in range(B)]
                              for t in range(max_length)], dtype=torch.int32, device=dev)
        seq = torch.zeros((max_length + 1, B), dtype=torch.int32, device=dev)
        seq[0] = torch.tensor(list(start_tokens), dtype=torch.int32, device=dev)
        qkv_dim = d.q_dim + 2 * d.kv_dim
        X = self._get("dX", (B, H))
        Xn = self._get("dXn", (B, H))
        QKV = self._get("dQKV", (B, qkv_dim))
        Q = self._get("dQ", (B, d.q_dim))
        O = self._get("dO", (B, d.q_dim))
        Hm = self._get("dHm", (B, I))
        V = w.lm_head.shape[0]
        logits = self._get("dlogits", (B, V))
        assert 1 <= B <= 64, "decode_tokens handles 1..64 samples per call"
        skw = self._get("skinny_ws_dec", (16 * B * max(2 * I, qkv_dim, V),), dtype=torch.float32)
        eps = d.rms_norm_eps

        steps_done = 0
        n_layers = len(w.layers)
        # one argument block for every step: only the three table pointers move (tables of all steps were built above)
        # scratch of the one-launch decode attention (tickets + partial sums), zero when first allocated and left
        # zero by every launch
        dws = None
        if K.decode_attention_supported(B, Hq, Hkv, D):
            dws = self._get("attn_ws_dec", (K.decode_attention_workspace_bytes(B, Hq, Hkv, D),), dtype=torch.uint8, zero=True)
            dws[:1024].zero_()   # the tickets: zero by contract; re-established here in case an earlier call was cut short
        rows = K.mot_rows(0, B, X, Xn, QKV, Q, O, Hm, cos, sin, row_cs[0], row_slot[0], tiles[0], B, S, skw, und_rounding=True,
                          decode_ws=dws, decode_max_kv=max(ctx_lens) + max_length)
        cs0, slot0, tile0 = row_cs.data_ptr(), row_slot.data_ptr(), tiles.data_ptr()
        table = w.layer_table()
        for t in range(max_length):
            K.gather_rows(w.embed_tokens, seq[t], X, B)
            rows.row_cs, rows.row_slot, rows.tiles = cs0 + 4 * B * t, slot0 + 4 * B * t, tile0 + 32 * B * t
            # 28 layers in one C call.  Up to four samples: five launches per layer (the RMSNorms run inside the
            # weight-streaming kernels that consume them; q/k norm, RoPE, KV write and attention are one kernel); more: o_proj / down_proj reduce their partial sums, add the
            # residual and apply the RMSNorm that follows in one launch
            K.mot_forward(self._cdims, table, n_layers, rows, kws, vtws)
            # final norm + lm_head in one weight-streaming launch; above 16 samples the tile kernel (every token-row
            # fragment shared through LDS) is ahead of streaming, as in the layers
            if B > 16:
                K.rmsnorm(X, Xn, w.norm[1], w.norm[1], B, eps)
                K.gemm(Xn, w.lm_head, logits)
            else:
                K.skinny_gemm_rmsnorm_in(X, w.norm[1], eps, w.lm_head, logits, skw, Xn=Xn)
            if logits_trace is not None:
                logits_trace.append(logits.clone())
            if sample_fn is None:
                K.argmax_rows(logits, seq[t + 1])
            else:
                seq[t + 1] = sample_fn(logits).to(torch.int32)
            steps_done = t + 1
            if end_token_id is not None and (t % 16 == 15 or t == max_length - 1):
                hit = (seq[1:t + 2, 0] == end_token_id).nonzero()
                if hit.numel() > 0:
                    steps_done = int(hit[0]) + 1   # the step that produced the end token is the last one run
                    break
        if end_token_id is not None:
            hit = (seq[1:steps_done + 1, 0] == end_token_id).nonzero()
            if hit.numel() > 0:
                steps_done = int(hit[0]) + 1
        return seq[:steps_done].to(torch.long)

    def set_latent_position_ids(self, pos_ids: torch.Tensor):
        self._pos_id = pos_ids.to(device=self.device, dtype=torch.int32)

    # ---- timestep embedding table -----------------------------------------------------------
    def timestep_table(self, timesteps: torch.Tensor) -> torch.Tensor:
        """t_emb rows for every flow step at once (TimestepEmbedder.forward, modeling_utils.py:109-112).
        The reference evaluates this MLP on Nv identical rows per step (bagel.py:1393-1395); here one row
        per step, all steps in one weight-streaming pass."""
        d, w = self.dims, self.w
        n = timesteps.numel()
        t_dev = timesteps.to(device=self.device, dtype=torch.float32).contiguous()
        sin = self._get("t_sin", (n, d.timestep_freq_dim))
        K.timestep_sinusoid(t_dev, sin)
        h = self._get("t_h", (n, d.hidden_size))
        out = torch.empty((n, d.hidden_size), device=self.device, dtype=BF16)
        skw = self._get("skinny_ws_t", (16 * 64 * d.hidden_size,), dtype=torch.float32)
        for r0 in range(0, n, 64):
            m = min(64, n - r0)
            K.skinny_gemm(sin[r0:r0 + m], w.t0_w, h[r0:r0 + m], skw, bias=w.t0_b, epilogue=K.SK_BIAS_SILU)
            K.skinny_gemm(h[r0:r0 + m], w.t2_w, out[r0:r0 + m], skw, bias=w.t2_b)
        return out
