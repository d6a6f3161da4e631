"""Build and load libbagel_hip.so (the C-ABI of include/bagel_hip.h) and wrap it for torch tensors.

There is no fallback: if the library is missing or an entry point fails, a RuntimeError is raised.
torch is used only to own device memory and to name the current HIP stream.
"""
from __future__ import annotations

import ctypes
import glob
import hashlib
import os
import re
import subprocess
from typing import Optional

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.path.join(HERE, "libbagel_hip.so")
HEADER = os.path.join(os.path.dirname(HERE), "include", "bagel_hip.h")
SOURCES = ["capi.hip", "gemm.hip", "skinny.hip", "attention.hip", "elementwise.hip", "vae.hip", "layer.hip", "decode_attention.hip"]

_lib: Optional[ctypes.CDLL] = None


STAMP_PATH = LIB_PATH + ".sha256"   # travels with the library


def source_digest() -> str:
    """SHA-256 over every file the library is compiled from (csrc/*.hip, csrc/*.h, the public header)."""
    h = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h"))) + [HEADER]:
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile csrc/*.hip for gfx950 into bagel_amd/libbagel_hip.so (cross-compiles without a GPU).

    The library is rebuilt whenever the digest of its sources differs from the one recorded next to it at
    the last build, so an edit to any kernel header is never silently ignored."""
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    digest = source_digest()
    if not force and os.path.exists(LIB_PATH) and os.path.exists(STAMP_PATH) and open(STAMP_PATH).read().strip() == digest:
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for s in srcs:
        o = os.path.join(HERE, "build", os.path.basename(s) + ".o")
        objs.append(o)
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-c", s, "-o", o]
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for cmd, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd) + "\n" + out.decode())
        if verbose and out:
            print(out.decode())
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("link failed: " + r.stdout.decode())
    with open(STAMP_PATH, "w") as f:
        f.write(digest + "\n")
    return LIB_PATH


def declared_symbols() -> list:
    """Entry points declared in include/bagel_hip.h."""
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bagel_[a-z0-9_]+)\s*\(", text)))


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(the HIP extension is the only compute path; there is no fallback)")
        stamp = open(STAMP_PATH).read().strip() if os.path.exists(STAMP_PATH) else None
        if stamp != source_digest() and os.environ.get("BAGEL_ALLOW_STALE_LIB") != "1":
            raise RuntimeError(f"{LIB_PATH} was not built from the sources in {CSRC} (digest mismatch): run "
                               "`python -c 'import __graft_entry__ as g; g.build()'`")
        # development: an A/B build of the same sources with other flags (tools/build_variant.py)
        variant = os.environ.get("BAGEL_LIB_VARIANT")
        _lib = ctypes.CDLL(variant if variant else LIB_PATH)
        _lib.bagel_last_error.restype = ctypes.c_char_p
        for name in ("bagel_skinny_gemm_workspace_bytes", "bagel_cfg_workspace_bytes", "bagel_groupnorm_workspace_bytes",
                     "bagel_decode_attention_workspace_bytes"):
            if hasattr(_lib, name):
                getattr(_lib, name).restype = ctypes.c_long
    return _lib


def _check(rc: int, what: str):
    if rc != 0:
        msg = lib().bagel_last_error().decode()
        if "is not supported" in msg and "cfg_renorm_type" in msg:
            raise NotImplementedError(msg)
        raise RuntimeError(f"{what} failed ({rc}): {msg}")


def _p(t: Optional[torch.Tensor]):
    if t is None:
        return ctypes.c_void_p(0)
    assert t.is_cuda, "device tensor expected"
    if t.device.index != torch.cuda.current_device():
        raise RuntimeError(f"tensor on {t.device} but the current device is cuda:{torch.cuda.current_device()}: kernels are "
                           "launched on the current device's stream (wrap the call in torch.cuda.device(...))")
    return ctypes.c_void_p(t.data_ptr())


def _stream():
    """The current HIP stream of the current device.  Every entry point launches on the current device, so
    tensors must live there: _p() checks it."""
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def on_device(fn):
    """Run a method with ``self.device`` as the current device: the kernels go to the current device's current
    stream, so a model placed on cuda:1 must not be driven while cuda:0 is current."""
    import functools

    @functools.wraps(fn)
    def wrapper(self, *args, **kwargs):
        dev = getattr(self, "device", None)
        if dev is None:
            dev = (getattr(self, "engine", None) or self._engine).device
        with torch.cuda.device(dev):
            return fn(self, *args, **kwargs)
    return wrapper


L = ctypes.c_long
I = ctypes.c_int
F = ctypes.c_float

EPI_BIAS, EPI_RESIDUAL, EPI_SWIGLU, EPI_BIAS_RESIDUAL, EPI_F32_SCALED, EPI_BIAS_GELU = 0, 1, 2, 3, 4, 5
SK_BIAS, SK_RESIDUAL, SK_SWIGLU, SK_BIAS_SILU = 0, 1, 2, 3
RENORM = {"global": 0, "channel": 1, "text_channel": 2}


def gemm_workspace_bytes(M, N, K, epilogue=EPI_BIAS):
    lib().bagel_gemm_workspace_bytes.restype = ctypes.c_long
    return lib().bagel_gemm_workspace_bytes(I(M), I(N), I(K), I(epilogue))


def gemm(A, W, C, bias=None, residual=None, epilogue=EPI_BIAS, M=None, alpha=1.0, workspace=None):
    """C[M,N] = epilogue(A[M,K] @ W[N,K]^T) (bf16 2-D tensors, rows may be strided).  workspace: a tensor the split-K
    form may use at 17 ... 1024 rows (bagel_gemm_ws_bf16)."""
    M = A.shape[0] if M is None else M
    N, K = W.shape
    assert A.dtype == W.dtype == torch.bfloat16 and A.shape[1] == K
    assert C.dtype == (torch.float32 if epilogue == EPI_F32_SCALED else torch.bfloat16)
    assert A.stride(1) == 1 and W.stride(1) == 1 and C.stride(1) == 1
    if workspace is not None:
        _check(lib().bagel_gemm_ws_bf16(_p(A), L(A.stride(0)), _p(W), L(W.stride(0)), _p(C), L(C.stride(0)), I(M), I(N), I(K),
                                        _p(bias), _p(residual), L(residual.stride(0) if residual is not None else 0),
                                        I(epilogue), F(alpha), _p(workspace), L(workspace.numel() * workspace.element_size()),
                                        _stream()), "bagel_gemm_ws_bf16")
        return C
    _check(lib().bagel_gemm_bf16(_p(A), L(A.stride(0)), _p(W), L(W.stride(0)), _p(C), L(C.stride(0)), I(M), I(N), I(K),
                                 _p(bias), _p(residual), L(residual.stride(0) if residual is not None else 0),
                                 I(epilogue), F(alpha), _stream()), "bagel_gemm_bf16")
    return C


def gemm_residual_rmsnorm(A, W, C, residual, norm_weight, Xn, eps, workspace):
    """C = residual + A @ W^T and Xn = RMSNorm(C) * norm_weight (bagel_gemm_ws_residual_rmsnorm_bf16)."""
    M, K = A.shape
    N = W.shape[0]
    _check(lib().bagel_gemm_ws_residual_rmsnorm_bf16(_p(A), L(A.stride(0)), _p(W), L(W.stride(0)), _p(C), L(C.stride(0)), I(M), I(N),
                                                     I(K), _p(residual), L(residual.stride(0)), _p(norm_weight), _p(Xn),
                                                     L(Xn.stride(0)), F(eps), _p(workspace),
                                                     L(workspace.numel() * workspace.element_size()), _stream()),
           "bagel_gemm_ws_residual_rmsnorm_bf16")
    return C


def skinny_workspace_bytes(M, N):
    return lib().bagel_skinny_gemm_workspace_bytes(I(M), I(N))


def skinny_gemm(X, W, C, workspace, bias=None, residual=None, epilogue=SK_BIAS, gemv=False):
    """gemv=True: the decode loop's one-launch form where the shape allows it (bagel_skinny_gemv_bf16)."""
    M, K = X.shape
    N = W.shape[0]
    assert X.dtype == W.dtype == C.dtype == torch.bfloat16 and W.shape[1] == K
    fn, name = (lib().bagel_skinny_gemv_bf16, "bagel_skinny_gemv_bf16") if gemv else \
               (lib().bagel_skinny_gemm_bf16, "bagel_skinny_gemm_bf16")
    _check(fn(_p(X), L(X.stride(0)), _p(W), L(W.stride(0)), _p(C), L(C.stride(0)), I(M), I(N), I(K), _p(bias), _p(residual),
              L(residual.stride(0) if residual is not None else 0), I(epilogue), _p(workspace),
              L(workspace.numel() * workspace.element_size()), _stream()), name)
    return C


def skinny_gemm_residual_rmsnorm(X, W, C, residual, norm_weight, Xn, eps, workspace):
    M, K = X.shape
    N = W.shape[0]
    assert X.dtype == W.dtype == C.dtype == Xn.dtype == torch.bfloat16 and W.shape[1] == K
    _check(lib().bagel_skinny_gemm_residual_rmsnorm_bf16(
        _p(X), L(X.stride(0)), _p(W), L(W.stride(0)), _p(C), L(C.stride(0)), I(M), I(N), I(K), _p(residual),
        L(residual.stride(0)), _p(norm_weight), _p(Xn), L(Xn.stride(0)), F(eps), _p(workspace),
        L(workspace.numel() * workspace.element_size()), _stream()), "bagel_skinny_gemm_residual_rmsnorm_bf16")
    return C


def skinny_gemm_rmsnorm_in(X, norm_weight, eps, W, C, workspace, Xn=None, bias=None, epilogue=SK_BIAS):
    """C = epilogue(RMSNorm(X) @ W^T) for at most 64 rows; Xn: scratch rows for shapes the fused kernel does not take."""
    M, K = X.shape
    N = W.shape[0]
    _check(lib().bagel_skinny_gemm_rmsnorm_in_bf16(_p(X), L(X.stride(0)), _p(norm_weight), F(eps), _p(W), L(W.stride(0)), _p(C),
                                                   L(C.stride(0)), I(M), I(N), I(K), _p(bias), I(epilogue), _p(Xn),
                                                   L(Xn.stride(0) if Xn is not None else 0), _p(workspace),
                                                   L(workspace.numel() * workspace.element_size()), _stream()),
           "bagel_skinny_gemm_rmsnorm_in_bf16")
    return C


def rmsnorm(x, out, w_first, w_second, split_row, eps, rows=None):
    rows = x.shape[0] if rows is None else rows
    _check(lib().bagel_rmsnorm_bf16(_p(x), L(x.stride(0)), _p(out), L(out.stride(0)), I(rows), I(x.shape[1]),
                                    _p(w_first), _p(w_second), I(split_row), F(eps), _stream()), "bagel_rmsnorm_bf16")
    return out


def qk_norm_rope_kv(qkv, M, Hq, Hkv, D, qn_first, qn_second, kn_first, kn_second, split_row, cos_t, sin_t, row_cs,
                    row_slot, Q, K_ws, Vt_ws, S, eps, und_rounding=False):
    _check(lib().bagel_qk_norm_rope_kv_bf16(_p(qkv), L(qkv.stride(0)), I(M), I(Hq), I(Hkv), I(D), _p(qn_first),
                                            _p(qn_second), _p(kn_first), _p(kn_second), I(split_row), _p(cos_t),
                                            _p(sin_t), _p(row_cs), _p(row_slot), _p(Q), _p(K_ws), _p(Vt_ws), L(S),
                                            F(eps), I(1 if und_rounding else 0), _stream()),
           "bagel_qk_norm_rope_kv_bf16")


def varlen_attention(Q, K_ws, Vt_ws, O, tiles, n_tiles, Hq, Hkv, D, S, softmax_scale=None):
    if softmax_scale is None:
        _check(lib().bagel_varlen_attention_bf16(_p(Q), _p(K_ws), _p(Vt_ws), _p(O), _p(tiles), I(n_tiles), I(Hq),
                                                 I(Hkv), I(D), L(S), _stream()), "bagel_varlen_attention_bf16")
    else:
        _check(lib().bagel_varlen_attention_scaled_bf16(_p(Q), _p(K_ws), _p(Vt_ws), _p(O), _p(tiles), I(n_tiles), I(Hq),
                                                        I(Hkv), I(D), L(S), F(softmax_scale), _stream()),
               "bagel_varlen_attention_scaled_bf16")
    return O


def layernorm(x, out, gamma, beta, eps, rows=None):
    rows = x.shape[0] if rows is None else rows
    _check(lib().bagel_layernorm_bf16(_p(x), L(x.stride(0)), _p(out), L(out.stride(0)), I(rows), I(x.shape[1]), _p(gamma),
                                      _p(beta), F(eps), _stream()), "bagel_layernorm_bf16")
    return out


def split_heads_pad(qkv, M, heads, d_in, d, row_slot, Q, K_ws, Vt_ws, S):
    _check(lib().bagel_split_heads_pad_bf16(_p(qkv), L(qkv.stride(0)), I(M), I(heads), I(d_in), I(d), _p(row_slot), _p(Q),
                                            _p(K_ws), _p(Vt_ws), L(S), _stream()), "bagel_split_heads_pad_bf16")


def embed_latent(lin, temb, pos_table, pos_id, src_row, X, rows):
    _check(lib().bagel_embed_latent_bf16(_p(lin), L(lin.stride(0)), _p(temb), _p(pos_table), _p(pos_id), _p(src_row),
                                         _p(X), L(X.stride(0)), I(rows), I(X.shape[1]), _stream()),
           "bagel_embed_latent_bf16")


def gather_rows(table, idx, out, rows):
    _check(lib().bagel_gather_rows_bf16(_p(table), L(table.stride(0)), _p(idx), _p(out), L(out.stride(0)), I(rows),
                                        I(out.shape[1]), _stream()), "bagel_gather_rows_bf16")


def timestep_sinusoid(t, out):
    _check(lib().bagel_timestep_sinusoid_bf16(_p(t), I(t.numel()), I(out.shape[1]), _p(out), _stream()),
           "bagel_timestep_sinusoid_bf16")


def cast_f32_bf16(x, out):
    _check(lib().bagel_cast_f32_bf16(_p(x), _p(out), L(x.numel()), _stream()), "bagel_cast_f32_bf16")


def taylor_update(feature, factors, n_new, distance):
    rows, H = feature.shape
    _check(lib().bagel_taylor_update_bf16(_p(feature), L(feature.stride(0)), I(rows), I(H), _p(factors),
                                          L(factors.stride(0)), I(n_new), I(distance), _stream()),
           "bagel_taylor_update_bf16")


def taylor_eval(factors, rows, H, n_factors, x, out):
    _check(lib().bagel_taylor_eval_bf16(_p(factors), L(factors.stride(0)), I(rows), I(H), I(n_factors), I(x), _p(out),
                                        L(out.stride(0)), _stream()), "bagel_taylor_eval_bf16")


def argmax_rows(logits, out):
    rows, n = logits.shape
    _check(lib().bagel_argmax_rows_bf16(_p(logits), L(logits.stride(0)), I(rows), I(n), _p(out), _stream()),
           "bagel_argmax_rows_bf16")
    return out


# ---- one layer / the whole stack per call --------------------------------------------------------
VP = ctypes.c_void_p


class MotDims(ctypes.Structure):
    _fields_ = [("hidden", I), ("intermediate", I), ("num_heads", I), ("num_kv_heads", I), ("head_dim", I), ("rms_eps", F)]


class MotExpert(ctypes.Structure):
    _fields_ = [(n, VP) for n in ("wqkv", "bqkv", "wo", "wgu", "wd", "ln_in", "ln_post", "q_norm", "k_norm")]


class MotLayer(ctypes.Structure):
    _fields_ = [("gen", MotExpert), ("und", MotExpert)]


class MotRows(ctypes.Structure):
    _fields_ = [("rows_gen", I), ("rows_und", I), ("X", VP), ("Xn", VP), ("QKV", VP), ("Q", VP), ("O", VP), ("Hm", VP),
                ("cos_table", VP), ("sin_table", VP), ("row_cs", VP), ("row_slot", VP), ("tiles", VP), ("n_tiles", I),
                ("S", L), ("skinny_ws", VP), ("skinny_ws_bytes", L), ("und_rounding", I),
                ("decode_ws", VP), ("decode_ws_bytes", L), ("decode_max_kv", L)]


def _addr(t):
    return None if t is None else _p(t).value


def mot_rows(rows_gen, rows_und, X, Xn, QKV, Q, O, Hm, cos_t, sin_t, row_cs, row_slot, tiles, n_tiles, S, skinny_ws,
             und_rounding=False, decode_ws=None, decode_max_kv=0) -> MotRows:
    """The per-forward argument block of mot_forward; the tensors must outlive the call.  decode_ws (zeroed once by the
    caller, decode_attention_workspace_bytes) lets a decode step run its attention in one launch."""
    return MotRows(rows_gen, rows_und, _addr(X), _addr(Xn), _addr(QKV), _addr(Q), _addr(O), _addr(Hm), _addr(cos_t),
                   _addr(sin_t), _addr(row_cs), _addr(row_slot), _addr(tiles), n_tiles, S, _addr(skinny_ws),
                   skinny_ws.numel() * skinny_ws.element_size(), 1 if und_rounding else 0, _addr(decode_ws),
                   0 if decode_ws is None else decode_ws.numel() * decode_ws.element_size(), decode_max_kv)


def decode_attention_supported(rows, Hq, Hkv, D) -> bool:
    return bool(lib().bagel_decode_attention_supported(I(rows), I(Hq), I(Hkv), I(D)))


def decode_attention_workspace_bytes(rows, Hq, Hkv, D) -> int:
    return int(lib().bagel_decode_attention_workspace_bytes(I(rows), I(Hq), I(Hkv), I(D)))


def decode_attention(qkv, Hq, Hkv, D, q_norm, k_norm, cos_t, sin_t, row_cs, row_slot, tiles, K_ws, Vt_ws, S, O, eps,
                     und_rounding, workspace, max_kv_len=0):
    """One new token per row: q/k norm + RoPE, KV write and attention in one launch (bagel_decode_attention_bf16)."""
    _check(lib().bagel_decode_attention_bf16(_p(qkv), L(qkv.stride(0)), I(qkv.shape[0]), I(Hq), I(Hkv), I(D), _p(q_norm),
                                             _p(k_norm), _p(cos_t), _p(sin_t), _p(row_cs), _p(row_slot), _p(tiles), _p(K_ws),
                                             _p(Vt_ws), L(S), _p(O), F(eps), I(1 if und_rounding else 0), L(max_kv_len), _p(workspace),
                                             L(workspace.numel() * workspace.element_size()), _stream()),
           "bagel_decode_attention_bf16")


def mot_forward(dims: MotDims, layers, n_layers, rows: MotRows, K_ws, Vt_ws, kv_only_last=False, final_norm_gen=None,
                final_norm_und=None, final_norm_rows=0):
    """All decoder layers in one call (bagel_mot_forward); K_ws / Vt_ws are [L, ...] workspaces."""
    assert K_ws.stride(0) == Vt_ws.stride(0)
    _check(lib().bagel_mot_forward(ctypes.byref(dims), layers, I(n_layers), ctypes.byref(rows), _p(K_ws), _p(Vt_ws),
                                   L(K_ws.stride(0)), I(1 if kv_only_last else 0), _p(final_norm_gen), _p(final_norm_und),
                                   I(final_norm_rows), _stream()), "bagel_mot_forward")


def mot_layer_forward(dims: MotDims, layer: MotLayer, rows: MotRows, K_ws, Vt_ws, flags=0, next_norm_gen=None,
                      next_norm_und=None):
    _check(lib().bagel_mot_layer_forward(ctypes.byref(dims), ctypes.byref(layer), ctypes.byref(rows), _p(K_ws), _p(Vt_ws),
                                         I(flags), _p(next_norm_gen), _p(next_norm_und), _stream()), "bagel_mot_layer_forward")


def profile_gate_up(on: bool):
    lib().bagel_profile_gate_up(I(1 if on else 0))


def profile_collect(cap=1 << 16):
    ms = (ctypes.c_double * cap)()
    fl = (ctypes.c_double * cap)()
    n = lib().bagel_profile_collect(ms, fl, I(cap))
    return [(ms[i], fl[i]) for i in range(n)]


def cfg_workspace_bytes(rows, P):
    return lib().bagel_cfg_workspace_bytes(I(rows), I(P))


def cfg_euler_update(v_cond, v_text, v_img, cfg_text_scale, cfg_img_scale, renorm_type, renorm_min, dt, x_t, x_bf16,
                     v_t_out, workspace):
    if renorm_type not in RENORM:
        raise NotImplementedError(f"{renorm_type} is not suppoprted")
    rows, P = x_t.shape
    _check(lib().bagel_cfg_euler_update(_p(v_cond), _p(v_text), _p(v_img), F(cfg_text_scale), F(cfg_img_scale),
                                        I(RENORM[renorm_type]), F(renorm_min), F(dt), _p(x_t), _p(x_bf16),
                                        _p(v_t_out), I(rows), I(P), _p(workspace),
                                        L(workspace.numel() * workspace.element_size()), _stream()),
           "bagel_cfg_euler_update")


def cfg_euler_update_static(v_cond, v_text, v_img, cfg_text_scale, cfg_img_scale, renorm_type, renorm_min, dt, x_t,
                            x_bf16, v_t_out):
    """cfg_euler_update on the library's built-in scratch: no workspace, no allocation (rows <= 65536)."""
    if renorm_type not in RENORM:
        raise NotImplementedError(f"{renorm_type} is not suppoprted")
    rows, P = x_t.shape
    _check(lib().bagel_cfg_euler_update_static(_p(v_cond), _p(v_text), _p(v_img), F(cfg_text_scale), F(cfg_img_scale),
                                               I(RENORM[renorm_type]), F(renorm_min), F(dt), _p(x_t), _p(x_bf16),
                                               _p(v_t_out), I(rows), I(P), _stream()),
           "bagel_cfg_euler_update_static")


# ---- FLUX VAE decoder ---------------------------------------------------------------------------
def conv3x3(x, Hin, Win, Cin, w, bias, out, zero_page, upsample=False, residual=None, downsample=False):
    Cout = w.shape[0]
    _check(lib().bagel_conv3x3_nhwc_bf16(_p(x), I(Hin), I(Win), I(Cin), I(2 if downsample else 1 if upsample else 0), _p(w), I(Cout), _p(bias),
                                         _p(residual), _p(out), _p(zero_page), _stream()), "bagel_conv3x3_nhwc_bf16")
    return out


def groupnorm_workspace_bytes(C):
    return lib().bagel_groupnorm_workspace_bytes(I(C))


def groupnorm_swish(x, npix, C, gamma, beta, out, workspace, swish=True, eps=1e-6):
    _check(lib().bagel_groupnorm_swish_nhwc_bf16(_p(x), L(npix), I(C), _p(gamma), _p(beta), F(eps), I(1 if swish else 0),
                                                 _p(out), _p(workspace), L(workspace.numel() * workspace.element_size()),
                                                 _stream()), "bagel_groupnorm_swish_nhwc_bf16")
    return out


def softmax_rows(scores, probs):
    rows, n = scores.shape
    _check(lib().bagel_softmax_rows_f32_bf16(_p(scores), L(scores.stride(0)), I(rows), I(n), _p(probs),
                                             L(probs.stride(0)), _stream()), "bagel_softmax_rows_f32_bf16")
    return probs


def transpose(x, out):
    R, C = x.shape
    _check(lib().bagel_transpose_bf16(_p(x), I(R), I(C), _p(out), _stream()), "bagel_transpose_bf16")
    return out


def vae_latent_im2col(z, scale_factor, shift_factor, out):
    _, Cz, H, W = z.shape
    _check(lib().bagel_vae_latent_im2col_bf16(_p(z), I(Cz), I(H), I(W), F(scale_factor), F(shift_factor), _p(out),
                                              I(out.shape[1]), _stream()), "bagel_vae_latent_im2col_bf16")
    return out


def vae_sample_latent(moments, noise, Cz, H, W, scale_factor, shift_factor, z):
    _check(lib().bagel_vae_sample_latent_bf16(_p(moments), I(moments.stride(0)), _p(noise), I(Cz), I(H), I(W),
                                              F(scale_factor), F(shift_factor), _p(z), _stream()),
           "bagel_vae_sample_latent_bf16")
    return z


def nhwc_to_nchw(x, npix, Cpad, C, out):
    _check(lib().bagel_nhwc_to_nchw_bf16(_p(x), L(npix), I(Cpad), I(C), _p(out), _stream()), "bagel_nhwc_to_nchw_bf16")
    return out
