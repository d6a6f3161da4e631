"""``AutoEncoder``: the FLUX VAE (modeling/autoencoder.py:122-272, 290-322) on HIP kernels.

Weights come from a state dict with the reference's parameter names (``decoder.*`` and, for ``encode``,
``encoder.*`` of ae.safetensors, fp32).  At load time convolution weights are rounded to bf16 (what bf16
autocast does on every call) and re-laid out ``[Cout][ky][kx][Cin]`` for the implicit-GEMM kernel;
GroupNorm affines stay fp32.  Activations are NHWC bf16 on the device.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from .. import _lib as K
from ..config import VAEDims

BF16 = torch.bfloat16


class AutoEncoder:
    def __init__(self, params: VAEDims, state_dict: Dict[str, torch.Tensor], device="cuda:0"):
        K.lib()
        self.params = params
        self.device = torch.device(device)
        self.scale_factor = params.scale_factor
        self.shift_factor = params.shift_factor
        self.w: Dict[str, torch.Tensor] = {}
        dev = self.device
        parts = ("decoder.", "encoder.")
        conv_names = {n[:-len(".weight")] for n, t in state_dict.items() if n.startswith(parts) and t.dim() == 4}
        self.has_encoder = any(n.startswith("encoder.") for n in state_dict)
        for name, t in state_dict.items():
            if not name.startswith(parts):
                continue
            t = t.detach().float()
            if t.dim() == 4:
                cout, cin, kh, kw = t.shape
                if name in ("decoder.conv_in.weight", "encoder.conv_in.weight"):
                    # dense GEMM on im2col rows: K = (ky, kx, c) zero-padded to a multiple of 64
                    w = t.permute(0, 2, 3, 1).reshape(cout, kh * kw * cin)
                    kpad = (w.shape[1] + 63) // 64 * 64
                    wp = torch.zeros((cout, kpad))
                    wp[:, :w.shape[1]] = w
                    self.w[name] = wp.to(BF16).to(dev)
                elif kh == 3:
                    w = t.permute(0, 2, 3, 1).contiguous()      # [Cout][ky][kx][Cin]
                    if cout % 4:
                        w = torch.cat([w, torch.zeros((4 - cout % 4,) + tuple(w.shape[1:]))])
                    self.w[name] = w.reshape(w.shape[0], -1).to(BF16).to(dev)
                else:
                    self.w[name] = t.reshape(cout, cin).to(BF16).contiguous().to(dev)
            elif name.endswith(".bias") and name[:-len(".bias")] in conv_names:
                b = t
                if b.numel() % 4:
                    b = torch.cat([b, torch.zeros(4 - b.numel() % 4)])
                self.w[name] = b.to(BF16).to(dev)
            else:
                self.w[name] = t.to(dev)                         # GroupNorm affine, fp32
        self.zero_page = torch.zeros(256, dtype=torch.uint8, device=dev)
        self._buf: Dict[str, torch.Tensor] = {}
        # query rows of the mid-block attention whose fp32 scores are held at once: all 16384 of a 1024 x 1024 image
        # (1 GiB of scores + 0.5 GiB of probabilities) - the P.V product then has 512 output tiles, two per CU,
        # where a 4096-row chunk left half the chip idle; results do not depend on the chunk
        self.score_chunk = 16384

    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    def _get(self, name, shape, dtype=BF16):
        n = 1
        for s in shape:
            n *= s
        cur = self._buf.get(name)
        if cur is None or cur.numel() < n or cur.dtype != dtype:
            cur = torch.empty(n, device=self.device, dtype=dtype)
            self._buf[name] = cur
        return cur[:n].view(*shape)

    # ---- blocks ---------------------------------------------------------------------------------
    def _gn(self, x, npix, C, name, out, swish=True):
        ws = self._get("gn_ws", (K.groupnorm_workspace_bytes(C),), torch.uint8)
        return K.groupnorm_swish(x, npix, C, self.w[name + ".weight"], self.w[name + ".bias"], out, ws, swish=swish)

    def _resblock(self, x, H, W, cin, cout, name, out_name):
        """ResnetBlock.forward (autoencoder.py:82-95); x [H*W, cin] -> [H*W, cout]."""
        npix = H * W
        t = self._gn(x, npix, cin, name + ".norm1", self._get("t0", (npix, cin)))
        h = K.conv3x3(t, H, W, cin, self.w[name + ".conv1.weight"], self.w[name + ".conv1.bias"],
                      self._get("t1", (npix, cout)), self.zero_page)
        t = self._gn(h, npix, cout, name + ".norm2", self._get("t0", (npix, cout)))
        out = self._get(out_name, (npix, cout))
        if cin == cout:
            return K.conv3x3(t, H, W, cout, self.w[name + ".conv2.weight"], self.w[name + ".conv2.bias"], out,
                             self.zero_page, residual=x)
        h2 = K.conv3x3(t, H, W, cout, self.w[name + ".conv2.weight"], self.w[name + ".conv2.bias"],
                       self._get("t1", (npix, cout)), self.zero_page)
        # x' = nin_shortcut(x); out = x' + h   (1x1 convolution = dense GEMM on NHWC rows)
        return K.gemm(x, self.w[name + ".nin_shortcut.weight"], out, bias=self.w[name + ".nin_shortcut.bias"],
                      residual=h2, epilogue=K.EPI_BIAS_RESIDUAL)

    def _attnblock(self, x, H, W, C, name, out_name):
        """AttnBlock.forward (autoencoder.py:50-65): one head over H*W tokens, head dim C."""
        n = H * W
        if n % 4:
            raise NotImplementedError(f"mid-block attention needs h*w % 4 == 0 (got {H}x{W} latent pixels)")
        npad = (n + 63) // 64 * 64      # keys padded to the GEMM's K step: zero probabilities x zero values
        hn = self._gn(x, n, C, name + ".norm", self._get("t0", (n, C)), swish=False)
        q = K.gemm(hn, self.w[name + ".q.weight"], self._get("aq", (n, C)), bias=self.w[name + ".q.bias"])
        k = K.gemm(hn, self.w[name + ".k.weight"], self._get("ak", (n, C)), bias=self.w[name + ".k.bias"])
        vbuf = self._get("av", (npad, C))
        if npad != n:
            vbuf[n:].zero_()
        K.gemm(hn, self.w[name + ".v.weight"], vbuf, bias=self.w[name + ".v.bias"])
        vt = K.transpose(vbuf, self._get("avt", (C, npad)))
        o = self._get("ao", (n, C))
        # scores in row chunks so the fp32 score matrix stays bounded (n can be 16384)
        chunk = min(n, self.score_chunk)
        s = self._get("as", (chunk, npad), torch.float32)
        p = self._get("ap", (chunk, npad))
        if npad != n:
            p.zero_()
        for r0 in range(0, n, chunk):
            m = min(chunk, n - r0)
            K.gemm(q[r0:r0 + m], k, s[:m], epilogue=K.EPI_F32_SCALED, alpha=float(C) ** -0.5)
            K.softmax_rows(s[:m, :n], p[:m, :n])
            K.gemm(p[:m], vt, o[r0:r0 + m])
        return K.gemm(o, self.w[name + ".proj_out.weight"], self._get(out_name, (n, C)),
                      bias=self.w[name + ".proj_out.bias"], residual=x, epilogue=K.EPI_BIAS_RESIDUAL)

    # ---- API --------------------------------------------------------------------------------------
    @K.on_device
    @torch.no_grad()
    def decode(self, z: torch.Tensor) -> torch.Tensor:
        """z [1, z_channels, h, w] (fp32) -> image [1, 3, 8h, 8w] (bf16), AutoEncoder.decode (autoencoder.py:320-322)."""
        v = self.params
        assert z.dim() == 4 and z.shape[0] == 1 and z.shape[1] == v.z_channels, "decode expects [1, z_channels, h, w]"
        z = z.to(device=self.device, dtype=torch.float32).contiguous()
        H, W = int(z.shape[2]), int(z.shape[3])
        nres = len(v.ch_mult)
        C = v.ch * v.ch_mult[nres - 1]
        w_in = self.w["decoder.conv_in.weight"]
        cols = K.vae_latent_im2col(z, v.scale_factor, v.shift_factor, self._get("im2col", (H * W, w_in.shape[1])))
        cur, other = "xa", "xb"
        h = K.gemm(cols, w_in, self._get(cur, (H * W, C)), bias=self.w["decoder.conv_in.bias"])

        def step(fn, *args):
            nonlocal cur, other, h
            h = fn(h, *args, other)
            cur, other = other, cur

        step(self._resblock, H, W, C, C, "decoder.mid.block_1")
        step(self._attnblock, H, W, C, "decoder.mid.attn_1")
        step(self._resblock, H, W, C, C, "decoder.mid.block_2")
        for lvl in reversed(range(nres)):
            cout = v.ch * v.ch_mult[lvl]
            for b in range(v.num_res_blocks + 1):
                step(self._resblock, H, W, C, cout, f"decoder.up.{lvl}.block.{b}")
                C = cout
            if lvl != 0:
                name = f"decoder.up.{lvl}.upsample.conv"
                h = K.conv3x3(h, H, W, C, self.w[name + ".weight"], self.w[name + ".bias"],
                              self._get(other, (4 * H * W, C)), self.zero_page, upsample=True)
                cur, other = other, cur
                H, W = 2 * H, 2 * W
        npix = H * W
        t = self._gn(h, npix, C, "decoder.norm_out", self._get("t0", (npix, C)))
        w_out = self.w["decoder.conv_out.weight"]
        o = K.conv3x3(t, H, W, C, w_out, self.w["decoder.conv_out.bias"], self._get("t1", (npix, w_out.shape[0])),
                      self.zero_page)
        img = torch.empty((1, v.out_ch, H, W), device=self.device, dtype=BF16)
        K.nhwc_to_nchw(o, npix, w_out.shape[0], v.out_ch, img)
        return img

    @K.on_device
    @torch.no_grad()
    def encode_moments(self, x: torch.Tensor) -> torch.Tensor:
        """Encoder.forward (autoencoder.py:172-193) for one image: x [1, 3, H, W] -> NHWC moments
        ``[H/8 * W/8, 2 z_channels]`` bf16 (mean channels first, then logvar)."""
        v = self.params
        if not self.has_encoder:
            raise ValueError("this AutoEncoder was built without encoder.* weights")
        assert x.dim() == 4 and x.shape[0] == 1 and x.shape[1] == v.in_channels, "encode_moments expects [1, 3, H, W]"
        nres = len(v.ch_mult)
        H, W = int(x.shape[2]), int(x.shape[3])
        if H % (1 << (nres - 1)) or W % (1 << (nres - 1)):
            raise ValueError(f"image size {H}x{W} must be a multiple of {1 << (nres - 1)}")
        x = x.to(device=self.device, dtype=torch.float32).contiguous()
        w_in = self.w["encoder.conv_in.weight"]
        cols = K.vae_latent_im2col(x, 1.0, 0.0, self._get("im2col", (H * W, w_in.shape[1])))
        C = v.ch
        cur, other = "xa", "xb"
        h = K.gemm(cols, w_in, self._get(cur, (H * W, C)), bias=self.w["encoder.conv_in.bias"])

        def step(fn, *args):
            nonlocal cur, other, h
            h = fn(h, *args, other)
            cur, other = other, cur

        for lvl in range(nres):
            cout = v.ch * v.ch_mult[lvl]
            for b in range(v.num_res_blocks):
                step(self._resblock, H, W, C, cout, f"encoder.down.{lvl}.block.{b}")
                C = cout
            if lvl != nres - 1:
                name = f"encoder.down.{lvl}.downsample.conv"
                h = K.conv3x3(h, H, W, C, self.w[name + ".weight"], self.w[name + ".bias"],
                              self._get(other, (H * W // 4, C)), self.zero_page, downsample=True)
                cur, other = other, cur
                H, W = H // 2, W // 2
        step(self._resblock, H, W, C, C, "encoder.mid.block_1")
        step(self._attnblock, H, W, C, "encoder.mid.attn_1")
        step(self._resblock, H, W, C, C, "encoder.mid.block_2")
        npix = H * W
        t = self._gn(h, npix, C, "encoder.norm_out", self._get("t0", (npix, C)))
        w_out = self.w["encoder.conv_out.weight"]
        out = torch.empty((npix, w_out.shape[0]), device=self.device, dtype=BF16)
        return K.conv3x3(t, H, W, C, w_out, self.w["encoder.conv_out.bias"], out, self.zero_page)

    @K.on_device
    @torch.no_grad()
    def encode(self, x: torch.Tensor, noise: Optional[torch.Tensor] = None) -> torch.Tensor:
        """AutoEncoder.encode (autoencoder.py:315-318): x [B, 3, H, W] -> z [B, z_channels, H/8, W/8] bf16.
        ``noise`` is the Gaussian draw of DiagonalGaussian.forward (:281-287), ``[B, z_channels, H/8, W/8]``;
        when omitted it is drawn on the device, as ``torch.randn_like(mean)`` does in the reference."""
        v = self.params
        B, H, W = int(x.shape[0]), int(x.shape[2]), int(x.shape[3])
        f = 1 << (len(v.ch_mult) - 1)
        h, w = H // f, W // f
        if noise is None:
            noise = torch.randn((B, v.z_channels, h, w), device=self.device, dtype=BF16)
        noise = noise.to(device=self.device, dtype=BF16).contiguous()
        z = torch.empty((B, v.z_channels, h, w), device=self.device, dtype=BF16)
        for b in range(B):
            mom = self.encode_moments(x[b:b + 1])
            K.vae_sample_latent(mom, noise[b], v.z_channels, h, w, v.scale_factor, v.shift_factor, z[b])
        return z
