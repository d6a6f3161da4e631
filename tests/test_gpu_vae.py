"""FLUX VAE decoder on HIP kernels vs the reference's AutoEncoder.decode (golden) and the CPU checker.

Bar: PSNR >= 40 dB on the decoded image (images span [-1, 1], peak-to-peak 2) against the reference's
bf16-autocast output; the reference's autocast output itself sits ~49.5 dB from its own fp32 output.
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from bagel_amd.config import VAEDims  # noqa: E402
from bagel_amd.weights import make_vae_decoder_state_dict  # noqa: E402
from tests.helpers import load_golden  # noqa: E402
from tests.test_oracle_vae import psnr  # noqa: E402

BF16 = torch.bfloat16
DEV = "cuda:0"


def _K():
    from bagel_amd import _lib
    return _lib


def _rand(shape, std=1.0, seed=0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * std).to(dtype)


@pytest.mark.parametrize("H,W,Cin,Cout,up,res", [(16, 16, 64, 128, False, False), (9, 13, 128, 64, False, True),
                                                 (8, 12, 64, 64, True, False), (32, 32, 512, 256, False, False),
                                                 (20, 24, 128, 4, False, False)])
def test_conv3x3(H, W, Cin, Cout, up, res):
    k = _K()
    x = _rand((1, Cin, H, W), 1.0, 1).to(BF16)
    w = _rand((Cout, Cin, 3, 3), (9 * Cin) ** -0.5, 2).to(BF16)
    b = _rand((Cout,), 0.1, 3).to(BF16)
    Ho, Wo = (2 * H, 2 * W) if up else (H, W)
    r = _rand((1, Cout, Ho, Wo), 1.0, 4).to(BF16) if res else None
    xin = F.interpolate(x.float(), scale_factor=2.0, mode="nearest") if up else x.float()
    ref = (F.conv2d(xin, w.float(), b.float(), padding=1)).to(BF16)
    if res:
        ref = (ref.float() + r.float()).to(BF16)
    x_nhwc = x[0].permute(1, 2, 0).contiguous().to(DEV)
    w_k = w.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous().to(DEV)
    r_nhwc = r[0].permute(1, 2, 0).reshape(Ho * Wo, Cout).contiguous().to(DEV) if res else None
    out = torch.full((Ho * Wo, Cout), float("nan"), dtype=BF16, device=DEV)
    zero = torch.zeros(256, dtype=torch.uint8, device=DEV)
    k.conv3x3(x_nhwc, H, W, Cin, w_k, b.to(DEV), out, zero, upsample=up, residual=r_nhwc)
    got = out.cpu().view(Ho, Wo, Cout).permute(2, 0, 1)[None]
    err = float((got.float() - ref.float()).abs().max())
    assert err <= 2.0 ** -7 * float(ref.float().abs().max()) + 1e-3, err


@pytest.mark.parametrize("npix,C,swish", [(64, 128, True), (1000, 256, True), (4096, 512, False)])
def test_groupnorm_swish(npix, C, swish):
    k = _K()
    x = (_rand((npix, C), 2.0, 5) + 0.5).to(BF16)
    gm, bt = _rand((C,), 0.1, 6) + 1, _rand((C,), 0.1, 7)
    out = torch.empty((npix, C), dtype=BF16, device=DEV)
    ws = torch.empty(k.groupnorm_workspace_bytes(C), dtype=torch.uint8, device=DEV)
    k.groupnorm_swish(x.to(DEV), npix, C, gm.to(DEV), bt.to(DEV), out, ws, swish=swish)
    ref = F.group_norm(x.float().t()[None], 32, gm, bt, eps=1e-6)[0].t()
    if swish:
        ref = ref * torch.sigmoid(ref)
    err = float((out.cpu().float() - ref).abs().max())
    assert err <= 2.0 ** -7 * float(ref.abs().max()) + 1e-3, err


def test_softmax_and_transpose():
    k = _K()
    s = _rand((37, 1024), 3.0, 8)
    p = torch.empty((37, 1024), dtype=BF16, device=DEV)
    k.softmax_rows(s.to(DEV), p)
    assert float((p.cpu().float() - torch.softmax(s, -1)).abs().max()) < 2e-3
    x = _rand((130, 70), 1.0, 9).to(BF16)
    xt = torch.empty((70, 130), dtype=BF16, device=DEV)
    k.transpose(x.to(DEV), xt)
    assert torch.equal(xt.cpu(), x.t().contiguous())


@pytest.mark.parametrize("name", ["vae_128", "vae_256", "vae_512x384", "vae_1024"])
def test_decode_matches_reference(name):
    """vae_1024 is the 128 x 128 latent -> 1024 x 1024 image the benchmark decodes: its mid-block attention runs
    over 16384 tokens in four score chunks (the golden file stores that case's images in fp16)."""
    from bagel_amd.modeling import AutoEncoder
    from tests.helpers import PARITY_LOG
    g = load_golden(name)
    g["image_autocast"], g["image_fp32"] = g["image_autocast"].float(), g["image_fp32"].float()
    vd = VAEDims()
    ae = AutoEncoder(vd, make_vae_decoder_state_dict(vd, seed=0), DEV)
    img = ae.decode(g["z"]).float().cpu()
    assert img.shape == g["image_autocast"].shape and torch.isfinite(img).all()
    p16, p32 = psnr(img, g["image_autocast"]), psnr(img, g["image_fp32"])
    print(f"{name}: HIP decode vs reference autocast {p16:.1f} dB, vs reference fp32 {p32:.1f} dB")
    PARITY_LOG.append({"golden": name, "tensor": "decoded image", "psnr_vs_reference_autocast_db": p16,
                       "psnr_vs_reference_fp32_db": p32,
                       "max_abs_vs_reference_autocast": float((img - g["image_autocast"]).abs().max())})
    assert p16 >= 40.0 and p32 >= 40.0
    meta = g["__meta__"]
    if "psnr_autocast_vs_fp32" in meta:
        # as close to the fp32 decode as the reference's own bf16 decode is (within 1 dB), and the same mean level
        assert p32 >= float(meta["psnr_autocast_vs_fp32"]) - 1.0, (p32, meta["psnr_autocast_vs_fp32"])
        assert abs(float(img.double().sum()) - float(meta["sum_fp32"])) <= 2e-3 * img.numel()
    u8 = ((img * 0.5 + 0.5).clamp(0, 1)[0].permute(1, 2, 0) * 255).to(torch.uint8)
    r8 = ((g["image_autocast"] * 0.5 + 0.5).clamp(0, 1)[0].permute(1, 2, 0) * 255).to(torch.uint8)
    assert float((u8.int() - r8.int()).abs().float().mean()) < 1.0


def test_decode_random_latent_sizes_match_cpu_checker():
    """Latent grids of random, mostly non-square sizes (rows x columns not a multiple of the GEMM's K step included)
    against the CPU checker's fp32 decode of the same latent.  BAGEL_FUZZ_VAE=n runs n sizes (default 3)."""
    import os
    import random
    from bagel_amd.modeling import AutoEncoder
    from oracle import bagel_oracle as O
    vd = VAEDims()
    sd = make_vae_decoder_state_dict(vd, seed=0)
    ae = AutoEncoder(vd, sd, DEV)
    rnd = random.Random(3)
    n_cases = int(os.environ.get("BAGEL_FUZZ_VAE", 3))
    for case in range(n_cases):
        if case and case % 10 == 0:
            print(f"  random VAE sizes: {case} / {n_cases}", flush=True)
        h, w = 2 * rnd.randint(1, 10), 2 * rnd.randint(1, 10)     # BAGEL latents are 2 x 2-patchified: even sides
        z = _rand((1, vd.z_channels, h, w), 1.0, 300 + case)
        img = ae.decode(z).float().cpu()
        with torch.no_grad():
            ref = O.vae_decode(sd, vd, z, bf16=False)
        assert img.shape == ref.shape and torch.isfinite(img).all(), (case, h, w)
        p = psnr(img, ref)
        assert p >= 40.0, (case, h, w, p)
    print(f"{n_cases} random latent sizes decode within 40 dB of the CPU checker's fp32 decode")


def test_mid_block_attention_chunking_is_exact():
    """The mid-block attention computes its fp32 scores in query-row chunks; the chunk size must not change a bit.
    64 x 64 latent (4096 tokens): one chunk by default, four chunks of 1024 rows when forced."""
    from bagel_amd.modeling import AutoEncoder
    vd = VAEDims()
    ae = AutoEncoder(vd, make_vae_decoder_state_dict(vd, seed=0), DEV)
    z = _rand((1, vd.z_channels, 64, 64), 1.0, 21)
    one = ae.decode(z).clone()
    ae.score_chunk = 1024
    four = ae.decode(z).clone()
    ae.score_chunk = 1000          # ragged last chunk
    ragged = ae.decode(z).clone()
    assert torch.equal(one, four) and torch.equal(one, ragged)


def test_decode_1024_is_deterministic():
    from bagel_amd.modeling import AutoEncoder
    vd = VAEDims()
    ae = AutoEncoder(vd, make_vae_decoder_state_dict(vd, seed=0), DEV)
    z = _rand((1, vd.z_channels, 128, 128), 1.0, 22)
    a = ae.decode(z).clone()
    b = ae.decode(z)
    assert torch.isfinite(a.float()).all() and torch.equal(a, b)
