"""Time AutoEncoder.decode on a 128 x 128 latent (a 1024 x 1024 image), FLUX dims, seeded weights."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bagel_amd.config import VAEDims  # noqa: E402
from bagel_amd.modeling import AutoEncoder  # noqa: E402
from bagel_amd.weights import make_vae_decoder_state_dict  # noqa: E402

if __name__ == "__main__":
    side = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    vd = VAEDims()
    ae = AutoEncoder(vd, make_vae_decoder_state_dict(vd, seed=0), "cuda:0")
    z = torch.randn(1, vd.z_channels, side, side, generator=torch.Generator().manual_seed(7))
    from bagel_amd import _lib as K
    ref = None
    # VAE_AB=1: also time the decode with every convolution on the 128 x 128 tile, in the same process
    for pipelined in ((1, 0, 1, 0) if os.environ.get("VAE_AB") else (1,)):
        K.lib().bagel_set_conv_pipelined(pipelined)
        for _ in range(2):
            img = ae.decode(z)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            img = ae.decode(z)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / iters
        same = "" if ref is None else f"; identical to the first: {bool(torch.equal(img, ref))}"
        ref = img.clone() if ref is None else ref
        print(f"decode {side * 8} x {side * 8} (pipelined convolutions {pipelined}): {dt * 1e3:.2f} ms per image; "
              f"checksum {float(img.float().sum()):.3f}{same}")
