"""bagel_cfg_euler_update_static (built-in device scratch, no allocation) against bagel_cfg_euler_update
(caller workspace), through the C-ABI: bit-identical outputs on two calls in a row, the second one smaller."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _inputs(rows, seed):
    g = torch.Generator(device="cuda:0").manual_seed(seed)
    mk = lambda s: (torch.randn((rows, 64), device="cuda:0", generator=g) * s).to(torch.bfloat16)  # noqa: E731
    return mk(1.0), mk(1.1), mk(0.9), torch.randn((rows, 64), device="cuda:0", generator=g)


def _run(static, rows, seed, renorm, rmin, s_text, s_img):
    from bagel_amd import _lib as K
    vc, vx, vi, x = _inputs(rows, seed)
    xb, vo = torch.empty_like(vc), torch.empty_like(vc)
    if static:
        K.cfg_euler_update_static(vc, vx, vi, s_text, s_img, renorm, rmin, 0.037, x, xb, vo)
    else:
        ws = torch.empty(K.cfg_workspace_bytes(rows, 64), dtype=torch.uint8, device="cuda:0")
        K.cfg_euler_update(vc, vx, vi, s_text, s_img, renorm, rmin, 0.037, x, xb, vo, ws)
    torch.cuda.synchronize()
    return x, xb, vo


@pytest.mark.parametrize("renorm,rmin,s_text,s_img", [("global", 0.0, 4.0, 1.5), ("channel", 0.3, 4.0, 1.0),
                                                      ("text_channel", 0.0, 3.0, 2.0), ("global", 0.0, 1.0, 1.0)])
def test_static_scratch_matches_workspace_entry(renorm, rmin, s_text, s_img):
    # two calls in a row on the shared scratch: a large one, then a smaller one that must not see its leftovers
    for rows, seed in ((3 * 4096, 1), (777, 2)):
        want = _run(False, rows, seed, renorm, rmin, s_text, s_img)
        got = _run(True, rows, seed, renorm, rmin, s_text, s_img)
        for name, a, b in zip(("x_t", "x_bf16", "v_t"), got, want):
            assert torch.equal(a, b), (renorm, rows, name)


def test_static_scratch_limits_and_no_allocation():
    from bagel_amd import _lib as K
    lib = K.lib()
    cap = lib.bagel_cfg_static_max_rows()
    assert cap == 65536
    vc, vx, vi, x = _inputs(8, 3)
    rc = lib.bagel_cfg_euler_update_static(K._p(vc), K._p(vx), K._p(vi), K.F(4.0), K.F(1.5), K.I(0), K.F(0.0), K.F(0.1),
                                           K._p(x), K._p(vc), ctypes.c_void_p(0), K.I(cap + 1), K.I(64), K._stream())
    assert rc != 0 and b"exceeds the built-in scratch" in lib.bagel_last_error()
    # the largest admitted size runs, and the call neither allocates nor frees device memory
    vc, vx, vi, x = _inputs(cap, 4)
    xb, vo = torch.empty_like(vc), torch.empty_like(vc)
    K.cfg_euler_update_static(vc, vx, vi, 4.0, 1.5, "global", 0.0, 0.02, x, xb, vo)   # first call: address lookup
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    stats0 = torch.cuda.memory_stats()["allocation.all.allocated"]
    K.cfg_euler_update_static(vc, vx, vi, 4.0, 1.5, "global", 0.0, 0.02, x, xb, vo)
    torch.cuda.synchronize()
    assert torch.cuda.memory_stats()["allocation.all.allocated"] == stats0
    assert torch.cuda.mem_get_info()[0] == free0
    assert torch.isfinite(x).all()
