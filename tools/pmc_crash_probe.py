"""Probe for the host-side SIGSEGV seen when bench.py runs under `rocprofv3 --pmc`: the fault is raised inside the
profiler / runtime at the first weight-streaming GEMM launch, but only in the full-size (28-layer) process.

    rocprofv3 --pmc SQ_WAVES -- python3 tools/pmc_crash_probe.py <GiB to allocate first> <chunks> <maps file>

Allocates and fills the given amount of device memory the way the weight generator does (chunks x normal_), dumps
/proc/self/maps, then issues the launch that crashed."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bagel_amd import _lib as K  # noqa: E402

gib, chunks, maps_path = float(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
dev = "cuda:0"
hold = []
if gib > 0:
    n = int(gib * (1 << 30) / 2 / chunks)
    for _ in range(chunks):
        t = torch.empty(n, device=dev, dtype=torch.bfloat16)
        t.normal_(0, 0.02)
        hold.append(t)
torch.cuda.synchronize()
H, F, rows = 3584, 256, 49
x = torch.randn((rows, F), device=dev).to(torch.bfloat16)
w = torch.randn((H, F), device=dev).to(torch.bfloat16)
b = torch.zeros(H, device=dev, dtype=torch.bfloat16)
out = torch.empty((rows, H), device=dev, dtype=torch.bfloat16)
ws = torch.empty(16 * 64 * H, device=dev, dtype=torch.float32)
torch.cuda.synchronize()
with open(maps_path, "w") as f:
    f.write(f"pointers: x={x.data_ptr():#x} w={w.data_ptr():#x} out={out.data_ptr():#x} ws={ws.data_ptr():#x}\n")
    f.write(open("/proc/self/maps").read())
print(f"allocated {gib} GiB in {chunks} chunks; launching", flush=True)
K.skinny_gemm(x, w, out, ws, bias=b, epilogue=K.SK_BIAS_SILU)
torch.cuda.synchronize()
print("skinny launch ok", flush=True)
