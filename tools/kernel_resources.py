"""Registers, scratch and occupancy of every kernel in a HIP source, from the compiler's own remarks:

    python tools/kernel_resources.py bagel_amd/csrc/gemm.hip [substring ...]
"""
import re
import subprocess
import sys


def kernel_resources(src, flags=()):
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", *flags,
           "--cuda-device-only", "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
    res, cur = {}, None
    for line in out.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = subprocess.run(["c++filt", m.group(1)], stdout=subprocess.PIPE, text=True).stdout.strip()
            res[cur] = {}
            continue
        m = re.search(r"remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|TotalSGPRs|LDS Size \[bytes/block\]): (\d+)", line)
        if m and cur:
            res[cur][m.group(1).split(" ")[0]] = int(m.group(2))
    return res


if __name__ == "__main__":
    for name, r in kernel_resources(sys.argv[1]).items():
        if all(s in name for s in sys.argv[2:]):
            print(f"{name[:90]:90s} VGPR {r.get('VGPRs', -1):4d} AGPR {r.get('AGPRs', 0):4d} scratch {r.get('ScratchSize', -1):4d} "
                  f"waves/SIMD {r.get('Occupancy', -1)}")
