"""Build a variant of libbagel_hip.so with extra compiler flags into tools/_ab/ (kept out of git; travels to the GPU
box) for one-process A/B runs:   python tools/build_variant.py <name> [-DFLAG ...]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bagel_amd import _lib  # noqa: E402


def main():
    name, flags = sys.argv[1], sys.argv[2:]
    out_dir = os.path.join(ROOT, "tools", "_ab")
    os.makedirs(os.path.join(out_dir, "obj_" + name), exist_ok=True)
    objs, procs = [], []
    for s in _lib.SOURCES:
        o = os.path.join(out_dir, "obj_" + name, s + ".o")
        objs.append(o)
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", *flags, "-c",
               os.path.join(_lib.CSRC, s), "-o", o]
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode:
            sys.exit(" ".join(cmd) + "\n" + out.decode())
    lib = os.path.join(out_dir, f"lib{name}.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    subprocess.call(["rm", "-rf", os.path.join(out_dir, "obj_" + name)])
    print(lib)


if __name__ == "__main__":
    main()
