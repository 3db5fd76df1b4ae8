"""Build libmtts.so (hipcc, gfx950) in-tree: moss-ttsd_amd/lib/libmtts.so."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = ["gemm.hip", "layer.hip", "attn.hip", "sampler.hip", "f32path.hip", "engine.hip", "codec.hip"]
OUT = os.path.join(HERE, "lib", "libmtts.so")


def build(force=False, verbose=True):
    srcs = [os.path.join(HERE, "csrc", s) for s in SRC]
    deps = srcs + [os.path.join(HERE, "csrc", "common.h"), os.path.join(HERE, "..", "include", "mtts.h")]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps):
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fgpu-rdc" if False else "-fno-gpu-rdc",
           "-Wno-unused-result", "-Wno-unused-value", "-o", OUT] + srcs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
