"""Build libmtts.so (hipcc, gfx950) in-tree: moss-ttsd_amd/lib/libmtts.so.

Every translation unit is compiled to its own object (in parallel, and only when it or a header changed), then
linked: a change in one kernel file costs one compile, not seven."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = ["gemm.hip", "layer.hip", "attn.hip", "sampler.hip", "f32path.hip", "engine.hip", "codec.hip", "codec_fused.hip"]
OUT = os.path.join(HERE, "lib", "libmtts.so")
OBJ = os.path.join(HERE, "build")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wno-unused-result", "-Wno-unused-value"]
FLAGS += os.environ.get("MTTS_BUILD_FLAGS", "").split()          # tuning experiments (-DNAME=value), with --force


def _newer(target, deps):
    return os.path.exists(target) and all(os.path.getmtime(target) >= os.path.getmtime(d) for d in deps)


def build(force=False, verbose=True):
    srcs = [os.path.join(HERE, "csrc", s) for s in SRC if os.path.exists(os.path.join(HERE, "csrc", s))]
    hdrs = [os.path.join(HERE, "csrc", "common.h"), os.path.join(HERE, "..", "include", "mtts.h")]
    hdrs += [os.path.join(HERE, "csrc", h) for h in os.listdir(os.path.join(HERE, "csrc")) if h.endswith(".h") and h != "common.h"]
    if not force and _newer(OUT, srcs + hdrs):
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    os.makedirs(OBJ, exist_ok=True)
    objs, jobs = [], []
    for s in srcs:
        o = os.path.join(OBJ, os.path.basename(s) + ".o")
        objs.append(o)
        if force or not _newer(o, [s] + hdrs):
            jobs.append(["hipcc"] + FLAGS + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc", "-o", OUT] + objs)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
