"""Builds kg_instance_segmentation_amd/libkgnet_hip.so for gfx950 with hipcc (in-tree, no JIT cache).

    python -m kg_instance_segmentation_amd.build [--force]

hipcc cross-compiles without a GPU.  postproc.hip is compiled with -ffp-contract=off (its float64
arithmetic must round exactly like the reference's NumPy operations).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libkgnet_hip.so")
LIB_F16 = os.path.join(HERE, "libkgnet_hip_f16.so")      # the same sources with IEEE-half rows (-DKG_F16, csrc/kg_common.h)
# sources with rows / packed-weight operands: built once per 16-bit format
ROWS_SOURCES = ("api.hip", "conv_igemm.hip", "conv_gather.hip", "conv_small.hip", "conv_halo.hip", "conv3_c64.hip", "conv1x1.hip",
                "conv_wgrad.hip", "wgrad_halo.hip", "norm_pool.hip", "loss.hip", "seg.hip", "conv_tiny.hip", "conv3_ws.hip", "conv7_narrow.hip")
SOURCES = {
    "api.hip": [],
    "conv_igemm.hip": [],
    "conv_gather.hip": [],
    "conv_small.hip": [],
    "conv_halo.hip": [],
    "conv3_c64.hip": [],
    "conv1x1.hip": [],
    "conv_wgrad.hip": [],
    "wgrad_halo.hip": [],
    "norm_pool.hip": [],
    "loss.hip": [],
    "postproc.hip": ["-ffp-contract=off"],
    "preproc.hip": ["-ffp-contract=off"],
    "evalmetrics.hip": [],
    "optim.hip": ["-ffp-contract=off"],
    "seg.hip": [],
    "paste.hip": ["-ffp-contract=off"],
    "gradscale.hip": [],
    "conv_tiny.hip": [],
    "conv3_ws.hip": [],
    "conv7_narrow.hip": [],
}
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", "-Wall", "-Wno-unused-function"]


def _stale(out, deps):
    return not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps)


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in sorted(os.listdir(CSRC)) if h.endswith(".h")]
    jobs = []
    variants = [("", [], list(SOURCES)), ("_f16", ["-DKG_F16"], [s for s in SOURCES if s in ROWS_SOURCES])]
    for suffix, defs, srcs in variants:
        for src in srcs:
            extra = SOURCES[src]
            s = os.path.join(CSRC, src)
            if not os.path.exists(s):
                continue
            o = os.path.join(OBJ, src.replace(".hip", suffix + ".o"))
            if force or _stale(o, [s, __file__] + hdrs):
                jobs.append((s, o, [HIPCC] + FLAGS + defs + extra + ["-c", s, "-o", o]))

    def run(job):
        s, o, cmd = job
        if verbose:
            print(" ".join(cmd))
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {s}:\n{r.stderr}")
        return r.stderr

    with ThreadPoolExecutor(max_workers=max(4, (os.cpu_count() or 4))) as ex:
        for warn in ex.map(run, jobs):
            if warn and verbose:
                print(warn)
    for (suffix, _, srcs), lib in zip(variants, (LIB, LIB_F16)):
        objs = [os.path.join(OBJ, s.replace(".hip", suffix + ".o")) for s in srcs if os.path.exists(os.path.join(CSRC, s))]
        if force or jobs or _stale(lib, objs):
            # -Bsymbolic: both libraries export the same C ABI; each must bind its own internal references
            cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-Bsymbolic", "-o", lib] + objs
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"link failed:\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
