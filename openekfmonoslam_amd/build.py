"""In-tree build of the HIP engine (libekf_engine.so) for gfx950 with hipcc.  No CPU fallback is built."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libekf_engine.so")
SOURCES = ["engine.cpp", "kernels_predict.hip", "kernels_match.hip", "kernels_ransac.hip", "kernels_update.hip",
           "kernels_pupdate.hip", "kernels_pexact.hip", "kernels_map.hip", "kernels_ncc.hip", "kernels_gemm.hip", "kernels_detect.hip"]
# the decision-making stages (projection, gates, dead-bands) are compiled without FMA contraction so their fp64
# arithmetic rounds like the reference's scalar C++; the GEMM-shaped kernels keep contraction
NO_CONTRACT = {"kernels_predict.hip", "kernels_match.hip", "kernels_ransac.hip", "kernels_ncc.hip", "kernels_detect.hip"}
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-x", "hip"]
FLAGS += os.environ.get("EKF_EXTRA_FLAGS", "").split()  # tuning experiments, e.g. -DPU_BK_VALUE=32


def _newer(src, dst):
    return (not os.path.exists(dst)) or os.path.getmtime(src) > os.path.getmtime(dst)


def build_engine(force=False, verbose=False):
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    deps += [os.path.join(HERE, "..", "include", f) for f in ("ekf_engine.h", "ekf_types.h")]
    objs, jobs = [], []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(CSRC, os.path.splitext(src)[0] + ".o")
        objs.append(op)
        if force or _newer(sp, op) or any(_newer(d, op) for d in deps):
            cmd = [HIPCC] + FLAGS + (["-ffp-contract=off"] if src in NO_CONTRACT else []) + ["-c", sp, "-o", op]
            jobs.append(cmd)
    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if jobs or not os.path.exists(LIB):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl", "-Wl,-rpath,/opt/rocm/lib"])
    return LIB


if __name__ == "__main__":
    print(build_engine(force="--force" in sys.argv, verbose=True))
