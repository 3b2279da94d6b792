"""Build libuncr_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

`python -m uncrtaints_amd.build` or `build_lib()`.  hipcc cross-compiles without a GPU; the .so is
git-ignored but travels to the GPU box with the working tree."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libuncr_hip.so")
DEV_LIB = os.path.join(LIBDIR, "libuncr_dev.so")     # development probes (include/uncr_dev.h): never loaded by the product path
# (source stem, extra flags, object stem); the split GEMM is compiled once per prologue kind (compile-time PRO)
SOURCES = [(s, [], s) for s in ["norm", "ew", "pw_gemm", "pw_wgrad_split", "pw_wgrad_a16", "dwconv", "dwconv_row", "se", "ltae", "ltae_fused", "aggregate", "mgnll", "metrics", "conv3", "attn_rows", "optim", "inconv", "anysize"]] + \
          [("pw_gemm_split", [f"-DPWS_PRO={p}"], f"pw_gemm_split_p{p}") for p in range(5)]
# -fno-slp-vectorize: the SLP vectoriser packs neighbouring scalar FMAs into v_pk_fma_f32 and pays for it with
# register-pair moves (depthwise row kernel: 370 vs 282 VALU instructions per row; whole step +1 %)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize"]


def source_sha() -> str:
    """sha256 (first 16 hex digits) over the kernel sources: measurements that depend on the kernels' code (the PMC traffic
    file under profiles/) record it, and bench.py only attaches them while it still matches."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".h")) and f != "dev_probes.hip":
            h.update(f.encode())
            h.update(open(os.path.join(CSRC, f), "rb").read())
    return h.hexdigest()[:16]


def _newer(src, dst):
    return (not os.path.exists(dst)) or os.path.getmtime(src) > os.path.getmtime(dst)


def build_lib(force: bool = False, verbose: bool = False) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")]
    jobs = []
    for s, extra, o in SOURCES:
        src, obj = os.path.join(CSRC, s + ".hip"), os.path.join(objdir, o + ".o")
        if force or _newer(src, obj) or any(_newer(h, obj) for h in hdrs):
            jobs.append([hipcc, *FLAGS, *extra, "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(13, len(jobs))) as ex:
            list(ex.map(run, jobs))
    objs = [os.path.join(objdir, o + ".o") for _, _, o in SOURCES]
    if force or jobs or not os.path.exists(LIB):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    dsrc = os.path.join(CSRC, "dev_probes.hip")
    if force or _newer(dsrc, DEV_LIB) or any(_newer(h, DEV_LIB) for h in hdrs):
        run([hipcc, *FLAGS, "-shared", "-o", DEV_LIB, dsrc])
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
