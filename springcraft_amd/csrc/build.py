"""
Builds libspringcraft_hip.so in-tree for gfx950 with hipcc (cross-compiles without a GPU).

    python springcraft_amd/csrc/build.py [--force] [--verbose]

One object per .hip source (compiled in parallel, rebuilt only when the source or a header
changed), linked into springcraft_amd/libspringcraft_hip.so.  assembly.hip is compiled with
-ffp-contract=off (bit-exact contact predicate, see the file header).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from os.path import abspath, dirname, exists, getmtime, join

HERE = dirname(abspath(__file__))
PKG = dirname(HERE)
REPO = dirname(PKG)
OBJ = join(HERE, "obj")
LIB = join(PKG, "libspringcraft_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

COMMON = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function", "-Werror=return-type",
          "-I", join(REPO, "include")] + os.environ.get("SC_EXTRA_HIPCC_FLAGS", "").split()
PER_FILE = {
    "assembly.hip": ["-ffp-contract=off"],
    # k_bt2_apply's diamond loop is 80 fully unrolled steps per half-diamond (every register index must be a constant):
    # above LLVM's default budget for `#pragma unroll`, below which the accumulators would live in scratch memory
    "twostage.hip": ["-mllvm", "-pragma-unroll-threshold=1000000"],
}


def sources():
    return sorted(f for f in os.listdir(HERE) if f.endswith(".hip") and not f.startswith("_"))


def headers_mtime():
    hs = [join(HERE, f) for f in os.listdir(HERE) if f.endswith(".h")]
    hs.append(join(REPO, "include", "springcraft_hip.h"))
    return max(getmtime(h) for h in hs)


def compile_one(src, force, verbose):
    obj = join(OBJ, src.replace(".hip", ".o"))
    spath = join(HERE, src)
    if not force and exists(obj) and getmtime(obj) > max(getmtime(spath), headers_mtime()):
        return obj, False
    cmd = [HIPCC, "-c", spath, "-o", obj] + COMMON + PER_FILE.get(src, [])
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    if verbose and r.stderr.strip():
        print(r.stderr)
    return obj, True


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    srcs = sources()
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        results = list(ex.map(lambda s: compile_one(s, force, verbose), srcs))
    objs = [o for o, _ in results]
    if force or not exists(LIB) or any(changed for _, changed in results):
        cmd = [HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    lib = build(force="--force" in sys.argv, verbose="--verbose" in sys.argv or "-v" in sys.argv)
    print(lib)
