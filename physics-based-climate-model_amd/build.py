"""Builds libclimate_hip.so (gfx950) from csrc/*.hip and *.cpp with hipcc, in-tree.

    python physics-based-climate-model_amd/build.py [--force]

hipcc cross-compiles without a GPU.  Objects are compiled in parallel and cached on (source, header) mtimes.
"""
import concurrent.futures as cf
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libclimate_hip.so")
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", f"--offload-arch={ARCH}", "-fPIC", "-munsafe-fp-atomics", "-std=c++17", "-I", os.path.join(ROOT, "include")]


def _sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")))


def _headers():
    return glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h"))


def _compile(src, force):
    obj = os.path.join(OBJ, os.path.basename(src) + ".o")
    deps = [src] + _headers()
    if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(d) for d in deps):
        return obj, None
    cmd = [HIPCC] + FLAGS + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    return obj, r.stderr


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    srcs = _sources()
    if not srcs:
        raise RuntimeError("no sources under " + CSRC)
    with cf.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        results = list(ex.map(lambda s: _compile(s, force), srcs))
    objs = [o for o, _ in results]
    rebuilt = any(log is not None for _, log in results)
    if rebuilt or not os.path.exists(LIB) or force:
        cmd = [HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"[build] {LIB} ({os.path.getsize(LIB) / 1e6:.1f} MB, {len(objs)} objects, rebuilt={rebuilt})")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
