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
# CM_BUILD_TAG=<tag> builds a second library beside the shipped one (libclimate_hip<tag>.so, objects under build<tag>/) with
# CM_HIPCC_EXTRA's flags appended -- for same-box A/B runs of a compiler option (CM_LIB_TAG selects it at import).
TAG = os.environ.get("CM_BUILD_TAG", "")
OBJ = os.path.join(HERE, "build" + TAG)
LIB = os.path.join(HERE, f"libclimate_hip{TAG}.so")
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", f"--offload-arch={ARCH}", "-fPIC", "-munsafe-fp-atomics", "-std=c++17", "-I", os.path.join(ROOT, "include")]
if TAG:
    FLAGS = FLAGS + os.environ.get("CM_HIPCC_EXTRA", "").split()
# Packed-FP32 instructions whose SECOND source has its halves swapped (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 with
# op_sel:[0,1] op_sel_hi:[1,0]) return wrong values in lanes 48-63 of a wave on MI355X while a wave of an MFMA kernel shares
# the SIMD (measured: tools/isa_lint.py, tools/coresidency_probe.py --pk, profiles/r03/coresidency/).  hip-clang's SLP
# vectorizer emitted that form in these translation units (the 7x7 tap loop of cm_block_tail_bwd, cm_gn_silu_bwd, the
# attention backward): they are compiled without it, and build() refuses a library in which the form appears anywhere.
NO_SLP = {"block_tail.hip", "norm_act.hip", "attention_mfma.hip"}
KEEP_SLP = bool(TAG and os.environ.get("CM_KEEP_SLP"))      # (A/B builds of the affected form only)
if KEEP_SLP:
    NO_SLP = set()


def _sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")))


def _headers():
    return glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h"))


def _compile(src, force):
    obj = os.path.join(OBJ, os.path.basename(src) + ".o")
    deps = [src] + _headers()
    if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(d) for d in deps):
        return obj, None
    extra = ["-fno-slp-vectorize"] if os.path.basename(src) in NO_SLP else []
    cmd = [HIPCC] + FLAGS + extra + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", src, "-o", obj]
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
    if (rebuilt or force) and not KEEP_SLP:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import isa_lint
        bad, _, total = isa_lint.lint(LIB)
        if bad:
            raise RuntimeError(f"{LIB}: {len(bad)} packed-fp32 instructions with the halves of src1 swapped (of {total}), "
                               f"e.g. {bad[0]}: see tools/isa_lint.py")
    if verbose:
        print(f"[build] {LIB} ({os.path.getsize(LIB) / 1e6:.1f} MB, {len(objs)} objects, rebuilt={rebuilt})")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
