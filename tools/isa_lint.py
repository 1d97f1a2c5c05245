"""ISA lint of the built library: no packed-FP32 instruction whose SECOND source has its halves swapped.

On MI355X (gfx950, ROCm 7.2) v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 with op_sel selecting the HIGH half of src1 for the
low result and op_sel_hi selecting the LOW half for the high result (`op_sel:[0,1] op_sel_hi:[1,0]`, for fma
`op_sel:[0,1,0] op_sel_hi:[1,0,1]`) returned wrong values in lanes 48-63 of a wave whenever a wave of one of the fp16x3 /
bf16x6 MFMA kernels shared its SIMD: 40 of 40 graph replays, against 0 of 40 for the same instruction alone, for the
src0-swapped, both-swapped, broadcast and unswizzled forms and for plain v_mul_f32 (tools/coresidency_probe.py --pk,
tools/_forms.py; profiles/r03/coresidency/).  hip-clang's SLP vectorizer produces the form from pairs of scalar
multiplies whose operands sit crosswise in two register pairs (the 7x7 tap loop of cm_block_tail_bwd was the case that
exposed it: its dmap came out wrong for 10-25 % of the samples whenever the ConvLSTM weight gradient ran beside it).
The library is built with -fno-slp-vectorize where the form appeared; this lint keeps it out.

    python tools/isa_lint.py [path/to/libclimate_hip.so]      exit code 1 and a listing if an offending instruction exists
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"

PK = re.compile(r"\b(v_pk_(?:mul|add|fma)_f32)\b(.*)")
SEL = re.compile(r"op_sel:\[([0-9,]+)\]")
SEL_HI = re.compile(r"op_sel_hi:\[([0-9,]+)\]")


def src1_swapped(operands: str) -> bool:
    """True when the instruction text selects (high, low) halves of src1 for the (low, high) results."""
    m, h = SEL.search(operands), SEL_HI.search(operands)
    sel = [int(v) for v in m.group(1).split(",")] if m else [0, 0, 0]
    hi = [int(v) for v in h.group(1).split(",")] if h else [1, 1, 1]
    return len(sel) > 1 and len(hi) > 1 and sel[1] == 1 and hi[1] == 0


def code_objects(lib: str):
    """The gfx950 code objects inside the library's .hip_fatbin section (one offload bundle per translation unit)."""
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
        blob = open(fat, "rb").read()
    pos = blob.find(MAGIC)
    while pos >= 0:
        n = struct.unpack_from("<Q", blob, pos + len(MAGIC))[0]
        q = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, q)
            triple = blob[q + 24:q + 24 + tlen].decode()
            q += 24 + tlen
            if "gfx950" in triple and size:
                yield blob[pos + off:pos + off + size]
        pos = blob.find(MAGIC, pos + len(MAGIC))


def lint(lib: str):
    bad, kernels, total = [], 0, 0
    for k, co in enumerate(code_objects(lib)):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            asm = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--mcpu=gfx950", f.name], capture_output=True,
                                 text=True, check=True).stdout
        cur = "?"
        for line in asm.splitlines():
            if line.endswith(">:"):
                cur = line.split("<")[-1][:-2]
                kernels += 1
                continue
            m = PK.search(line)
            if m:
                total += 1
                if src1_swapped(m.group(2)):
                    bad.append((cur, line.split("//")[0].strip()))
    return bad, kernels, total


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "physics-based-climate-model_amd", "libclimate_hip.so")
    bad, kernels, total = lint(lib)
    print(f"{lib}: {kernels} functions, {total} packed-fp32 mul/add/fma instructions, {len(bad)} with the halves of src1 swapped")
    for fn, ins in bad[:40]:
        print(f"    {fn[:90]}: {ins}")
    sys.exit(1 if bad else 0)
