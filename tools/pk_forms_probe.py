"""Which packed-FP32 instruction encodings return wrong values beside an MFMA kernel?  Each form runs 3000 times per thread on
fixed operands inside tools/canary's arithmetic canary (256-thread workgroups) and is compared with the same arithmetic done
by scalar instructions; the canary is recorded into one hipGraph beside a conv3x3 fp16x3 launch on a forked stream and
replayed 40 times.  Result of round 3 (profiles/r03/coresidency/16_packed_fp32_forms.txt): v_pk_mul_f32 / v_pk_add_f32 /
v_pk_fma_f32 with the halves of src1 swapped fail in lanes 48-63 in 40 of 40 replays; every other form 0 of 40.

    python tools/pk_forms_probe.py gpurun_out/pk_forms.txt
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from climate_amd import ops
import coresidency_probe as cp
cp.REPLAYS = 20
out = open(sys.argv[1], "w")
def say(*a):
    line = " ".join(str(v) for v in a); print(line); out.write(line + "\n"); out.flush()
n = 96
x = torch.randn(n, 32, 48, 72, device="cuda")
wt = torch.randn(32, 32, 3, 3, device="cuda") * 0.02
wph, winv = ops.pack_conv3x3_h3(wt)
agg = lambda: ops.conv3x3(x, None, 32, wph=wph, winv=winv, config=ops.H3_BASE + 7)
agg(); torch.cuda.synchronize()
names = {6: "v_mul_f32 x2 (reference path too: no packed instruction anywhere)",
         0: "v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0] (src1 halves swapped)",
         13: "v_pk_mul_f32 op_sel:[1,0] op_sel_hi:[0,1] (src0 halves swapped)",
         16: "v_pk_mul_f32 op_sel:[1,1] op_sel_hi:[0,0] (both swapped)",
         14: "v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0] (src1 halves swapped)",
         2: "v_pk_add_f32 op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1] (src0 swapped; the conv kernels' form)",
         15: "v_pk_fma_f32 op_sel:[0,1,0] op_sel_hi:[1,0,1] (src1 halves swapped)",
         3: "v_pk_fma_f32 op_sel:[1,0,0] (src0 high half broadcast)",
         1: "v_pk_mul_f32 op_sel:[1,0] (src0 high half broadcast)",
         5: "v_pk_fma_f32 op_sel_hi:[0,1,1] (src0 low half broadcast)",
         4: "v_pk_mul_f32 (no swizzle)",
         17: "v_pk_mov_b32 op_sel:[1,0] (lo <- src0.hi, hi <- src1.lo; the compiler's half-swap, 589 in the shipped library)",
         18: "v_pk_mov_b32 op_sel:[0,1]", 19: "v_pk_mov_b32 op_sel:[1,1]", 20: "v_pk_mov_b32 op_sel:[0,0]"}
cp.REPLAYS = 40
for form, nm in names.items():
    d = cp.pk_canary_case(form, threads=256)
    cp.paired_canary(d, None, say, f"{nm} in the loop, alone")
    cp.paired_canary(d, agg, say, f"{nm} in the loop, beside conv3x3 fp16x3 cfg 7 32->32 @48x72")
