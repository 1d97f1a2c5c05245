"""Does a single instruction class of the fp16x3 kernels, looped in an otherwise empty kernel, disturb the canary?  (DPP
row_bcast / row_shr, the f16 MFMA, readlane: none does -- profiles/r03/coresidency/14_synthetic_aggressors_clean.txt; the
disturbance needs the real kernels' mix of LDS / MFMA / VALU traffic.)

    python tools/synthetic_aggressors.py gpurun_out/synthetic_aggressors.txt
"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import coresidency_probe as cp
cp.REPLAYS = 20
lib = cp.canary_lib()
lib.aggressor_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
sink = torch.zeros(1024, device="cuda")
names = {0: "DPP row_bcast:15 + row_bcast:31 (row masks 0xa / 0xc, bound_ctrl)", 1: "DPP row_shr:1 / row_shr:8", 2: "v_mfma_f32_32x32x16_f16",
         3: "v_readlane / v_readfirstlane", 4: "DPP row_bcast:15 alone", 5: "DPP row_bcast:31 alone", 6: "DPP row_bcast:15, all rows, no bound_ctrl"}
out = open(sys.argv[1], "w")
def say(*a):
    line = " ".join(str(v) for v in a); print(line); out.write(line + "\n"); out.flush()
dets = {"arithmetic canary (v_pk_mul swizzle, 256 threads)": cp.pk_canary_case(0, threads=256),
        "arithmetic canary (v_mul, 256 threads)": cp.pk_canary_case(6, threads=256),
        "arithmetic canary (v_mul, 1024 threads)": cp.pk_canary_case(6, threads=1024),
        "parked-register canary (256 threads)": cp.canary_case(56, 256, 7700)}
for mode, nm in names.items():
    for blocks, iters in ((2048, 4000),):
        def agg(mode=mode, blocks=blocks, iters=iters):
            rc = lib.aggressor_launch(sink.data_ptr(), mode, blocks, iters, torch.cuda.current_stream().cuda_stream)
            assert rc == 0
        for dn, d in dets.items():
            cp.paired_canary(d, agg, say, f"aggressor = {nm} ({blocks} workgroups x {iters} iterations) | {dn}")
