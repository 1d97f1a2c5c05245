"""Times cm_se_excite_bwd at the benchmark's seven ConvBlock shapes under graph replay."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from climate_amd._lib import lib, check
tot = 0.0
for (n, c) in [(192, 32), (192, 64), (192, 128), (192, 256), (32, 128), (32, 64), (32, 32)]:
    cr = c // 8
    R = lambda *s: torch.randn(*s, device="cuda")
    ds, s, z, pooled, w1, w2 = R(n, c), torch.sigmoid(R(n, c)), R(n, cr), R(n, c), R(cr, c), R(c, cr)
    dsig, dz, dpool, dw1, dw2 = R(n, c), R(n, cr), R(n, c), torch.zeros(cr, c, device="cuda"), torch.zeros(c, cr, device="cuda")
    f = lambda: check(lib.cm_se_excite_bwd(ds.data_ptr(), s.data_ptr(), z.data_ptr(), pooled.data_ptr(), w1.data_ptr(), w2.data_ptr(),
                                           dsig.data_ptr(), dz.data_ptr(), dpool.data_ptr(), dw1.data_ptr(), dw2.data_ptr(), n, c, cr, torch.cuda.current_stream().cuda_stream))
    g = torch.cuda.CUDAGraph()
    for _ in range(3): f()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(20): f()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 50
    tot += t
    print(f"n={n} c={c}: {t:.1f} us")
print(f"total {tot:.1f} us")
