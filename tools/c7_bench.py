"""Times cm_conv7_bwd at the benchmark shapes.  (r01 experiment behind its design: single-copy atomics 42/21/14/8.5 us,
partial copies + ticket + __threadfence 123/50/31/18 us, plain partial stores 27/10/6/4.5 us + fold.)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from climate_amd._lib import lib, check
for (n, h, w) in [(192, 48, 72), (192, 24, 36), (192, 12, 18), (192, 6, 9), (32, 48, 72)]:
    dg = torch.randn(n, h, w, device="cuda"); fm = torch.randn(n, 2, h, w, device="cuda")
    w7 = torch.randn(98, device="cuda"); dmap = torch.empty(n, 2, h, w, device="cuda"); dw7 = torch.zeros(98, device="cuda")
    scr = torch.zeros(4096 + 98 * 4096, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    f = lambda: check(lib.cm_conv7_bwd(dg.data_ptr(), fm.data_ptr(), w7.data_ptr(), dmap.data_ptr(), dw7.data_ptr(), scr.data_ptr(), n, h, w, st))
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize()
    print((n, h, w), f"{e0.elapsed_time(e1) * 20:.1f} us")
