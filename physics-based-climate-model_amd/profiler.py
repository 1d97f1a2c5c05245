"""Per-launch HIP-event timing of the ``cm_*`` launchers (used by bench.py for the roofline figures).

Events are recorded on the stream each kernel is launched on (torch's current stream, whose handle is what the
launchers receive), immediately before and after the launch.  Algorithmic flops / bytes are derived from the launch
arguments, not from counters.
"""
import collections
import contextlib

import torch

# Call-site tag of the launches being enqueued (set by the engine around the ConvLSTM cell, ...): lets bench.py report a
# roofline for a PART of the step (north_star: "the ConvLSTM cell") made of several launcher families.
REGION = None


@contextlib.contextmanager
def region(name):
    global REGION
    prev, REGION = REGION, name
    try:
        yield
    finally:
        REGION = prev


def _conv_flops(a):        # cm_conv3x3(in0, st0, c0, in1, st1, c1, wp, bias, resid, st_resid, out, st_out, n, h, w, cout, ..)
    return 2.0 * a[12] * a[13] * a[14] * a[15] * (a[2] + a[5]) * 9


def _conv_bytes(a):
    n, h, w, cout, cin = a[12], a[13], a[14], a[15], a[2] + a[5]
    return 4.0 * (n * h * w * (cin + cout) + 9 * cin * cout)


def _conv_h3_flops(a):     # cm_conv3x3_h3(in0, st0, c0, in1, st1, c1, wps, winv, bias, resid, st_resid, out, st_out, be, be_stride, n, h, w, cout, ..)
    return 2.0 * a[15] * a[16] * a[17] * a[18] * (a[2] + a[5]) * 9


def _conv_h3_bytes(a):
    n, h, w, cout, cin = a[15], a[16], a[17], a[18], a[2] + a[5]
    return 4.0 * (n * h * w * (cin + cout) + 9 * cin * cout)


def _wgrad_flops(a):       # cm_wgrad3x3(x0, sx0, c0, x1, sx1, c1, dy, sdy, g, ctot, c_off, n, h, w, cout, config, stream)
    return 2.0 * a[11] * a[12] * a[13] * a[14] * (a[2] + a[5]) * 9


def _wgrad_bytes(a):
    n, h, w, cout, cin = a[11], a[12], a[13], a[14], a[2] + a[5]
    return 4.0 * (n * h * w * (cin + cout) + 9 * cin * cout)


def _lstm_fwd_bytes(a):    # (gates, sg, c_prev, scp, c_out, sco, h_out, sho, b, ch, hw, stream)
    return 4.0 * a[8] * a[9] * a[10] * (4 + (1 if a[2] else 0) + 4 + 2)


def _lstm_bwd_bytes(a):    # (gates, sg, c_prev, scp, c_cur, scc, dh_a, sa, dh_b, sb, dc, first, b, ch, hw, stream)
    reads = 4 + (1 if a[2] else 0) + 1 + (1 if a[6] else 0) + (1 if a[8] else 0) + (0 if a[11] else 1)
    return 4.0 * a[12] * a[13] * a[14] * (reads + 5)


def _lstm_fwd_parts_bytes(a):   # (gates, sg, parts, sp, zs, nparts, c_prev, scp, c_out, sco, h_out, sho, b, ch, hw, stream)
    return 4.0 * a[12] * a[13] * a[14] * (4 + 4 * a[5] + (1 if a[6] else 0) + 4 + 2)


def _lstm_bwd_parts_bytes(a):   # (gates, sg, c_prev, scp, c_cur, scc, dh_a, sa, dh_parts, sb, zb, nparts, dc, first, b, ch, hw, stream)
    reads = 4 + (1 if a[2] else 0) + 1 + (1 if a[6] else 0) + a[11] + (0 if a[13] else 1)
    return 4.0 * a[14] * a[15] * a[16] * (reads + 5)


def _wgrad_h3_flops(a):    # cm_wgrad3x3_h3(x0, sx0, c0, x1, sx1, c1, dy, sdy, be_x, be_y, g, ctot, c_off, n, h, w, cout, config, stream)
    return _wgrad_flops(a[:8] + a[10:])


def _wgrad_h3_bytes(a):
    return _wgrad_bytes(a[:8] + a[10:])


def _gemm_flops(a):        # cm_gemm_h3(a, lda, ta, b, ldb, tb, c, ldc, bias, resid, ldr, res_rows, mask, ldm, mask_scale, relu, rng, site, p, m, n, k, ksplit, tile, stream)
    return 2.0 * a[19] * a[20] * a[21]


def _gemm_bytes(a):
    m, n, k = a[19], a[20], a[21]
    return 4.0 * (m * k + n * k + m * n)


def _gemm_wgrad_flops(a):  # cm_gemm_h3_wgrad(dy, ld_dy, x, ldx, dw, ld_dw, dbias, n_out, k_in, tokens, ksplit, tile, stream)
    return 2.0 * a[7] * a[8] * a[9]


def _gemm_wgrad_bytes(a):
    n, k, t = a[7], a[8], a[9]
    return 4.0 * (t * n + t * k + n * k)


def _gemm_pb_flops(a):     # cm_gemm_h3_pb(a, lda, bp, be, c, ldc, bias, resid, ldr, res_rows, mask, ldm, mask_scale, relu, rng, site, p, m, n, k, stream)
    return 2.0 * a[17] * a[18] * a[19]


def _gemm_pb_bytes(a):
    m, n, k = a[17], a[18], a[19]
    return 4.0 * (m * k + n * k + m * n)


MODELS = {
    "cm_gemm_h3": (_gemm_flops, _gemm_bytes),           # ALGORITHMIC flops (x3 are executed)
    "cm_gemm_h3_pb": (_gemm_pb_flops, _gemm_pb_bytes),
    "cm_gemm_h3_wgrad": (_gemm_wgrad_flops, _gemm_wgrad_bytes),
    "cm_conv3x3": (_conv_flops, _conv_bytes),
    "cm_conv3x3_split": (_conv_flops, _conv_bytes),     # same argument positions; ALGORITHMIC flops (x6 are executed)
    "cm_conv3x3_h3": (_conv_h3_flops, _conv_h3_bytes),  # ALGORITHMIC flops (x3 are executed)
    "cm_wgrad3x3_h3": (_wgrad_h3_flops, _wgrad_h3_bytes),
    "cm_wgrad3x3": (_wgrad_flops, _wgrad_bytes),
    "cm_wgrad3x3_split": (_wgrad_flops, _wgrad_bytes),  # same argument positions; ALGORITHMIC flops
    "cm_lstm_gates_fwd": (None, _lstm_fwd_bytes),
    "cm_lstm_gates_bwd": (None, _lstm_bwd_bytes),
    "cm_lstm_gates_fwd_parts": (None, _lstm_fwd_parts_bytes),
    "cm_lstm_gates_bwd_parts": (None, _lstm_bwd_parts_bytes),
}


class KernelTimer:
    """with KernelTimer() as kt: ...launches...; kt.summary() -> {name: dict(calls, ms, flops, bytes)}"""

    def __init__(self):
        self.records = []

    def wrap(self, name, fn):
        if name in ("cm_version", "cm_arch") or "pick_config" in name or "num_configs" in name or "packed_elems" in name or "scratch_elems" in name \
                or "supported" in name or "packed_bytes" in name:
            return fn

        def timed(*a):
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            st = torch.cuda.current_stream()
            e0.record(st)
            rc = fn(*a)
            e1.record(st)
            fl, by = MODELS.get(name, (None, None))
            self.records.append((name, e0, e1, fl(a) if fl else 0.0, by(a) if by else 0.0, REGION))
            return rc
        return timed

    def __enter__(self):
        from ._lib import lib
        lib.set_hook(self)
        return self

    def __exit__(self, *exc):
        from ._lib import lib
        lib.set_hook(None)
        return False

    def summary(self):
        torch.cuda.synchronize()
        out = collections.OrderedDict()
        for name, e0, e1, fl, by, _reg in self.records:
            d = out.setdefault(name, dict(calls=0, ms=0.0, flops=0.0, bytes=0.0))
            d["calls"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["flops"] += fl
            d["bytes"] += by
        return out

    def by_region(self):
        """{region: dict(calls, ms)} over the tagged launches (profiler.region)."""
        torch.cuda.synchronize()
        out = collections.OrderedDict()
        for name, e0, e1, fl, by, reg in self.records:
            if reg is None:
                continue
            d = out.setdefault(reg, dict(calls=0, ms=0.0))
            d["calls"] += 1
            d["ms"] += e0.elapsed_time(e1)
        return out
