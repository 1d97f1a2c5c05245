"""Tensor-level wrappers over the C ABI (one Python function per ``cm_*`` launcher).

torch supplies device memory and the current HIP stream only; every function enqueues hand-written HIP kernels from
libclimate_hip.so.  There is deliberately no CPU implementation here: calling these with CPU tensors raises.
"""
import os

import torch

from ._lib import check, lib

GN_GROUPS = 8
GN_EPS = 1e-5


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    if t is None:
        return None
    if not t.is_cuda or t.dtype != torch.float32:
        raise RuntimeError("climate_hip ops need float32 tensors on the GPU (no CPU fallback)")
    return t.data_ptr()


# --------------------------------------------------------------------------------------------------- autotuning
# Tile configurations are chosen empirically per call signature the first time it is seen outside graph capture
# (3 timed launches per configuration into scratch outputs, then 10 more for the four leaders); the choice is cached
# for the life of the process and can be saved / pre-loaded (save_tuned / load_tuned, CM_TUNE_CACHE).
AUTOTUNE = True
_TUNED = {}


def _pick(key, candidates, launch, fallback):
    """launch(cfg) enqueues one launch with that configuration into scratch memory."""
    if key in _TUNED:
        return _TUNED[key]
    if not AUTOTUNE or torch.cuda.is_current_stream_capturing():
        return fallback
    torch.cuda.synchronize()      # drain every stream so the candidates are timed alone on the device

    def timed(cfg, reps):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            launch(cfg)
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / reps

    scored = []
    for cfg in (range(candidates) if isinstance(candidates, int) else candidates):
        if launch(cfg) != 0:          # configuration not applicable to this call (argument error): skip it
            continue
        scored.append((timed(cfg, 3), cfg))
    best = fallback
    if scored:
        scored.sort()
        # second look at the leaders with more repetitions: 3-launch timings of ~50 us kernels are noisy
        finals = [(timed(cfg, 10), cfg) for _, cfg in scored[:4]]
        best = min(finals)[1]
    _TUNED[key] = best
    return best


def tuned_table():
    return dict(_TUNED)


def save_tuned(path):
    """Write the tuned configuration table (call signature -> configuration id) as text, one entry per line."""
    with open(path, "w") as f:
        for k, v in sorted(_TUNED.items(), key=repr):
            f.write(f"{k!r}\t{v}\n")


def load_tuned(path):
    """Pre-load a table written by save_tuned (same GPU model and library build): calls found in it are not re-timed,
    which keeps autotuner trial launches out of a profiled run."""
    import ast
    n = 0
    with open(path) as f:
        for line in f:
            k, v = line.rstrip("\n").split("\t")
            _TUNED[ast.literal_eval(k)] = int(v)
            n += 1
    return n


if os.environ.get("CM_TUNE_CACHE") and os.path.exists(os.environ["CM_TUNE_CACHE"]):
    load_tuned(os.environ["CM_TUNE_CACHE"])


def _contig(t):
    if not t.is_contiguous():
        raise RuntimeError("climate_hip ops need contiguous tensors")
    return t


# ----------------------------------------------------------------------------------------------------- conv3x3
def pack_conv3x3(w, c_off=0, cin=None, dgrad=False):
    cout, cin_total = w.shape[0], w.shape[1]
    cin = cin_total - c_off if cin is None else cin
    n = lib.cm_conv3x3_packed_elems(cout if dgrad else cin, cin if dgrad else cout)
    wp = torch.empty(n, device=w.device, dtype=torch.float32)
    check(lib.cm_pack_conv3x3(_p(_contig(w)), cout, cin_total, c_off, cin, int(dgrad), _p(wp), _stream()), "pack")
    return wp


SPLIT_BASE = 1 << 20   # tuned configuration ids >= SPLIT_BASE select the bf16x6 kernel (cm_conv3x3_split)
H3_BASE = 1 << 21      # ... ids >= H3_BASE the fp16x3 kernel (cm_conv3x3_h3 / cm_wgrad3x3_h3)
ACC_CHANNELS = int(os.environ.get("CM_ACC_CHANNELS", "0"))   # see conv3x3: longest reduction one accumulator takes (0: no bound)
SMALLC_CFG = 1 << 22   # tuned configuration id of the few-input-channels kernels (cm_conv3x3_smallc / cm_wgrad3x3_smallc)
LAST_CONV_CONFIG = -1   # configuration the most recent conv3x3() call ran with (the engine prunes unused weight packs)


class SampleExponents:
    """Per-sample biased exponents of max |x| ([n] int32, ``stride`` entries apart, zero-initialised by the owner) that a
    cm_conv3x3_h3 launch publishes for the tensor it reads and cm_wgrad3x3_h3 consumes.  ``valid`` is set by the launch
    wrapper: only the fp16x3 kernel publishes, and a consumer must not trust a table nobody wrote."""
    __slots__ = ("t", "stride", "valid")

    def __init__(self, t, stride=1):
        self.t, self.stride, self.valid = t, int(stride), False

    @staticmethod
    def measure(x):
        """Stand-alone measurement (one launch) for an x [N, C, H, W] no fp16x3 conv has read."""
        n = x.shape[0]
        se = SampleExponents(torch.zeros(n, device=x.device, dtype=torch.int32))
        per = x[0].numel()
        if x.dim() != 4 or x.stride(3) != 1 or x.stride(2) != x.shape[3] or x.stride(1) != x.shape[2] * x.shape[3]:
            raise RuntimeError("SampleExponents.measure needs [N, C, H, W] with dense channel planes")
        check(lib.cm_sample_exponents(_p(x), x.stride(0), n, per, _p_any(se.t), 1, _stream()), "sample_exponents")
        se.valid = True
        return se


# GroupNorm statistics from the conv epilogue (cm_conv3x3_h3_gn -> cm_gn_silu_fwd_stats): built and parity-tested, OFF by
# default -- measured 0.8 % SLOWER at config 2 (tools/ab_bench.py, same box: 8376 vs 8447 samples/s): the two extra
# barriers and reductions sit on every conv workgroup's critical path, while the pass they save re-reads the conv output
# from L2 in a launch that is short anyway.  CM_GN_EPILOGUE=1 switches it on.
GN_EPILOGUE = os.environ.get("CM_GN_EPILOGUE", "0") != "0"


class GnPartials:
    """GroupNorm(8) partial statistics of a conv output, written by the conv's own epilogue (cm_conv3x3_h3_gn) when the
    tile configuration the call runs with can produce them: ``t`` [n, 8, slots, 3] or None.  Handed to gn_silu_fwd."""
    __slots__ = ("t", "slots")

    def __init__(self):
        self.t, self.slots = None, 0


def conv3x3(x0, wp, cout, x1=None, bias=None, resid=None, out=None, config=-1, wps=None, w_raw=None,
            out_zeroed=False, wph=None, winv=None, be_out=None, gn_out=None):
    """out = conv3x3(cat(x0, x1), wp) + bias + resid.  x0/x1: [N,C,H,W] (sample stride may exceed C*H*W).

    ``wp`` is the fp32-MFMA operand (cm_pack_conv3x3); ``wps`` (optional) the bf16x6 operand of the same weight;
    ``w_raw`` (optional) the unpacked [cout, cin, 3, 3] parameter, which adds the few-input-channels kernel
    (cm_conv3x3_smallc) to the candidates when cin * 9 <= 64.  ``out_zeroed``: the caller has filled ``out`` with
    zeros (one fill for several launches), so a K-split bf16x6 launch skips its own zero-fill launch.
    ``wph`` / ``winv`` (optional) the fp16x3 operand of the same weight and its inverse-scale scalar (pack_conv3x3_h3).
    ``be_out`` (optional SampleExponents): filled with the per-sample magnitudes of the input when the fp16x3 kernel runs.
    With config < 0 the autotuner times every kernel family it has an operand for on this call signature and keeps
    the fastest."""
    n, c0, h, w = x0.shape
    c1 = 0 if x1 is None else x1.shape[1]
    if out is None:
        out = torch.empty(n, cout, h, w, device=x0.device, dtype=torch.float32)
    for t in (x0, x1, out, resid):
        if t is not None and (t.stride(3) != 1 or t.stride(2) != w or t.stride(1) != h * w):
            raise RuntimeError("conv3x3 needs dense HxW planes with channel stride H*W")
    st1 = 0 if x1 is None else x1.stride(0)
    if config < 0:
        def launch(cfg, _scratch=[None]):
            if _scratch[0] is None:
                _scratch[0] = torch.empty(n, cout, h, w, device=x0.device, dtype=torch.float32)
            if cfg == SMALLC_CFG:
                return lib.cm_conv3x3_smallc(_p(x0), x0.stride(0), c0, _p(w_raw), _p(bias), _p(_scratch[0]),
                                             _scratch[0].stride(0), n, h, w, cout, _stream())
            if cfg >= H3_BASE:
                return lib.cm_conv3x3_h3(_p(x0), x0.stride(0), c0, _p(x1), st1, c1, _p(wph), _p(winv), _p(bias), None, 0,
                                         _p(_scratch[0]), _scratch[0].stride(0), None, 0, n, h, w, cout, cfg - H3_BASE,
                                         _stream())
            if cfg >= SPLIT_BASE:
                return lib.cm_conv3x3_split(_p(x0), x0.stride(0), c0, _p(x1), st1, c1, _p(wps), _p(bias), None, 0,
                                            _p(_scratch[0]), _scratch[0].stride(0), n, h, w, cout, cfg - SPLIT_BASE,
                                            _stream())
            return lib.cm_conv3x3(_p(x0), x0.stride(0), c0, _p(x1), st1, c1, _p(wp), _p(bias), None, 0,
                                  _p(_scratch[0]), _scratch[0].stride(0), n, h, w, cout, cfg, _stream())
        nchunks = (c0 + c1 + 7) // 8
        splits = [1] + [k for k in (2, 4, 8) if nchunks >= 4 * k and not (resid is not None and resid.data_ptr() == out.data_ptr())]
        cands = [c + ((k if k > 1 else 0) << 8) for c in range(lib.cm_conv3x3_num_configs()) for k in splits]
        if wp is None:              # the caller holds only the bf16x6 operand of this weight
            cands = []
        use_split = wps is not None and (c1 == 0 or c0 % 16 == 0)
        if use_split:
            ssplits = [1] + [k for k in (2, 4) if (c0 + c1) // 16 >= 4 * k and len(splits) > 1]
            cands += [SPLIT_BASE + c + ((k if k > 1 else 0) << 8) for c in range(lib.cm_conv3x3_split_num_configs())
                      for k in ssplits]
        use_h3 = wph is not None and (c1 == 0 or c0 % 16 == 0)
        if use_h3:
            hsplits = [1] + [k for k in (2, 4, 8, 16) if (c0 + c1) // 16 >= 2 * k and len(splits) > 1]
            # rounding noise of the fp32 accumulators grows as sqrt(MFMA updates per accumulator) (tools/op_noise.py:
            # 3.7e-7 at 128 input channels, 7.4e-7 at 512, 1.05e-6 at 1024, halving per 4x reduction split).
            # CM_ACC_CHANNELS=<n> bounds the input channels one accumulator takes by forcing reduction splits on wider
            # layers; off by default: at model level it bought nothing measurable (tools/noise_probe.py, worst gradient
            # 1.9e-5 vs 2.0e-5) for 1.3 % of the step.
            kmin = 1
            while ACC_CHANNELS > 0 and kmin * ACC_CHANNELS < c0 + c1:
                kmin *= 2
            if any(k >= kmin for k in hsplits):
                hsplits = [k for k in hsplits if k >= kmin]
            cands += [H3_BASE + c + ((k if k > 1 else 0) << 8) for c in range(lib.cm_conv3x3_split_num_configs())
                      for k in hsplits]
        use_smallc = (w_raw is not None and c1 == 0 and c0 * 9 <= 64 and w <= 320 and resid is None
                      and tuple(w_raw.shape) == (cout, c0, 3, 3) and w_raw.is_contiguous())
        if use_smallc:
            cands.append(SMALLC_CFG)
        if not cands:
            raise RuntimeError("conv3x3: no operand form usable for this call")
        # (same cache key with or without the fp32 operand: a caller that dropped it did so because the cached
        #  choice for its calls is a bf16x6 configuration)
        key = (("conv3x3", n, h, w, c0, c1, cout, len(splits), use_split) + (("smallc",) if use_smallc else ())
               + (("h3", ACC_CHANNELS) if use_h3 else ()))
        mfma16 = H3_BASE if use_h3 else SPLIT_BASE
        config = _pick(key, cands, launch, -1 if wp is not None else mfma16)
        if wp is None and config < SPLIT_BASE:      # cached while the fp32 operand still existed: tune the others only
            config = _pick(key + ("no-fp32",), cands, launch, mfma16)
    global LAST_CONV_CONFIG
    LAST_CONV_CONFIG = config
    if config == SMALLC_CFG:
        check(lib.cm_conv3x3_smallc(_p(x0), x0.stride(0), c0, _p(w_raw), _p(bias), _p(out), out.stride(0), n, h, w, cout,
                                    _stream()), "conv3x3_smallc")
        return out
    if config >= H3_BASE:
        cfg = config - H3_BASE
        slots = lib.cm_conv3x3_h3_gn_slots(cfg, h, w, cout) if (gn_out is not None and GN_EPILOGUE) else 0
        if slots > 0:
            gn_out.t = torch.empty(n, GN_GROUPS, slots, 3, device=x0.device, dtype=torch.float32)
            gn_out.slots = slots
            check(lib.cm_conv3x3_h3_gn(_p(x0), x0.stride(0), c0, _p(x1), st1, c1, _p(wph), _p(winv), _p(bias), _p(resid),
                                       0 if resid is None else resid.stride(0), _p(out), out.stride(0),
                                       None if be_out is None else _p_any(be_out.t), 0 if be_out is None else be_out.stride,
                                       _p(gn_out.t), slots, n, h, w, cout, cfg, _stream()), "conv3x3_h3_gn")
            if be_out is not None:
                be_out.valid = True
            return out
        if out_zeroed and (cfg >> 8) > 1:
            cfg |= 1 << 30
        check(lib.cm_conv3x3_h3(_p(x0), x0.stride(0), c0, _p(x1), st1, c1, _p(wph), _p(winv), _p(bias), _p(resid),
                                0 if resid is None else resid.stride(0), _p(out), out.stride(0),
                                None if be_out is None else _p_any(be_out.t), 0 if be_out is None else be_out.stride,
                                n, h, w, cout, cfg, _stream()), "conv3x3_h3")
        if be_out is not None:
            be_out.valid = True
        return out
    if config >= SPLIT_BASE:
        cfg = config - SPLIT_BASE
        if out_zeroed and (cfg >> 8) > 1:
            cfg |= 1 << 30
        check(lib.cm_conv3x3_split(_p(x0), x0.stride(0), c0, _p(x1), st1, c1, _p(wps), _p(bias), _p(resid),
                                   0 if resid is None else resid.stride(0), _p(out), out.stride(0), n, h, w, cout,
                                   cfg, _stream()), "conv3x3_split")
        return out
    check(lib.cm_conv3x3(_p(x0), x0.stride(0), c0, _p(x1), st1, c1, _p(wp), _p(bias),
                         _p(resid), 0 if resid is None else resid.stride(0), _p(out), out.stride(0), n, h, w, cout,
                         config, _stream()), "conv3x3")
    return out


PART_SLICES = (2, 4, 8)      # reduction splits the partial-slices form of a conv may use


def conv3x3_parts(x0, cout, wph, winv, bias=None, parts=None, config=-1, be_out=None, x1=None):
    """"Partial slices" form of the fp16x3 conv for launches that are too small to fill the chip (the ConvLSTM's
    recurrent projection and its data gradient): the reduction is split k ways over blockIdx.z and share z STORES its
    partial sums into slice z of ``parts`` [max(PART_SLICES), N, cout, H, W] -- no zero fill, no atomics (they were half
    of such a launch's time), a fixed summation order.  Returns (parts, k): the consumer adds the first k slices
    (lstm_gates_fwd / lstm_gates_bwd with ``parts``).  ``bias`` lands in slice 0; ``be_out`` as in conv3x3 (every share
    posts the maxima of the channels it read).  Returns None when the reduction has
    fewer than two 16-channel k-steps (the caller uses conv3x3)."""
    n, c0, h, w = x0.shape
    c1 = 0 if x1 is None else x1.shape[1]
    nsteps = (c0 + c1 + 15) // 16
    ks = [k for k in PART_SLICES if nsteps % k == 0 and k <= nsteps]
    if not ks or wph is None or (c1 and c0 % 16):
        return None
    if x0.stride(3) != 1 or x0.stride(2) != w or x0.stride(1) != h * w:
        raise RuntimeError("conv3x3_parts needs dense HxW planes with channel stride H*W")
    if parts is None:
        parts = torch.empty(max(PART_SLICES), n, cout, h, w, device=x0.device, dtype=torch.float32)
    if parts.shape[0] < max(ks) or tuple(parts.shape[1:]) != (n, cout, h, w) or not parts.is_contiguous():
        raise RuntimeError("conv3x3_parts: parts must be a contiguous [>= k, N, cout, H, W] stack")

    def call(cfg, dst, be=None):
        return lib.cm_conv3x3_h3(_p(x0), x0.stride(0), c0, _p(x1), 0 if x1 is None else x1.stride(0), c1, _p(wph),
                                 _p(winv), _p(bias), None, 0, _p(dst), dst.stride(1),
                                 None if be is None else _p_any(be.t), 0 if be is None else be.stride,
                                 n, h, w, cout, (cfg - H3_BASE) | (1 << 29), _stream())
    if config < 0:
        cands = [H3_BASE + c + (k << 8) for c in range(lib.cm_conv3x3_split_num_configs()) for k in ks]
        config = _pick(("conv3x3p", n, h, w, c0, c1, cout), cands, lambda cfg: call(cfg, parts), H3_BASE + (ks[0] << 8))
    check(call(config, parts, be_out), "conv3x3_h3 (partial slices)")
    if be_out is not None:
        be_out.valid = True
    return parts, (config - H3_BASE) >> 8


def pack_conv3x3_split(w, c_off=0, cin=None, dgrad=False):
    """bf16x6 operand form of a 3x3 weight (single-job use of the batched packer; the engine batches all jobs)."""
    cout, cin_total = w.shape[0], w.shape[1]
    cin = cin_total - c_off if cin is None else cin
    nbytes = lib.cm_conv3x3_split_packed_bytes(cout if dgrad else cin, cin if dgrad else cout)
    wps = torch.empty(nbytes // 4, device=w.device, dtype=torch.float32)
    blocks = max(1, min(512, nbytes // 48 // 256 + 1))
    table = torch.tensor([[w.data_ptr(), wps.data_ptr(), cout, cin_total, c_off, cin, int(dgrad), 0],
                          [0, 0, 0, 0, 0, 0, 0, blocks]], dtype=torch.int64).to(w.device)
    check(lib.cm_pack_conv3x3_split_batch(_p_any(table), 1, blocks, _stream()), "pack_split")
    return wps


def _p_any(t):
    return t.data_ptr()


def pack_conv3x3_h3(w, c_off=0, cin=None, dgrad=False):
    """fp16x3 operand form of a 3x3 weight: returns (wph, winv) -- two fp16 pieces scaled by a power of two and the
    device scalar that undoes the scale (single-job use of the batched packer; the engine batches all jobs)."""
    cout, cin_total = w.shape[0], w.shape[1]
    cin = cin_total - c_off if cin is None else cin
    nbytes = lib.cm_conv3x3_h3_packed_bytes(cout if dgrad else cin, cin if dgrad else cout)
    wph = torch.empty(nbytes // 4, device=w.device, dtype=torch.float32)
    blocks = max(1, min(512, nbytes // 32 // 256 + 1))
    scratch = torch.empty(1 + blocks, device=w.device, dtype=torch.float32)
    table = torch.tensor([[w.data_ptr(), wph.data_ptr(), cout, cin_total, c_off, cin, int(dgrad), 0],
                          [0, 0, 0, 0, 0, 0, 0, blocks]], dtype=torch.int64).to(w.device)
    check(lib.cm_pack_conv3x3_h3_batch(_p_any(table), 1, blocks, _p(scratch), _stream()), "pack_h3")
    return wph, scratch[:1]


def conv3x3_split(x0, wps, cout, x1=None, bias=None, resid=None, out=None, config=0):
    n, c0, h, w = x0.shape
    c1 = 0 if x1 is None else x1.shape[1]
    if out is None:
        out = torch.empty(n, cout, h, w, device=x0.device, dtype=torch.float32)
    check(lib.cm_conv3x3_split(_p(x0), x0.stride(0), c0, _p(x1), 0 if x1 is None else x1.stride(0), c1, _p(wps),
                               _p(bias), _p(resid), 0 if resid is None else resid.stride(0), _p(out), out.stride(0),
                               n, h, w, cout, config, _stream()), "conv3x3_split")
    return out


# Matrix-core numerics of the WEIGHT GRADIENT (CM_WGRAD_NUMERICS): follows the convolutions -- fp16x3 by default, bf16x6
# under CM_CONV_NUMERICS=bf16x6, fp32 under fp32.  The reduction of a weight gradient mixes samples, and the reference's
# left-padded windows put all-zero frames (gradients amplified by rstd = 1/sqrt(eps) = 316 per GroupNorm) next to real
# ones, 2^28 apart: cm_wgrad3x3_h3 therefore scales PER SAMPLE with a constant product scale (csrc/wgrad3x3_split.hip),
# fed by the per-sample magnitudes the fp16x3 convolutions publish for the tensors they read (SampleExponents).
_cn = os.environ.get("CM_CONV_NUMERICS", "fp16x3")
_WG_NUM = os.environ.get("CM_WGRAD_NUMERICS", _cn if _cn in ("fp16x3", "bf16x6", "fp32") else "bf16x6")
if _WG_NUM not in ("fp16x3", "bf16x6", "fp32"):
    raise RuntimeError(f"CM_WGRAD_NUMERICS={_WG_NUM!r}: expected fp16x3, bf16x6 or fp32")
WGRAD_ROUNDS = tuple(int(v) for v in os.environ.get("CM_WGRAD_ROUNDS", "2,4,8").split(","))   # grid sizes the tuner tries
WGRAD_BF16X6 = _WG_NUM == "bf16x6"
WGRAD_H3 = _WG_NUM == "fp16x3"


def _wgrad_call(x0, dy, g, c_off, x1, config, be_x=None, be_y=None):
    """One launch of a weight-gradient family: ids >= H3_BASE select the fp16x3 kernel (cm_wgrad3x3_h3, needs the
    per-sample exponent tables), ids >= SPLIT_BASE the bf16x6 kernel (cm_wgrad3x3_split), SMALLC_CFG the first-layer
    kernel (cm_wgrad3x3_smallc)."""
    n, c0, h, w = x0.shape
    c1 = 0 if x1 is None else x1.shape[1]
    cout, ctot = g.shape[0], g.shape[2]
    st1 = 0 if x1 is None else x1.stride(0)
    if config == SMALLC_CFG:
        ws = torch.empty(int(lib.cm_wgrad3x3_smallc_scratch_elems(n, h, w, cout)), device=x0.device, dtype=torch.float32)
        return lib.cm_wgrad3x3_smallc(_p(x0), x0.stride(0), c0, _p(dy), dy.stride(0), _p(g), ctot, c_off, n, h, w, cout,
                                      _p(ws), _stream())
    if config >= H3_BASE:
        return lib.cm_wgrad3x3_h3(_p(x0), x0.stride(0), c0, _p(x1), st1, c1, _p(dy), dy.stride(0), _p_any(be_x.t), _p_any(be_y.t),
                                  _p(g), ctot, c_off, n, h, w, cout, config - H3_BASE, _stream())
    elif config >= SPLIT_BASE:
        fn, cfg = lib.cm_wgrad3x3_split, config - SPLIT_BASE
    else:
        fn, cfg = lib.cm_wgrad3x3, config
    return fn(_p(x0), x0.stride(0), c0, _p(x1), st1, c1, _p(dy), dy.stride(0), _p(g), ctot, c_off, n, h, w, cout, cfg,
              _stream())


def wgrad3x3(x0, dy, g, c_off=0, x1=None, config=-1, be_x=None, be_y=None):
    """g[cout][9][ctot] += wgrad(cat(x0, x1), dy) for input-channel range [c_off, ...).

    ``be_x`` / ``be_y`` (SampleExponents, optional): per-sample magnitudes of cat(x0, x1) / dy for the fp16x3 kernel,
    normally published by the conv launches that read the same tensors; measured here (one launch each) when missing.

    With config < 0 the autotuner times the fp32-MFMA configurations and those of the split-operand family that
    CM_WGRAD_NUMERICS selects (fp16x3 by default) on this call signature and keeps the fastest."""
    n, c0, h, w = x0.shape
    c1 = 0 if x1 is None else x1.shape[1]
    cout = g.shape[0]
    be = [be_x, be_y]

    def tables():
        """Per-sample magnitudes for the fp16x3 kernel: the published ones, or measured here (one launch per operand)."""
        if be[0] is None or not be[0].valid or be[0].stride != 1:
            be[0] = SampleExponents.measure(x0)
            if x1 is not None:
                check(lib.cm_sample_exponents(_p(x1), x1.stride(0), n, x1[0].numel(), _p_any(be[0].t), 1, _stream()),
                      "sample_exponents")
        if be[1] is None or not be[1].valid or be[1].stride != 1:
            be[1] = SampleExponents.measure(dy)
        return be
    if config < 0:
        def launch(cfg, _scratch=[None]):
            if _scratch[0] is None:
                _scratch[0] = torch.empty_like(g)
            return _wgrad_call(x0, dy, _scratch[0], c_off, x1, cfg, *(tables() if cfg >= H3_BASE and cfg != SMALLC_CFG
                                                                       else (None, None)))
        cands = [c + (u << 8) for c in range(lib.cm_wgrad3x3_num_configs()) for u in (2, 3, 4, 6, 8)]
        if (WGRAD_BF16X6 or WGRAD_H3) and (c1 == 0 or c0 % 32 == 0):
            base = H3_BASE if WGRAD_H3 else SPLIT_BASE
            cands = cands + [base + c + (u << 8) for c in range(lib.cm_wgrad3x3_split_num_configs()) for u in WGRAD_ROUNDS]
        if c1 == 0 and c0 * 9 <= 64 and w % 4 == 0 and w <= 320 and dy.stride(0) % 4 == 0:
            cands.append(SMALLC_CFG)
        config = _pick(("wgrad3x3", n, h, w, c0, c1, cout, _WG_NUM), cands, launch, -1)
    if config >= H3_BASE and config != SMALLC_CFG:
        tables()
    check(_wgrad_call(x0, dy, g, c_off, x1, config, be[0], be[1]), "wgrad3x3")
    return g


def wgrad3x3_unpack(g, scale=1.0):
    cout, _, ctot = g.shape
    dw = torch.empty(cout, ctot, 3, 3, device=g.device, dtype=torch.float32)
    check(lib.cm_wgrad3x3_unpack(_p(g), _p(dw), cout, ctot, scale, _stream()), "unpack")
    return dw


# ----------------------------------------------------------------------------------------------------- GN + SiLU
def gn_silu_fwd(x, gamma, beta, want_pooled=False, parts=None, gn=None):
    """parts = (stack [>= k, N, C, H, W], k) from conv3x3_parts instead of x: the launch adds the slices and returns the
    summed conv output as a fourth value (x is ignored).  gn (GnPartials with .t set): the statistics the producing conv's
    epilogue wrote -- the launch merges them instead of making its own statistics pass over x."""
    if gn is not None and gn.t is not None and parts is None:
        n, c, h, w = x.shape
        y = torch.empty_like(_contig(x))
        stats = torch.empty(n * GN_GROUPS * 2, device=x.device, dtype=torch.float32)
        pooled = torch.empty(n, c, device=x.device, dtype=torch.float32) if want_pooled else None
        check(lib.cm_gn_silu_fwd_stats(_p(x), _p(gn.t), gn.slots, _p(gamma), _p(beta), _p(y), _p(stats), _p(pooled), n, c,
                                       h * w, GN_GROUPS, GN_EPS, _stream()), "gn_silu_fwd_stats")
        return y, stats, pooled
    if parts is not None:
        pt, k = parts
        _, n, c, h, w = pt.shape
        xs = torch.empty(n, c, h, w, device=pt.device, dtype=torch.float32)
        y = torch.empty_like(xs)
        stats = torch.empty(n * GN_GROUPS * 2, device=pt.device, dtype=torch.float32)
        pooled = torch.empty(n, c, device=pt.device, dtype=torch.float32) if want_pooled else None
        check(lib.cm_gn_silu_fwd_parts(_p(pt), pt.stride(0), k, _p(xs), _p(gamma), _p(beta), _p(y), _p(stats), _p(pooled),
                                       n, c, h * w, GN_GROUPS, GN_EPS, _stream()), "gn_silu_fwd_parts")
        return y, stats, pooled, xs
    n, c, h, w = x.shape
    y = torch.empty_like(_contig(x))
    stats = torch.empty(n * GN_GROUPS * 2, device=x.device, dtype=torch.float32)
    pooled = torch.empty(n, c, device=x.device, dtype=torch.float32) if want_pooled else None
    check(lib.cm_gn_silu_fwd(_p(x), _p(gamma), _p(beta), _p(y), _p(stats), _p(pooled), n, c, h * w, GN_GROUPS, GN_EPS,
                             _stream()), "gn_silu_fwd")
    return y, stats, pooled


def gn_silu_apply(x, gamma, beta, stats):
    """SiLU(GroupNorm(x)) from stored statistics (bit-identical to gn_silu_fwd's output)."""
    n, c, h, w = x.shape
    y = torch.empty_like(_contig(x))
    check(lib.cm_gn_silu_apply(_p(x), _p(gamma), _p(beta), _p(stats), _p(y), n, c, h * w, GN_GROUPS, _stream()),
          "gn_silu_apply")
    return y


def gn_silu_bwd(x, gamma, beta, stats, dA, dgamma, dbeta):
    n, c, h, w = x.shape
    dx = torch.empty_like(_contig(x))
    check(lib.cm_gn_silu_bwd(_p(x), _p(gamma), _p(beta), _p(stats), _p(dA), dA.stride(0), _p(dx), _p(dgamma),
                             _p(dbeta), n, c, h * w, GN_GROUPS, _stream()), "gn_silu_bwd")
    return dx


def gn_silu_bwd_gated(x, gamma, beta, stats, a2, dout, gate, dmap, umax, cnt, s, dpool, dgamma, dbeta, se=None):
    """``se`` = (dsig, dz, z, pooled, dw1, dw2) hands the SE weight gradients of the preceding gates_bwd(...,
    defer_se_wgrad=True) to this launch as a side duty (one launch less)."""
    n, c, h, w = x.shape
    dx = torch.empty_like(_contig(x))
    sd = se if se is not None else (None,) * 6
    cr = 0 if se is None else se[1].shape[1]
    check(lib.cm_gn_silu_bwd_gated(_p(x), _p(gamma), _p(beta), _p(stats), _p(a2), _p(_contig(dout)), _p(gate),
                                   _p(dmap), _p(umax), _p(cnt), _p(s), _p(dpool), _p(dx), _p(dgamma), _p(dbeta), n, c,
                                   h * w, GN_GROUPS, _p(sd[0]), _p(sd[1]), _p(sd[2]), _p(sd[3]), _p(sd[4]), _p(sd[5]), cr,
                                   _stream()), "gn_silu_bwd_gated")
    return dx


# ----------------------------------------------------------------------------------------------------- gates
def se_excite_fwd(pooled, w1, w2):
    n, c = pooled.shape
    cr = w1.shape[0]
    z = torch.empty(n, cr, device=pooled.device, dtype=torch.float32)
    s = torch.empty(n, c, device=pooled.device, dtype=torch.float32)
    check(lib.cm_se_excite_fwd(_p(pooled), _p(_contig(w1)), _p(_contig(w2)), _p(z), _p(s), n, c, cr, _stream()), "se")
    return z, s


def spatial_gate_fwd(a2, s, w7):
    n, c, h, w = a2.shape
    fmap = torch.empty(n, 2, h, w, device=a2.device, dtype=torch.float32)
    gate = torch.empty(n, h, w, device=a2.device, dtype=torch.float32)
    out = torch.empty_like(_contig(a2))
    check(lib.cm_spatial_stats(_p(a2), _p(s), _p(fmap), n, c, h * w, _stream()), "spatial_stats")
    check(lib.cm_spatial_apply(_p(a2), _p(s), _p(fmap), _p(_contig(w7)), _p(gate), _p(out), None, n, c, h, w,
                               _stream()), "spatial_apply")
    return out, fmap, gate


def se_spatial_gate_fwd(a2, pooled, w1, w2, w7, pool_out=False):
    """SE excite + spatial gate of one ConvBlock (src/unet.py:45-47) in two launches; returns out, z, s, fmap, gate
    (and MaxPool2d(2)(out) when ``pool_out``: the encoder's next input, written by the same pass)."""
    n, c, h, w = a2.shape
    cr = w1.shape[0]
    dev = a2.device
    z = torch.empty(n, cr, device=dev, dtype=torch.float32)
    s = torch.empty(n, c, device=dev, dtype=torch.float32)
    fmap = torch.empty(n, 2, h, w, device=dev, dtype=torch.float32)
    gate = torch.empty(n, h, w, device=dev, dtype=torch.float32)
    out = torch.empty_like(_contig(a2))
    st = _stream()
    check(lib.cm_se_spatial_stats(_p(pooled), _p(_contig(w1)), _p(_contig(w2)), _p(a2), _p(z), _p(s), _p(fmap), n, c, cr,
                                  h * w, st), "se_spatial_stats")
    mp = torch.empty(n, c, h // 2, w // 2, device=dev, dtype=torch.float32) if pool_out else None
    check(lib.cm_spatial_apply(_p(a2), _p(s), _p(fmap), _p(_contig(w7)), _p(gate), _p(out), _p(mp), n, c, h, w, st),
          "spatial_apply")
    if pool_out:
        return out, z, s, fmap, gate, mp
    return out, z, s, fmap, gate


def gates_bwd(dout, a2, s, z, pooled, gate, fmap, w1, w2, w7, dw1, dw2, dw7, defer_se_wgrad=False):
    """Backward of SE + spatial gate up to (but excluding) the GroupNorm; returns the maps cm_gn_silu_bwd_gated needs:
    dmap, (umax, cnt), dpool (plus (dsig, dz) when ``defer_se_wgrad``: the SE weight gradients are then left to
    gn_silu_bwd_gated(se=...)).  (umax, cnt) is the backward's own channel-maximum / tie-count pair."""
    n, c, h, w = a2.shape
    cr = w1.shape[0]
    dev = a2.device
    dgpre = torch.empty(n, h, w, device=dev, dtype=torch.float32)
    cnt = torch.empty(n, h, w, device=dev, dtype=torch.float32)
    umax = torch.empty(n, h, w, device=dev, dtype=torch.float32)
    dmap = torch.empty(n, 2, h, w, device=dev, dtype=torch.float32)
    ds = torch.empty(n, c, device=dev, dtype=torch.float32)
    dsig = torch.empty(n, c, device=dev, dtype=torch.float32)
    dz = torch.empty(n, cr, device=dev, dtype=torch.float32)
    dpool = torch.empty(n, c, device=dev, dtype=torch.float32)
    c7ws = torch.empty(int(lib.cm_conv7_bwd_scratch_elems(n, h)), device=dev, dtype=torch.float32)
    st = _stream()
    check(lib.cm_gate_bwd_reduce(_p(_contig(dout)), _p(a2), _p(s), _p(gate), _p(dgpre), _p(cnt), _p(umax), n, c, h * w,
                                 st), "gate_bwd_reduce")
    # dW7 partials are folded by the first workgroups of se_bwd_reduce (one launch less than folding in conv7_bwd)
    check(lib.cm_conv7_bwd(_p(dgpre), _p(fmap), _p(_contig(w7)), _p(dmap), None, _p(c7ws), n, h, w, st), "conv7_bwd")
    check(lib.cm_se_bwd_reduce(_p(dout), _p(a2), _p(s), _p(gate), _p(dmap), _p(umax), _p(cnt), _p(ds), n, c, h * w,
                               _p(c7ws), c7ws.numel() // 98, _p(dw7), st), "se_bwd_reduce")
    check(lib.cm_se_excite_bwd(_p(ds), _p(s), _p(z), _p(pooled), _p(_contig(w1)), _p(_contig(w2)), _p(dsig), _p(dz),
                               _p(dpool), None if defer_se_wgrad else _p(dw1), None if defer_se_wgrad else _p(dw2), n, c,
                               cr, st), "se_excite_bwd")
    if defer_se_wgrad:
        return dmap, (umax, cnt), dpool, (dsig, dz)
    return dmap, (umax, cnt), dpool


# ----------------------------------------------------------------------------------------------------- fused tail
BLOCK_TAIL = os.environ.get("CM_BLOCK_TAIL", "1") != "0"      # sample-resident ConvBlock tail (csrc/block_tail.hip)


BLOCK_TAIL_MIN_N = int(os.environ.get("CM_BLOCK_TAIL_MIN_N", "32"))    # fewer samples than this: keep the launch chain


def block_tail_supported(c, cr, h, w, n=None):
    if n is not None and n < BLOCK_TAIL_MIN_N:
        return False
    return BLOCK_TAIL and bool(lib.cm_block_tail_supported(c, cr, h, w))


def block_tail_fwd(y2, gamma, beta, w1, w2, w7, pool_out=False, parts=None):
    """GroupNorm+SiLU -> SE -> spatial gate (-> MaxPool2d(2)) of one ConvBlock in one launch (cm_block_tail_fwd).
    ``parts`` = (stack, k) from conv3x3_parts instead of y2.  Returns (y2, stats, pooled, z, s, fmap, gate, out, mp);
    the activation a2 is not materialised (gn_silu_apply(y2, gamma, beta, stats) reproduces it)."""
    if parts is not None:
        pt, k = parts
        _, n, c, h, w = pt.shape
        y2 = torch.empty(n, c, h, w, device=pt.device, dtype=torch.float32)
    else:
        n, c, h, w = _contig(y2).shape
    dev = y2.device
    cr = w1.shape[0]
    stats = torch.empty(n * GN_GROUPS * 2, device=dev, dtype=torch.float32)
    pooled = torch.empty(n, c, device=dev, dtype=torch.float32)
    z = torch.empty(n, cr, device=dev, dtype=torch.float32)
    s = torch.empty(n, c, device=dev, dtype=torch.float32)
    fmap = torch.empty(n, 2, h, w, device=dev, dtype=torch.float32)
    gate = torch.empty(n, h, w, device=dev, dtype=torch.float32)
    out = torch.empty(n, c, h, w, device=dev, dtype=torch.float32)
    mp = torch.empty(n, c, h // 2, w // 2, device=dev, dtype=torch.float32) if pool_out else None
    if parts is not None:
        check(lib.cm_block_tail_fwd(None, _p(pt), pt.stride(0), k, _p(y2), _p(gamma), _p(beta), _p(_contig(w1)),
                                    _p(_contig(w2)), _p(_contig(w7)), _p(stats), _p(pooled), _p(z), _p(s), _p(fmap),
                                    _p(gate), _p(out), _p(mp), n, c, cr, h, w, GN_EPS, _stream()), "block_tail_fwd")
    else:
        check(lib.cm_block_tail_fwd(_p(y2), None, 0, 0, None, _p(gamma), _p(beta), _p(_contig(w1)), _p(_contig(w2)),
                                    _p(_contig(w7)), _p(stats), _p(pooled), _p(z), _p(s), _p(fmap), _p(gate), _p(out),
                                    _p(mp), n, c, cr, h, w, GN_EPS, _stream()), "block_tail_fwd")
    return y2, stats, pooled, z, s, fmap, gate, out, mp


def block_tail_bwd(dout, y2, stats, gamma, beta, s, z, gate, fmap, w1, w2, w7, dw7):
    """Whole-sample reductions of the tail's backward in one launch (cm_block_tail_bwd): returns dmap, (umax, cnt),
    dpool, (dsig, dz) -- the tuple gates_bwd(..., defer_se_wgrad=True) returns."""
    n, c, h, w = y2.shape
    cr = w1.shape[0]
    dev = y2.device
    dmap = torch.empty(n, 2, h, w, device=dev, dtype=torch.float32)
    umax = torch.empty(n, h, w, device=dev, dtype=torch.float32)
    cnt = torch.empty(n, h, w, device=dev, dtype=torch.float32)
    dpool = torch.empty(n, c, device=dev, dtype=torch.float32)
    dsig = torch.empty(n, c, device=dev, dtype=torch.float32)
    dz = torch.empty(n, cr, device=dev, dtype=torch.float32)
    check(lib.cm_block_tail_bwd(_p(_contig(y2)), _p(stats), _p(gamma), _p(beta), _p(s), _p(z), _p(gate), _p(fmap),
                                _p(_contig(w1)), _p(_contig(w2)), _p(_contig(w7)), _p(_contig(dout)), _p(dmap),
                                _p(umax), _p(cnt), _p(dpool), _p(dsig), _p(dz), _p(dw7), n, c, cr, h, w, _stream()),
          "block_tail_bwd")
    return dmap, (umax, cnt), dpool, (dsig, dz)


# ----------------------------------------------------------------------------------------------------- pool / skip
def maxpool2_fwd(x):
    n, c, h, w = x.shape
    y = torch.empty(n, c, h // 2, w // 2, device=x.device, dtype=torch.float32)
    check(lib.cm_maxpool2_fwd(_p(_contig(x)), _p(y), n * c, h, w, _stream()), "maxpool")
    return y


def maxpool2_bwd(x, dy, dskip=None, t=1):
    n, c, h, w = x.shape
    dx = torch.empty_like(_contig(x))
    check(lib.cm_maxpool2_bwd(_p(x), _p(_contig(dy)), _p(dskip), 0 if dskip is None else dskip.stride(0), _p(dx), n, c,
                              h, w, t, 1.0 / t, _stream()), "maxpool_bwd")
    return dx


def time_mean(x, b, t):
    chw = x[0].numel()
    y = torch.empty((b,) + tuple(x.shape[1:]), device=x.device, dtype=torch.float32)
    check(lib.cm_time_mean(_p(_contig(x)), _p(y), b, t, chw, _stream()), "time_mean")
    return y


def channel_sum(x, out):
    n, c, h, w = x.shape
    check(lib.cm_channel_sum(_p(x), x.stride(0), _p(out), n, c, h * w, _stream()), "channel_sum")
    return out


# ----------------------------------------------------------------------------------------------------- convT
def convT2x2_fwd(x, w, b):
    n, ci, h, wd = x.shape
    co = w.shape[1]
    y = torch.empty(n, co, 2 * h, 2 * wd, device=x.device, dtype=torch.float32)
    check(lib.cm_convT2x2_fwd(_p(x), x.stride(0), _p(_contig(w)), _p(b), _p(y), y.stride(0), n, ci, co, h, wd,
                              _stream()), "convT_fwd")
    return y


def convT2x2_bwd(x, w, dy, dw, db):
    n, ci, h, wd = x.shape
    co = w.shape[1]
    dx = torch.empty(n, ci, h, wd, device=x.device, dtype=torch.float32)
    st = _stream()
    check(lib.cm_convT2x2_bwd_data(_p(dy), dy.stride(0), _p(_contig(w)), _p(dx), dx.stride(0), n, ci, co, h, wd, st),
          "convT_bwd_data")
    check(lib.cm_convT2x2_bwd_weight(_p(x), x.stride(0), _p(dy), dy.stride(0), _p(dw), _p(db), n, ci, co, h, wd, st),
          "convT_bwd_weight")
    return dx


# ----------------------------------------------------------------------------------------------------- lstm / head
def lstm_gates_fwd(gates, c_prev, c_out, h_out, parts=None):
    """parts = (stack [>= k, B, 4 ch, h, w], k) from conv3x3_parts: the recurrent projection as k partial sums."""
    b, ch4, h, w = gates.shape
    ch = ch4 // 4
    if parts is not None:
        pt, k = parts
        check(lib.cm_lstm_gates_fwd_parts(_p(gates), gates.stride(0), _p(pt), pt.stride(1), pt.stride(0), k, _p(c_prev),
                                          0 if c_prev is None else c_prev.stride(0), _p(c_out), c_out.stride(0),
                                          _p(h_out), h_out.stride(0), b, ch, h * w, _stream()), "lstm_fwd_parts")
        return
    check(lib.cm_lstm_gates_fwd(_p(gates), gates.stride(0), _p(c_prev), 0 if c_prev is None else c_prev.stride(0),
                                _p(c_out), c_out.stride(0), _p(h_out), h_out.stride(0), b, ch, h * w, _stream()),
          "lstm_fwd")


LSTM_STEP = os.environ.get("CM_LSTM_STEP", "1") != "0"        # fused ConvLSTM step (csrc/lstm_step.hip)


def lstm_step_supported(b, ch, h, w):
    return LSTM_STEP and bool(lib.cm_lstm_step_supported(b, ch, h, w))


def lstm_step_fwd(hprev, wph, winv, gates, c_prev, c_out, h_out):
    """One ConvLSTM step t >= 1 in one launch: gates (x-projection in, activations out) += conv3x3(hprev, W_h), then the
    gate nonlinearities and the state update (cm_lstm_step_fwd)."""
    b, ch4, h, w = gates.shape
    ch = ch4 // 4
    for t in (hprev, gates, c_prev, c_out, h_out):
        if t.stride(3) != 1 or t.stride(2) != w or t.stride(1) != h * w:
            raise RuntimeError("lstm_step_fwd needs dense HxW planes with channel stride H*W")
    check(lib.cm_lstm_step_fwd(_p(hprev), hprev.stride(0), _p(wph), _p(winv), _p(gates), gates.stride(0), _p(c_prev),
                               c_prev.stride(0), _p(c_out), c_out.stride(0), _p(h_out), h_out.stride(0), b, ch, h, w,
                               _stream()), "lstm_step_fwd")


# The backward step kernel (cm_lstm_step_bwd) is built and parity-tested but OFF by default: its reduction is 4x deeper than
# the forward's (K = 4 Ch * 9) for a quarter of the output rows, so each of its 64 workgroups streams 590 KB of weight
# fragments and runs 216 MFMAs per wave with nothing beside it on the CU -- 20.3 us against 17.3 us for the K-split
# partial-slices launch + gate kernel it would replace (tools/tail_bench.py; step: 8247 vs 8268 samples/s same box).
LSTM_STEP_BWD = os.environ.get("CM_LSTM_STEP_BWD", "0") != "0"


def lstm_step_bwd_supported(b, ch, h, w):
    return LSTM_STEP and LSTM_STEP_BWD and bool(lib.cm_lstm_step_bwd_supported(b, ch, h, w))


def lstm_step_bwd(dA_next, wpd, winv, gates, c_prev, c_cur, dh_ext, dc):
    """One BPTT step t < T-1 in one launch: recurrent data gradient of dA_next + gate backward of this step
    (cm_lstm_step_bwd); ``gates`` turns from activations into d(pre-activations), ``dc`` is updated in place."""
    b, ch4, h, w = gates.shape
    ch = ch4 // 4
    for t in (dA_next, gates, c_prev, c_cur, dh_ext):
        if t is not None and (t.stride(3) != 1 or t.stride(2) != w or t.stride(1) != h * w):
            raise RuntimeError("lstm_step_bwd needs dense HxW planes with channel stride H*W")
    check(lib.cm_lstm_step_bwd(_p(dA_next), dA_next.stride(0), _p(wpd), _p(winv), _p(gates), gates.stride(0), _p(c_prev),
                               0 if c_prev is None else c_prev.stride(0), _p(c_cur), c_cur.stride(0), _p(dh_ext),
                               0 if dh_ext is None else dh_ext.stride(0), _p(_contig(dc)), b, ch, h, w, _stream()),
          "lstm_step_bwd")


def lstm_gates_bwd(gates, c_prev, c_cur, dh_a, dh_b, dc, first):
    """dh_b: a tensor [B, ch, h, w], None, or (stack [>= k, B, ch, h, w], k) from conv3x3_parts."""
    b, ch4, h, w = gates.shape
    ch = ch4 // 4
    if isinstance(dh_b, tuple):
        pt, k = dh_b
        check(lib.cm_lstm_gates_bwd_parts(_p(gates), gates.stride(0), _p(c_prev), 0 if c_prev is None else c_prev.stride(0),
                                          _p(c_cur), c_cur.stride(0), _p(dh_a), 0 if dh_a is None else dh_a.stride(0),
                                          _p(pt), pt.stride(1), pt.stride(0), k, _p(_contig(dc)), int(first), b, ch, h * w,
                                          _stream()), "lstm_bwd_parts")
        return
    check(lib.cm_lstm_gates_bwd(_p(gates), gates.stride(0), _p(c_prev), 0 if c_prev is None else c_prev.stride(0),
                                _p(c_cur), c_cur.stride(0), _p(dh_a), 0 if dh_a is None else dh_a.stride(0), _p(dh_b),
                                0 if dh_b is None else dh_b.stride(0), _p(_contig(dc)), int(first), b, ch, h * w,
                                _stream()), "lstm_bwd")


def head_fwd(x, w, b):
    n, c, h, wd = x.shape
    oc = w.shape[0]
    pred = torch.empty(n, oc, h, wd, device=x.device, dtype=torch.float32)
    check(lib.cm_head_fwd(_p(x), x.stride(0), _p(_contig(w)), _p(b), _p(pred), n, c, oc, h * wd, _stream()), "head_fwd")
    return pred


def mse_loss(pred, y, want_grad=True):
    loss = torch.empty(1, device=pred.device, dtype=torch.float32)
    dpred = torch.empty_like(pred) if want_grad else None
    check(lib.cm_mse_loss(_p(_contig(pred)), _p(_contig(y)), _p(loss), _p(dpred), pred.numel(), _stream()), "mse")
    return loss, dpred


def head_bwd(dpred, x, w, dw, db):
    n, c, h, wd = x.shape
    oc = w.shape[0]
    dx = torch.empty(n, c, h, wd, device=x.device, dtype=torch.float32)
    check(lib.cm_head_bwd(_p(_contig(dpred)), _p(x), x.stride(0), _p(_contig(w)), _p(dx), dx.stride(0), _p(dw), _p(db),
                          n, c, oc, h * wd, _stream()), "head_bwd")
    return dx


def head_mse_bwd(x, w, b, y, loss, dw, db, pred_out=None):
    """Head + MSE + head backward in one launch (fused step): loss[0] += mse, dw/db += grads; returns d(x)."""
    n, c, h, wd = x.shape
    oc = w.shape[0]
    dx = torch.empty(n, c, h, wd, device=x.device, dtype=torch.float32)
    check(lib.cm_head_mse_bwd(_p(x), x.stride(0), _p(_contig(w)), _p(b), _p(_contig(y)), _p(pred_out), _p(loss), _p(dx),
                              dx.stride(0), _p(dw), _p(db), n, c, oc, h * wd, _stream()), "head_mse_bwd")
    return dx


def adam_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, grad_scale=1.0):
    check(lib.cm_adam_step(_p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, weight_decay, step,
                           grad_scale, _stream()), "adam")


# ----------------------------------------------------------------------------------------------------- cnn_transformer
def _drop_args(drop):
    """drop = None | (rng int32[2] device tensor, site id, p) -> (rng pointer, site, p) for the launchers."""
    if drop is None or drop[2] <= 0.0:
        return None, 0, 0.0
    rng, site, p = drop
    return _p_any(rng), int(site), float(p)


def gemm(a, b, m, n, k, trans_a=False, trans_b=False, bias=None, resid=None, res_rows=0, mask=None, relu=False,
         out=None, ksplit=1, lda=None, ldb=None, drop=None, mask_scale=1.0, tile=0):
    """out[m, n] = relu?(op(a) op(b) + bias) + resid (masked) on the fp16x3 GEMM kernel (cm_gemm_h3).  a / b are 2-D
    row-major tensors: a is [m, k] (or [k, m] when trans_a), b is [n, k] -- an nn.Linear weight, out = a b^T -- (or
    [k, n] when trans_b).  ksplit > 1 ACCUMULATES into ``out`` (which the caller zeroed).  ``drop`` = (rng, site, p):
    counter-based dropout after bias / ReLU, before the residual; ``mask_scale`` multiplies what ``mask`` keeps."""
    if out is None:
        out = torch.empty(m, n, device=a.device, dtype=torch.float32)
    lda = a.stride(0) if lda is None else lda
    ldb = b.stride(0) if ldb is None else ldb
    check(lib.cm_gemm_h3(_p(a), lda, int(trans_a), _p(b), ldb, int(trans_b), _p(out), out.stride(0), _p(bias), _p(resid),
                         0 if resid is None else resid.stride(0), res_rows, _p(mask),
                         0 if mask is None else mask.stride(0), float(mask_scale), int(relu), *_drop_args(drop), m, n, k,
                         ksplit, int(tile), _stream()), "gemm_h3")
    return out


class PackedWeights:
    """fp16x3 B operands of a set of weight matrices, split once per step (cm_gemm_h3_pack_b_batch): ``add(key, w, n, k,
    trans)`` registers B[n][k] = w[n, k] (trans False) or w[k, n] (trans True) of a 2-D view ``w``; ``pack()`` (3 launches)
    refreshes all of them from the current values; ``gemm_pb`` consumes ``self[key]``."""

    def __init__(self, device):
        self.device = device
        self.jobs = []
        self.index = {}
        self.table = None

    def add(self, key, w, n, k, trans=False):
        if w.dim() != 2 or w.stride(1) != 1:
            raise RuntimeError("PackedWeights.add needs a 2-D view with unit column stride")
        self.index[key] = len(self.jobs)
        self.jobs.append((w, int(n), int(k), bool(trans)))
        self.table = None

    def _build(self):
        nb = [((n + 63) // 64) * ((k + 63) // 64) * 2 for _, n, k, _ in self.jobs]
        self.out = [torch.empty(b * 512 * 4, device=self.device, dtype=torch.float32) for b in nb]
        self.be = torch.zeros(len(self.jobs), device=self.device, dtype=torch.int32)
        self.slot = list(range(len(self.jobs)))
        rows, first, owner = [], 0, {}
        for j, (w, n, k, trans) in enumerate(self.jobs):
            tkey = (w.data_ptr(), w.stride(0), tuple(w.shape))
            o = owner.setdefault(tkey, j)          # the first job of a tensor measures its magnitude; later ones reuse it
            self.slot[j] = o
            rows.append([w.data_ptr(), self.out[j].data_ptr(), self.be.data_ptr() + 4 * o, n, k, w.stride(0),
                         int(trans) | (2 if o != j else 0), first])
            first += nb[j]
        self.total = first
        self.table = torch.tensor(rows, dtype=torch.int64).to(self.device)

    def pack(self):
        if self.table is None:
            self._build()
        check(lib.cm_zero(self.be.data_ptr(), self.be.numel() * 4, _stream()), "zero")
        check(lib.cm_gemm_h3_pack_b_batch(_p_any(self.table), len(self.jobs), self.total, 8, _stream()), "gemm_pack_b")

    def __getitem__(self, key):
        if self.table is None:
            self._build()
        j = self.index[key]
        o = self.slot[j]
        return self.out[j], self.be[o:o + 1], self.jobs[j][1], self.jobs[j][2]


def gemm_pb(a, packed, m, bias=None, resid=None, res_rows=0, mask=None, relu=False, out=None, drop=None, mask_scale=1.0):
    """out[m, n] = epilogue(a [m, k] @ B^T) with ``packed`` = PackedWeights[key] (the B operand split once per step)."""
    bp, be, n, k = packed
    if out is None:
        out = torch.empty(m, n, device=a.device, dtype=torch.float32)
    check(lib.cm_gemm_h3_pb(_p(a), a.stride(0), _p(bp), _p_any(be), _p(out), out.stride(0), _p(bias), _p(resid),
                            0 if resid is None else resid.stride(0), res_rows, _p(mask),
                            0 if mask is None else mask.stride(0), float(mask_scale), int(relu), *_drop_args(drop), m, n, k,
                            _stream()), "gemm_h3_pb")
    return out


def gemm_wgrad(dy, x, dw, n_out, k_in, tokens, ksplit, dbias=None, tile=0):
    """dw [n_out, k_in] += dy^T x (dy [tokens, n_out], x [tokens, >= k_in]) and, with ``dbias``, dbias [n_out] += the column
    sums of dy in the same launch (cm_gemm_h3_wgrad).  ksplit <= 0: the autotuner picks the reduction split."""
    if ksplit <= 0:
        # reduction split chosen by timing (the best count depends on how many output tiles there are: 768 x 256 x 6912
        # runs 36 us at 16 splits and 42 us at 27, 256 x 256 x 6912 the other way round)
        def launch(ks, _scratch=[None]):
            if _scratch[0] is None:
                _scratch[0] = torch.zeros_like(dw)
            return lib.cm_gemm_h3_wgrad(_p(dy), dy.stride(0), _p(x), x.stride(0), _p(_scratch[0]), _scratch[0].stride(0), None,
                                        n_out, k_in, tokens, ks, int(tile), _stream())
        cands = [k for k in (4, 8, 12, 16, 20, 24, 32, 48, 64) if k * 64 <= tokens] or [1]
        ksplit = _pick(("gemm_wgrad", n_out, k_in, tokens, int(tile)), cands, launch, max(1, min(16, tokens // 256)))
    check(lib.cm_gemm_h3_wgrad(_p(dy), dy.stride(0), _p(x), x.stride(0), _p(dw), dw.stride(0), _p(dbias), n_out, k_in,
                               tokens, ksplit, int(tile), _stream()), "gemm_h3_wgrad")
    return dw


def layernorm_fwd(x, resid, gamma, beta, eps=1e-5):
    m, e = x.shape
    s = torch.empty_like(x)
    y = torch.empty_like(x)
    stats = torch.empty(m, 2, device=x.device, dtype=torch.float32)
    check(lib.cm_layernorm_fwd(_p(_contig(x)), _p(resid), _p(gamma), _p(beta), _p(s), _p(y), _p(stats), m, e, eps,
                               _stream()), "layernorm_fwd")
    return y, s, stats


def layernorm_bwd(s, stats, gamma, dy, dgamma, dbeta, drop=None, dbias=None):
    """ds = gradient wrt the LayerNorm's input.  With ``dbias`` (and optionally ``drop`` = (rng, site, p)) the launch also
    produces the gradient wrt the sublayer output that was added through x + dropout(sublayer(x)) -- returned as a second
    tensor (ds itself when there is no dropout) -- and accumulates its column sums into ``dbias``
    (cm_layernorm_bwd_sublayer: one launch instead of three)."""
    m, e = s.shape
    ds = torch.empty_like(s)
    if dbias is None and drop is None:
        check(lib.cm_layernorm_bwd(_p(s), _p(stats), _p(gamma), _p(_contig(dy)), _p(ds), _p(dgamma), _p(dbeta), m, e,
                                   _stream()), "layernorm_bwd")
        return ds
    rng, site, p = _drop_args(drop)
    dd = torch.empty_like(s) if p > 0.0 else None
    check(lib.cm_layernorm_bwd_sublayer(_p(s), _p(stats), _p(gamma), _p(_contig(dy)), _p(ds), _p(dgamma), _p(dbeta),
                                        _p(dd), _p(dbias), rng, site, p, m, e, _stream()), "layernorm_bwd_sublayer")
    return ds, (dd if dd is not None else ds)


ATTENTION_MFMA = os.environ.get("CM_ATTENTION", "mfma") != "valu"     # CM_ATTENTION=valu: the fp32 VALU kernels everywhere


def attention_fwd(qkv, b, s, e, h, drop=None):
    """softmax(Q K^T / sqrt(d)) V per (sample, head) on qkv [b*s, 3e].  Returns (saved, o): ``saved`` is what
    attention_bwd needs -- the row statistics [b, h, s, 2] on the matrix-core path (head_dim 32, s <= 224: the
    probabilities are recomputed), the probabilities [b, h, s, s] on the fp32 VALU path (other head sizes)."""
    qkv = _contig(qkv)
    o = torch.empty(b * s, e, device=qkv.device, dtype=torch.float32)
    if ATTENTION_MFMA and e % h == 0 and e // h == 32 and s <= 224 and (drop is None or drop[2] <= 0.75):
        stats = torch.empty(b, h, s, 2, device=qkv.device, dtype=torch.float32)
        check(lib.cm_attention_mfma_fwd(_p(qkv), _p(stats), _p(o), *_drop_args(drop), b, s, e, h, _stream()),
              "attention_mfma_fwd")
        return stats, o
    p = torch.empty(b, h, s, s, device=qkv.device, dtype=torch.float32)
    check(lib.cm_attention_fwd(_p(qkv), _p(p), _p(o), *_drop_args(drop), b, s, e, h, _stream()), "attention_fwd")
    return p, o


def attention_bwd(qkv, saved, d_o, b, s, e, h, drop=None, o=None):
    """``o``: the forward's output (needed on the matrix-core path: rowsum(dO o O) replaces a sweep over the keys)."""
    dqkv = torch.empty_like(qkv)
    if saved.shape[-1] == 2 and saved.shape[-1] != s:          # row statistics: the matrix-core path
        if o is None:
            raise RuntimeError("attention_bwd on the matrix-core path needs the forward's output `o`")
        dsum = torch.empty(b, h, s, device=qkv.device, dtype=torch.float32)
        check(lib.cm_attention_mfma_bwd(_p(qkv), _p(saved), _p(_contig(o)), _p(_contig(d_o)), _p(dsum), _p(dqkv),
                                        *_drop_args(drop), b, s, e, h, _stream()), "attention_mfma_bwd")
        return dqkv
    scratch = torch.empty_like(saved)
    check(lib.cm_attention_bwd(_p(qkv), _p(saved), _p(_contig(d_o)), _p(scratch), _p(dqkv), *_drop_args(drop), b, s, e, h,
                               _stream()), "attention_bwd")
    return dqkv


def dropout(x, drop):
    """x * mask / (1 - p) with the counter-based mask of drop = (rng, site, p) (element index = flat index of x)."""
    x = _contig(x)
    out = torch.empty_like(x)
    rng, site, p = drop
    check(lib.cm_dropout(_p(x), _p(out), x.numel(), _p_any(rng), int(site), float(p), _stream()), "dropout")
    return out


def rng_advance(rng):
    check(lib.cm_rng_advance(_p_any(rng), _stream()), "rng_advance")


def im2col_s2(x, b, cin, h, w, ldc, tokens_in):
    col = torch.empty(b * (h // 2) * (w // 2), ldc, device=x.device, dtype=torch.float32)
    check(lib.cm_im2col_s2(_p(_contig(x)), _p(col), b, cin, h, w, ldc, int(tokens_in), _stream()), "im2col_s2")
    return col


def col2im_s2(dcol, b, cin, h, w):
    dx = torch.empty(b * h * w, cin, device=dcol.device, dtype=torch.float32)
    check(lib.cm_col2im_s2(_p(dcol), _p(dx), b, cin, h, w, dcol.stride(0), _stream()), "col2im_s2")
    return dx


def transpose_batched(x, batch, rows, cols):
    out = torch.empty(batch, cols, rows, device=x.device, dtype=torch.float32)
    check(lib.cm_transpose_batched(_p(_contig(x)), _p(out), batch, rows, cols, _stream()), "transpose")
    return out


def relu_(x):
    check(lib.cm_relu(_p(x), x.numel(), _stream()), "relu")
    return x


def relu_mask_(g, y):
    check(lib.cm_relu_mask(_p(g), _p(y), g.numel(), _stream()), "relu_mask")
    return g


def rowgroup_sum(x, out, period=1):
    """out[r % period, c] += sum_r x[r, c]  (x 2-D contiguous)."""
    rows, cols = x.shape
    check(lib.cm_rowgroup_sum(_p(_contig(x)), _p(out), rows, cols, period, _stream()), "rowgroup_sum")
    return out
