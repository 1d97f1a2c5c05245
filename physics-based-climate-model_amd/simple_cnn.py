"""``SimpleCNN`` on the MI355X HIP path (BASELINE.json configs[0]).

Mirrors reference src/models.py:44-123 -- stem conv-BN-ReLU, ``depth`` ResidualBlocks (conv-BN-ReLU-conv-BN + identity or
1x1 conv-BN skip, add, ReLU; channels double until the last block), Dropout2d, conv-BN-ReLU, 1x1 conv -- with the
reference's exact ``state_dict`` (parameters AND the BatchNorm buffers live in the same stock containers, which are never
called), driven by the same autograd bridge / fused trainer as the hot-path model.

Schedule:
  3x3 convolutions (bias=True)      cm_conv3x3_h3 forward / data gradient, cm_wgrad3x3_h3 weight gradient (fp16x3 matrix
                                    cores), cm_channel_sum bias gradient
  1x1 skip convolutions             the same kernels on the weight embedded at the centre tap of a zero 3x3 kernel
                                    (cm_embed_center_tap / cm_extract_center_tap)
  BatchNorm2d (+ residual + ReLU)   cm_bn_fwd / cm_bn_bwd: train-mode batch statistics per rank (no SyncBN, as the
                                    reference under DDP), running buffers updated in place, eval mode on the buffers
  Dropout2d                         per-(sample, channel) multipliers from the counter-based generator (cm_dropout on a
                                    vector of ones, csrc/common.h) applied by cm_scale_planes; in distribution, not bit
                                    for bit, the reference's masks -- tests impose the reference's own mask
  final 1x1 convolution             the hot path's head kernels (cm_head_fwd / cm_head_bwd, fused with the MSE in the
                                    trainer)
BatchNorm couples the samples of a batch, so this model must not be run as two micro-batches (``batch_coupled``).
"""
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import ops
from ._lib import check, lib
from .model import _HipModule, _Holder

Tensor = torch.Tensor
BN_MOMENTUM = 0.1     # nn.BatchNorm2d defaults
BN_EPS = 1e-5


def _st():
    return torch.cuda.current_stream().cuda_stream


class _ConvSet:
    """Packed fp16x3 operands (forward + data-gradient form) of a list of 3x3 weights and the tap-major staging tensors of
    their weight gradients: one batched pack launch pair per step.  Valid while the weight storage stays where it is."""

    def __init__(self, units: "List[tuple]", need_input_grad: bool):
        # units: (key, weight [co, ci, 3, 3])
        dev = units[0][1].device
        self.sig = tuple(w.data_ptr() for _, w in units) + (need_input_grad,)
        jobs = []
        for i, (key, w) in enumerate(units):
            jobs.append((key + "/f", w, 0))
            if i > 0 or need_input_grad:          # the first conv's data gradient is d(input)
                jobs.append((key + "/d", w, 1))
        sizes = [lib.cm_conv3x3_h3_packed_bytes(w.shape[0] if dg else w.shape[1], w.shape[1] if dg else w.shape[0]) // 4
                 for _, w, dg in jobs]
        self.arena = torch.empty(sum(sizes), device=dev, dtype=torch.float32)
        self.pkh: Dict[str, Tensor] = {}
        rec, o, blk = [], 0, 0
        for (key, w, dg), sz in zip(jobs, sizes):
            self.pkh[key] = self.arena[o:o + sz]
            rec.append([w.data_ptr(), self.pkh[key].data_ptr(), w.shape[0], w.shape[1], 0, w.shape[1], dg, blk])
            blk += max(1, min(512, (sz // 8 + 255) // 256))
            o += sz
        rec.append([0, 0, 0, 0, 0, 0, 0, blk])
        self.table = torch.tensor(rec, dtype=torch.int64).to(dev)
        self.n, self.blocks = len(jobs), blk
        self.scratch = torch.zeros(len(jobs) + blk, device=dev, dtype=torch.float32)
        self.winv = {j[0]: self.scratch[i:i + 1] for i, j in enumerate(jobs)}
        self.stage = {key: torch.empty(w.shape[0], 9, w.shape[1], device=dev, dtype=torch.float32) for key, w in units}
        self._stage_flat = None

    def pack(self):
        check(lib.cm_pack_conv3x3_h3_batch(self.table.data_ptr(), self.n, self.blocks, self.scratch.data_ptr(), _st()),
              "pack_h3_batch")

    def conv(self, key, x, cout, dgrad=False, bias=None, resid=None):
        k = key + ("/d" if dgrad else "/f")
        return ops.conv3x3(x, None, cout, bias=bias, resid=resid, wph=self.pkh[k], winv=self.winv[k])

    def zero_staging(self):
        for t in self.stage.values():
            check(lib.cm_zero(t.data_ptr(), t.numel() * 4, _st()), "zero")


class _Saved:
    pass


def _bn_fwd(x, p, bufs, q, training, relu, resid=None):
    n, c, h, w = x.shape
    y = torch.empty_like(x)
    save = torch.empty(c, 2, device=x.device, dtype=torch.float32)
    check(lib.cm_bn_fwd(x.data_ptr(), p[q + ".weight"].data_ptr(), p[q + ".bias"].data_ptr(),
                        None if resid is None else resid.data_ptr(), y.data_ptr(), save.data_ptr(),
                        bufs[q + ".running_mean"].data_ptr(), bufs[q + ".running_var"].data_ptr(), BN_MOMENTUM, BN_EPS,
                        int(relu), int(training), n, c, h * w, _st()), "bn_fwd")
    if training:
        bufs[q + ".num_batches_tracked"].add_(1)          # (bookkeeping only: momentum is not None)
    return y, save


def _bn_bwd(x, y, dy, p, g, q, save, training, relu, want_dres=False):
    n, c, h, w = x.shape
    dx = torch.empty_like(x)
    dres = torch.empty_like(x) if want_dres else None
    check(lib.cm_bn_bwd(x.data_ptr(), None if y is None else y.data_ptr(), dy.data_ptr(), p[q + ".weight"].data_ptr(),
                        save.data_ptr(), dx.data_ptr(), None if dres is None else dres.data_ptr(),
                        g[q + ".weight"].data_ptr(), g[q + ".bias"].data_ptr(), int(relu), int(training), n, c, h * w,
                        _st()), "bn_bwd")
    return dx, dres


class ResidualBlock(_Holder):
    """Parameter container with the reference's attribute names (src/models.py:44-59); never called."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1):
        super().__init__()
        pad = kernel_size // 2
        self.conv1 = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=pad)
        self.bn1 = nn.BatchNorm2d(out_channels)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(out_channels, out_channels, kernel_size, padding=pad)
        self.bn2 = nn.BatchNorm2d(out_channels)
        self.skip = nn.Sequential()
        if stride != 1 or in_channels != out_channels:
            self.skip = nn.Sequential(nn.Conv2d(in_channels, out_channels, kernel_size=1, stride=stride),
                                      nn.BatchNorm2d(out_channels))


class SimpleCNN(_HipModule):
    """Drop-in for the reference's ``SimpleCNN`` (same constructor, same state_dict) on the HIP path.
    x [B, n_input_channels, H, W] -> [B, n_output_channels, H, W]."""

    _head_param_names = ("final.3.weight", "final.3.bias")
    batch_coupled = True           # BatchNorm statistics run over the batch: no micro-batch split

    def __init__(self, n_input_channels, n_output_channels, kernel_size=3, init_dim=64, depth=4, dropout_rate=0.2):
        super().__init__()
        if kernel_size != 3:
            raise ValueError("the HIP SimpleCNN serves kernel_size=3 (the reference's configs/model/SimpleCNN.yaml)")
        pad = kernel_size // 2
        # registration order == reference __init__ (src/models.py:91-115)
        self.initial = nn.Sequential(nn.Conv2d(n_input_channels, init_dim, kernel_size=kernel_size, padding=pad),
                                     nn.BatchNorm2d(init_dim), nn.ReLU(inplace=True))
        self.res_blocks = nn.ModuleList()
        dim = init_dim
        for i in range(depth):
            last = i == depth - 1
            self.res_blocks.append(ResidualBlock(dim, dim if last else dim * 2))
            if not last:
                dim *= 2
        self.dropout = nn.Dropout2d(dropout_rate)
        self.final = nn.Sequential(nn.Conv2d(dim, dim // 2, kernel_size=kernel_size, padding=pad),
                                   nn.BatchNorm2d(dim // 2), nn.ReLU(inplace=True),
                                   nn.Conv2d(dim // 2, n_output_channels, kernel_size=1))
        self.dropout_p = float(dropout_rate)
        self.depth = depth
        self._finish_init()

    # ------------------------------------------------------------------ dropout state (as CNNTransformer)
    def _rng_state(self, device) -> Tensor:
        rng = self.__dict__.get("_rng")
        if rng is None or rng.device != device:
            seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if rng is None else int(rng[0].item())
            rng = torch.tensor([seed, 0], dtype=torch.int32, device=device)
            self.__dict__["_rng"] = rng
        return rng

    def reseed_dropout(self, seed: int, counter: int = 0) -> None:
        dev = next(self.parameters()).device
        self.__dict__["_rng"] = torch.tensor([int(seed) & 0x7fffffff, int(counter)], dtype=torch.int32, device=dev)

    def impose_dropout_mask(self, mult: Optional[Tensor]) -> None:
        """Test aid: use these per-(sample, channel) multipliers [B, C] (0 or 1/(1-p)) instead of drawing them."""
        self.__dict__["_forced_mask"] = mult

    def dropout_multipliers(self, b: int, c: int, device) -> Tensor:
        """This step's Dropout2d multipliers [b, c] from the counter-based generator (advances the device counter)."""
        forced = self.__dict__.get("_forced_mask")
        if forced is not None:
            return forced.to(device=device, dtype=torch.float32).contiguous()
        rng = self._rng_state(device)
        ops.rng_advance(rng)
        return ops.dropout(torch.ones(b * c, device=device, dtype=torch.float32), (rng, 0, self.dropout_p)).view(b, c)

    def _state_snapshot(self):
        """Forward-side state a throw-away pass must not leave changed (the trainer's warm-up before graph capture)."""
        rng = self.__dict__.get("_rng")
        return {k: v.clone() for k, v in self.named_buffers()}, None if rng is None else rng.clone()

    def _state_restore(self, state) -> None:
        bufs, rng = state
        with torch.no_grad():
            for k, v in self.named_buffers():
                v.copy_(bufs[k])
        if rng is not None:
            self.__dict__["_rng"].copy_(rng)
        elif self.__dict__.get("_rng") is not None:
            self.__dict__["_rng"][1] = 0          # the seed was drawn during the pass: keep it, rewind the counter

    # ------------------------------------------------------------------ engine
    def _bn_buffers(self) -> Dict[str, Tensor]:
        return dict(self.named_buffers())

    def _conv_units(self, p):
        """(key, 3x3 weight) of every convolution that runs on the 3x3 kernels; 1x1 skips through their embeddings."""
        emb = self.__dict__.setdefault("_emb", {})
        units = [("initial.0", p["initial.0.weight"])]
        for i in range(self.depth):
            q = f"res_blocks.{i}."
            units.append((q + "conv1", p[q + "conv1.weight"]))
            units.append((q + "conv2", p[q + "conv2.weight"]))
            if q + "skip.0.weight" in p:
                w1 = p[q + "skip.0.weight"]
                key = (q, w1.data_ptr())
                if key not in emb:
                    emb[key] = torch.empty(w1.shape[0], w1.shape[1], 3, 3, device=w1.device, dtype=torch.float32)
                units.append((q + "skip.0", emb[key]))
        units.append(("final.0", p["final.0.weight"]))
        return units

    def _conv_set(self, p, need_dx) -> _ConvSet:
        units = self._conv_units(p)
        sig = tuple(w.data_ptr() for _, w in units) + (bool(need_dx),)
        cs = self.__dict__.get("_cs")
        if cs is None or cs.sig != sig:
            cs = _ConvSet(units, bool(need_dx))
            self.__dict__["_cs"] = cs
        # refresh the centre-tap embeddings of the 1x1 skip weights, then pack everything (two launches)
        for key, w3 in units:
            if key.endswith("skip.0"):
                w1 = p[key + ".weight"]
                check(lib.cm_embed_center_tap(w1.data_ptr(), w3.data_ptr(), w1.shape[0], w1.shape[1], _st()), "embed")
        cs.pack()
        return cs

    def _engine_forward(self, p, pk, x, save=True, head=True):
        if x.dim() != 4:
            raise RuntimeError("expected x of shape [B, C, H, W]")
        if x.shape[1] != p["initial.0.weight"].shape[1]:
            raise RuntimeError(f"channel mismatch: input has {x.shape[1]} channels, the stem expects "
                               f"{p['initial.0.weight'].shape[1]}")
        x = x.contiguous()
        training = self.training
        bufs = self._bn_buffers()
        cs = self._conv_set(p, need_dx=True)
        sv = _Saved() if save else None
        y0 = cs.conv("initial.0", x, p["initial.0.weight"].shape[0], bias=p["initial.0.bias"])
        a, st0 = _bn_fwd(y0, p, bufs, "initial.1", training, relu=True)
        blocks = []
        for i in range(self.depth):
            q = f"res_blocks.{i}."
            co = p[q + "conv1.weight"].shape[0]
            y1 = cs.conv(q + "conv1", a, co, bias=p[q + "conv1.bias"])
            a1, s1 = _bn_fwd(y1, p, bufs, q + "bn1", training, relu=True)
            y2 = cs.conv(q + "conv2", a1, co, bias=p[q + "conv2.bias"])
            if q + "skip.0.weight" in p:
                ys = cs.conv(q + "skip.0", a, co, bias=p[q + "skip.0.bias"])
                r, ss = _bn_fwd(ys, p, bufs, q + "skip.1", training, relu=False)
            else:
                ys, r, ss = None, a, None
            out, s2 = _bn_fwd(y2, p, bufs, q + "bn2", training, relu=True, resid=r)
            blocks.append((a, y1, a1, s1, y2, s2, ys, ss, out))
            a = out
        mult = None
        d = a
        if training and self.dropout_p > 0.0:
            mult = self.dropout_multipliers(a.shape[0], a.shape[1], a.device)
            d = torch.empty_like(a)
            check(lib.cm_scale_planes(a.data_ptr(), mult.data_ptr(), d.data_ptr(), a.shape[0] * a.shape[1],
                                      a.shape[2] * a.shape[3], _st()), "scale_planes")
        cf = p["final.0.weight"].shape[0]
        yf = cs.conv("final.0", d, cf, bias=p["final.0.bias"])
        af, sf = _bn_fwd(yf, p, bufs, "final.1", training, relu=True)
        pred = ops.head_fwd(af, p["final.3.weight"], p["final.3.bias"]) if head else None
        if save:
            sv.x, sv.y0, sv.a0, sv.st0, sv.blocks, sv.mult, sv.d, sv.yf, sv.af, sv.sf = x, y0, blocks[0][0], st0, blocks, \
                mult, d, yf, af, sf
            sv.d1 = af                       # input of the 1x1 head: what the fused trainer's head+MSE launch reads
            sv.cs, sv.training = cs, training
        return pred, sv

    def _conv_bwd(self, cs, key, pkey, p, g, x, dy, need_dx=True, resid=None):
        """Bias, weight and data gradient of one 3x3 convolution (the 1x1 skip through its embedding)."""
        ops.channel_sum(dy, g[pkey + ".bias"])
        ops.wgrad3x3(x, dy, cs.stage[key])
        return cs.conv(key, dy, x.shape[1], dgrad=True, resid=resid) if need_dx else None

    def _engine_backward(self, p, pk, g, sv, dpred, need_dx=False, dd1=None):
        cs, training = sv.cs, sv.training
        cs.zero_staging()
        if dd1 is None:
            dd1 = ops.head_bwd(dpred, sv.af, p["final.3.weight"], g["final.3.weight"], g["final.3.bias"])
        dyf, _ = _bn_bwd(sv.yf, sv.af, dd1, p, g, "final.1", sv.sf, training, relu=True)
        dd = self._conv_bwd(cs, "final.0", "final.0", p, g, sv.d, dyf)
        if sv.mult is not None:
            da = torch.empty_like(dd)
            check(lib.cm_scale_planes(dd.data_ptr(), sv.mult.data_ptr(), da.data_ptr(), dd.shape[0] * dd.shape[1],
                                      dd.shape[2] * dd.shape[3], _st()), "scale_planes")
        else:
            da = dd
        for i in range(self.depth - 1, -1, -1):
            q = f"res_blocks.{i}."
            a_in, y1, a1, s1, y2, s2, ys, ss, out = sv.blocks[i]
            dy2, dres = _bn_bwd(y2, out, da, p, g, q + "bn2", s2, training, relu=True, want_dres=True)
            da1 = self._conv_bwd(cs, q + "conv2", q + "conv2", p, g, a1, dy2)
            dy1, _ = _bn_bwd(y1, a1, da1, p, g, q + "bn1", s1, training, relu=True)
            if ys is not None:
                dys, _ = _bn_bwd(ys, None, dres, p, g, q + "skip.1", ss, training, relu=False)
                dskip = self._conv_bwd(cs, q + "skip.0", q + "skip.0", p, g, a_in, dys)
            else:
                dskip = dres
            da = self._conv_bwd(cs, q + "conv1", q + "conv1", p, g, a_in, dy1, resid=dskip)
        dy0, _ = _bn_bwd(sv.y0, sv.a0, da, p, g, "initial.1", sv.st0, training, relu=True)
        dx = self._conv_bwd(cs, "initial.0", "initial.0", p, g, sv.x, dy0, need_dx=need_dx)
        # staged weight gradients -> parameter layout
        for key, stage in cs.stage.items():
            co, _, ci = stage.shape
            if key.endswith("skip.0"):
                check(lib.cm_extract_center_tap(stage.data_ptr(), g[key + ".weight"].data_ptr(), co, ci, _st()), "extract")
            else:
                check(lib.cm_wgrad3x3_unpack(stage.data_ptr(), g[key + ".weight"].data_ptr(), co, ci, 1.0, _st()), "unpack")
        return dx
