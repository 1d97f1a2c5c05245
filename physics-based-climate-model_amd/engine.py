"""Hot-path engine: forward and backward of the unet_convlstm_attention model as a sequence of HIP launches.

This is the host-side schedule that replaces ``AttUNetConvLSTM.forward`` (reference src/unet_convlstm_attention.py:60-104)
and the autograd graph behind ``loss.backward()`` (main_final.py:556-561).  It only orders ``cm_*`` launches on the
current HIP stream and keeps the tensors the backward needs; there is no math in Python and no CPU fallback.  Because
every launch is capture-safe, a whole training step can be recorded into one hipGraph (see ``trainer.GraphedStep``).

Layout decisions:
  * the frame loop of the encoder is folded into the batch: encoder tensors are [B*T, C, H, W], sample n = b*T + t
    (x_seq [B,T,C,H,W] is viewed, not copied; GroupNorm / SE / spatial gate are per-sample so results are unchanged);
  * the ConvLSTM input projection W_x * s4_t has no recurrence and runs once for all T; only W_h * h_{t-1} is
    sequential.  Gate buffers are [B,T,4*Ch,h,w]; step t works on the strided slice [:, t] in place;
  * torch.cat([up, skip]) is never materialised: the conv kernels take two input pointers.
"""
import os
from typing import Dict, Optional

import torch

from . import ops, profiler

Tensor = torch.Tensor
Params = Dict[str, Tensor]


KEEP_ACTIVATION2 = False      # debugging / tests (HotPathTrainer.keep_saved sets it): see _BlockCtx.activation2


class _BlockCtx:
    """Tensors one ConvBlock keeps for its backward.  ``a2`` (the activation entering the SE block) is None when the
    sample-resident tail ran: it is never stored there and ``activation2()`` recomputes it bit-exactly."""
    __slots__ = ("x0", "x1", "y1", "a1", "st1", "y2", "a2", "st2", "pooled", "z", "s", "fmap", "gate", "out", "be",
                 "gn2", "a2_dbg")

    def activation2(self):
        """SiLU(GroupNorm(y2)): the stored tensor, the copy the forward made under KEEP_ACTIVATION2, or its recomputation
        from the stored statistics (only valid while the GroupNorm parameters are the forward's: a fused trainer step
        has already run Adam on them when it returns -- tests that look at a2 afterwards set KEEP_ACTIVATION2)."""
        if self.a2 is not None:
            return self.a2
        if self.a2_dbg is not None:
            return self.a2_dbg
        return ops.gn_silu_apply(self.y2, self.gn2[0], self.gn2[1], self.st2)


def _conv_jobs(p: Params, need_input_grad: bool):
    """(key, weight name, c_off, cin, dgrad) for every MFMA operand the step needs."""
    jobs = []
    for name, w in p.items():
        if name.endswith("body.0.weight") or name.endswith("body.3.weight"):
            jobs.append((name + "/f", name, 0, w.shape[1], 0))
            if name != "enc1.body.0.weight" or need_input_grad:
                jobs.append((name + "/d", name, 0, w.shape[1], 1))
    wl = p.get("convlstm.cell.conv.weight")
    if wl is not None:
        ch = wl.shape[0] // 4
        cx = wl.shape[1] - ch
        for key, off, cin, dg in (("lstm.x/f", 0, cx, 0), ("lstm.h/f", cx, ch, 0), ("lstm.x/d", 0, cx, 1),
                                  ("lstm.h/d", cx, ch, 1)):
            jobs.append((key, "convlstm.cell.conv.weight", off, cin, dg))
    return jobs


# Matrix-core numerics of the 3x3 convolutions (forward, data gradient, weight gradient), all fp32-equivalent:
#   "fp16x3" (default): two fp16 pieces per operand, three products, exact power-of-two scaling (csrc/split_f16.h);
#   "bf16x6": three bf16 pieces, six products (csrc/split_bf16.h);  "fp32": the exact fp32-MFMA kernels only.
NUMERICS = os.environ.get("CM_CONV_NUMERICS", "fp16x3")
if NUMERICS not in ("fp16x3", "bf16x6", "fp32"):
    raise RuntimeError(f"CM_CONV_NUMERICS={NUMERICS!r}: expected fp16x3, bf16x6 or fp32")
USE_SPLIT = NUMERICS == "bf16x6"
USE_H3 = NUMERICS == "fp16x3"


# The ConvLSTM recurrence is a serial chain of small launches (B samples at H/8 x W/8): its h-projection and the
# data gradient of it run as "partial slices" launches (reduction shares stored, not added with atomics) whose slices the
# pointwise gate kernels add while they read their other operands.  CM_LSTM_PARTS=0 restores the atomic form.
PARTIAL_SLICES = os.environ.get("CM_LSTM_PARTS", "1") != "0"
CONV_PARTS = os.environ.get("CM_CONV_PARTS", "1") != "0"      # the same form for the small forward convs of a ConvBlock


class _Packs:
    """Both operand forms of the packed 3x3 weights; ``conv(key, ...)`` lets the tuner choose the kernel family."""

    def __init__(self, pk, pks, uses_fp32=None, packed_fp32=None, raw=None, pkh=None, winv=None):
        self.pk, self.pks = pk, pks
        self.pkh, self.winv = pkh or {}, winv or {}
        self.raw = raw or {}                      # key -> unpacked weight (few-input-channels forward kernel)
        self.uses_fp32 = uses_fp32 if uses_fp32 is not None else {}
        self.packed_fp32 = packed_fp32            # keys whose fp32-MFMA operand was re-packed this step (None = all)

    def conv(self, key, x0, cout, **kw):
        """kw: x1, bias, resid, out, out_zeroed, be_out, gn_out (ops.conv3x3)."""
        wp = self.pk[key] if (self.packed_fp32 is None or key in self.packed_fp32) else None
        out = ops.conv3x3(x0, wp, cout, wps=self.pks.get(key), w_raw=self.raw.get(key), wph=self.pkh.get(key),
                          winv=self.winv.get(key), **kw)
        # (-1 = untuned fallback under graph capture, which runs the fp32-MFMA family)
        self.uses_fp32[key] = self.uses_fp32.get(key, False) or ops.LAST_CONV_CONFIG < ops.SPLIT_BASE
        return out

    def conv_parts(self, key, x0, cout, parts=None, be_out=None, x1=None):
        """The same conv as k stored partial sums (ops.conv3x3_parts) -- or None when this weight has no fp16x3 operand
        or the reduction is too short; the caller then uses ``conv``."""
        if not PARTIAL_SLICES or key not in self.pkh:
            return None
        res = ops.conv3x3_parts(x0, cout, self.pkh[key], self.winv[key], parts=parts, be_out=be_out, x1=x1)
        if res is not None:
            self.uses_fp32.setdefault(key, False)
        return res


class Plan:
    """Persistent per-model scratch: the packed-weight arena + its batched pack job table, and (for training) the
    tap-major weight-gradient staging arena + its batched unpack table.  One launch each per step for pack, zero
    and unpack instead of ~60 tiny ones.  Valid while the parameter / gradient storage addresses stay the same."""

    def __init__(self, p: Params, g: Optional[Params], need_input_grad: bool):
        from ._lib import lib
        dev = next(iter(p.values())).device
        self.key = self.signature(p, g, need_input_grad)
        jobs = _conv_jobs(p, need_input_grad)
        sizes = []
        for key, name, off, cin, dg in jobs:
            w = p[name]
            sizes.append(lib.cm_conv3x3_packed_elems(w.shape[0] if dg else cin, cin if dg else w.shape[0]))
        self.wp_arena = torch.empty(sum(sizes), device=dev, dtype=torch.float32)
        self.pk: Dict[str, Tensor] = {}
        rec, o, blk = [], 0, 0
        for (key, name, off, cin, dg), sz in zip(jobs, sizes):
            w = p[name]
            if not w.is_contiguous():
                raise RuntimeError("parameters must be contiguous")
            self.pk[key] = self.wp_arena[o:o + sz]
            rec.append([w.data_ptr(), self.pk[key].data_ptr(), w.shape[0], w.shape[1], off, cin, dg, blk])
            blk += max(1, min(512, (sz + 1023) // 1024))
            o += sz
        rec.append([0, 0, 0, 0, 0, 0, 0, blk])
        self.pack_table = torch.tensor(rec, dtype=torch.int64).to(dev)
        self.pack_n, self.pack_blocks = len(jobs), blk
        self._fp32_rec = rec[:-1]
        self._job_keys = [j[0] for j in jobs]
        # forward convs whose whole reduction (cin * 9) fits one 64-wide MFMA column block also offer the unpacked weight
        self.raw = {key: p[name] for key, name, off, cin, dg in jobs
                    if not dg and off == 0 and cin == p[name].shape[1] and cin * 9 <= 64}
        self.uses_fp32: Dict[str, bool] = {}      # per conv key: did any call run the fp32-MFMA family?
        self._pruned = None                       # (frozenset of keys, table, n, blocks)
        # bf16x6 operand forms of the same weights (the autotuner picks the kernel family per layer)
        self.pks: Dict[str, Tensor] = {}
        if USE_SPLIT:
            ssz = []
            for key, name, off, cin, dg in jobs:
                w = p[name]
                ssz.append(lib.cm_conv3x3_split_packed_bytes(w.shape[0] if dg else cin, cin if dg else w.shape[0]) // 4)
            self.wps_arena = torch.empty(sum(ssz), device=dev, dtype=torch.float32)
            rec, o, blk = [], 0, 0
            for (key, name, off, cin, dg), sz in zip(jobs, ssz):
                w = p[name]
                self.pks[key] = self.wps_arena[o:o + sz]
                rec.append([w.data_ptr(), self.pks[key].data_ptr(), w.shape[0], w.shape[1], off, cin, dg, blk])
                blk += max(1, min(512, (sz // 12 + 255) // 256))
                o += sz
            rec.append([0, 0, 0, 0, 0, 0, 0, blk])
            self.spack_table = torch.tensor(rec, dtype=torch.int64).to(dev)
            self.spack_blocks = blk
        # fp16x3 operand forms + their per-job inverse scales (scratch[njobs + j])
        self.pkh: Dict[str, Tensor] = {}
        self.winv: Dict[str, Tensor] = {}
        if USE_H3:
            hsz = []
            for key, name, off, cin, dg in jobs:
                w = p[name]
                hsz.append(lib.cm_conv3x3_h3_packed_bytes(w.shape[0] if dg else cin, cin if dg else w.shape[0]) // 4)
            self.wph_arena = torch.empty(sum(hsz), device=dev, dtype=torch.float32)
            rec, o, blk = [], 0, 0
            self._winv_slot = {}
            for j, ((key, name, off, cin, dg), sz) in enumerate(zip(jobs, hsz)):
                w = p[name]
                self.pkh[key] = self.wph_arena[o:o + sz]
                self._winv_slot[key] = j
                rec.append([w.data_ptr(), self.pkh[key].data_ptr(), w.shape[0], w.shape[1], off, cin, dg, blk])
                blk += max(1, min(512, (sz // 8 + 255) // 256))
                o += sz
            rec.append([0, 0, 0, 0, 0, 0, 0, blk])
            self.hpack_table = torch.tensor(rec, dtype=torch.int64).to(dev)
            self.hpack_blocks = blk
            self.h3_scratch = torch.zeros(len(jobs) + blk, device=dev, dtype=torch.float32)
            self.winv = {k: self.h3_scratch[j:j + 1] for k, j in self._winv_slot.items()}
        self.gw: Dict[str, Tensor] = {}
        if g is not None:
            names = [n for n in p if n.endswith("body.0.weight") or n.endswith("body.3.weight")]
            if "convlstm.cell.conv.weight" in p:
                names.append("convlstm.cell.conv.weight")
            total = sum(p[n].numel() for n in names)
            self.g_arena = torch.empty(total, device=dev, dtype=torch.float32)
            rec, o, blk = [], 0, 0
            for n in names:
                co, ci = p[n].shape[0], p[n].shape[1]
                sz = p[n].numel()
                self.gw[n] = self.g_arena[o:o + sz].view(co, 9, ci)
                rec.append([self.gw[n].data_ptr(), g[n].data_ptr(), co, ci, 0, 0, 0, blk])
                blk += max(1, min(512, (sz // 9 + 255) // 256))
                o += sz
            rec.append([0, 0, 0, 0, 0, 0, 0, blk])
            self.unpack_table = torch.tensor(rec, dtype=torch.int64).to(dev)
            self.unpack_n, self.unpack_blocks = len(names), blk
            # the same jobs split into the two gradient buckets of the data-parallel exchange: "late" = encoder (its
            # backward runs last), "early" = ConvLSTM + decoder (ready first)
            self._unpack_part = {}
            for part, sel in (("late", lambda n: n.startswith("enc")), ("early", lambda n: not n.startswith("enc"))):
                prec, pblk = [], 0
                for n_, r_ in zip(names, rec[:-1]):
                    if sel(n_):
                        nb = max(1, min(512, (p[n_].numel() // 9 + 255) // 256))
                        prec.append(r_[:7] + [pblk])
                        pblk += nb
                if prec:
                    prec.append([0, 0, 0, 0, 0, 0, 0, pblk])
                    self._unpack_part[part] = (torch.tensor(prec, dtype=torch.int64).to(dev), len(prec) - 1, pblk)

    @staticmethod
    def signature(p: Params, g: Optional[Params], need_input_grad: bool):
        return (tuple(t.data_ptr() for t in p.values()), None if g is None else tuple(t.data_ptr() for t in g.values()),
                bool(need_input_grad))

    def pack(self):
        from ._lib import check, lib
        st = torch.cuda.current_stream().cuda_stream
        table, n, blocks = self.pack_table, self.pack_n, self.pack_blocks
        packed = None
        if (self.pks or self.pkh) and len(self.uses_fp32) == len(self._job_keys):
            # every conv has run at least once: pack the fp32-MFMA form only where that family is actually used
            need = frozenset(k for k in self._job_keys if self.uses_fp32[k])
            if self._pruned is None or self._pruned[0] != need:
                if not torch.cuda.is_current_stream_capturing():
                    rec, blk = [], 0
                    for k, r in zip(self._job_keys, self._fp32_rec):
                        if k in need:
                            nb = max(1, min(512, (self.pk[k].numel() + 1023) // 1024))
                            rec.append(r[:7] + [blk])
                            blk += nb
                    rec.append([0, 0, 0, 0, 0, 0, 0, blk])
                    self._pruned = (need, torch.tensor(rec, dtype=torch.int64).to(self.pack_table.device), len(rec) - 1,
                                    blk)
            if self._pruned is not None and self._pruned[0] == need:
                packed, table, n, blocks = self._pruned
        if n > 0:
            check(lib.cm_pack_conv3x3_batch(table.data_ptr(), n, blocks, st), "pack_batch")
        if self.pks:
            check(lib.cm_pack_conv3x3_split_batch(self.spack_table.data_ptr(), self.pack_n, self.spack_blocks, st),
                  "pack_split_batch")
        if self.pkh:
            check(lib.cm_pack_conv3x3_h3_batch(self.hpack_table.data_ptr(), self.pack_n, self.hpack_blocks,
                                               self.h3_scratch.data_ptr(), st), "pack_h3_batch")
        return _Packs(self.pk, self.pks, self.uses_fp32, packed, self.raw, self.pkh, self.winv)

    def zero_staging(self):
        _zero_(self.g_arena)

    def unpack(self, part=None):
        """Transpose the staged weight gradients into parameter layout: everything, or one exchange bucket."""
        from ._lib import check, lib
        table, n, blocks = (self.unpack_table, self.unpack_n, self.unpack_blocks) if part is None \
            else self._unpack_part.get(part, (None, 0, 0))
        if n > 0:
            check(lib.cm_wgrad3x3_unpack_batch(table.data_ptr(), n, blocks, 1.0,
                                               torch.cuda.current_stream().cuda_stream), "unpack_batch")


_PLANS: "Dict[tuple, Plan]" = {}      # insertion order == recency (re-inserted on every hit)
_PLANS_MAX = 8


def get_plan(p: Params, g: Optional[Params] = None, need_input_grad: bool = False) -> Plan:
    """Plan for this (parameter storage, gradient storage) pair.  The cache is a small LRU; a plan is evicted only
    when nothing else holds it (``sys.getrefcount``): captured hipGraphs carry raw pointers into a plan's arenas and
    tables, so ``HotPathTrainer`` keeps strong references to the plans its graphs were recorded with and those are
    never dropped from under a replay (several models in one process, e.g. an ensemble, stay safe)."""
    import sys
    key = Plan.signature(p, g, need_input_grad)
    plan = _PLANS.pop(key, None)
    if plan is None:
        plan = Plan(p, g, need_input_grad)
        for k in list(_PLANS):
            if len(_PLANS) < _PLANS_MAX:
                break
            if sys.getrefcount(_PLANS[k]) <= 2:      # the dict's reference + getrefcount's argument
                del _PLANS[k]
    _PLANS[key] = plan
    return plan


def pack_weights(p: Params, need_input_grad: bool = False) -> Dict[str, Tensor]:
    """MFMA operand layouts of every 3x3 weight: forward and data-gradient forms (re-run whenever params change)."""
    return get_plan(p, None, need_input_grad).pack()


# ------------------------------------------------------------------------------------------------- ConvBlock
class _BeArena:
    """Per-forward pool of per-sample exponent tables (ops.SampleExponents): the fp16x3 convs publish the magnitudes of
    the tensors they read, the fp16x3 weight gradients of the same tensors consume them.  One zero fill per step."""

    def __init__(self, device, capacity: int):
        self.buf = _zeros(capacity, device=device).view(torch.int32)
        self.used = 0

    def take(self, n: int) -> "ops.SampleExponents":
        if self.used + n > self.buf.numel():
            raise RuntimeError("exponent-table arena too small")
        t = self.buf[self.used:self.used + n]
        self.used += n
        return ops.SampleExponents(t)


def _small_launch(x: Tensor, cin: int, cout: int) -> bool:
    """A forward conv whose grid cannot fill the chip and whose reduction is a chain of >= 4 stages (the 16-frame decoder
    launches and the H/8 level at the micro-batch shapes): it runs as stored partial slices over more workgroups
    (stage_cost_probe: 4.4 us + 1.7 us per 16-channel stage for one workgroup per CU) and GroupNorm adds the slices."""
    n, _, h, w = x.shape
    return PARTIAL_SLICES and CONV_PARTS and cin >= 64 and cin % 16 == 0 and n * h * w * cout <= 1400000


def _block_fwd(p: Params, pk, prefix: str, x0: Tensor, x1: Optional[Tensor], save: bool, pool: bool = False,
               bea: Optional[_BeArena] = None):
    co = p[prefix + "body.0.weight"].shape[0]
    # tables: [x of conv 1, x of conv 2 (a1), dy of conv 1, dy of conv 2]
    be = [bea.take(x0.shape[0]) for _ in range(4)] if (save and bea is not None) else [None] * 4
    # launches too small to fill the chip run as partial slices that the GroupNorm launch adds (see _small_launch)
    ci = x0.shape[1] + (0 if x1 is None else x1.shape[1])
    r1 = pk.conv_parts(prefix + "body.0.weight/f", x0, co, x1=x1, be_out=be[0]) if _small_launch(x0, ci, co) else None
    if r1 is not None:
        a1, st1, _, y1 = ops.gn_silu_fwd(None, p[prefix + "body.1.weight"], p[prefix + "body.1.bias"], parts=r1)
    else:
        gn1 = ops.GnPartials()         # GroupNorm statistics from the conv's epilogue where its tile configuration allows
        y1 = pk.conv(prefix + "body.0.weight/f", x0, co, x1=x1, be_out=be[0], gn_out=gn1)
        a1, st1, _ = ops.gn_silu_fwd(y1, p[prefix + "body.1.weight"], p[prefix + "body.1.bias"], gn=gn1)
    r2 = pk.conv_parts(prefix + "body.3.weight/f", a1, co, be_out=be[1]) if _small_launch(a1, co, co) else None
    g2, b2 = p[prefix + "body.4.weight"], p[prefix + "body.4.bias"]
    w1, w2, w7 = p[prefix + "se.fc.0.weight"], p[prefix + "se.fc.2.weight"], p[prefix + "spat.conv.weight"]
    if ops.block_tail_supported(co, w1.shape[0], a1.shape[2], a1.shape[3], n=a1.shape[0]):
        # the whole tail in one launch, one workgroup per sample (csrc/block_tail.hip); a2 is not materialised
        y2 = None if r2 is not None else pk.conv(prefix + "body.3.weight/f", a1, co, be_out=be[1])
        y2, st2, pooled, z, s, fmap, gate, out, mp = ops.block_tail_fwd(y2, g2, b2, w1, w2, w7, pool_out=pool, parts=r2)
        a2 = None
    else:
        if r2 is not None:
            a2, st2, pooled, y2 = ops.gn_silu_fwd(None, g2, b2, want_pooled=True, parts=r2)
        else:
            gn2 = ops.GnPartials()
            y2 = pk.conv(prefix + "body.3.weight/f", a1, co, be_out=be[1], gn_out=gn2)
            a2, st2, pooled = ops.gn_silu_fwd(y2, g2, b2, want_pooled=True, gn=gn2)
        res = ops.se_spatial_gate_fwd(a2, pooled, w1, w2, w7, pool_out=pool)
        out, z, s, fmap, gate = res[:5]
        mp = res[5] if pool else None
    ctx = None
    if save:
        ctx = _BlockCtx()
        ctx.x0, ctx.x1, ctx.y1, ctx.a1, ctx.st1, ctx.y2, ctx.a2, ctx.st2 = x0, x1, y1, a1, st1, y2, a2, st2
        ctx.pooled, ctx.z, ctx.s, ctx.fmap, ctx.gate, ctx.out = pooled, z, s, fmap, gate, out
        ctx.be = be
        ctx.gn2 = (g2, b2)
        # (tests / probes that inspect a2 after a whole trainer step: materialise it now, one extra launch per block)
        ctx.a2_dbg = ops.gn_silu_apply(y2, g2, b2, st2) if (a2 is None and KEEP_ACTIVATION2) else None
    if pool:
        return out, ctx, mp
    return out, ctx


def _block_bwd(p: Params, pk, g: Params, gw: Params, ss: "_SideStream", prefix: str, ctx: _BlockCtx, dout: Tensor,
               need_dx: bool = True):
    """Returns d(input) as one tensor [N, C0+C1, H, W] (or None); parameter gradients are accumulated into ``g``."""
    co = ctx.y1.shape[1]
    w1, w2, w7 = p[prefix + "se.fc.0.weight"], p[prefix + "se.fc.2.weight"], p[prefix + "spat.conv.weight"]
    if ctx.a2 is None:      # the sample-resident tail ran in the forward: its backward reductions are one launch too
        dmap, (umax, cnt), dpool, (dsig, dz) = ops.block_tail_bwd(dout, ctx.y2, ctx.st2, p[prefix + "body.4.weight"],
                                                                  p[prefix + "body.4.bias"], ctx.s, ctx.z, ctx.gate,
                                                                  ctx.fmap, w1, w2, w7, g[prefix + "spat.conv.weight"])
    else:
        dmap, (umax, cnt), dpool, (dsig, dz) = ops.gates_bwd(dout, ctx.a2, ctx.s, ctx.z, ctx.pooled, ctx.gate, ctx.fmap,
                                                             w1, w2, w7, g[prefix + "se.fc.0.weight"],
                                                             g[prefix + "se.fc.2.weight"],
                                                             g[prefix + "spat.conv.weight"], defer_se_wgrad=True)
    # (the SE weight gradients ride along with the GroupNorm backward launch)
    dy2 = ops.gn_silu_bwd_gated(ctx.y2, p[prefix + "body.4.weight"], p[prefix + "body.4.bias"], ctx.st2, ctx.a2, dout,
                                ctx.gate, dmap, umax, cnt, ctx.s, dpool, g[prefix + "body.4.weight"],
                                g[prefix + "body.4.bias"],
                                se=(dsig, dz, ctx.z, ctx.pooled, g[prefix + "se.fc.0.weight"], g[prefix + "se.fc.2.weight"]))
    # (the data gradient first: it publishes the per-sample magnitudes of dy that the fp16x3 weight gradient scales by)
    be = ctx.be
    da1 = pk.conv(prefix + "body.3.weight/d", dy2, co, be_out=be[3])
    ss.run(lambda: ops.wgrad3x3(ctx.a1, dy2, gw[prefix + "body.3.weight"], be_x=be[1], be_y=be[3]), ctx.a1, dy2)
    dy1 = ops.gn_silu_bwd(ctx.y1, p[prefix + "body.1.weight"], p[prefix + "body.1.bias"], ctx.st1, da1,
                          g[prefix + "body.1.weight"], g[prefix + "body.1.bias"])
    ci = ctx.x0.shape[1] + (0 if ctx.x1 is None else ctx.x1.shape[1])
    dx = pk.conv(prefix + "body.0.weight/d", dy1, ci, be_out=be[2]) if need_dx else None
    ss.run(lambda: ops.wgrad3x3(ctx.x0, dy1, gw[prefix + "body.0.weight"], x1=ctx.x1, be_x=be[0], be_y=be[2]),
           ctx.x0, ctx.x1, dy1)
    return dx


def _zeros(*shape, device) -> Tensor:
    """Zero-filled scratch via hipMemsetAsync on the current stream (a memset node under graph capture)."""
    from ._lib import check, lib
    t = torch.empty(*shape, device=device, dtype=torch.float32)
    check(lib.cm_zero(t.data_ptr(), t.numel() * 4, torch.cuda.current_stream().cuda_stream), "zero")
    return t


def _zero_(t: Tensor) -> None:
    from ._lib import check, lib
    check(lib.cm_zero(t.data_ptr(), t.numel() * 4, torch.cuda.current_stream().cuda_stream), "zero")


def _unpack_into(gw: Tensor, dst: Tensor) -> None:
    from ._lib import check, lib
    cout, _, ctot = gw.shape
    check(lib.cm_wgrad3x3_unpack(gw.data_ptr(), dst.data_ptr(), cout, ctot, 1.0,
                                 torch.cuda.current_stream().cuda_stream), "unpack")


# ------------------------------------------------------------------------------------------------- whole model
class _SideStream:
    """Second HIP stream for the weight-gradient GEMMs.  They depend only on (layer input, upstream gradient) and
    feed nothing but the final unpack, so they run beside the main chain (data gradients + the HBM-bound
    normalisation / gate kernels) and fill the matrix pipes while those stream memory.  Works eagerly and under
    hipGraph capture (fork/join through events).  Tensors handed to the side stream are kept alive until the join.

    One side stream per (device, PARENT stream): the two micro-batch halves of the trainer run on two streams, and each
    must fork into a child of its own.  (Round 2 kept one side stream per device: both halves then forked into -- and
    joined from -- the SAME child, which cross-links the two capture branches: the child waits on an event of half A,
    then on one of half B, and both halves join on it.  That is the topology behind the recorded hipStreamEndCapture
    crash, profiles/r03/capture_fork_probe.txt.)"""

    _streams: Dict[tuple, "torch.cuda.Stream"] = {}

    @staticmethod
    def child_of(device, parent: "torch.cuda.Stream") -> "torch.cuda.Stream":
        idx = device.index if device.index is not None else torch.cuda.current_device()
        key = (idx, parent.cuda_stream)
        if key not in _SideStream._streams:
            _SideStream._streams[key] = torch.cuda.Stream(device=device)
        return _SideStream._streams[key]

    def __init__(self, device, enabled: bool):
        self.enabled = enabled
        self.keep = []
        if enabled:
            self.main = torch.cuda.current_stream(device)
            self.side = _SideStream.child_of(device, self.main)

    def run(self, fn, *tensors):
        if not self.enabled:
            fn()
            return
        self.keep.extend(t for t in tensors if t is not None)
        self.side.wait_stream(self.main)
        with torch.cuda.stream(self.side):
            fn()

    # the stream a graph capture started on while the trainer runs two micro-batches on two streams (trainer._run_parts)
    origin: "Optional[torch.cuda.Stream]" = None

    def deferred_join(self) -> bool:
        """Under capture, a NON-origin stream that waits for its own child (fork + join between two forked streams)
        crashes hipStreamEndCapture; fork-only edges are fine and so is a join into the origin (tools/capture_fork_probe.py,
        profiles/r03/capture_fork_probe.txt).  In that situation the join is left to the origin (trainer._run_parts joins
        every child directly), which is sufficient as long as nothing on the parent consumes the child's results first --
        the weight-gradient unpack therefore runs ON the child (``run_last``)."""
        o = _SideStream.origin
        return (self.enabled and o is not None and torch.cuda.is_current_stream_capturing()
                and self.main.cuda_stream != o.cuda_stream)

    def run_last(self, fn):
        """fn consumes what the side stream produced (the unpack of the staged weight gradients): run it behind that work
        on the side stream when the join is deferred, on the parent after a normal join otherwise."""
        if self.deferred_join():
            with torch.cuda.stream(self.side):
                fn()
            self.keep.clear()
        else:
            self.join()
            fn()

    def join(self):
        if self.enabled:
            if self.deferred_join():
                raise RuntimeError("a side-stream join inside the second micro-batch's stream cannot be captured "
                                   "(hipStreamEndCapture crash, profiles/r03/capture_fork_probe.txt)")
            self.main.wait_stream(self.side)
        self.keep.clear()


# Both side-stream overlaps are OFF and cannot be switched on from the environment (module attributes that only a test or a
# probe sets, e.g. tests/test_model_gpu.py::test_side_stream_overlap_three_steps): they measure 0 to -32 % on config 2
# (a third and fourth stream oversubscribe the hardware queues; profiles/r03/ab_bench.txt).
# History: round 1 saw non-finite values in EAGER mode with the weight-gradient overlap on, rounds 2-3 gradients 1e-4..1e-3
# away from the serial schedule.  Root cause (round 3, profiles/r03/coresidency/): not a missing dependency -- with the
# ConvLSTM weight gradient on the side stream, cm_block_tail_bwd / cm_gn_silu_bwd ran beside an MFMA kernel, and a
# packed-fp32 instruction form the compiler had emitted in them (v_pk_mul_f32 with the halves of src1 swapped) returns wrong
# values in lanes 48-63 of a wave in that situation.  The form is kept out of the library now (build.py, tools/isa_lint.py)
# and both ways of issuing the schedule agree with the serial one to 1e-6.
OVERLAP_WGRAD = False
# Under graph capture a stream that ENTERS the capture from a non-origin stream (a fork inside a forked stream: the child of
# the second micro-batch's stream) crashes hipStreamEndCapture on this runtime; children that entered as first-level forks of
# the origin and only pick up dependency edges from other captured streams later are fine (tools/capture_fork_probe.py,
# profiles/r03/capture_fork_probe.txt).  The trainer pre-forks the side streams from the origin when this is set.
PREFORK_OK = True

# The ConvLSTM recurrence is a serial chain of small launches (N = B samples at 6x9: <= 256 workgroups each) that leaves
# most of the chip idle.  Work that does not depend on it can run beside it on the side stream: the three time-mean
# skips in the forward, the decoder's (deferred) weight gradients in the backward (~0.5 % of the step).  Off, see above.
OVERLAP_LSTM = False


class _Deferred:
    """Collects launches (same ``run`` interface as _SideStream) to be issued later in one go."""

    def __init__(self, enabled: bool):
        self.enabled = enabled
        self.jobs = []
        self.keep = []

    def run(self, fn, *tensors):
        if not self.enabled:
            fn()
            return
        self.jobs.append(fn)
        self.keep.extend(t for t in tensors if t is not None)

    def flush(self):
        for fn in self.jobs:
            fn()
        self.jobs.clear()
        self.keep.clear()


class _LstmCtx:
    """What the ConvLSTM backward needs from its forward."""
    __slots__ = ("B", "T", "s4", "gx", "hprev", "call", "bott", "be")


def convlstm_fwd(p: Params, pk, s4: Tensor, B: int, T: int, save: bool = True, bea: Optional[_BeArena] = None):
    """ConvLSTM.forward (reference src/convlstm.py:27-35) on the folded encoder output s4 [B*T, Cx, h, w]
    (sample n = b*T + t).  Returns (h_last [B, Ch, h, w], ctx); ctx.hprev[:, t] = h_{t-1} (slot 0 = 0), so the hidden
    state of step t < T-1 is ctx.hprev[:, t+1] and the last one is h_last."""
    wl, bl = p["convlstm.cell.conv.weight"], p["convlstm.cell.conv.bias"]
    ch = wl.shape[0] // 4
    h8, w8 = s4.shape[2], s4.shape[3]
    dev = s4.device
    # exponent tables (sample n = b*T + t): [s4, h_{t-1}, d(pre-activations)]
    be = [bea.take(B * T) for _ in range(3)] if (save and bea is not None) else [None] * 3
    gx = pk.conv("lstm.x/f", s4, 4 * ch, bias=bl, be_out=be[0]).view(B, T, 4 * ch, h8, w8)
    hprev = _zeros(B, T, ch, h8, w8, device=dev)             # hprev[:, t] = h_{t-1}; slot 0 stays 0
    call = torch.empty(B, T, ch, h8, w8, device=dev, dtype=torch.float32)
    bott = torch.empty(B, ch, h8, w8, device=dev, dtype=torch.float32)
    pbuf = None                     # slice stack of the partial-slices form, reused by every step
    fused = "lstm.h/f" in pk.pkh and ops.lstm_step_supported(B, ch, h8, w8)
    for t in range(T):
        parts = None
        if t > 0 and fused:
            # projection + gates + state update of step t in ONE launch (csrc/lstm_step.hip).  (The per-sample magnitudes
            # of h_{t-1} for the fp16x3 weight gradient are not published on this path: |h| < 1, the consumer measures.)
            ops.lstm_step_fwd(hprev[:, t], pk.pkh["lstm.h/f"], pk.winv["lstm.h/f"], gx[:, t], call[:, t - 1], call[:, t],
                              hprev[:, t + 1] if t + 1 < T else bott)
            pk.uses_fp32.setdefault("lstm.h/f", False)
            if be[1] is not None:
                be[1].valid = False
            continue
        if t > 0:
            # (slot t of every sample's row of the [B, T] table: first entry t, stride T)
            be_t = None if be[1] is None else ops.SampleExponents(be[1].t[t:], T)
            parts = pk.conv_parts("lstm.h/f", hprev[:, t], 4 * ch, parts=pbuf, be_out=be_t)
            if parts is not None:
                pbuf = parts[0]
            else:
                pk.conv("lstm.h/f", hprev[:, t], 4 * ch, resid=gx[:, t], out=gx[:, t], be_out=be_t)
            if be_t is not None:
                be[1].valid = be_t.valid and (t == 1 or be[1].valid)
        ops.lstm_gates_fwd(gx[:, t], call[:, t - 1] if t > 0 else None, call[:, t],
                           hprev[:, t + 1] if t + 1 < T else bott, parts=parts)
    ctx = None
    if save:
        ctx = _LstmCtx()
        ctx.B, ctx.T, ctx.s4, ctx.gx, ctx.hprev, ctx.call, ctx.bott = B, T, s4, gx, hprev, call, bott
        ctx.be = be
    return bott, ctx


def convlstm_bwd(p: Params, pk, g: Params, gw: Params, ss, ctx: _LstmCtx, dbott: Optional[Tensor],
                 dh_all_steps: Optional[Tensor] = None, before_chain=None):
    """BPTT through the ConvLSTM.  ``dbott``: gradient wrt the last hidden state (the only one the model consumes,
    src/unet_convlstm_attention.py:88); ``dh_all_steps`` [B, T, Ch, h, w] (optional) adds an external gradient to every
    step's hidden state (generic ConvLSTM use; parity fixture convlstm_alldy.npz).  Accumulates the cell's weight /
    bias gradients and returns d(s4) [B*T, Cx, h, w].  ``before_chain`` (optional callable) is issued right before the
    serial chain starts (work for a side stream)."""
    B, T = ctx.B, ctx.T
    gx, hprev, call = ctx.gx, ctx.hprev, ctx.call
    ch = hprev.shape[2]
    cx = ctx.s4.shape[1]
    h8, w8 = hprev.shape[3], hprev.shape[4]
    dev = gx.device
    dc = torch.empty(B, ch, h8, w8, device=dev, dtype=torch.float32)
    dhrec = None
    # the recurrent data gradients are tiny K-split launches: stored partial slices (no fill at all), or -- atomic form --
    # one zero fill for all T-1 outputs instead of one each
    dh_rec_all = None
    pbuf = None
    if before_chain is not None:
        before_chain()
    fused = "lstm.h/d" in pk.pkh and ops.lstm_step_bwd_supported(B, ch, h8, w8)
    for t in range(T - 1, -1, -1):
        ext = None
        if dh_all_steps is not None:
            ext = dh_all_steps[:, t]
        if fused and t < T - 1:
            # recurrent data gradient of step t+1's dA + gate backward of step t in ONE launch (csrc/lstm_step.hip)
            ops.lstm_step_bwd(gx[:, t + 1], pk.pkh["lstm.h/d"], pk.winv["lstm.h/d"], gx[:, t],
                              call[:, t - 1] if t > 0 else None, call[:, t], ext, dc)
            pk.uses_fp32.setdefault("lstm.h/d", False)
            continue
        if t == T - 1 and dbott is not None:
            if ext is None:
                ext = dbott
            else:
                ext = ext + dbott          # (test-only combination; the model passes dbott alone)
        # dh_t = external part + recurrent part (either may be absent); dc carries dL/dc_t
        ops.lstm_gates_bwd(gx[:, t], call[:, t - 1] if t > 0 else None, call[:, t], ext, dhrec, dc, first=(t == T - 1))
        if t > 0 and not fused:
            dhrec = pk.conv_parts("lstm.h/d", gx[:, t], ch, parts=pbuf) if dh_rec_all is None else None
            if dhrec is not None:
                pbuf = dhrec[0]
            else:
                if dh_rec_all is None:
                    dh_rec_all = _zeros(max(T - 1, 1), B, ch, h8, w8, device=dev)
                dhrec = pk.conv("lstm.h/d", gx[:, t], ch, out=dh_rec_all[t - 1], out_zeroed=True)
    dA = gx.view(B * T, 4 * ch, h8, w8)            # now holds d(pre-activations) for every (b, t)
    gl = gw["convlstm.cell.conv.weight"]

    be = ctx.be
    ds4 = pk.conv("lstm.x/d", dA, cx, be_out=be[2])       # (first: publishes the per-sample magnitudes of dA)

    ss.run(lambda: ops.wgrad3x3(ctx.s4, dA, gl, c_off=0, be_x=be[0], be_y=be[2]), ctx.s4, dA)
    if T > 1:   # hprev[:, 0] == 0 contributes nothing
        ss.run(lambda: ops.wgrad3x3(hprev.view(B * T, ch, h8, w8), dA, gl, c_off=cx, be_x=be[1], be_y=be[2]), hprev, dA)
    ops.channel_sum(dA, g["convlstm.cell.conv.bias"])
    return ds4


def up_fwd(p: Params, pk, prefix: str, x: Tensor, skip: Tensor, save: bool = True, bea: Optional[_BeArena] = None):
    """Up.forward (reference src/unet.py:66-69): ConvTranspose2d(2, s2) -> cat([up, skip]) -> ConvBlock; the concat is
    virtual.  Returns (out, (block ctx, x))."""
    u = ops.convT2x2_fwd(x, p[prefix + "up.weight"], p[prefix + "up.bias"])
    out, ctx = _block_fwd(p, pk, prefix + "conv.", u, skip, save, bea=bea)
    return out, (ctx, x)


def up_bwd(p: Params, pk, g: Params, gw: Params, ss, prefix: str, saved, dout: Tensor):
    """Returns (d(x), d(cat)) -- d(skip) is d(cat)[:, c_up:], consumed in place by the caller."""
    ctx, x = saved
    dcat = _block_bwd(p, pk, g, gw, ss, prefix + "conv.", ctx, dout)
    b = ctx.x0.shape[1]
    dx = ops.convT2x2_bwd(x, p[prefix + "up.weight"], dcat[:, :b], g[prefix + "up.weight"], g[prefix + "up.bias"])
    return dx, dcat


class Saved:
    """Everything the backward needs from one forward."""
    __slots__ = ("B", "T", "enc", "lstm", "ups", "d1", "x_shape")


def forward(p: Params, pk, x_seq: Tensor, save: bool = True, head: bool = True):
    """x_seq [B,T,C,H,W] (contiguous, fp32, GPU) -> pred [B,out_ch,H,W], Saved.

    ``head=False`` stops before the 1x1 output head (pred is None; ``Saved.d1`` holds its input): the fused training
    step runs head + loss + head backward as one launch (ops.head_mse_bwd) and passes the result to ``backward``."""
    if x_seq.dim() != 5:
        raise RuntimeError("expected x_seq of shape [B, T, C, H, W]")
    B, T, C, H, W = x_seq.shape
    if C != p["enc1.body.0.weight"].shape[1]:
        raise RuntimeError(f"channel mismatch: input has {C} channels, enc1 expects "
                           f"{p['enc1.body.0.weight'].shape[1]}")
    if H % 8 or W % 8:
        raise RuntimeError("H and W must be divisible by 8 (three 2x2 poolings)")
    x = x_seq.contiguous().view(B * T, C, H, W)
    sv = Saved() if save else None

    # ---- encoder, all frames at once -------------------------------------------------------------------
    # per-sample exponent tables for the fp16x3 weight gradients: 4 per block, 3 for the ConvLSTM (one zero fill)
    bea = _BeArena(x.device, 19 * B * T + 12 * B) if (save and ops.WGRAD_H3) else None
    s1, c1, p1 = _block_fwd(p, pk, "enc1.", x, None, save, pool=True, bea=bea)     # Down = MaxPool2d(2) + ConvBlock
    s2, c2, p2 = _block_fwd(p, pk, "enc2.conv.", p1, None, save, pool=True, bea=bea)
    s3, c3, p3 = _block_fwd(p, pk, "enc3.conv.", p2, None, save, pool=True, bea=bea)
    s4, c4 = _block_fwd(p, pk, "enc4.conv.", p3, None, save, bea=bea)

    # ---- time-mean skips (beside the recurrence when the side stream is on) + ConvLSTM bottleneck --------
    skips = []
    side = _SideStream(x.device, OVERLAP_LSTM)
    side.run(lambda: skips.extend(ops.time_mean(sk, B, T) for sk in (s1, s2, s3)), s1, s2, s3)
    with profiler.region("lstm_cell_fwd"):
        bott, lctx = convlstm_fwd(p, pk, s4, B, T, save, bea=bea)
    side.join()
    k1, k2, k3 = skips

    # ---- decoder ---------------------------------------------------------------------------------------
    d3, u3 = up_fwd(p, pk, "up3.", bott, k3, save, bea=bea)
    d2, u2 = up_fwd(p, pk, "up2.", d3, k2, save, bea=bea)
    d1, u1 = up_fwd(p, pk, "up1.", d2, k1, save, bea=bea)
    pred = ops.head_fwd(d1, p["head.weight"], p["head.bias"]) if head else None

    if save:
        sv.B, sv.T, sv.x_shape = B, T, tuple(x_seq.shape)
        sv.enc = (c1, c2, c3, c4)
        sv.lstm = lctx
        sv.ups = (u3, u2, u1)
        sv.d1 = d1
    return pred, sv


class _BwdState:
    """What the encoder half of the backward needs from the decoder / ConvLSTM half."""
    __slots__ = ("plan", "ss", "side", "ds4", "dcat", "split")


def backward_decoder_lstm(p: Params, pk, g: Params, sv: Saved, dpred: Optional[Tensor], need_dx: bool = False,
                          dd1: Optional[Tensor] = None, bucketed: bool = False) -> "_BwdState":
    """First half of the backward: head, decoder, ConvLSTM (BPTT).  When ``bucketed`` the weight gradients of these
    layers are final (transposed into parameter layout) when this returns, so the data-parallel exchange of that
    bucket can start while ``backward_encoder`` runs."""
    plan = get_plan(p, g, need_dx)
    gw = plan.gw
    plan.zero_staging()
    dev = (dpred if dpred is not None else dd1).device
    ss = _SideStream(dev, OVERLAP_WGRAD)
    u3, u2, u1 = sv.ups
    if dd1 is None:
        dd1 = ops.head_bwd(dpred, sv.d1, p["head.weight"], g["head.weight"], g["head.bias"])
    dec = _Deferred(OVERLAP_LSTM and not OVERLAP_WGRAD)      # decoder weight gradients: issued beside the LSTM chain
    dss = dec if dec.enabled else ss
    dd2, dcat1 = up_bwd(p, pk, g, gw, dss, "up1.", u1, dd1)
    dd3, dcat2 = up_bwd(p, pk, g, gw, dss, "up2.", u2, dd2)
    dbott, dcat3 = up_bwd(p, pk, g, gw, dss, "up3.", u3, dd3)
    side = _SideStream(dev, dec.enabled)
    with profiler.region("lstm_cell_bwd"):
        ds4 = convlstm_bwd(p, pk, g, gw, ss, sv.lstm, dbott,
                           before_chain=lambda: side.run(dec.flush, *list(dec.keep)))   # (kept alive until the join)
    st = _BwdState()
    st.plan, st.ss, st.side, st.ds4, st.dcat, st.split = plan, ss, side, ds4, (dcat1, dcat2, dcat3), bucketed
    if bucketed:
        side.join()
        ss.run_last(lambda: plan.unpack("early"))
    return st


def backward_encoder(p: Params, pk, g: Params, sv: Saved, st: "_BwdState", need_dx: bool = False):
    """Second half: the four encoder blocks over all frames (+ the skips' gradients, the pooling backward)."""
    plan, ss = st.plan, st.ss
    gw = plan.gw
    T = sv.T
    c1, c2, c3, c4 = sv.enc
    u3, u2, u1 = sv.ups
    dcat1, dcat2, dcat3 = st.dcat
    b1, b2, b3 = u1[0].x0.shape[1], u2[0].x0.shape[1], u3[0].x0.shape[1]
    dp3 = _block_bwd(p, pk, g, gw, ss, "enc4.conv.", c4, st.ds4)
    ds3 = ops.maxpool2_bwd(c3.out, dp3, dcat3[:, b3:], t=T)
    dp2 = _block_bwd(p, pk, g, gw, ss, "enc3.conv.", c3, ds3)
    ds2 = ops.maxpool2_bwd(c2.out, dp2, dcat2[:, b2:], t=T)
    dp1 = _block_bwd(p, pk, g, gw, ss, "enc2.conv.", c2, ds2)
    ds1 = ops.maxpool2_bwd(c1.out, dp1, dcat1[:, b1:], t=T)
    dx = _block_bwd(p, pk, g, gw, ss, "enc1.", c1, ds1, need_dx=need_dx)
    st.side.join()
    ss.run_last(lambda: plan.unpack("late" if st.split else None))
    return dx.view(sv.x_shape) if dx is not None else None


def backward(p: Params, pk, g: Params, sv: Saved, dpred: Optional[Tensor], need_dx: bool = False,
             dd1: Optional[Tensor] = None):
    """Accumulates every parameter gradient into ``g`` (same keys as ``p``; the caller zeroes them) and returns
    d(x_seq) when ``need_dx``.  ``post_conv.*`` is untouched (never used by the forward, as in the reference).
    Either ``dpred`` (gradient w.r.t. the prediction) or ``dd1`` (gradient w.r.t. the head's input, with the head's
    own parameter gradients already accumulated by ops.head_mse_bwd) is given."""
    st = backward_decoder_lstm(p, pk, g, sv, dpred, need_dx=need_dx, dd1=dd1)
    return backward_encoder(p, pk, g, sv, st, need_dx=need_dx)


# ------------------------------------------------------------------------------------------------- plain UNet
class SavedUNet:
    """What the backward of the single-frame UNet needs."""
    __slots__ = ("enc", "bott", "ups", "d1", "x_shape")


def unet_forward(p: Params, pk, x: Tensor, save: bool = True, head: bool = True):
    """UNet.forward (reference src/unet.py:99-109): x [B,C,H,W] -> [B,out_ch,H,W].  Same blocks as the hot-path model
    (ConvBlock / Down / Up, src/unet.py:32-69) without the frame loop and the ConvLSTM: a ConvBlock bottleneck at H/8
    and direct (not time-averaged) skips."""
    if x.dim() != 4:
        raise RuntimeError("expected x of shape [B, C, H, W]")
    B, C, H, W = x.shape
    if C != p["enc1.body.0.weight"].shape[1]:
        raise RuntimeError(f"channel mismatch: input has {C} channels, enc1 expects "
                           f"{p['enc1.body.0.weight'].shape[1]}")
    if H % 8 or W % 8:
        raise RuntimeError("H and W must be divisible by 8 (three 2x2 poolings)")
    x = x.contiguous()
    bea = _BeArena(x.device, 32 * B) if (save and ops.WGRAD_H3) else None
    s1, c1, p1 = _block_fwd(p, pk, "enc1.", x, None, save, pool=True, bea=bea)
    s2, c2, p2 = _block_fwd(p, pk, "enc2.conv.", p1, None, save, pool=True, bea=bea)
    s3, c3, p3 = _block_fwd(p, pk, "enc3.conv.", p2, None, save, pool=True, bea=bea)
    s4, c4 = _block_fwd(p, pk, "enc4.conv.", p3, None, save, bea=bea)
    bt, cb = _block_fwd(p, pk, "bott.", s4, None, save, bea=bea)
    d3, u3 = up_fwd(p, pk, "up3.", bt, s3, save, bea=bea)
    d2, u2 = up_fwd(p, pk, "up2.", d3, s2, save, bea=bea)
    d1, u1 = up_fwd(p, pk, "up1.", d2, s1, save, bea=bea)
    pred = ops.head_fwd(d1, p["head.weight"], p["head.bias"]) if head else None
    sv = None
    if save:
        sv = SavedUNet()
        sv.enc, sv.bott, sv.ups, sv.d1, sv.x_shape = (c1, c2, c3, c4), cb, (u3, u2, u1), d1, tuple(x.shape)
    return pred, sv


def unet_backward(p: Params, pk, g: Params, sv: SavedUNet, dpred: Optional[Tensor], need_dx: bool = False,
                  dd1: Optional[Tensor] = None):
    plan = get_plan(p, g, need_dx)
    gw = plan.gw
    plan.zero_staging()
    dev = (dpred if dpred is not None else dd1).device
    ss = _SideStream(dev, OVERLAP_WGRAD)
    c1, c2, c3, c4 = sv.enc
    u3, u2, u1 = sv.ups
    if dd1 is None:
        dd1 = ops.head_bwd(dpred, sv.d1, p["head.weight"], g["head.weight"], g["head.bias"])
    dd2, dcat1 = up_bwd(p, pk, g, gw, ss, "up1.", u1, dd1)
    dd3, dcat2 = up_bwd(p, pk, g, gw, ss, "up2.", u2, dd2)
    dbt, dcat3 = up_bwd(p, pk, g, gw, ss, "up3.", u3, dd3)
    b1, b2, b3 = u1[0].x0.shape[1], u2[0].x0.shape[1], u3[0].x0.shape[1]
    ds4 = _block_bwd(p, pk, g, gw, ss, "bott.", sv.bott, dbt)
    dp3 = _block_bwd(p, pk, g, gw, ss, "enc4.conv.", c4, ds4)
    ds3 = ops.maxpool2_bwd(c3.out, dp3, dcat3[:, b3:], t=1)       # pooling backward + the skip's own gradient
    dp2 = _block_bwd(p, pk, g, gw, ss, "enc3.conv.", c3, ds3)
    ds2 = ops.maxpool2_bwd(c2.out, dp2, dcat2[:, b2:], t=1)
    dp1 = _block_bwd(p, pk, g, gw, ss, "enc2.conv.", c2, ds2)
    ds1 = ops.maxpool2_bwd(c1.out, dp1, dcat1[:, b1:], t=1)
    dx = _block_bwd(p, pk, g, gw, ss, "enc1.", c1, ds1, need_dx=need_dx)
    ss.run_last(plan.unpack)
    return dx.view(sv.x_shape) if dx is not None else None
