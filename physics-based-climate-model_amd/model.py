"""Drop-in model classes and ``get_model`` for the reference's model seam, running on hand-written HIP kernels.

Mirrors (names, argument meaning, state_dict layout, error behaviour):
  * ``AttUNetConvLSTM(in_ch, out_ch, base, seq_len)``  -- reference src/unet_convlstm_attention.py:27-104 (the hot path)
  * ``UNet(in_ch, out_ch, base)``                      -- reference src/unet.py:72-109 (same blocks, single frame)
  * ``get_model(cfg)``                                 -- reference src/models.py:7-38 (all four ``model.type`` values)

A HIP module owns exactly the reference's parameters (same names / shapes / registration order, so checkpoints load
both ways, and the same default initialisation under ``torch.manual_seed``): they live in stock ``nn.Conv2d`` /
``nn.GroupNorm`` / ``nn.ConvTranspose2d`` *containers* that are never called.  ``forward`` hands the parameter
tensors to the engine (HIP kernels) through one ``autograd.Function`` so that the harness's ``loss.backward()``
works unchanged.  There is no CPU path for these classes: a CPU input raises ``RuntimeError``.

``cnn_transformer`` (BASELINE.json configs[3]) lives in ``cnn_transformer.py``, ``SimpleCNN`` (configs[0]) in
``simple_cnn.py`` -- both on the HIP path as well.
"""
from typing import Dict, List

import torch
import torch.nn as nn

from . import engine

_ALIGN = 64  # floats; every parameter starts on a 256-byte boundary of the flat buffers


class _Holder(nn.Module):
    """A parameter container whose layers are only there to own (and default-initialise) tensors."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter container: compute goes through climate_amd.engine")


def _gated_block(ci: int, co: int) -> _Holder:
    # registration order == reference ConvBlock.__init__ (src/unet.py:33-42): body, se, spat
    blk = _Holder()
    blk.body = nn.Sequential(
        nn.Conv2d(ci, co, 3, padding=1, bias=False), nn.GroupNorm(8, co), nn.SiLU(),
        nn.Conv2d(co, co, 3, padding=1, bias=False), nn.GroupNorm(8, co), nn.SiLU())
    blk.se = _Holder()
    blk.se.fc = nn.Sequential(nn.Conv2d(co, co // 8, 1, bias=False), nn.ReLU(),
                              nn.Conv2d(co // 8, co, 1, bias=False), nn.Sigmoid())
    blk.spat = _Holder()
    blk.spat.conv = nn.Conv2d(2, 1, kernel_size=7, padding=3, bias=False)
    return blk


def _pooled_block(ci: int, co: int) -> _Holder:
    h = _Holder()
    h.conv = _gated_block(ci, co)      # the pooling itself has no parameters
    return h


def _up_block(ci: int, cskip: int, co: int) -> _Holder:
    h = _Holder()
    h.up = nn.ConvTranspose2d(ci, co, 2, stride=2)
    h.conv = _gated_block(co + cskip, co)
    return h


class _HotPathFunction(torch.autograd.Function):
    """forward/backward of a whole model as one autograd node (inputs: x + the parameters with gradients)."""

    @staticmethod
    def forward(ctx, module, x, *tensors):
        names = module._grad_names
        p = module._param_dict()
        # (grad mode is always off inside Function.forward: decide from what autograd says it will ask for)
        need_x = bool(ctx.needs_input_grad[1])
        save = any(ctx.needs_input_grad)
        pk = engine.get_plan(p, None, need_x).pack()
        pred, sv = module._engine_forward(p, pk, x, save=save)
        ctx.module, ctx.sv, ctx.pk, ctx.p, ctx.need_x, ctx.names = module, sv, pk, p, need_x, names
        return pred

    @staticmethod
    def backward(ctx, dpred):
        m = ctx.module
        if ctx.sv is None:
            raise RuntimeError("backward through this forward was already run (its saved activations are freed)")
        flat = m._grad_workspace(dpred.device)
        engine._zero_(flat)
        g = m._views(flat)
        dx = m._engine_backward(ctx.p, ctx.pk, g, ctx.sv, dpred.contiguous(), need_dx=ctx.need_x)
        ctx.sv = None
        # Autograd may ADOPT what is returned here as p.grad (AccumulateGrad steals a contiguous gradient), so it must
        # not alias the persistent workspace the next backward zeroes and rewrites: hand out views of a fresh flat copy
        # (one 4*P-byte device copy).  Gradient accumulation over micro-batches, zero_grad(set_to_none=False) and
        # several model calls in one autograd graph all depend on this.
        out = m._views(flat.clone())
        return (None, dx) + tuple(out[n] for n in ctx.names)


class _HipModule(nn.Module):
    """Shared machinery of the HIP-backed models: flat parameter / gradient buffers and the autograd bridge."""

    _INPUT_DIMS = 4
    _NO_GRAD_PREFIXES = ()            # parameters the forward never uses (grad stays None, as in the reference)

    def _finish_init(self):
        self._grad_names: List[str] = [n for n, _ in self.named_parameters()
                                       if not n.startswith(self._NO_GRAD_PREFIXES or ("\0",))]
        self._flat = None          # flat parameter buffer once flatten_parameters_() ran
        self._flat_g = None
        self._layout = None

    # ------------------------------------------------------------------ flat buffers
    def _build_layout(self):
        if self._layout is None:
            off, lay = 0, {}
            named = dict(self.named_parameters())
            order = self._grad_names + [n for n in named if n not in set(self._grad_names)]
            for n in order:
                k = named[n].numel()
                lay[n] = (off, k, tuple(named[n].shape))
                off += (k + _ALIGN - 1) // _ALIGN * _ALIGN
                if n == self._grad_names[-1]:
                    self._n_trainable = off
            self._n_total = off
            self._layout = lay
        return self._layout

    @property
    def n_flat_trainable(self) -> int:
        """Length (floats, padded) of the flat prefix that carries gradients."""
        self._build_layout()
        return self._n_trainable

    def _views(self, flat: torch.Tensor) -> Dict[str, torch.Tensor]:
        lay = self._build_layout()
        return {n: flat[o:o + k].view(s) for n, (o, k, s) in lay.items() if o + k <= flat.numel()}

    def _grad_workspace(self, device) -> torch.Tensor:
        self._build_layout()
        if self._flat_g is None or self._flat_g.device != device:
            self._flat_g = torch.zeros(self._n_trainable, device=device, dtype=torch.float32)
        return self._flat_g

    def flatten_parameters_(self) -> torch.Tensor:
        """Re-home every parameter into one flat fp32 buffer (trainable ones first) and return it.

        Enables the fused multi-tensor Adam and the single-buffer RCCL all-reduce.  Parameter objects keep their
        identity (only ``.data`` is re-pointed), values are preserved.
        """
        lay = self._build_layout()
        named = dict(self.named_parameters())
        dev = next(iter(named.values())).device
        flat = torch.zeros(self._n_total, device=dev, dtype=torch.float32)
        for n, (o, k, s) in lay.items():
            flat[o:o + k].copy_(named[n].data.reshape(-1))
            named[n].data = flat[o:o + k].view(s)
        self._flat = flat
        return flat

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)      # .to()/.cuda() re-allocate parameters: flat views are gone
        self._flat = None
        self._flat_g = None
        return out

    def _param_dict(self) -> Dict[str, torch.Tensor]:
        return {n: t for n, t in self.named_parameters()}

    # ------------------------------------------------------------------ forward
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not x.is_cuda:
            raise RuntimeError(f"{type(self).__name__} (climate_amd) runs on the MI355X HIP path only; got a CPU "
                               "tensor.  The CPU restatement lives in oracle/ and is test infrastructure.")
        if x.dtype != torch.float32:
            raise RuntimeError("expected float32 input (trainer.precision: 32)")
        named = dict(self.named_parameters())
        tensors = [named[n] for n in self._grad_names]
        if any(not t.is_cuda for t in tensors):
            raise RuntimeError("module parameters are on the CPU; call .cuda() first")
        return _HotPathFunction.apply(self, x, *tensors)


class AttUNetConvLSTM(_HipModule):
    """Per-frame attention-UNet encoder, ConvLSTM bottleneck over T, time-mean skips, UNet decoder, 1x1 head.

    x_seq [B, T, in_ch, H, W] (H, W divisible by 8) -> [B, out_ch, H, W].  ``seq_len`` is accepted and, as in the
    reference, not enforced: T is taken from the input.
    """

    _INPUT_DIMS = 5
    _NO_GRAD_PREFIXES = ("post_conv.",)

    def __init__(self, in_ch: int = 5, out_ch: int = 2, base: int = 16, seq_len: int = 3):
        super().__init__()
        if base % 8:
            raise ValueError("base must be a multiple of 8 (GroupNorm(8, base), SE ratio 8)")
        self.seq_len = seq_len
        self.in_ch, self.out_ch, self.base = in_ch, out_ch, base
        b = base
        # registration order == reference __init__ (src/unet_convlstm_attention.py:33-56)
        self.enc1 = _gated_block(in_ch, b)
        self.enc2 = _pooled_block(b, 2 * b)
        self.enc3 = _pooled_block(2 * b, 4 * b)
        self.enc4 = _pooled_block(4 * b, 8 * b)
        self.convlstm = _Holder()
        self.convlstm.cell = _Holder()
        self.convlstm.cell.conv = nn.Conv2d(8 * b + 4 * b, 4 * 4 * b, 3, padding=1)
        # defined-but-unused in the reference (src/unet_convlstm_attention.py:46-49); kept for state_dict parity
        self.post_conv = nn.Sequential(nn.Conv2d(4 * b, 4 * b, kernel_size=3, padding=1), nn.ReLU())
        self.up3 = _up_block(4 * b, 4 * b, 4 * b)
        self.up2 = _up_block(4 * b, 2 * b, 2 * b)
        self.up1 = _up_block(2 * b, b, b)
        self.head = nn.Conv2d(b, out_ch, kernel_size=1)
        self._finish_init()

    def _engine_forward(self, p, pk, x, save=True, head=True):
        return engine.forward(p, pk, x, save=save, head=head)

    def _engine_backward(self, p, pk, g, sv, dpred, need_dx=False, dd1=None):
        return engine.backward(p, pk, g, sv, dpred, need_dx=need_dx, dd1=dd1)

    # two-bucket form for the data-parallel trainer: the flat gradient buffer is laid out in registration order, so the
    # encoder's gradients are its PREFIX [0, bucket_boundary) and ConvLSTM + decoder + head its SUFFIX
    @property
    def bucket_boundary(self) -> int:
        lay = self._build_layout()
        return lay["convlstm.cell.conv.weight"][0]

    def _engine_backward_early(self, p, pk, g, sv, dd1):
        return engine.backward_decoder_lstm(p, pk, g, sv, None, need_dx=False, dd1=dd1, bucketed=True)

    def _engine_backward_late(self, p, pk, g, sv, st):
        return engine.backward_encoder(p, pk, g, sv, st, need_dx=False)


class UNet(_HipModule):
    """Depth-4 attention UNet on one frame (reference src/unet.py:72-109): x [B, in_ch, H, W] -> [B, out_ch, H, W].
    Same HIP kernels as the hot-path model; 82 parameters (enc1-4, bott, up3-1, head), all trained."""

    def __init__(self, in_ch: int = 5, out_ch: int = 2, base: int = 16):
        super().__init__()
        if base % 8:
            raise ValueError("base must be a multiple of 8 (GroupNorm(8, base), SE ratio 8)")
        self.in_ch, self.out_ch, self.base = in_ch, out_ch, base
        b = base
        # registration order == reference __init__ (src/unet.py:81-97); Down = MaxPool2d + ConvBlock under `.conv`
        self.enc1 = _gated_block(in_ch, b)
        self.enc2 = _pooled_block(b, 2 * b)
        self.enc3 = _pooled_block(2 * b, 4 * b)
        self.enc4 = _pooled_block(4 * b, 8 * b)
        self.bott = _gated_block(8 * b, 8 * b)
        self.up3 = _up_block(8 * b, 4 * b, 4 * b)
        self.up2 = _up_block(4 * b, 2 * b, 2 * b)
        self.up1 = _up_block(2 * b, b, b)
        self.head = nn.Conv2d(b, out_ch, kernel_size=1)
        self._finish_init()

    def _engine_forward(self, p, pk, x, save=True, head=True):
        return engine.unet_forward(p, pk, x, save=save, head=head)

    def _engine_backward(self, p, pk, g, sv, dpred, need_dx=False, dd1=None):
        return engine.unet_backward(p, pk, g, sv, dpred, need_dx=need_dx, dd1=dd1)


# ---------------------------------------------------------------------------------------------------- factory
def _get(cfg, key, default=None):
    try:
        v = cfg[key] if not hasattr(cfg, key) else getattr(cfg, key)
    except (KeyError, AttributeError, TypeError):
        return default
    return default if v is None else v


def _items(node):
    return node.items() if hasattr(node, "items") else vars(node).items()


def get_model(cfg):
    """Model factory keyed on ``cfg.model.type`` (reference src/models.py:7-38); every type the reference knows is
    served, unknown ones raise the reference's ``ValueError``.

      * ``unet_convlstm_attention`` -> AttUNetConvLSTM on the HIP engine (the hot path);
      * ``unet``                    -> UNet on the HIP engine (same kernels, single frame);
      * ``cnn_transformer``         -> CNNTransformer on the HIP path (cnn_transformer.py: fp16x3 GEMMs, attention,
                                       LayerNorm kernels; dropout-free function, see that module);
      * ``SimpleCNN``               -> SimpleCNN on the HIP path (simple_cnn.py: BatchNorm2d / Dropout2d / residual kernels
                                       around the same fp16x3 convolutions; BASELINE configs[0]).

    Differences from the reference, both deliberate: for ``unet_convlstm_attention`` ``in_ch`` is
    ``cfg.model.in_ch`` when present, else ``len(cfg.data.input_vars)`` (the reference hard-codes 7, which cannot run
    with its own 5-variable data config); ``cfg.model.seq_len`` / ``cfg.data.seq_len`` is forwarded when present.
    """
    mtype = cfg.model.type
    n_in, n_out = len(cfg.data.input_vars), len(cfg.data.output_vars)
    if mtype == "unet_convlstm_attention":
        in_ch = _get(cfg.model, "in_ch", None) or n_in
        seq_len = _get(cfg.model, "seq_len", None) or _get(cfg.data, "seq_len", 3)
        return AttUNetConvLSTM(in_ch=int(in_ch), out_ch=n_out, base=int(cfg.model.base_channels), seq_len=int(seq_len))
    if mtype == "unet":
        return UNet(in_ch=n_in, out_ch=n_out, base=int(cfg.model.base_channels))
    if mtype == "SimpleCNN":
        from .simple_cnn import SimpleCNN
        kwargs = {k: v for k, v in _items(cfg.model) if k != "type"}      # src/models.py:9-13
        return SimpleCNN(n_input_channels=n_in, n_output_channels=n_out, **kwargs)
    if mtype == "cnn_transformer":
        from .cnn_transformer import CNNTransformer
        return CNNTransformer(in_channels=n_in, out_channels=n_out, embed_dim=int(cfg.model.embed_dim),
                              depth=int(cfg.model.depth), n_heads=int(cfg.model.n_heads),
                              mlp_dim=int(cfg.model.mlp_dim), dropout=float(cfg.model.dropout))
    raise ValueError(f"Unknown model type: {mtype}")
