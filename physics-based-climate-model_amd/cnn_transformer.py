"""``CNNTransformer`` on the MI355X HIP path (BASELINE.json configs[3]: embed 256, depth 6, 8 heads, 48x72, batch 64).

Mirrors reference src/cnn_transformer.py:4-54 -- CNN encoder (two 3x3 stride-2 convs + ReLU: 48x72 -> 12x18 = 216
tokens), learned positional embedding, ``depth`` post-norm ``nn.TransformerEncoderLayer`` (batch_first, ReLU MLP,
LayerNorm eps 1e-5), CNN decoder (two 2x2 stride-2 transposed convs + ReLU, 1x1 conv) -- with the reference's exact
``state_dict`` (the parameters live in the same stock containers, which are never called), driven by the same
autograd bridge / fused trainer as the hot-path model.

Schedule (every contraction on the f16 matrix cores with fp32-equivalent accuracy, ``cm_gemm_h3``):
  conv 3x3 s2  = cm_im2col_s2 + GEMM (+ bias, ReLU)        token-major activations [B*S, C] from the first conv on
  in_proj / out_proj / linear1 / linear2 = GEMM (F.linear), residual + LayerNorm = cm_layernorm_fwd
  softmax(Q K^T / sqrt(d)) V = cm_attention_fwd             probabilities kept for the backward
  decoder      = transpose to NCHW + the hot path's cm_convT2x2_* and 1x1 head kernels
Backward: the same kernels' gradient forms (weight gradients are split-K GEMMs accumulating straight into the flat
gradient buffer).

DROPOUT: the reference trains this model with ``dropout=0.1`` -- four sites per encoder layer (attention
probabilities, after the attention block, inside the MLP, after the MLP).  In training mode the kernels apply
counter-based masks (csrc/common.h: a hash of seed, step counter, site and element index; nothing is stored, the backward
regenerates them; the counter lives on the device so a replayed hipGraph draws fresh masks every step).  The masks are
the reference's in distribution, not bit for bit (torch's Philox stream is not reproduced): the parity tests export the
device's masks, apply them in the oracle and compare every gradient; ``eval()`` is the dropout-free function.
"""
import math
from typing import Dict

import torch
import torch.nn as nn

from . import ops
from .model import _HipModule

Tensor = torch.Tensor


class _Saved:
    pass


def _weight_views(p: Dict[str, Tensor]):
    """name -> 2-D [out, in] view of every weight that is the B operand of a GEMM (linear layers; the two stride-2 convs as
    im2col GEMMs)."""
    out = {"encoder.0.weight": p["encoder.0.weight"].view(p["encoder.0.weight"].shape[0], -1),
           "encoder.2.weight": p["encoder.2.weight"].view(p["encoder.2.weight"].shape[0], -1)}
    for k, v in p.items():
        if k.startswith("transformer.layers.") and k.endswith(("in_proj_weight", "out_proj.weight", "linear1.weight",
                                                                 "linear2.weight")):
            out[k] = v
    return out


def pack_weights(p: Dict[str, Tensor], device) -> "ops.PackedWeights":
    """fp16x3 B operands of every GEMM weight, both orientations: key ``name`` for the forward (rows = out features) and
    ``name + "/d"`` for the data gradient (rows = in features).  ``.pack()`` refreshes them from the current parameters."""
    pw = ops.PackedWeights(device)
    for k, w in _weight_views(p).items():
        n_out, k_in = w.shape
        pw.add(k, w, n_out, k_in, trans=False)
        pw.add(k + "/d", w, k_in, n_out, trans=True)
    return pw


def _lin(x, p, name, pw, m, n, k, key=None, **kw):
    """x @ B^T with B = the weight ``name`` ([n, k] view) -- from the packed operands when they are given."""
    if pw is not None:
        return ops.gemm_pb(x, pw[key or name], m, **kw)
    w = p[name]
    return ops.gemm(x, w.view(w.shape[0], -1), m, n, k, **kw)


def forward(p: Dict[str, Tensor], x: Tensor, n_heads: int, save: bool = True, head: bool = True, drop=None, pw=None):
    """x [B, Cin, H, W] (H, W multiples of 4) -> pred [B, out, H, W], saved activations.

    ``drop`` = (rng, p) turns on the four dropouts of every encoder layer (attention probabilities, after the attention
    block, inside the MLP, after the MLP: nn.TransformerEncoderLayer(dropout=p), reference src/cnn_transformer.py:26-33)
    with counter-based masks: site 4*layer + {0, 1, 2, 3}; rng is this call's {seed, counter} snapshot (kept in the
    saved state: the backward regenerates the masks from it).  ``pw``: ops.PackedWeights of this parameter set, packed from
    the CURRENT values (pack_weights(p).pack()); without it every GEMM splits its weight operand itself."""
    if x.dim() != 4:
        raise RuntimeError("expected x of shape [B, C, H, W]")
    B, Cin, H, W = x.shape
    if Cin != p["encoder.0.weight"].shape[1]:
        raise RuntimeError(f"channel mismatch: input has {Cin} channels, the encoder expects "
                           f"{p['encoder.0.weight'].shape[1]}")
    S = (H // 4) * (W // 4)
    if H % 4 or W % 4 or S != p["pos_embedding"].shape[1]:
        raise RuntimeError(f"input grid {H}x{W} gives {S} tokens, the positional embedding has "
                           f"{p['pos_embedding'].shape[1]} (reference: 48x72 -> 12x18 = 216)")
    E2, E = p["encoder.0.weight"].shape[0], p["encoder.2.weight"].shape[0]
    M1, M = B * (H // 2) * (W // 2), B * S
    x = x.contiguous()
    k1 = Cin * 9
    col1 = ops.im2col_s2(x, B, Cin, H, W, (k1 + 3) // 4 * 4, tokens_in=False)
    y1 = _lin(col1, p, "encoder.0.weight", pw, M1, E2, k1, bias=p["encoder.0.bias"], relu=True)
    col2 = ops.im2col_s2(y1, B, E2, H // 2, W // 2, E2 * 9, tokens_in=True)
    pos = p["pos_embedding"].view(S, E)
    t0 = _lin(col2, p, "encoder.2.weight", pw, M, E, E2 * 9, bias=p["encoder.2.bias"], relu=True)
    t = _add_pos(t0, pos, S)            # the ReLU output t0 is kept: it is the mask of the conv's backward
    depth = 1 + max(int(k.split(".")[2]) for k in p if k.startswith("transformer.layers."))
    layers = []
    for i in range(depth):
        q = f"transformer.layers.{i}."
        mlp = p[q + "linear1.weight"].shape[0]
        dr = [None] * 4 if drop is None else [(drop[0], 4 * i + k, drop[1]) for k in range(4)]
        qkv = _lin(t, p, q + "self_attn.in_proj_weight", pw, M, 3 * E, E, bias=p[q + "self_attn.in_proj_bias"])
        P, o = ops.attention_fwd(qkv, B, S, E, n_heads, drop=dr[0])
        a = _lin(o, p, q + "self_attn.out_proj.weight", pw, M, E, E, bias=p[q + "self_attn.out_proj.bias"], drop=dr[1])
        t1, s1, st1 = ops.layernorm_fwd(a, t, p[q + "norm1.weight"], p[q + "norm1.bias"])
        h1 = _lin(t1, p, q + "linear1.weight", pw, M, mlp, E, bias=p[q + "linear1.bias"], relu=True, drop=dr[2])
        m2 = _lin(h1, p, q + "linear2.weight", pw, M, E, mlp, bias=p[q + "linear2.bias"], drop=dr[3])
        t2, s2, st2 = ops.layernorm_fwd(m2, t1, p[q + "norm2.weight"], p[q + "norm2.bias"])
        if save:
            layers.append((t, qkv, P, o, s1, st1, t1, h1, s2, st2))
        t = t2
    z = ops.transpose_batched(t, B, S, E).view(B, E, H // 4, W // 4)
    d1 = ops.relu_(ops.convT2x2_fwd(z, p["decoder.0.weight"], p["decoder.0.bias"]))
    d2 = ops.relu_(ops.convT2x2_fwd(d1, p["decoder.2.weight"], p["decoder.2.bias"]))
    pred = ops.head_fwd(d2, p["decoder.4.weight"], p["decoder.4.bias"]) if head else None
    sv = None
    if save:
        sv = _Saved()
        sv.shape, sv.col1, sv.y1, sv.col2, sv.t0, sv.layers, sv.z = (B, Cin, H, W), col1, y1, col2, t0, layers, z
        sv.dec1, sv.dec2 = d1, d2         # decoder activations (post-ReLU)
        sv.d1 = d2                        # input of the 1x1 head: what the fused trainer's head+MSE launch reads
        sv.drop = drop
        sv.pw = pw
    return pred, sv


def _add_pos(t0: Tensor, pos: Tensor, S: int) -> Tensor:
    """t0 [B*S, E] + pos [S, E] broadcast over the batch (x + self.pos_embedding, src/cnn_transformer.py:48)."""
    from ._lib import check, lib
    out = torch.empty_like(t0)
    check(lib.cm_add_rowgroup(t0.data_ptr(), pos.data_ptr(), out.data_ptr(), t0.shape[0], t0.shape[1], S,
                              torch.cuda.current_stream().cuda_stream), "add_rowgroup")
    return out


def _dlin(dy, p, name, pw, m, n, k, **kw):
    """dy @ W (the data gradient of x W^T: B rows = in features, stored across the weight's rows)."""
    if pw is not None:
        return ops.gemm_pb(dy, pw[name + "/d"], m, **kw)
    w = p[name]
    return ops.gemm(dy, w.view(w.shape[0], -1), m, n, k, trans_b=True, **kw)


def _wgrad(dy: Tensor, x: Tensor, dw: Tensor, n_out: int, k_in: int, tokens: int, dbias: Tensor = None):
    """dw [n_out, k_in] += dy^T x  (dy [tokens, n_out], x [tokens, >= k_in]): split-K GEMM over the tokens; with ``dbias``
    the same launch adds the column sums of dy into it (the layer's bias gradient)."""
    # (reduction split 0: chosen by the autotuner per shape -- 768 x 256 over 6912 tokens is fastest at 16 splits, 256 x 256
    #  at 27-32)
    ops.gemm_wgrad(dy, x, dw.view(n_out, k_in), n_out, k_in, tokens, 0, dbias=dbias)


def backward(p: Dict[str, Tensor], g: Dict[str, Tensor], sv, n_heads: int, dpred=None, dd_head=None, need_dx=False):
    """Accumulates every parameter gradient into ``g`` (zeroed by the caller).  Either ``dpred`` or ``dd_head`` (the
    gradient wrt the 1x1 head's input, head gradients already accumulated) is given."""
    B, Cin, H, W = sv.shape
    S = (H // 4) * (W // 4)
    E2, E = p["encoder.0.weight"].shape[0], p["encoder.2.weight"].shape[0]
    M1, M = B * (H // 2) * (W // 2), B * S
    pw = getattr(sv, "pw", None)
    if dd_head is None:
        dd_head = ops.head_bwd(dpred, sv.dec2, p["decoder.4.weight"], g["decoder.4.weight"], g["decoder.4.bias"])
    dd2 = ops.relu_mask_(dd_head, sv.dec2)
    dd1 = ops.relu_mask_(ops.convT2x2_bwd(sv.dec1, p["decoder.2.weight"], dd2, g["decoder.2.weight"],
                                          g["decoder.2.bias"]), sv.dec1)
    dz = ops.convT2x2_bwd(sv.z, p["decoder.0.weight"], dd1, g["decoder.0.weight"], g["decoder.0.bias"])
    dt = ops.transpose_batched(dz.view(B, E, S), B, E, S).view(M, E)
    for i in range(len(sv.layers) - 1, -1, -1):
        q = f"transformer.layers.{i}."
        t_in, qkv, P, o, s1, st1, t1, h1, s2, st2 = sv.layers[i]
        mlp = h1.shape[1]
        drop = getattr(sv, "drop", None)
        dr = [None] * 4 if drop is None else [(drop[0], 4 * i + k, drop[1]) for k in range(4)]
        keep_scale = 1.0 if drop is None else 1.0 / (1.0 - drop[1])
        # ds2 = gradient wrt (t1 + dropout(m2)): the residual branch takes it as it is, the MLP branch through the mask;
        # the same launch adds the column sums of the masked gradient into linear2's bias gradient
        ds2, dm2 = ops.layernorm_bwd(s2, st2, p[q + "norm2.weight"], dt, g[q + "norm2.weight"], g[q + "norm2.bias"],
                                     drop=dr[3], dbias=g[q + "linear2.bias"])
        _wgrad(dm2, h1, g[q + "linear2.weight"], E, mlp, M)
        # through linear2, the MLP dropout and the ReLU: h1 is the DROPPED activation, so h1 > 0 <=> kept and positive
        dh1 = _dlin(dm2, p, q + "linear2.weight", pw, M, mlp, E, mask=h1, mask_scale=keep_scale)
        _wgrad(dh1, t1, g[q + "linear1.weight"], mlp, E, M, dbias=g[q + "linear1.bias"])
        dt1 = _dlin(dh1, p, q + "linear1.weight", pw, M, E, mlp, resid=ds2, res_rows=M)   # + residual branch
        ds1, da = ops.layernorm_bwd(s1, st1, p[q + "norm1.weight"], dt1, g[q + "norm1.weight"], g[q + "norm1.bias"],
                                    drop=dr[1], dbias=g[q + "self_attn.out_proj.bias"])
        _wgrad(da, o, g[q + "self_attn.out_proj.weight"], E, E, M)
        d_o = _dlin(da, p, q + "self_attn.out_proj.weight", pw, M, E, E)
        dqkv = ops.attention_bwd(qkv, P, d_o, B, S, E, n_heads, drop=dr[0], o=o)
        _wgrad(dqkv, t_in, g[q + "self_attn.in_proj_weight"], 3 * E, E, M, dbias=g[q + "self_attn.in_proj_bias"])
        dt = _dlin(dqkv, p, q + "self_attn.in_proj_weight", pw, M, E, 3 * E, resid=ds1, res_rows=M)
    ops.rowgroup_sum(dt, g["pos_embedding"].view(S, E), period=S)
    dt0 = ops.relu_mask_(dt, sv.t0)
    _wgrad(dt0, sv.col2, g["encoder.2.weight"], E, E2 * 9, M, dbias=g["encoder.2.bias"])
    dcol2 = _dlin(dt0, p, "encoder.2.weight", pw, M, E2 * 9, E)
    dy1 = ops.relu_mask_(ops.col2im_s2(dcol2, B, E2, H // 2, W // 2), sv.y1)
    _wgrad(dy1, sv.col1, g["encoder.0.weight"], E2, Cin * 9, M1, dbias=g["encoder.0.bias"])
    if not need_dx:
        return None
    dcol1 = _dlin(dy1, p, "encoder.0.weight", pw, M1, Cin * 9, E2)
    dx_tok = ops.col2im_s2(dcol1, B, Cin, H, W)                     # token-major [B*H*W, Cin]
    return ops.transpose_batched(dx_tok.view(B, H * W, Cin), B, H * W, Cin).view(B, Cin, H, W)


class CNNTransformer(_HipModule):
    """Drop-in for the reference's ``CNNTransformer`` (same constructor, same state_dict) on the HIP path."""

    _head_param_names = ("decoder.4.weight", "decoder.4.bias")

    def __init__(self, in_channels=5, out_channels=2, embed_dim=128, depth=4, n_heads=4, mlp_dim=256, dropout=0.1):
        super().__init__()
        if embed_dim % n_heads or embed_dim // n_heads not in (8, 16, 32):
            raise ValueError("head_dim = embed_dim / n_heads must be 8, 16 or 32 on the HIP attention kernel")
        if embed_dim % 4 or embed_dim > 1024:
            raise ValueError("embed_dim must be a multiple of 4 and <= 1024")
        half, quarter = embed_dim // 2, embed_dim // 4
        # registration order == reference __init__ (src/cnn_transformer.py:8-42); these containers are never called
        self.encoder = nn.Sequential(nn.Conv2d(in_channels, half, kernel_size=3, stride=2, padding=1), nn.ReLU(),
                                     nn.Conv2d(half, embed_dim, kernel_size=3, stride=2, padding=1), nn.ReLU())
        self.height, self.width = 12, 18
        self.num_tokens = self.height * self.width
        self.embed_dim, self.n_heads, self.dropout_p = embed_dim, n_heads, float(dropout)
        self.pos_embedding = nn.Parameter(torch.randn(1, self.num_tokens, embed_dim))
        layer = nn.TransformerEncoderLayer(d_model=embed_dim, nhead=n_heads, dim_feedforward=mlp_dim, dropout=dropout,
                                           batch_first=True)
        self.transformer = nn.TransformerEncoder(layer, num_layers=depth)
        self.decoder = nn.Sequential(nn.ConvTranspose2d(embed_dim, half, kernel_size=2, stride=2), nn.ReLU(),
                                     nn.ConvTranspose2d(half, quarter, kernel_size=2, stride=2), nn.ReLU(),
                                     nn.Conv2d(quarter, out_channels, kernel_size=1))
        self._finish_init()

    def _rng_state(self, device) -> Tensor:
        """{seed, step counter} on the device (int32 x 2).  The seed is drawn from torch's global generator the first
        time it is needed (so ``torch.manual_seed`` / ``cfg.seed`` decide it); ``reseed_dropout`` sets it explicitly."""
        rng = self.__dict__.get("_rng")
        if rng is None or rng.device != device:
            seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if rng is None else int(rng[0].item())
            rng = torch.tensor([seed, 0], dtype=torch.int32, device=device)
            self.__dict__["_rng"] = rng
        return rng

    def reseed_dropout(self, seed: int, counter: int = 0) -> None:
        dev = next(self.parameters()).device
        self.__dict__["_rng"] = torch.tensor([int(seed) & 0x7fffffff, int(counter)], dtype=torch.int32, device=dev)

    def _packed_weights(self, p, device):
        """This parameter set's packed GEMM operands (built once per flat parameter buffer; ``.pack()`` refreshes them)."""
        key = (str(device), p["encoder.0.weight"].data_ptr())
        cache = self.__dict__.setdefault("_pw_cache", {})
        if key not in cache:
            cache.clear()
            cache[key] = pack_weights(p, device)
        return cache[key]

    def _engine_forward(self, p, pk, x, save=True, head=True, rng_snapshot=None, packed=False):
        # the weight operands are split once per forward (3 launches) -- or once per STEP by the caller that runs several
        # forwards of the same parameters side by side (_micro_prepare: packed=True)
        pw = self._packed_weights(p, x.device)
        if not packed:
            pw.pack()
        drop = None
        if self.training and self.dropout_p > 0.0:
            if rng_snapshot is None:
                rng = self._rng_state(x.device)
                ops.rng_advance(rng)
                # this call's snapshot: a later forward (gradient accumulation) must not change the masks of this backward
                rng_snapshot = rng.clone()
            drop = (rng_snapshot, self.dropout_p)
        return forward(p, x, self.n_heads, save=save, head=head, drop=drop, pw=pw)

    def _micro_prepare(self, device, parts: int):
        """Per-micro-batch keyword arguments of ``_engine_forward`` for forwards that run CONCURRENTLY (the trainer's
        micro-batch overlap): the dropout counter is advanced once per part here, on the caller's stream, and every part
        gets its own {seed, counter} snapshot -- two forwards advancing the device counter from two streams would race
        and could draw the same masks.  The weight operands are packed here too, once for all parts."""
        self._packed_weights(self._param_dict(), device).pack()       # once, on the caller's stream, for all parts
        if not (self.training and self.dropout_p > 0.0):
            return [{"packed": True} for _ in range(parts)]
        rng = self._rng_state(device)
        out = []
        for _ in range(parts):
            ops.rng_advance(rng)
            out.append({"rng_snapshot": rng.clone(), "packed": True})
        return out

    def _engine_backward(self, p, pk, g, sv, dpred, need_dx=False, dd1=None):
        return backward(p, g, sv, self.n_heads, dpred=dpred, dd_head=dd1, need_dx=need_dx)
