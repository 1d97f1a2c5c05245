/* climate_hip.h -- C ABI of libclimate_hip.so: the MI355X (gfx950) hot path of the unet_convlstm_attention
 * climate emulator.
 *
 * The reference (ZhenmanShen/Physics-Based-Climate-Model) has no FFI: its seam is Python
 * (src/models.py:7-38 get_model -> nn.Module.forward, main_final.py:538-561,737-747 LightningModule).  This header is
 * the native boundary underneath that seam: plain pointers, sizes and a HIP stream handle; no torch types.
 * Each entry cites the reference call site(s) (paths relative to the reference root) that it replaces.
 *
 * Conventions
 *  - all tensors fp32, NCHW, channel stride == H*W, rows contiguous;
 *  - every tensor pointer is followed (where it can be a slice) by its per-sample stride in ELEMENTS;
 *  - `stream` is a hipStream_t passed as void*; every call only enqueues work on it (graph-capture safe: no
 *    allocation, no synchronisation);
 *  - return value 0 = ok, negative = -errno style argument error, positive = hipError_t.
 */
#ifndef CLIMATE_HIP_H
#define CLIMATE_HIP_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef void* cm_stream;

/* ---- library ---------------------------------------------------------------------------------------------- */
int cm_version(void);                 /* ABI version, bumped on any signature change */
const char* cm_arch(void);            /* "gfx950" */

/* ---- conv3x3 (implicit GEMM on fp32 MFMA) ------------------------------------------------------------------ *
 * nn.Conv2d(ci, co, 3, padding=1): src/unet.py:36,38; gate conv src/convlstm.py:9,13; its data gradient.         */
int cm_conv3x3_num_configs(void);
int cm_conv3x3_pick_config(int n, int h, int w, int cout);
long long cm_conv3x3_packed_elems(int k_channels, int out_channels);
/* w: [cout][cin_total][3][3]; packs input-channel range [c_off, c_off+cin).
 * dgrad=0: operand for y = conv(x, w).  dgrad=1: operand for dx = conv(dy, flip/transposed w). */
int cm_pack_conv3x3(const float* w, int cout, int cin_total, int c_off, int cin, int dgrad, float* wp,
                    cm_stream stream);
/* One launch for many cm_pack_conv3x3 jobs.  descs_dev: device array of (ndesc + 1) records of 8 int64
 * {w ptr, wp ptr, cout, cin_total, c_off, cin, dgrad, first block}; record ndesc carries total_blocks in field 7. */
int cm_pack_conv3x3_batch(const void* descs_dev, int ndesc, int total_blocks, cm_stream stream);
/* out[n, :cout] = conv3x3(cat(in0[:, :c0], in1[:, :c1]), wp) (+ bias) (+ resid).  in1 may be NULL (c1 = 0).
 * resid (nullable) uses the addressing of out (st_resid must equal st_out; resid may alias out).
 * config < 0 picks a tile configuration automatically; else bits 0-7 = tile configuration, bits 8.. = split of
 * the input-channel reduction over workgroups (0/1 = none; > 1 accumulates with atomics into a zeroed output and
 * is not allowed with an in-place residual). */
int cm_conv3x3(const float* in0, long long st0, int c0, const float* in1, long long st1, int c1, const float* wp,
               const float* bias, const float* resid, long long st_resid, float* out, long long st_out, int n, int h,
               int w, int cout, int config, cm_stream stream);

/* ---- conv3x3 on the bf16 matrix cores with fp32-equivalent accuracy ("bf16x6") --------------------------------- *
 * Same contract as cm_conv3x3 (same reference call sites), operands split into three bf16 pieces each, six bf16
 * MFMAs per 16-deep k-step, fp32 accumulation.  wps comes from cm_pack_conv3x3_split_batch (descriptor records as
 * cm_pack_conv3x3_batch, with the wp field pointing at cm_conv3x3_split_packed_bytes() bytes).  When in1 is given,
 * c0 must be a multiple of 16.  config bits 0-7 in [0, cm_conv3x3_split_num_configs()), bits 8.. = K split (as in
 * cm_conv3x3: the output is zeroed and accumulated with float atomics; not with an in-place residual); bit 30 set =
 * the caller has already zeroed `out` (lets it batch the fills of several launches into one).                    */
int cm_conv3x3_split_num_configs(void);
long long cm_conv3x3_split_packed_bytes(int k_channels, int out_channels);
int cm_pack_conv3x3_split_batch(const void* descs_dev, int ndesc, int total_blocks, cm_stream stream);
int cm_conv3x3_split(const float* in0, long long st0, int c0, const float* in1, long long st1, int c1,
                     const void* wps, const float* bias, const float* resid, long long st_resid, float* out,
                     long long st_out, int n, int h, int w, int cout, int config, cm_stream stream);

/* ---- conv3x3 on the f16 matrix cores with fp32-equivalent accuracy ("fp16x3") -------------------------------------- *
 * Same contract, reference call sites and config encoding as cm_conv3x3_split, half its matrix work: every operand is
 * split into TWO fp16 pieces (22 mantissa bits), three products hi*hi + hi*lo + lo*hi, fp32 accumulation (measured
 * 2e-7..5e-7 relative, the level of an fp32 convolution).  fp16's 5 exponent bits are handled by exact power-of-two
 * scaling: the weights per packed slice (cm_pack_conv3x3_h3_batch measures max |w|), the input per workgroup from the
 * running maximum of what it has staged (no history, no calibration, any magnitude; inf / NaN propagate).
 * cm_pack_conv3x3_h3_batch: descriptor records as cm_pack_conv3x3_batch with the wp field pointing at
 * cm_conv3x3_h3_packed_bytes() bytes; scratch = ndesc + total_blocks floats; scratch[job] is that job's wscale_inv.   */
long long cm_conv3x3_h3_packed_bytes(int k_channels, int out_channels);
int cm_pack_conv3x3_h3_batch(const void* descs_dev, int ndesc, int total_blocks, float* scratch, cm_stream stream);
/* sample_be (optional; [n] entries be_stride apart, zeroed by the caller): entry i is raised (atomic max) to the biased
 * exponent (float bits >> 23) of max |input of sample i| over both input tensors -- the per-sample magnitudes
 * cm_wgrad3x3_h3 needs, produced for free by the launch that reads the same tensor. */
int cm_conv3x3_h3(const float* in0, long long st0, int c0, const float* in1, long long st1, int c1, const void* wps,
                  const float* wscale_inv, const float* bias, const float* resid, long long st_resid, float* out,
                  long long st_out, unsigned* sample_be, long long be_stride, int n, int h, int w, int cout, int config,
                  cm_stream stream);
/* GroupNorm(8) statistics in the convolution's epilogue (every ConvBlock conv feeds nn.GroupNorm(8, cout), src/unet.py:36-39):
 * cm_conv3x3_h3_gn is cm_conv3x3_h3 (no reduction split, one sample per workgroup) that also writes, per (sample, group),
 * gn_slots partial records {count, mean, sum of squares about that mean} of its OUTPUT -- a local two-pass over the
 * accumulators of each workgroup's tile -- into gn_part [n][8][gn_slots][3]; cm_gn_silu_fwd_stats merges them, which
 * removes GroupNorm's own statistics pass over the conv output.  cm_conv3x3_h3_gn_slots: records per (sample, group) this
 * tile configuration writes, 0 = the configuration cannot (sample groups, reduction split, cout / 8 not a power of two >= 4). */
int cm_conv3x3_h3_gn_slots(int config, int h, int w, int cout);
int cm_conv3x3_h3_gn(const float* in0, long long st0, int c0, const float* in1, long long st1, int c1, const void* wps,
                     const float* wscale_inv, const float* bias, const float* resid, long long st_resid, float* out,
                     long long st_out, unsigned* sample_be, long long be_stride, float* gn_part, int gn_slots, int n, int h,
                     int w, int cout, int config, cm_stream stream);

/* Forward conv for VERY FEW input channels (cin * 9 <= 64; the first layer, src/unet.py:36 at
 * src/unet_convlstm_attention.py:35): the reduction index is the (input channel, tap) pair, fp32 MFMA, weights read
 * UNPACKED ([cout][cin][3][3]).  out = conv(x, w) + bias.  w_ <= 320. */
int cm_conv3x3_smallc(const float* x, long long sx, int cin, const float* w, const float* bias, float* out,
                      long long st_out, int n, int h, int w_, int cout, cm_stream stream);

/* ---- conv3x3 weight gradient ------------------------------------------------------------------------------- *
 * convolution_backward (weight) of the convs above.  Accumulates (fp32 atomics) into a tap-major staging buffer
 * g[cout][9][ctot] that the caller zeroes once per step; cm_wgrad3x3_unpack transposes it to [cout][ctot][3][3].
 * x = cat(x0[:, :c0], x1[:, :c1]) occupies input-channel range [c_off, c_off+c0+c1) of the full weight.
 * config: < 0 automatic; else bits 0-7 = tile configuration, bits 8.. = grid size in quarter rounds of the
 * resident workgroup slots (0 = one full round).                                                                */
int cm_wgrad3x3_num_configs(void);
int cm_wgrad3x3_pick_config(int n, int h, int w, int cout);
int cm_wgrad3x3(const float* x0, long long sx0, int c0, const float* x1, long long sx1, int c1, const float* dy,
                long long sdy, float* g, int ctot, int c_off, int n, int h, int w, int cout, int config,
                cm_stream stream);
/* bf16x6 form of cm_wgrad3x3 (same arguments, same staging format, fp32-equivalent accuracy; see
 * csrc/wgrad3x3_split.hip): three bf16 pieces per fp32 operand, six v_mfma_f32_32x32x16_bf16 products, 8 samples per
 * 16-byte LDS record so that the nine tap shifts stay aligned.  c0 must be a multiple of 32 when c1 > 0.
 * config in [0, cm_wgrad3x3_split_num_configs()), bits 8.. = grid size in quarter rounds (0 = one round). */
int cm_wgrad3x3_split_num_configs(void);
int cm_wgrad3x3_split(const float* x0, long long sx0, int c0, const float* x1, long long sx1, int c1, const float* dy,
                      long long sdy, float* g, int ctot, int c_off, int n, int h, int w, int cout, int config,
                      cm_stream stream);
/* fp16x3 form of cm_wgrad3x3_split: same configurations, half the matrix work.  The reduction mixes samples, so the
 * power-of-two scaling is per sample and product balanced (csrc/wgrad3x3_split.hip): be_x / be_y [n] = biased exponent of
 * max |x| / max |dy| of every sample (0: all zeros), from cm_conv3x3_h3's sample_be or cm_sample_exponents; values above
 * the true exponent are safe (they only cost precision).  A left-padded window (main_final.py:127-131) next to real
 * ones keeps its accuracy this way. */
int cm_wgrad3x3_h3(const float* x0, long long sx0, int c0, const float* x1, long long sx1, int c1, const float* dy,
                   long long sdy, const unsigned* be_x, const unsigned* be_y, float* g, int ctot, int c_off, int n,
                   int h, int w, int cout, int config, cm_stream stream);
/* be[i * be_stride] = max(be[i * be_stride], biased exponent of max |x[i * stride .. + len)|), i < n; be zeroed by the
 * caller.  For callers whose x / dy were not read by a cm_conv3x3_h3 launch. */
int cm_sample_exponents(const float* x, long long stride, int n, long long len, unsigned* be, long long be_stride,
                        cm_stream stream);
/* Weight gradient for VERY FEW input channels (cin * 9 <= 64, the first layer: src/unet.py:36 at
 * src/unet_convlstm_attention.py:35): GEMM columns are the (input channel, tap) pairs, fp32 MFMA, same staging
 * format G[cout][9][ctot].  w % 4 == 0, w <= 320, st_dy % 4 == 0.  scratch: cm_wgrad3x3_smallc_scratch_elems()
 * floats of workspace (per-workgroup partial tiles, folded by a second launch inside the call). */
long long cm_wgrad3x3_smallc_scratch_elems(int n, int h, int w, int cout);
int cm_wgrad3x3_smallc(const float* x, long long sx, int cin, const float* dy, long long sdy, float* g, int ctot,
                       int c_off, int n, int h, int w, int cout, float* scratch, cm_stream stream);
int cm_wgrad3x3_unpack(const float* g, float* dw, int cout, int ctot, float scale, cm_stream stream);
/* batched form: records of 8 int64 {g ptr, dw ptr, cout, ctot, 0, 0, 0, first block}, as cm_pack_conv3x3_batch */
int cm_wgrad3x3_unpack_batch(const void* descs_dev, int ndesc, int total_blocks, float scale, cm_stream stream);

/* ---- GroupNorm(8) + SiLU ----------------------------------------------------------------------------------- *
 * nn.GroupNorm(8, c), nn.SiLU: src/unet.py:37,39.  stats[n*groups+g] = {mean, rstd}.  pooled (nullable) receives
 * mean_hw(y) per (n,c) = SEBlock's AdaptiveAvgPool2d(1) (src/unet.py:10).  x, y, dx are contiguous [n,c,hw].     */
int cm_gn_silu_fwd(const float* x, const float* gamma, const float* beta, float* y, float* stats, float* pooled,
                   int n, int c, int hw, int groups, float eps, cm_stream stream);
/* cm_gn_silu_fwd fed by a "partial slices" convolution (cm_conv3x3_h3 config bit 29): x arrives as nparts contiguous
 * [n,c,hw] slices zs apart; the launch adds them in slice order, writes the sum to xsum [n,c,hw] (the conv output the
 * backward reads) and proceeds as cm_gn_silu_fwd(xsum, ...). */
int cm_gn_silu_fwd_parts(const float* parts, long long zs, int nparts, float* xsum, const float* gamma,
                         const float* beta, float* y, float* stats, float* pooled, int n, int c, int hw, int groups,
                         float eps, cm_stream stream);
/* cm_gn_silu_fwd whose statistics come from the producing convolution's epilogue (src/unet.py:36-39: conv -> GroupNorm):
 * gpart [n][groups][gslots][3] = {count, mean, sum of squares about that mean} of the parts of each (sample, group), as
 * cm_conv3x3_h3_gn writes them; they are merged (parallel-variance formula) and x is read ONCE. */
int cm_gn_silu_fwd_stats(const float* x, const float* gpart, int gslots, const float* gamma, const float* beta, float* y,
                         float* stats, float* pooled, int n, int c, int hw, int groups, float eps, cm_stream stream);
/* dA = gradient wrt y (sample stride st_dA); dgamma/dbeta are ACCUMULATED (atomics). */
int cm_gn_silu_bwd(const float* x, const float* gamma, const float* beta, const float* stats, const float* dA,
                   long long st_dA, float* dx, float* dgamma, float* dbeta, int n, int c, int hw, int groups,
                   cm_stream stream);
/* y = SiLU(GroupNorm(x)) from the stored statistics of cm_gn_silu_fwd: the recomputation cm_gn_silu_bwd_gated performs
 * inline; bit-identical to what cm_gn_silu_fwd wrote, which the amax tie test depends on. */
int cm_gn_silu_apply(const float* x, const float* gamma, const float* beta, const float* stats, float* y, int n, int c,
                     int hw, int groups, cm_stream stream);
/* cm_gn_silu_bwd with the gradient wrt y rebuilt on the fly from the SE / spatial-gate backward maps (see below).
 * `a2` is unused (recomputed; may be NULL); umax / cnt [n,hw] come from cm_gate_bwd_reduce.  Optional side duty (se_dsig != NULL): the SE weight gradients of the
 * preceding cm_se_excite_bwd call -- dw2 [c,cr] += dsig^T relu(z), dw1 [cr,c] += dz^T pooled -- computed by this
 * launch's workgroups, which saves the separate launch (call cm_se_excite_bwd with dw1 = dw2 = NULL then). */
int cm_gn_silu_bwd_gated(const float* x, const float* gamma, const float* beta, const float* stats,
                         const float* a2, const float* dout, const float* gate, const float* dmap, const float* umax,
                         const float* cnt, const float* s, const float* dpool, float* dx, float* dgamma,
                         float* dbeta, int n, int c, int hw, int groups, const float* se_dsig, const float* se_dz,
                         const float* se_z, const float* se_pooled, float* se_dw1, float* se_dw2, int se_cr,
                         cm_stream stream);

/* ---- SE channel gate + CBAM spatial gate ------------------------------------------------------------------- *
 * SEBlock src/unet.py:6-17, SpatialGate src/unet.py:19-29.  a2 [n,c,hw] = activation entering SE; s [n,c] SE scale;
 * z [n,cr] = W1 pooled (pre-ReLU); map [n,2,hw] = [mean_c, max_c](a2*s); gate [n,hw]; out = a2*s*gate.           */
int cm_se_excite_fwd(const float* pooled, const float* w1, const float* w2, float* z, float* s, int n, int c, int cr,
                     cm_stream stream);
int cm_spatial_stats(const float* a2, const float* s, float* map, int n, int c, int hw, cm_stream stream);
/* cm_se_excite_fwd + cm_spatial_stats in one launch (same results bit for bit): z [n,cr], s [n,c], map [n,2,hw]. */
int cm_se_spatial_stats(const float* pooled, const float* w1, const float* w2, const float* a2, float* z, float* s,
                        float* map, int n, int c, int cr, int hw, cm_stream stream);
/* pooled (nullable, [n,c,h/2,w/2], h and w even): also writes nn.MaxPool2d(2) of `out` (the encoder's Down input,
 * src/unet_convlstm_attention.py:21,25) from the same pass. */
int cm_spatial_apply(const float* a2, const float* s, const float* map, const float* w7, float* gate, float* out,
                     float* pooled, int n, int c, int h, int w, cm_stream stream);
/* backward chain: gate_bwd_reduce -> conv7_bwd -> se_bwd_reduce -> se_excite_bwd -> cm_gn_silu_bwd_gated.
 * amax backward (src/unet.py:27 under autograd: gradient split equally among tied channels): cm_gate_bwd_reduce
 * re-derives umax [n,hw] = max_c(a2*s) and cnt [n,hw] = #{c: a2*s == umax} (>= 1 by construction) in ONE pass from
 * the products it computes itself; the two consumers below test their own a2*s against THAT umax, so the tie logic
 * does not depend on the forward's stored map being reproduced bit for bit and can never divide by zero. */
int cm_gate_bwd_reduce(const float* dout, const float* a2, const float* s, const float* gate, float* dgpre, float* cnt,
                       float* umax, int n, int c, int hw, cm_stream stream);
/* `scratch`: cm_conv7_bwd_scratch_elems(n, h) floats of workspace (per-workgroup partial dW7 sums, no initialisation
 * needed): thousands of workgroups adding into the same 98 addresses serialise in the L2, so the partials are stored
 * and folded by a second tiny kernel inside the same call, or -- with dw7 = NULL -- by the following cm_se_bwd_reduce. */
long long cm_conv7_bwd_scratch_elems(int n, int h);
int cm_conv7_bwd(const float* dgpre, const float* map, const float* w7, float* dmap, float* dw7 /* accumulated */,
                 float* scratch, int n, int h, int w, cm_stream stream);
/* c7_partials (nullable): the scratch of a preceding cm_conv7_bwd(..., dw7 = NULL, ...) with c7_rows =
 * n * ceil(h/8) rows; the first workgroups then also fold those partial sums into dw7 (saves the fold launch). */
int cm_se_bwd_reduce(const float* dout, const float* a2, const float* s, const float* gate, const float* dmap,
                     const float* umax, const float* cnt, float* ds, int n, int c, int hw, const float* c7_partials,
                     int c7_rows, float* dw7, cm_stream stream);
/* dsig [n,c], dz [n,cr], dpool [n,c] are outputs; dw1 [cr,c], dw2 [c,cr] are ACCUMULATED (both NULL: left to the side
 * duty of cm_gn_silu_bwd_gated). */
int cm_se_excite_bwd(const float* ds, const float* s, const float* z, const float* pooled, const float* w1,
                     const float* w2, float* dsig, float* dz, float* dpool, float* dw1, float* dw2, int n, int c,
                     int cr, cm_stream stream);

/* ---- Sample-resident ConvBlock tail (csrc/block_tail.hip) --------------------------------------------------- *
 * Everything of ConvBlock.forward after its second convolution -- GroupNorm(8) -> SiLU (src/unet.py:39), SEBlock
 * (src/unet.py:6-17, called at :46), SpatialGate (src/unet.py:19-29, called at :47) and the encoder's MaxPool2d(2) of
 * the result (src/unet_convlstm_attention.py:21,25) -- as ONE launch: a workgroup keeps one sample's conv output
 * y2 [c,h,w] in registers, so y2 is read once and `out` written once.  Replaces the chain cm_gn_silu_fwd ->
 * cm_se_spatial_stats -> cm_spatial_apply with the same outputs (stats, pooled, z, s, map, gate, out, mp) EXCEPT the
 * activation a2 = SiLU(GroupNorm(y2)), which is not stored (cm_gn_silu_apply reproduces it bit for bit).
 * Input: y2, or (y2 = NULL) `parts` = nparts partial slices of it zs floats apart (cm_conv3x3_h3 config bit 29), whose
 * sum the launch also writes to ysum.  mp nullable ([n,c,h/2,w/2]).
 * cm_block_tail_supported: 1 when a sample fits the register-resident layout (c % 8 == 0, h*w even, c*h*w <= 65536,
 * <= 1024 pixel units: the H/2..H/8 levels at base 32, H/4..H/8 at base 64); otherwise callers keep the chain above. */
int cm_block_tail_supported(int c, int cr, int h, int w);
int cm_block_tail_fwd(const float* y2, const float* parts, long long zs, int nparts, float* ysum, const float* gamma,
                      const float* beta, const float* w1, const float* w2, const float* w7, float* stats,
                      float* pooled, float* z, float* s, float* fmap, float* gate, float* out, float* mp, int n, int c,
                      int cr, int h, int w, float eps, cm_stream stream);
/* The whole-sample reductions of that tail's backward in one launch (replaces cm_gate_bwd_reduce -> cm_conv7_bwd ->
 * cm_se_bwd_reduce -> cm_se_excite_bwd; autograd of src/unet.py:16-17,26-29): from d(out), y2 and the forward's stats /
 * s / z / gate / map it produces exactly what cm_gn_silu_bwd_gated consumes -- dmap [n,2,hw], umax / cnt [n,hw] (the
 * backward's own channel maximum and tie count, cnt >= 1), dpool [n,c], dsig [n,c], dz [n,cr] -- and ACCUMULATES dw7 [98]. */
int cm_block_tail_bwd(const float* y2, const float* stats, const float* gamma, const float* beta, const float* s,
                      const float* z, const float* gate, const float* fmap, const float* w1, const float* w2,
                      const float* w7, const float* dout, float* dmap, float* umax, float* cnt, float* dpool,
                      float* dsig, float* dz, float* dw7, int n, int c, int cr, int h, int w, cm_stream stream);

/* ---- SimpleCNN: BatchNorm2d (+ residual, + ReLU), Dropout2d, 1x1 skip convolutions (csrc/batchnorm.hip) ------ *
 * ResidualBlock / SimpleCNN, src/models.py:44-123.  x, y, resid, dy, dx, dres are contiguous [n,c,hw].
 * cm_bn_fwd: y = relu?(BatchNorm2d(x) + resid) (resid nullable: `out += self.skip(identity)`, src/models.py:72; relu:
 * src/models.py:66,73,93,115).  training != 0: batch statistics (mean, biased variance over n*hw per channel), and
 * running_mean / running_var (nullable) move by `momentum` towards the batch mean / UNBIASED variance (nn.BatchNorm2d
 * defaults: momentum 0.1, eps 1e-5); training == 0: the running buffers normalise.  save [c][2] = {mean, rstd} used.
 * cm_bn_bwd: g = dy * [y > 0] (relu != 0, y = the forward's output); dgamma += sum g*xhat, dbeta += sum g (one owner per
 * channel, plain +=); dx = gamma*rstd*(g - mean(g) - xhat*mean(g*xhat)) (training) or gamma*rstd*g (eval);
 * dres (nullable) = g, the gradient of the residual input. */
int cm_bn_fwd(const float* x, const float* gamma, const float* beta, const float* resid, float* y, float* save,
              float* running_mean, float* running_var, float momentum, float eps, int relu, int training, int n, int c,
              int hw, cm_stream stream);
int cm_bn_bwd(const float* x, const float* y, const float* dy, const float* gamma, const float* save, float* dx,
              float* dres, float* dgamma, float* dbeta, int relu, int training, int n, int c, int hw,
              cm_stream stream);
/* out[p][:] = x[p][:] * mult[p] for `planes` planes of hw floats: nn.Dropout2d (src/models.py:109,119) with the
 * per-(sample, channel) multipliers 0 or 1/(1-p) in `mult`, forward and backward alike. */
int cm_scale_planes(const float* x, const float* mult, float* out, long long planes, int hw, cm_stream stream);
/* The 1x1 skip convolution of a ResidualBlock (src/models.py:57-59) runs on the 3x3 kernels: w3 [cout,cin,3,3] = w1
 * [cout,cin,1,1] at the centre tap, zero elsewhere; its weight gradient is the centre tap of the 3x3 staging tensor
 * G [cout][9][ctot] (cm_wgrad3x3*), ADDED to dw1 [cout,ctot]. */
int cm_embed_center_tap(const float* w1, float* w3, int cout, int cin, cm_stream stream);
int cm_extract_center_tap(const float* g, float* dw1, int cout, int ctot, cm_stream stream);

/* ---- MaxPool2d(2), time-mean skips, per-channel sums ------------------------------------------------------- *
 * nn.MaxPool2d(2): src/unet_convlstm_attention.py:21,25.  stack(...).mean(0): src/unet_convlstm_attention.py:91-93. */
int cm_maxpool2_fwd(const float* x, float* y, long long planes, int h, int w, cm_stream stream);
/* dx = maxpool_backward(dy) + scale * dskip[n / t] (dskip nullable; [n/t, c, h, w] with sample stride st_dskip) */
int cm_maxpool2_bwd(const float* x, const float* dy, const float* dskip, long long st_dskip, float* dx, int n, int c,
                    int h, int w, int t, float scale, cm_stream stream);
int cm_time_mean(const float* x, float* y, int b, int t, long long chw, cm_stream stream);
/* out[c] += sum_{n,p} x[n,c,p]  (bias gradients) */
int cm_channel_sum(const float* x, long long st, float* out, int n, int c, int hw, cm_stream stream);

/* ---- ConvTranspose2d(2, stride 2) -------------------------------------------------------------------------- *
 * nn.ConvTranspose2d(ci, co, 2, stride=2): src/unet.py:63,67.  w: [ci][co][2][2]; x [n,ci,h,w] -> y [n,co,2h,2w]. */
int cm_convT2x2_fwd(const float* x, long long sx, const float* w, const float* b, float* y, long long sy, int n,
                    int ci, int co, int h, int w_, cm_stream stream);
int cm_convT2x2_bwd_data(const float* dy, long long sdy, const float* w, float* dx, long long sdx, int n, int ci,
                         int co, int h, int w_, cm_stream stream);
/* dw [ci,co,2,2] and (nullable) db [co] are ACCUMULATED. */
int cm_convT2x2_bwd_weight(const float* x, long long sx, const float* dy, long long sdy, float* dw, float* db, int n,
                           int ci, int co, int h, int w_, cm_stream stream);

/* ---- ConvLSTM cell pointwise stage ------------------------------------------------------------------------- *
 * ConvLSTMCell.forward src/convlstm.py:14-19 (gate order i,f,o,g).  gates [b,4ch,hw]: pre-activations in,
 * activations out (forward); activations in, d(pre-activations) out (backward).  c_prev nullable (= 0).          */
int cm_lstm_gates_fwd(float* gates, long long sg, const float* c_prev, long long scp, float* c_out, long long sco,
                      float* h_out, long long sho, int b, int ch, int hw, cm_stream stream);
/* dh = dh_a + dh_b (either nullable); dc [b,ch,hw] contiguous: in dL/dc_t (ignored if first), out dL/dc_{t-1}.   */
int cm_lstm_gates_bwd(float* gates, long long sg, const float* c_prev, long long scp, const float* c_cur,
                      long long scc, const float* dh_a, long long sa, const float* dh_b, long long sb, float* dc,
                      int first, int b, int ch, int hw, cm_stream stream);

/* One ConvLSTM time step t >= 1 as ONE launch (csrc/lstm_step.hip): gates = x-projection [+ bias] (already in `gates`) +
 * conv3x3(h_{t-1}, W_h) on the f16 matrix cores (fp16x3), then the stage above -- ConvLSTMCell.forward
 * src/convlstm.py:11-19 without the x-part of its convolution (computed for all T at once).  wph / wscale_inv: the fp16x3
 * operand of the weight's h-columns from cm_pack_conv3x3_h3_batch (k channels = ch, outputs = 4 ch).  hprev, c_prev,
 * c_out, h_out are [b,ch,h*w] with their own sample strides; gates [b,4ch,h*w] is overwritten with i, f, o, g.
 * cm_lstm_step_supported: ch in {64, 128, 256} and h*w <= 64 (the H/8 level of BASELINE configs 2 and 3). */
int cm_lstm_step_supported(int b, int ch, int h, int w);
int cm_lstm_step_fwd(const float* hprev, long long sh, const void* wph, const float* wscale_inv, float* gates,
                     long long sg, const float* c_prev, long long scp, float* c_out, long long sco, float* h_out,
                     long long sho, int b, int ch, int h, int w, cm_stream stream);

/* One BPTT step t < T-1 as ONE launch: dh_t = dh_ext + conv3x3(dA_{t+1}, W_h^T flipped) (fp16x3), then the gate backward
 * of step t (autograd of src/convlstm.py:13-19): gates [b,4ch,hw] holds the step's activations on entry and
 * d(pre-activations) on exit; dA_next = the already-overwritten gates of step t+1; dc [b,ch,hw] carries dL/dc; dh_ext
 * nullable.  wpd / wscale_inv: the data-gradient operand of the weight's h-columns (cm_pack_conv3x3_h3_batch, dgrad = 1).
 * Replaces cm_conv3x3_h3 (partial slices) + cm_lstm_gates_bwd_parts.  Supported: ch in {64, 128}, h*w <= 64. */
int cm_lstm_step_bwd_supported(int b, int ch, int h, int w);
int cm_lstm_step_bwd(const float* dA_next, long long sdn, const void* wpd, const float* wscale_inv, float* gates,
                     long long sg, const float* c_prev, long long scp, const float* c_cur, long long scc,
                     const float* dh_ext, long long sde, float* dc, int b, int ch, int h, int w, cm_stream stream);

/* The same two stages fed by a "partial slices" recurrent projection (cm_conv3x3_h3 config bit 29: the reduction shares
 * of the h-projection / its data gradient are STORED as nparts slices instead of being added with atomics -- the
 * recurrence is a chain of small launches whose cost is latency, and the atomics were half of it):
 * forward: pre-activation = gates + sum_z parts[z] (slices zs apart, sample stride sp, [b,4ch,hw] each);
 * backward: dh = dh_a + sum_z dh_parts[z] (slices zb apart, sample stride sb, [b,ch,hw] each).  Fixed summation order. */
int cm_lstm_gates_fwd_parts(float* gates, long long sg, const float* parts, long long sp, long long zs, int nparts,
                            const float* c_prev, long long scp, float* c_out, long long sco, float* h_out,
                            long long sho, int b, int ch, int hw, cm_stream stream);
int cm_lstm_gates_bwd_parts(float* gates, long long sg, const float* c_prev, long long scp, const float* c_cur,
                            long long scc, const float* dh_a, long long sa, const float* dh_parts, long long sb,
                            long long zb, int nparts, float* dc, int first, int b, int ch, int hw, cm_stream stream);

/* ---- head + loss ------------------------------------------------------------------------------------------- *
 * nn.Conv2d(base, out_ch, 1): src/unet_convlstm_attention.py:56,104.  nn.MSELoss(): main_final.py:544,559.       */
int cm_head_fwd(const float* x, long long sx, const float* w, const float* b, float* pred, int n, int c, int oc,
                int hw, cm_stream stream);
/* *loss = mean((pred-y)^2) (zeroed inside); dpred (nullable) = 2 (pred-y) / total */
int cm_mse_loss(const float* pred, const float* y, float* loss, float* dpred, long long total, cm_stream stream);
/* dw [oc,c], db [oc] ACCUMULATED */
int cm_head_bwd(const float* dpred, const float* x, long long sx, const float* w, float* dx, long long sdx, float* dw,
                float* db, int n, int c, int oc, int hw, cm_stream stream);
/* cm_head_fwd + cm_mse_loss + cm_head_bwd in one pass over x (the fused training step): loss [1] is ACCUMULATED
 * (caller zeroes it), dw/db accumulated, dx written; pred_out nullable.  Same per-pixel arithmetic as the three
 * separate launchers. */
int cm_head_mse_bwd(const float* x, long long sx, const float* w, const float* b, const float* y, float* pred_out,
                    float* loss, float* dx, long long sdx, float* dw, float* db, int n, int c, int oc, int hw,
                    cm_stream stream);

/* ---- optimizer --------------------------------------------------------------------------------------------- *
 * optim.Adam: main_final.py:742-746 (torch defaults).  Flat, 16-byte aligned buffers.  grad_scale multiplies g.  */
int cm_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                 float eps, float weight_decay, int step, float grad_scale, cm_stream stream);
/* graph-replay safe: the step counter and bias corrections live in state[4] on the device and advance per call */
int cm_adam_step_dev(float* p, const float* g, float* m, float* v, long long n, float* state, float lr, float beta1,
                     float beta2, float eps, float weight_decay, float grad_scale, cm_stream stream);
int cm_scale(float* x, long long n, float s, cm_stream stream);
int cm_zero(void* p, size_t bytes, cm_stream stream);

/* ---- the callers either side of the path: window builder and evaluation -------- *
 * cm_build_windows: ClimateDataset.__getitem__ + collation, main_final.py:97-154: x[b,t] = inputs[idx[b]-T+1+t] (zeros
 * where that index is < 0 or >= total: the left padding template, main_final.py:76,127-131), y[b] = outputs[idx[b]].
 * inputs [total, chw_in], outputs [total, chw_out] device resident; idx_dev: b int64 on the device.               */
int cm_build_windows(const float* inputs, const float* outputs, const long long* idx_dev, float* x, float* y, int b,
                     int t, long long chw_in, long long chw_out, long long total, cm_stream stream);
/* validation_step / _evaluate_predictions, main_final.py:563-668.  params_dev: c x 4 doubles {method, a, b, lambda}
 * per output variable, the inverse of Normalizer.normalize (src/utils_final.py:130-206): method 0 pass-through,
 * 1 zscore (a = mean, b = std), 2 minimax (a = min, b = max), 3 log1p (a, b = mean / std of the logged data),
 * 4 sqrt, 5 pow (lambda).  moments: c x 5 x hw doubles, zeroed by the caller before the first batch, accumulated per
 * call: sum p, sum p^2, sum t, sum t^2, sum (p-t)^2 of the DE-NORMALISED values.  target_is_normalized = 0: targets are
 * already in physical units (the test split, main_final.py:454-459).                                              */
int cm_eval_accumulate(const float* pred, const float* target, const double* params_dev, double* moments, int n, int c,
                       int hw, int target_is_normalized, cm_stream stream);
/* out_dev [c][3] = area-weighted monthly RMSE, time-mean RMSE, time-stddev MAE (src/utils_final.py:282-302; identical
 * to _climate_kaggle_metric.py:109-142) from the moments of `count` time steps; lat_w_dev [h] = cos(latitude) weights
 * (any normalisation: src/utils_final.py:387-406 uses mean 1, the Kaggle metric sum 1).                            */
int cm_eval_finalize(const double* moments, const double* lat_w_dev, double count, double* out_dev, int c, int h, int w,
                     cm_stream stream);

/* ---- cnn_transformer (BASELINE.json configs[3]; reference src/cnn_transformer.py:4-54) ---------------------------- *
 * Dense GEMM on the f16 matrix cores, fp32-equivalent ("fp16x3", in-kernel power-of-two scaling, see cm_conv3x3_h3):
 *   C[m,n] = relu?(op(A)[m,k] op(B)[k,n] + bias[n]) + resid[m % res_rows][n], then zeroed where mask[m][n] <= 0.
 * trans_a = 0: A stored [m][k] (lda); 1: stored [k][m].  trans_b = 0: B stored [n][k] (ldb) -- an nn.Linear weight
 * [out][in], so C = A W^T (F.linear: nn.TransformerEncoderLayer's in_proj / out_proj / linear1 / linear2,
 * src/cnn_transformer.py:26-33) --; 1: stored [k][n].  ksplit > 1: K split over workgroups, raw sums accumulated with
 * atomics into a C the caller zeroed (weight gradients: C = dY^T X with trans_a = trans_b = 1).
 * Dropout (nn.Dropout in the encoder layer, src/cnn_transformer.py:26-33): with rng != null and drop_p > 0 the value is
 * multiplied by the counter-based mask of (rng, site) -- see cm_dropout -- after bias / ReLU and BEFORE the residual
 * (x + dropout(sublayer(x))); mask_scale multiplies what `mask` keeps (ReLU backward through a dropped activation).
 * tile: workgroup tile, 0 = the library's choice (64 x 64 at these sizes), 1 = 128 x 128, 2 = 128 x 64, 3 = 64 x 64.      */
int cm_gemm_h3(const float* a, long long lda, int trans_a, const float* b, long long ldb, int trans_b, float* c,
               long long ldc, const float* bias, const float* resid, long long ldr, int res_rows, const float* mask,
               long long ldm, float mask_scale, int relu, const unsigned* rng, unsigned site, float drop_p, int m, int n,
               int k, int ksplit, int tile, cm_stream stream);
/* Weight and bias gradient of a linear layer in one launch (nn.Linear inside nn.TransformerEncoderLayer and the im2col
 * convolutions, src/cnn_transformer.py:9-13,26-33): dw[n_out][k_in] += dy^T x over `tokens` rows (split-K, atomics, as
 * cm_gemm_h3 with trans_a = trans_b = 1) and, when dbias != null, dbias[n_out] += column sums of dy -- the workgroups of
 * the first column of tiles add up the dy tiles they stage anyway (replaces a cm_rowgroup_sum launch).  ksplit = 1
 * STORES dw (as cm_gemm_h3 does); dbias always accumulates.                                                            */
int cm_gemm_h3_wgrad(const float* dy, long long ld_dy, const float* x, long long ldx, float* dw, long long ld_dw,
                     float* dbias, int n_out, int k_in, int tokens, int ksplit, int tile, cm_stream stream);
/* The same GEMM with the B operand split ONCE per step instead of inside every workgroup of every launch (the weight of a
 * linear layer is the B operand of its forward and of its data-gradient GEMM; F.linear and its backward,
 * src/cnn_transformer.py:26-33).  cm_gemm_h3_pack_b_batch: job table on the device, int64 x 8 per job = {w, out, be, N, K,
 * ld, flags, first_block}: B[n][k] = w[n * ld + k] (flags bit 0 clear: the forward's W [out][in]) or w[k * ld + n] (bit 0 set:
 * the data gradient's view of the same storage); `be` (one unsigned per TENSOR, zeroed by the caller; flags bit 1: another
 * job of the same tensor fills it -- the two orientations of a weight share one) receives the biased exponent of max |w|, `out` (cm_gemm_h3_packed_b_bytes(N, K) bytes) the fp16 pieces scaled by 2^(140 - max(be, 13)) in the kernel's
 * LDS image order; first_block = running sum of ceil(N/64) * ceil(K/64) * 2 over the jobs, total_blocks the sum over all.
 * cm_gemm_h3_pb: C = epilogue(A B^T) as cm_gemm_h3 with trans_a = trans_b = 0, ksplit = 1, 64 x 64 tiles.                */
long long cm_gemm_h3_packed_b_bytes(int n, int k);
int cm_gemm_h3_pack_b_batch(const long long* table, int njobs, int total_blocks, int amax_blocks, cm_stream stream);
int cm_gemm_h3_pb(const float* a, long long lda, const void* b_packed, const unsigned* b_exp, float* c, long long ldc,
                  const float* bias, const float* resid, long long ldr, int res_rows, const float* mask, long long ldm,
                  float mask_scale, int relu, const unsigned* rng, unsigned site, float drop_p, int m, int n, int k,
                  cm_stream stream);
/* post-norm residual LayerNorm, eps as given (nn.LayerNorm default 1e-5): sum_out = x + resid (nullable resid; kept for
 * the backward), y = LN(sum_out) * gamma + beta, stats[m][2] = {mean, rstd}.  e <= 1024.                          */
int cm_layernorm_fwd(const float* x, const float* resid, const float* gamma, const float* beta, float* sum_out, float* y,
                     float* stats, int m, int e, float eps, cm_stream stream);
/* ds = gradient wrt sum_in (flows to both residual branches); dgamma / dbeta ACCUMULATED.                         */
int cm_layernorm_bwd(const float* sum_in, const float* stats, const float* gamma, const float* dy, float* ds,
                     float* dgamma, float* dbeta, int m, int e, cm_stream stream);
/* cm_layernorm_bwd with the side outputs of the sublayer that fed this LayerNorm through x + dropout(sublayer(x))
 * (nn.TransformerEncoderLayer._sa_block / _ff_block, src/cnn_transformer.py:27-31): ds_drop (nullable) = ds times the
 * dropout multipliers of (rng, site, drop_p) (element index m*e + column; drop_p 0: a copy is NOT written, pass NULL) and
 * dbias (nullable, ACCUMULATED) += column sums of that gradient = the bias gradient of the sublayer's last linear layer.
 * Replaces a cm_dropout + a cm_rowgroup_sum launch. */
int cm_layernorm_bwd_sublayer(const float* sum_in, const float* stats, const float* gamma, const float* dy, float* ds,
                              float* dgamma, float* dbeta, float* ds_drop, float* dbias, const unsigned* rng,
                              unsigned site, float drop_p, int m, int e, cm_stream stream);
/* Multi-head self-attention core of nn.MultiheadAttention (batch_first, no mask, dropout off): qkv [b*s, 3e] = packed
 * in_proj output; p [b, h, s, s] = softmax(q k^T / sqrt(d)) (kept for the backward); o [b*s, e] = p v, heads
 * concatenated.  head_dim e/h in {8, 16, 32}, s <= 256.                                                           */
/* rng / site / drop_p: dropout on the attention probabilities (nn.MultiheadAttention(dropout=p)); p holds the UNdropped
 * probabilities, the mask is regenerated from the element index by the backward.  drop_p = 0: off.                   */
int cm_attention_fwd(const float* qkv, float* p, float* o, const unsigned* rng, unsigned site, float drop_p, int b, int s,
                     int e, int h, cm_stream stream);
/* dqkv [b*s, 3e] (all three column blocks written) from d_o [b*s, e]; ds_scratch [b, h, s, s] workspace.          */
int cm_attention_bwd(const float* qkv, const float* p, const float* d_o, float* ds_scratch, float* dqkv,
                     const unsigned* rng, unsigned site, float drop_p, int b, int s, int e, int h, cm_stream stream);
/* The same attention core on the f16 matrix cores ("fp16x3", fp32-equivalent) for head_dim 32 and s <= 224 (BASELINE
 * configs[3]: "MFMA attention path"; csrc/attention_mfma.hip): nothing of size s x s is stored -- the forward keeps
 * stats [b, h, s, 2] = {row maximum of the scaled scores, row sum of exp} and the backward recomputes the probabilities
 * (o: the forward's output, dq_rowsum [b, h, s]: workspace).  drop_p <= 0.75.  cm_attention_mfma_supported: 1 when the shape qualifies.        */
int cm_attention_mfma_supported(int b, int s, int e, int h);
int cm_attention_mfma_fwd(const float* qkv, float* stats, float* o, const unsigned* rng, unsigned site, float drop_p,
                          int b, int s, int e, int h, cm_stream stream);
int cm_attention_mfma_bwd(const float* qkv, const float* stats, const float* o, const float* d_o, float* dq_rowsum,
                          float* dqkv, const unsigned* rng, unsigned site, float drop_p, int b, int s, int e, int h,
                          cm_stream stream);
/* Counter-based dropout: out[i] = in[i] * (hash(i ^ key(rng[0] = seed, rng[1] = step counter, site)) < p * 2^32 ? 0 :
 * 1 / (1 - p)).  Nothing is stored: every consumer regenerates the decision from the element index (row-major index of
 * the logical [m, n] GEMM output; flat [b, h, s, s] index for attention probabilities).  cm_rng_advance: counter += 1,
 * once per training forward, on the device so that a replayed hipGraph draws fresh masks.  The stream is NOT torch's
 * Philox stream: masks match the reference statistically, not bitwise.                                 */
int cm_dropout(const float* in, float* out, long long n, const unsigned* rng, unsigned site, float p, cm_stream stream);
int cm_rng_advance(unsigned* rng, cm_stream stream);
/* nn.Conv2d(cin, cout, 3, stride=2, padding=1) as im2col + cm_gemm_h3 (src/cnn_transformer.py:9-13): col
 * [b*(h/2)*(w/2)][ldc], column ci*9 + tap, zero beyond cin*9; x NCHW (tokens_in = 0) or token-major [b*h*w][cin].
 * cm_col2im_s2: the data gradient, token-major out.                                                                */
int cm_im2col_s2(const float* x, float* col, int b, int cin, int h, int w, int ldc, int tokens_in, cm_stream stream);
int cm_col2im_s2(const float* dcol, float* dx_tokens, int b, int cin, int h, int w, int ldc, cm_stream stream);
/* out[batch][cols][rows] = in[batch][rows][cols]: tokens [b][s][e] <-> NCHW [b][e][s] (src/cnn_transformer.py:47,52) */
int cm_transpose_batched(const float* in, float* out, int batch, int rows, int cols, cm_stream stream);
int cm_relu(float* x, long long n, cm_stream stream);                         /* in place */
int cm_relu_mask(float* g, const float* y, long long n, cm_stream stream);    /* g = y > 0 ? g : 0 (y = ReLU output) */
/* out[r % period][c] += x[r][c] summed over r: period 1 = column sums (bias gradients), period s = the gradient of the
 * positional embedding broadcast over the batch (src/cnn_transformer.py:48).  rows % period == 0.                  */
int cm_rowgroup_sum(const float* x, float* out, long long rows, int cols, int period, cm_stream stream);
/* out[r][c] = x[r][c] + add[r % period][c]: tokens + positional embedding (src/cnn_transformer.py:48) */
int cm_add_rowgroup(const float* x, const float* add, float* out, long long rows, int cols, int period, cm_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* CLIMATE_HIP_H */
