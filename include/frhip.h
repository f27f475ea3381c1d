/* frhip.h -- C ABI of libfrhip.so: the MI355X (gfx950) kernels behind the face-embedding training path.
 *
 * The reference (aanna0701/face-recognition-pytorch) is pure Python over torch; it has no FFI of its own
 * (SURVEY.md section 8b).  Each entry point below replaces the device work that one reference call site hands to
 * cuDNN / cuBLAS / ATen, cited as /root/reference file:line.  INTEGRATION.md shows the ctypes stub a reference
 * maintainer would add at that call site.
 *
 * Conventions
 *   - plain pointers to DEVICE memory owned by the caller (PyTorch caching allocator), sizes as int / size_t;
 *     no torch types; the library allocates nothing.
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*); no hidden synchronisation.
 *   - returns 0 on success, a negative FRHIP_E* code otherwise; frhip_last_error() gives the message
 *     (thread-local).  Nothing throws across the boundary.
 *   - dtype: 0 = bf16 storage + bf16 MFMA (fp32 accumulate), 1 = fp32 storage + exact-fp32 MFMA (validation mode).
 *   - activations are NHWC ("channels last"); conv weights are [K][R][S][C] which is the physical layout of a
 *     torch channels_last [K,C,R,S] tensor.
 */
#ifndef FRHIP_H
#define FRHIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef FRHIP_STREAM_T
#define FRHIP_STREAM_T
#ifdef __HIP_PLATFORM_AMD__
#include <hip/hip_runtime_api.h>
typedef hipStream_t frhip_stream_t;
#else
typedef void* frhip_stream_t;
#endif
#endif

#define FRHIP_OK 0
#define FRHIP_EINVAL (-1)
#define FRHIP_ELAUNCH (-2)
#define FRHIP_DT_BF16 0
#define FRHIP_DT_F32 1

/* Threading contract.  Every compute entry point is asynchronous on the stream it is given, keeps no state between calls and may
 * be called concurrently from several host threads on different streams (frhip_last_error() is per thread).
 * The frhip_set_* functions are TEST / TUNING HOOKS, not part of that contract: they flip process-global kernel-selection
 * switches (which tile, which kernel variant -- never the arithmetic contract beyond what each hook documents), are not
 * synchronised, and must only be called while no other thread is inside the library.  The product path (nets/, model/,
 * bench.py) never changes them (one read-only query, frhip_set_winattn_mfma(-1)); the defaults are the measured-best settings (environment overrides are read once, at load). */
const char* frhip_last_error(void);
int frhip_abi_version(void);
/* stream-concurrency probe: one wave that occupies `stream`'s hardware queue for `ticks` of the 100-MHz wall clock (<= 1e8).
 * nets/_backbone.py times two of them on two streams to find a side stream that does not share the main stream's hardware queue. */
int frhip_spin(long long ticks, frhip_stream_t stream);

/* ---- convolution = MFMA implicit GEMM.  nn.Conv2d(bias=False): nets/resnet.py:23-46, used at :89-103, :232 ---- */
/* y[n,ho,wo,k] = conv(x[n,h,w,c], w[k,r,s,c]); stats_partial (may be NULL) receives per-row-tile
 * {sum, sum of squares} per output channel: [frhip_conv_stat_rows(...)][2][k] -- the BN batch statistics
 * (nets/resnet.py:90-91) come out of the conv epilogue instead of a second pass. */
int frhip_nt_block_m(int k);
/* number of partial rows frhip_conv_fwd writes into stats_partial for this convolution (m = n*ho*wo output pixels) */
int frhip_conv_stat_rows(int dtype, int m, int k, int h, int w, int c, int r, int s, int stride, int pad);
/* test hook: 1 (default) = 3x3 / stride-1 bf16 weight gradients on the nine-tap kernel, 0 = on the per-tap gather kernel;
 * < 0 queries.  Returns the old value */
int frhip_set_wgrad_taps9(int enabled);
/* test hook: 0 disables the LDS-halo 3x3/s1 kernel (generic NT kernel is used instead), 1 automatic tile choice,
 * 2 forces the 4-wave 256x64 tile (two workgroups per CU), 3 the 8-wave 256x128 tile.  Returns the old value */
int frhip_set_conv_halo(int enabled);
/* test / tuning hook: which 3x3/s1 launches of the W <= 28 layers (bf16, automatic mode) may use the 64 x 128-per-wave halo tile:
 * bit 0 forward, bit 1 data-gradient (default 2); < 0 queries.  Returns the old value */
int frhip_set_halo_wide_dirs(int dirs);
/* test hook: 1 (default) = launches whose tiles are all whole and whose layout is dense (bf16) run the LEAN kernel instantiations
 * (32-bit buffer-offset store epilogue, BatchNorm sums on the matrix pipe); 0 = always the general store epilogue; < 0 queries.
 * Outputs are bit-identical either way, the per-tile partial sums agree to fp32 summation order.  Returns the old value */
int frhip_set_epi_lean(int enabled);
/* test / micro-benchmark hook: force the NT tile (0 auto, 1 128x128, 2 256x64, 3 256x128, 4 256x256); returns the old value */
int frhip_set_nt_tile(int tile);
/* Inference (model/FR_PartialFC.py:205-211: encoder.eval(); nets/resnet.py:89-103 with BatchNorm in eval mode): convolution with the
 * eval-mode BatchNorm folded into its store epilogue, y = [relu](conv(x, w) * scale[k] + shift[k] + residual), scale / shift from
 * frhip_bn_eval_affine, residual (optional) shaped like y.  The convolution result is rounded to the compute dtype before the
 * affine map, as the unfused pair frhip_conv_fwd + frhip_bn_apply stores it: same values, one tensor pass less per BatchNorm. */
int frhip_conv_fwd_affine(int dtype, const void* x, const void* w, void* y, const float* scale, const float* shift, int relu,
                          const void* residual, int n, int h, int wd, int c, int k, int r, int s, int stride, int pad,
                          frhip_stream_t stream);
int frhip_conv_fwd(int dtype, const void* x, const void* w, void* y, float* stats_partial,
                   int n, int h, int wd, int c, int k, int r, int s, int stride, int pad, frhip_stream_t stream);
/* dx[n,h,w,c] = conv_transpose(dy[n,ho,wo,k], w) (+ residual[n,h,w,c] if not NULL); wt = frhip_pack_wt(w) = [c][r][s][k].
 * autograd of nn.Conv2d w.r.t. input; the residual add is the gradient fan-in of `out += residual` (nets/resnet.py:101). */
int frhip_conv_dgrad(int dtype, const void* dy, const void* wt, void* dx, const void* residual,
                     int n, int h, int wd, int c, int k, int r, int s, int stride, int pad, frhip_stream_t stream);
/* frhip_conv_dgrad with the NEXT BatchNorm-backward reduction fused into its epilogue: dx is the upstream gradient of a
 * BatchNorm (nets/resnet.py:91, :98) whose saved input is y_bn [n,h,w,c]; stats_partial[frhip_dgrad_stat_rows(...)][2][c]
 * receives per-row-tile { sum d, sum d*(y_bn-mean)*invstd } with d = dx * (y_bn*mask_scale + mask_shift > 0) (mask_scale may
 * be NULL: no ReLU between the BN and this gradient) -- what frhip_bn_bwd_reduce would compute in a second pass over dx
 * and y_bn; feed it to frhip_bn_bwd_finalize. */
int frhip_dgrad_stat_rows(int dtype, int n, int h, int wd, int c, int k, int r, int s, int stride, int pad);
int frhip_conv_dgrad_bnred(int dtype, const void* dy, const void* wt, void* dx, const void* residual,
                           const void* y_bn, const float* mean, const float* invstd, const float* mask_scale,
                           const float* mask_shift, float* stats_partial, int n, int h, int wd, int c, int k,
                           int r, int s, int stride, int pad, frhip_stream_t stream);
/* the general form: residual_stride 1 = dense residual [n,h,w,c]; 2 = residual is the COMPACT gradient of a stride-2 1x1
 * shortcut, [n,(h+1)/2,(w+1)/2,c], added on the even pixels only (the downsample branch of nets/resnet.py:96-101: its
 * zero-stuffed full-size gradient is never built).  y_bn == NULL: no BatchNorm reduction (stats_partial ignored). */
int frhip_conv_dgrad_fused(int dtype, const void* dy, const void* wt, void* dx, const void* residual, int residual_stride,
                           const void* y_bn, const float* mean, const float* invstd, const float* mask_scale,
                           const float* mask_shift, float* stats_partial, int n, int h, int wd, int c, int k,
                           int r, int s, int stride, int pad, frhip_stream_t stream);
/* frhip_conv_dgrad_fused for a BatchNorm under stochastic depth (nets/AlterNet_SwinV2_FAN.py:407-450: x + drop_path(norm(f(x))), the
 * BatchNorm's incoming gradient is dx * rowscale[sample]): rowscale[g] (fp32: 0 for a dropped sample, keep_scale = 1 / keep-probability for
 * a kept one) covers the rows_per consecutive rows of sample g; the partial sums are those frhip_bn_bwd_reduce_rs would produce from the
 * stored dx, dx itself is stored unscaled.  Replaces that separate pass over (dx, y_bn) in the attention blocks' backward. */
int frhip_conv_dgrad_fused_rs(int dtype, const void* dy, const void* wt, void* dx, const void* residual, int residual_stride,
                              const void* y_bn, const float* mean, const float* invstd, const float* mask_scale,
                              const float* mask_shift, const float* rowscale, int rows_per, float keep_scale,
                              float* stats_partial, int n, int h, int wd, int c, int k, int r, int s, int stride, int pad,
                              frhip_stream_t stream);
/* dw[k,r,s,c] (fp32, caller-zeroed) += sum over output pixels dy * x.  autograd of nn.Conv2d w.r.t. weight.
 * splits <= 0: library picks the split-K factor.  workspace (may be NULL): caller-owned scratch used by THIS call only
 * (one per stream); when splits * sizeof(dw) fits, each K split stores a private slab with plain stores and one reduce
 * pass adds them (deterministic, no same-address atomic contention); otherwise fp32 atomics are used. */
int frhip_conv_wgrad(int dtype, const void* dy, const void* x, float* dw, int n, int h, int w, int c,
                     int k, int r, int s, int stride, int pad, int splits, float* workspace, size_t workspace_bytes,
                     frhip_stream_t stream);
/* Chained weight gradients of 3x3 / stride-1 bf16 convolutions on 14 x 14 maps (the rows kernel): launch i adds the K-split slabs of
 * launch i - 1 to THAT launch's dw in its own prologue, so no reduce launch sits between two weight gradients of a stream (inside the
 * training step the separate reduce -- 8 us alone -- took 50 - 65 us: it got onto the CUs only as the other stream's workgroups retired).
 * frhip_conv_wgrad_chain_ok: 1 when the shape is served.  frhip_conv_wgrad_chain writes this launch's slabs to `slabs` (>= *splits_out *
 * k*9*c floats; must differ from prev_slabs), stores the split count in *splits_out (host memory) and does NOT touch this launch's dw:
 * pass (dw, slabs, k, c, *splits_out) as the prev_* arguments of the next launch on the same stream, or to
 * frhip_conv_wgrad_chain_finish, which adds them with the ordinary reduce kernels.  prev_splits = 0: nothing to add.
 * Same sums in the same order as frhip_conv_wgrad on these shapes (bit-identical dw). */
int frhip_conv_wgrad_chain_ok(int dtype, int n, int h, int w, int c, int k, int r, int s, int stride, int pad);
int frhip_conv_wgrad_chain(int dtype, const void* dy, const void* x, int n, int h, int w, int c, int k, float* slabs,
                           size_t slab_bytes, float* prev_dw, const float* prev_slabs, int prev_k, int prev_c, int prev_splits,
                           int* splits_out, frhip_stream_t stream);
int frhip_conv_wgrad_chain_finish(float* dw, float* slabs, int k, int c, int splits, frhip_stream_t stream);
/* nn.Linear forward with its epilogue fused (nets/SwinV2.py:16-32 Mlp, :150-176 qkv / proj): out[m][n] = a[m][k] . w[n][k] +
 * bias[n] (bias may be NULL), stored in `dtype`; act_out (may be NULL) = gelu(out), exact erf form, of the stored value;
 * stats_partial (may be NULL): per-tile BatchNorm partial sums of `out`, rows = frhip_conv_stat_rows(dtype, m, n, 1,1,k,1,1,1,0) */
int frhip_linear_fwd(int dtype, const void* a, const void* w, const float* bias, void* out, void* act_out,
                     float* stats_partial, int m, int n, int k, frhip_stream_t stream);
/* data-gradient of the Linear that feeds a GELU (Mlp.fc2 backward, nets/SwinV2.py:16-32): dx[m][n] = (dy[m][k] . wt[n][k]) *
 * gelu'(pre[m][n]) with pre = the saved fc1 output; stats_partial (may be NULL, rows as frhip_linear_fwd): [.][0][n] sums to
 * the column sums of dx = the gradient of fc1.bias */
int frhip_linear_dgrad_gelu(int dtype, const void* dy, const void* wt, const void* pre, void* dx, float* stats_partial,
                            int m, int n, int k, frhip_stream_t stream);
/* out[m][n] = sum_k a[m][k]*b[n][k].  atomic_f32 = 0: out has `dtype`, overwritten (splits ignored);
 * atomic_f32 = 1: out is fp32, caller-zeroed, K is split `splits` ways and added atomically.  nn.Linear: nets/resnet.py:244 */
int frhip_gemm_nt(int dtype, const void* a, const void* b, void* out, int m, int n, int k, int splits,
                  int atomic_f32, frhip_stream_t stream);
/* the same product with a deterministic K split (the fc of nets/resnet.py:244 has M = N = 512, K = 25 088): every split
 * stores a private fp32 slab into the caller's workspace, one pass adds them in split order and adds bias[n] (or NULL) */
int frhip_gemm_nt_splitk(int dtype, const void* a, const void* b, const float* bias, float* out, int m, int n, int k,
                         int splits, float* slabs, size_t slab_bytes, frhip_stream_t stream);
/* out[kc][c] (fp32, caller-zeroed) += sum_m p[m][0..kc) * q[m][0..c);  p has row pitch ldp elements. */
int frhip_gemm_tn(int dtype, const void* p, const void* q, float* out, int m, int kc, int ldp, int c,
                  int splits, float* workspace, size_t workspace_bytes, frhip_stream_t stream);
/* the same with out = (not +=): `out` need not be initialised; with a single K split (m <= 512 rows) the tiles are stored
 * plainly -- no zero fill and no atomic read-modify-write pass (the head's dW = dT^T E is 250 MB, nets/PartialFC.py:201) */
int frhip_gemm_tn_overwrite(int dtype, const void* p, const void* q, float* out, int m, int kc, int ldp, int c,
                            float* workspace, size_t workspace_bytes, frhip_stream_t stream);

/* ---- BatchNorm + element-wise glue.  nn.BatchNorm2d/1d: nets/resnet.py:81-86, :187, :196-199 ---- */
int frhip_colreduce_blocks(int rows, int c, int dtype);   /* number of partial rows frhip_colstats / _bn_bwd_reduce write */
int frhip_colstats(int dtype, const void* x, int rows, int c, float* partial, frhip_stream_t stream);
/* partial[nparts][2][c] -> batch mean / invstd, affine scale/shift, running-stat update (running_* may be NULL).
 * scratch: 64*2*c floats. */
int frhip_bn_finalize(const float* partial, int nparts, float* scratch, int c, float count,
                      const float* gamma, const float* beta, float* running_mean, float* running_var,
                      float momentum, float eps, float* mean, float* invstd, float* scale, float* shift,
                      frhip_stream_t stream);
int frhip_bn_eval_affine(int c, const float* gamma, const float* beta, const float* running_mean,
                         const float* running_var, float eps, float* scale, float* shift, frhip_stream_t stream);
/* stand-in BatchNorm state with which the generic BatchNorm-backward reduction over the stem's POOLED map (bn1 -> relu -> maxpool,
 * nets/resnet.py:232-235) yields the stem's reduction: mean = beta, invstd = gamma / (gamma^2 + (k beta)^2 + 1e-20), scale = 1, shift = 0 */
int frhip_bn_standin_state(int c, const float* gamma, const float* beta, float k, float* mean, float* invstd, float* scale,
                           float* shift, frhip_stream_t stream);
/* out = act(y*scale+shift [+ res | + res*res_scale+res_shift]);  BasicBlock tail: nets/resnet.py:92, :96-101 */
int frhip_bn_apply(int dtype, const void* y, const float* scale, const float* shift, const void* res,
                   const float* res_scale, const float* res_shift, int relu, void* out, int rows, int c,
                   frhip_stream_t stream);
/* backward of BN (optionally preceded by the ReLU mask recomputed from y*mask_scale+mask_shift > 0) */
int frhip_bn_bwd_reduce(int dtype, const void* dout, const void* y, const float* mean, const float* invstd,
                        const float* mask_scale, const float* mask_shift, int rows, int c, float* partial,
                        frhip_stream_t stream);
int frhip_bn_bwd_finalize(const float* partial, int nparts, float* scratch, int c, float count,
                          const float* gamma, const float* mean, const float* invstd, float* dgamma,
                          float* dbeta, float* ca, float* cb, float* cc, frhip_stream_t stream);
int frhip_bn_bwd_apply(int dtype, const void* dout, const void* y, const float* ca, const float* cb,
                       const float* cc, const float* mask_scale, const float* mask_shift, void* dy,
                       int rows, int c, frhip_stream_t stream);
/* stochastic depth around a normalised branch (timm DropPath as used by nets/AlterNet_SwinV2_FAN.py:407-450: x + drop_path(norm(f(x)))):
 * rowscale[g] (0 or 1/keep, fp32) multiplies the rows_per consecutive rows of sample g -- out = res + rowscale * (y*scale+shift) in the
 * forward pass, d = dout * rowscale in the BatchNorm-backward reduction and apply.  One pass each instead of separate full-tensor
 * multiply / add launches. */
int frhip_bn_apply_rs(int dtype, const void* y, const float* scale, const float* shift, const void* res, const float* rowscale,
                      int rows_per, void* out, int rows, int c, frhip_stream_t stream);
int frhip_bn_bwd_reduce_rs(int dtype, const void* dout, const void* y, const float* mean, const float* invstd, const float* rowscale,
                           int rows_per, int rows, int c, float* partial, frhip_stream_t stream);
int frhip_bn_bwd_apply_rs(int dtype, const void* dout, const void* y, const float* ca, const float* cb, const float* cc,
                          const float* rowscale, int rows_per, void* dy, int rows, int c, frhip_stream_t stream);
/* out_accum[c] += sum over partial rows of partial[p][which][c]  (fc bias gradient = column sum) */
int frhip_sum_partials(const float* partial, int nparts, int c, int which, float* out_accum, frhip_stream_t stream);
int frhip_add_bias(float* x, const float* bias, int rows, int c, frhip_stream_t stream);
int frhip_cast_from_f32(int dtype, const float* src, void* dst, size_t n, frhip_stream_t stream);
int frhip_cast_to_f32(int dtype, const void* src, float* dst, size_t n, frhip_stream_t stream);

/* ---- stem.  conv1 + bn1 + relu + maxpool: nets/resnet.py:186-189, :232-235 ---- */
/* x NCHW fp32 [b,3,h,w] -> col [b*ho*wo][64 (bf16) | 32 (f32)], k = (r*3+s)*3+ci, zero beyond 27; stride 1
 * (nets/resnet.py:186) or 2 (nets/AlterNet_SwinV2_FAN.py:652) */
int frhip_stem_im2col(int dtype, const float* x, void* col, int b, int h, int w, int stride, frhip_stream_t stream);
int frhip_bn_relu_maxpool_fwd(int dtype, const void* y, const float* scale, const float* shift, void* out,
                              uint8_t* argmax, int b, int h, int w, int c, frhip_stream_t stream);
int frhip_maxpool_bwd(int dtype, const void* dpool, const uint8_t* argmax, void* da, int b, int h, int w,
                      int c, frhip_stream_t stream);

/* ---- operand packs ---- */
int frhip_pack_wt(int dtype, const float* w, void* wt, int k, int rs, int c, frhip_stream_t stream);
/* out[cols][ld_out] = in[rows][cols]^T, ld_out >= rows, pad columns zero-filled */
int frhip_transpose2d(int dtype_in, int dtype_out, const void* in, void* out, int rows, int cols, int ld_out,
                      frhip_stream_t stream);
int frhip_pack_stem(int dtype, const float* w, void* wp, int k, int kin, int kp, frhip_stream_t stream);
int frhip_unpack_stem_grad(const float* dwp, float* dw, int k, int kin, int kp, frhip_stream_t stream);
/* fc weight columns: reference flattens NCHW (x.view(B,-1), nets/resnet.py:243); this backbone is NHWC */
int frhip_fc_permute(int dtype, const float* w, void* wp, int nout, int c, int hw, frhip_stream_t stream);
int frhip_fc_unpermute_grad(const float* dwp, float* dw, int nout, int c, int hw, frhip_stream_t stream);
/* PartialFC sampled rows: weight[index] gather (nets/PartialFC.py:120-121) / write-back (:142-143) */
/* PartialFC.sample (nets/PartialFC.py:108-121) + the shard-relative labels (:188-193) in one launch: labels[n] int64 GLOBAL class ids
 * of the gathered batch, u[num_local] the uniform draws (torch.rand on the CPU generator, :110), num_sample rows to keep.  Writes
 * *n_positive = distinct classes of this shard in the batch; if that is <= num_sample: index_out[num_sample] = the sampled rows
 * ascending (all positives + the rows with the largest draws; equal draws at the cut: lowest row first) and rel_out[n] = position of
 * the label's class in index_out, or -1 when another rank owns it.  If n_positive > num_sample (reference: `index = positive`, another
 * output length) n_positive is written, rel_out = -1 and index_out = 0 .. num_sample-1 (valid rows, to be discarded), and the caller
 * takes that branch itself.  num_local <=
 * frhip_pfc_sample_max_local(). */
int frhip_pfc_sample_max_local(void);
int frhip_pfc_sample(const int64_t* labels, int n, long long class_start, int num_local, const float* u, int num_sample,
                     int64_t* index_out, int* rel_out, int64_t* n_positive, frhip_stream_t stream);
int frhip_gather_rows(const float* src, const int64_t* index, float* dst, int n, int d, frhip_stream_t stream);
int frhip_scatter_rows(const float* src, const int64_t* index, float* dst, int n, int d, frhip_stream_t stream);

/* ---- margin-softmax head.  nets/PartialFC.py:198-207, nets/ArcFace.py:76-91, nets/PartialFC.py:441-484 ---- */
/* Head: gradient of the class centres in ONE launch (autograd of F.normalize(weight) + F.linear, nets/PartialFC.py:464-484):
 * dw[c][:] = (g[c][:] - what[c][:] * <g[c], what[c]>) * out_scale / wnorm[c] with g = dT^T ehat (dT [n][ldt] from frhip_head_bwd_dt,
 * ehat [n][512], what [classes][512] all `dtype`; wnorm, dw fp32) -- replaces frhip_gemm_tn_overwrite + frhip_l2norm_bwd, whose
 * intermediate g (250 MB at 122 000 classes) crossed HBM twice.  bf16 and d == 512 only (frhip_head_dw_ok). */
int frhip_head_dw_ok(int dtype, int n, int classes, int d);
int frhip_head_dw(int dtype, const void* dt, int ldt, const void* ehat, const void* what, const float* wnorm, float* dw,
                  int n, int classes, int d, float out_scale, frhip_stream_t stream);
/* F.normalize rows (also model/FR_PartialFC.py:171) */
int frhip_l2norm_rows(int dtype, const float* x, void* xhat, float* norms, int rows, int d, float eps,
                      frhip_stream_t stream);
int frhip_l2norm_bwd(int dtype, const float* dxhat, const void* xhat, const float* norms, float* dx,
                     int rows, int d, float out_scale, frhip_stream_t stream);
int frhip_head_groups(int num_classes);     /* rows of part_max / part_sum */
/* fused normalised-GEMM -> clamp -> ArcFace margin -> x s -> per-row max & sum-exp (this shard);
 * labels: int32 shard-relative, -1 = another shard owns the class. */
int frhip_head_fwd(int dtype, const void* ehat, const void* what, const int* labels, int n, int classes,
                   int d, float s, float m, float* part_max, float* part_sum, float* ztarget,
                   float* rowmax, float* rowsum, frhip_stream_t stream);
int frhip_head_rescale(float* rowsum, const float* local_max, const float* global_max, int n, frhip_stream_t stream);
int frhip_head_target_prob(const float* ztarget, const int* labels, const float* rowmax, const float* rowsum,
                           float* q, int n, frhip_stream_t stream);
/* the three per-row all-reduces of the distributed CE (nets/PartialFC.py:448 MAX, :453 SUM, :459 SUM) as ONE all-gather:
 * every rank packs {local max, local sum-exp, target logit | -inf} per row ([n][3] fp32), the blocks of all ranks are
 * all-gathered ([world_size][n][3]) and merged in rank order -> global max, global sum-exp, target probability q */
int frhip_head_pack_stats(const float* ztarget, const int* labels, const float* rowmax, const float* rowsum,
                          float* packed, int n, frhip_stream_t stream);
int frhip_head_merge_stats(const float* gathered, int world_size, int n, float* rowmax, float* rowsum, float* q,
                           frhip_stream_t stream);
int frhip_head_loss(const float* q, int n, float* loss, frhip_stream_t stream);
/* dT[n][ldt] = d loss / d cos (after clamp/margin/scale chain rule); gscale = 1 / N_global, times the device
 * scalar *upstream when it is not NULL (the reference syncs the host here: loss_gradient.item(), nets/PartialFC.py:484).
 * dtt (may be NULL): the same matrix class-major, dtt[classes][ldtt] (ldtt >= n, a multiple of the 16-byte vector; columns >= n zero),
 * which the embedding-gradient GEMM (contraction over classes) reads -- written from the same tiles, no transpose pass */
int frhip_head_bwd_dt(int dtype, const void* ehat, const void* what, const int* labels, int n, int classes,
                      int d, float s, float m, const float* rowmax, const float* rowsum, float gscale,
                      const float* upstream, void* dt, int ldt, void* dtt, int ldtt, frhip_stream_t stream);

/* ---- bn1 -> relu -> conv2 of a BasicBlock (nets/resnet.py:91-93) WITHOUT the activated tensor: the BatchNorm-apply + ReLU
 * is folded into the operand path of the convolution (forward) and of its weight gradient, which read the saved BatchNorm
 * INPUT x and form relu(x * in_scale[c] + in_shift[c]) in LDS.  bf16, 3x3 / stride 1 / pad 1; *_fusable() tells whether a
 * shape is served (otherwise use frhip_bn_apply + the plain entry points).  Results are bit-identical to the unfused pair. ---- */
int frhip_conv_bnrelu_fusable(int dtype, int h, int wd, int c, int k, int r, int s, int stride, int pad);
/* act_out (may be NULL): [n,h,wd,c] like x -- the activated tensor relu(x * in_scale + in_shift) is written there on the way (each
 * row once, by the workgroups of output-channel column 0) for a backward pass that wants it: the forward pass then has neither
 * the BatchNorm-apply launch nor its re-read of x, and the backward pass is the unfused one */
int frhip_conv_fwd_bnrelu(int dtype, const void* x, const float* in_scale, const float* in_shift, const void* w, void* y,
                          float* stats_partial, void* act_out, int n, int h, int wd, int c, int k, int r, int s, int stride,
                          int pad, frhip_stream_t stream);
int frhip_conv_wgrad_bnrelu_fusable(int dtype, int n, int h, int w, int c, int k, int r, int s, int stride, int pad);
int frhip_conv_wgrad_bnrelu(int dtype, const void* dy, const void* x, const float* in_scale, const float* in_shift,
                            float* dw, int n, int h, int w, int c, int k, int r, int s, int stride, int pad, int splits,
                            float* workspace, size_t workspace_bytes, frhip_stream_t stream);

/* ---- fp8 weight path (BASELINE cfg 5; the reference has no fp8 arithmetic, see csrc/igemm_fp8.hip).  Forward convolutions
 * (nets/AlterNet_SwinV2_FAN.py:520-568, nets/resnet.py:23-46) and linears (:263-302) with both MFMA operands in OCP fp8
 * e4m3 on v_mfma_scale_f32_16x16x128_f8f6f4; bf16 output = fp32 accumulator x act_scale x w_scale[k] ---- */
/* w [k][rowlen] fp32 (a conv weight in its physical [K][R][S][C] order) -> w8 fp8 + scale[k] = amax(row) / 448 */
int frhip_quant_fp8_weights(const float* w, void* w8, float* scale, int k, int rowlen, frhip_stream_t stream);
/* the same for all weight tensors of a step in ONE launch (the weights change every optimizer step, model/FR_PartialFC.py:166-170):
 * row_begin = number of output rows of the tensors before this one, nrows = the total; rowlen % 4 == 0.  Table in device memory */
typedef struct { const float* w; void* w8; float* scale; int32_t k, rowlen, row_begin, pad_; } frhip_q8w;
int frhip_quant_fp8_weights_multi(const frhip_q8w* table, int ntensors, int nrows, frhip_stream_t stream);
/* debug counter of SATURATED activations in the fp8 quantisers (frhip_quant_fp8, frhip_bn_apply_q8 clamp at +-448 silently; the
 * activation scale is static): op 1 zeroes and arms it, op 0 disarms, op 2 returns the number of elements that exceeded e4m3's range
 * since it was armed (synchronises the device; capped at INT_MAX).  Process-global test hook, outside the threading contract */
int frhip_fp8_saturation(int op);
/* x8 = fp8(x * inv_scale), n % 16 == 0 */
int frhip_quant_fp8(int dtype, const void* x, void* x8, size_t n, float inv_scale, frhip_stream_t stream);
/* frhip_bn_apply that also writes the fp8 copy out8 = fp8(out * inv_q) the next fp8 GEMM reads (bf16 tensors) */
int frhip_bn_apply_q8(int dtype, const void* y, const float* scale, const float* shift, const void* res,
                      const float* res_scale, const float* res_shift, int relu, void* out, void* out8, float inv_q,
                      int rows, int c, frhip_stream_t stream);
/* y bf16 [n,ho,wo,k] = conv(x8 [n,h,w,c] fp8, w8 [k,r,s,c] fp8) * act_scale * wscale[k]; c % 128 == 0;
 * stats_partial: [frhip_fp8_stat_rows(n*ho*wo, k)][2][k] BatchNorm partial sums of the stored values, or NULL */
int frhip_conv_fwd_fp8(const void* x8, const void* w8, const float* wscale, float act_scale, void* y,
                       float* stats_partial, int n, int h, int wd, int c, int k, int r, int s, int stride, int pad,
                       frhip_stream_t stream);
int frhip_fp8_stat_rows(int m, int k);                                               /* linears / generic kernel */
int frhip_fp8_conv_stat_rows(int m, int k, int h, int w, int c, int r, int s, int stride, int pad);   /* convolutions */
int frhip_set_fp8_halo(int enabled);     /* test hook: 3x3 stride-1 fp8 convs on the LDS-halo kernel (1, default) or the generic one (0); < 0: query */
/* out bf16 [m][n] = a8 [m][k] x w8 [n][k]^T * act_scale * wscale[n] + bias[n] */
int frhip_linear_fwd_fp8(const void* a8, const void* w8, const float* wscale, float act_scale, const float* bias,
                         void* out, float* stats_partial, int m, int n, int k, frhip_stream_t stream);

/* ---- continuous position bias of the window attention (nets/SwinV2.py:88-125, :150-158; nets/AlterNet_SwinV2_FAN.py:210-283):
 * bias[h][i][j] = 16 sigmoid(cpb_mlp(coords)[index[i][j]][h]), scale[h] = exp(min(logit_scale[h], ln 100)), and the gradients of
 * cpb_mlp.0.weight / .0.bias / .2.weight / logit_scale, for ALL attention blocks of a step in one launch each way.
 * One descriptor per block (device array); every pointer is a device address, fp32 unless noted. ---- */
typedef struct frhip_cpb_block {
    const void* coords;        /* [entries][2]  relative_coords_table */
    const void* index;         /* int64 [tokens*tokens]  relative_position_index */
    const void* w0;            /* [512][2] */
    const void* b0;            /* [512] */
    const void* w2;            /* [heads][512] */
    const void* logit_scale;   /* [heads] */
    void* bias;                /* out: [heads][tokens][tokens] */
    void* scale;               /* out: [heads] */
    const void* dbias;         /* backward in: [heads][tokens][tokens] */
    const void* dscale;        /* backward in: [heads] */
    void* dw0;                 /* backward, accumulated into: same shapes as w0 / b0 / w2 / logit_scale */
    void* db0;
    void* dw2;
    void* dlogit_scale;
    int entries, tokens, heads, pad_;
} frhip_cpb_block;
int frhip_cpb_limits(int* max_entries, int* max_heads, int* hidden);
int frhip_cpb_scratch_floats(int nblocks);      /* size of the caller-owned scratch both calls need (keep it from fwd to bwd: not required) */
int frhip_cpb_fwd(const frhip_cpb_block* blocks_dev, int nblocks, float* scratch, frhip_stream_t stream);
int frhip_cpb_bwd(const frhip_cpb_block* blocks_dev, int nblocks, float* scratch, frhip_stream_t stream);

/* ---- explicit-logit margin and softmax-CE (stand-alone use of nets/ArcFace.py:76-105, nets/PartialFC.py:441-484) ---- */
/* logits[n][c] fp32 in place: target entry -> margin (kind 0 ArcFace, 1 CosFace), everything x s; tsave[n] = raw target cos.
 * labels int64 [n], -1 = no target in this shard */
int frhip_margin_fwd(float* logits, const int64_t* labels, int n, int c, float s, float m, int kind, float* tsave,
                     frhip_stream_t stream);
int frhip_margin_bwd(const float* gout, const int64_t* labels, const float* tsave, int n, int c, float s, float m,
                     int kind, float* gin, frhip_stream_t stream);
int frhip_rows_max(const float* x, int n, int c, float* rowmax, frhip_stream_t stream);
int frhip_rows_exp_sum(float* x, int n, int c, const float* rowmax, float* rowsum, frhip_stream_t stream);
int frhip_rows_normalize(float* x, int n, int c, const float* rowsum, const int64_t* labels, float* ptarget,
                         frhip_stream_t stream);
int frhip_ce_grad(float* p, int n, int c, const int64_t* labels, float inv_n, const float* upstream,
                  frhip_stream_t stream);

/* ---- SwinV2 window attention (ws x ws windows, ws <= 7, head dim 32).  nets/SwinV2.py:139-179 and
 * nets/AlterNet_SwinV2_FAN.py:263-302; window_partition/reverse, the cyclic roll and the SW-MSA mask
 * (nets/AlterNet_SwinV2_FAN.py:375-397, :420-440) are index arithmetic inside the kernel ---- */
/* qkv [b*h*w][3c] (pixel order), bias fp32 [heads][n][n] (n = ws*ws) = 16*sigmoid(cpb table)[index], scale fp32 [heads] =
 * exp(min(logit_scale, ln 100)); shift = 0 (W-MSA) or ws/2 (SW-MSA); out [b*h*w][c] */
int frhip_winattn_fwd(int dtype, const void* qkv, const float* bias, const float* scale, void* out, int b, int h,
                      int w, int c, int heads, int ws, int shift, frhip_stream_t stream);
/* dqkv [b*h*w][3c]; dbias [heads][n][n] and dscale [heads] are fp32, caller-zeroed, accumulated atomically */
int frhip_winattn_bwd(int dtype, const void* qkv, const void* dout, const float* bias, const float* scale,
                      void* dqkv, float* dbias, float* dscale, int b, int h, int w, int c, int heads,
                      int ws, int shift, frhip_stream_t stream);
/* frhip_winattn_bwd that also accumulates dqkv_colsum[3c] (fp32, caller-zeroed) += column sums of the stored dqkv = the
 * gradients of q_bias / v_bias (nets/SwinV2.py:150-154).  bf16 with the MFMA kernels only; FRHIP_EINVAL otherwise */
int frhip_winattn_bwd_colsum(int dtype, const void* qkv, const void* dout, const float* bias, const float* scale,
                             void* dqkv, float* dbias, float* dscale, float* dqkv_colsum, int b, int h, int w, int c,
                             int heads, int ws, int shift, frhip_stream_t stream);
/* frhip_winattn_bwd_colsum with the q third of the column sums added into dq_bias[c] and the v third into dv_bias[c] (fp32
 * gradient accumulators of q_bias / v_bias, nets/SwinV2.py:150-154; either may be NULL); the k third is not formed */
int frhip_winattn_bwd_qvbias(int dtype, const void* qkv, const void* dout, const float* bias, const float* scale,
                             void* dqkv, float* dbias, float* dscale, float* dq_bias, float* dv_bias, int b, int h, int w,
                             int c, int heads, int ws, int shift, frhip_stream_t stream);
/* 1 (default): bf16 calls run the MFMA-tile kernels (bf16 GEMM operands, fp32 scores / softmax -- the reference's autocast
 * numerics); 0: the fp32-arithmetic VALU kernels for every dtype.  Negative: query.  Returns the old value */
int frhip_set_winattn_mfma(int enabled);
/* y[rows][c] += bias (in place); act_out (may be NULL) = gelu(y).  Mlp fc1 + GELU: nets/SwinV2.py:16-32 */
int frhip_bias_gelu_fwd(int dtype, void* y, const float* bias, void* act_out, int rows, int c, frhip_stream_t stream);
int frhip_gelu_bwd(int dtype, const void* da, const void* h, void* dh, size_t n, frhip_stream_t stream);

/* ---- verification metrics: utils/eval.py:68-99 pair_score.  scores[n] float64, hist_idx[n] = int(99999*score),
 * hist_genuine / hist_imposter int32[100001] (caller-zeroed) ---- */
int frhip_pair_score(const float* e1, const float* e2, const int64_t* labels, int n, int d, double* scores,
                     int* hist_idx, int* hist_genuine, int* hist_imposter, frhip_stream_t stream);
/* utils/eval.py:102-137 cross_score: all pairs j < i of ONE embedding set e [n][d] in the reference's order
 * l = i (i - 1) / 2 + j; scores / pair_labels float64 [n (n - 1) / 2] (pair label 1.0 = same identity), hist_idx int32 */
int frhip_cross_score(const float* e, const int64_t* labels, int n, int d, double* scores, double* pair_labels,
                      int* hist_idx, int* hist_genuine, int* hist_imposter, frhip_stream_t stream);

/* ---- recompute-style stem (stride 1): conv3x3(3->64) -> BN -> ReLU -> MaxPool(3,2,1) of nets/resnet.py:232-235 without ever
 * writing the conv output map or an im2col matrix; every pass recomputes the conv from x [b,3,h,w] fp32 NCHW and
 * wp = frhip_pack_stem(w, 64, 27, 32) ([64][32] in `dtype`).  frhip_stem_blocks(b,h,w) = partial rows / slabs the
 * grid-stride kernels write.
 *   stats      : partial[blocks][2][64] = per-workgroup { sum y, sum y^2 }            -> frhip_bn_finalize
 *   fwd        : pooled [b,hp,wp,64] `dtype`, argmax uint8 [b,hp,wp,64] (first max in row-major window order)
 *   bwd_reduce : partial[blocks][2][64] = { sum d, sum d*(y-mean)*invstd }, d = maxpool_bwd(dpool) * (y*scale+shift > 0)
 *                                                                                     -> frhip_bn_bwd_finalize
 *   bwd_wgrad  : dy = ca*d + cb*y + cc; dw[64][27] (fp32) += sum_pixels dy x im2col; slabs: blocks*64*32 floats of scratch */
int frhip_stem_blocks(int b, int h, int w);
/* test hook: 0 = gather-form backward kernels for bf16 as well (fp32 always uses them); returns the old value */
int frhip_set_stem_scatter(int enabled);
int frhip_stem_stats(int dtype, const float* x, const void* wp, int b, int h, int w, float* partial, frhip_stream_t stream);
int frhip_stem_fwd(int dtype, const float* x, const void* wp, const float* scale, const float* shift, void* pooled,
                   uint8_t* argmax, int b, int h, int w, frhip_stream_t stream);
int frhip_stem_bwd_reduce(int dtype, const float* x, const void* wp, const void* dpool, const uint8_t* argmax,
                          const float* mean, const float* invstd, const float* scale, const float* shift, int b, int h,
                          int w, float* partial, frhip_stream_t stream);
int frhip_stem_bwd_wgrad(int dtype, const float* x, const void* wp, const void* dpool, const uint8_t* argmax,
                         const float* ca, const float* cb, const float* cc, const float* scale, const float* shift, int b,
                         int h, int w, float* slabs, float* dw, frhip_stream_t stream);
/* The same weight gradient without recomputing the convolution (csrc/stem_algebra.hip): dy = ca d + cb y + cc and y = W col give
 * dW = ca D + cb (W G) + cc s with G = sum_p col col^T (27 x 27) and s = sum_p col, functions of the input batch alone, and
 * D = sum over pooled elements with pooled > 0 of dpool x col[arg-max pixel].  frhip_stem_gram fills gram[frhip_stem_gram_floats()]
 * = {G, s} from x (partial: scratch of (frhip_stem_gram_blocks(b,h,w) + 1) x 567 floats); it needs nothing else of the step, so a caller
 * can run it on a second stream during the forward pass.  frhip_stem_bwd_wgrad_gram then needs dpool, the pooled map the forward
 * pass returned (ReLU mask: pooled > 0), the arg-max bytes and (ca, cb, cc) of frhip_bn_bwd_finalize; slabs as above. */
int frhip_stem_gram_floats(void);
int frhip_stem_gram_blocks(int b, int h, int w);
int frhip_stem_gram(int dtype, const float* x, int b, int h, int w, float* partial, float* gram, frhip_stream_t stream);
int frhip_stem_bwd_wgrad_gram(int dtype, const float* x, const void* wp, const void* dpool, const void* pooled,
                              const uint8_t* argmax, const float* gram, const float* ca, const float* cb, const float* cc,
                              int b, int h, int w, float* slabs, float* dw, frhip_stream_t stream);

/* ---- per-step operand preparation of ALL conv weights in one launch (the reference casts them implicitly under autocast,
 * nets/resnet.py:23-46 + model/FR_PartialFC.py:166-170).  Per tensor w[k][rs][c] fp32: wc[k][rs][c] (forward operand) and
 * wt[c][rs][k] (data-gradient operand) in `dtype`; either destination may be NULL.  tile_begin = number of 32x32 tiles
 * (ceil(k/32)*ceil(c/32)*rs per tensor) of the tensors before this one; ntiles = the total.  Table in device memory. */
typedef struct { const float* w; void* wc; void* wt; int32_t k, rs, c, tile_begin; } frhip_wprep;
int frhip_prep_conv_weights(int dtype, const frhip_wprep* table, int ntensors, int ntiles, frhip_stream_t stream);

/* Dropout mask of the backbone tails (nn.Dropout() before fc: nets/SwinV2.py:559, nets/AlterNet_SwinV2_FAN.py:743): mask[i] = 1/keep
 * with probability keep, else 0, in `dtype`; Philox-4x32-10, key = seed, counter = i / 4 (same seed -> same mask).  The caller draws
 * the seed from torch's generator, so torch.manual_seed still pins a run */
int frhip_dropout_mask(int dtype, void* mask, size_t n, float keep, long long seed, frhip_stream_t stream);

/* ---- device input pipeline: Resize -> HorizontalFlip -> Normalize(0.5,0.5) -> CoarseDropout -> CHW fp32 of the reference's
 * albumentations chain (utils/data_partial.py:134-164) in one kernel.  in: uint8 [b,hin,win,3] (HWC, device);
 * out: fp32 [b,3,size,size]; flip: int32 [b] (may be NULL); holes: int32 [b][nholes][4] = x1,y1,x2,y2 in output
 * coordinates, exclusive upper bounds, x2 <= x1 marks an unused slot (may be NULL when nholes == 0).  Dropped pixels are 0
 * (CoarseDropout fill_value 0 after Normalize).  Resize = OpenCV INTER_LINEAR (8-bit fixed point), identity when sizes match. */
int frhip_augment_u8(const uint8_t* in, float* out, const int32_t* flip, const int32_t* holes, int nholes,
                     int b, int hin, int win, int size, frhip_stream_t stream);
/* the same with RandomGamma in front (utils/data_partial.py:137-138): lut uint8 [b][256] (may be NULL), the per-image table
 * ((i / 255) ** gamma * 255 truncated to uint8, gamma = uniform(gamma_limit) / 100) applied to the source bytes before Resize */
int frhip_augment_u8_lut(const uint8_t* in, const uint8_t* lut, float* out, const int32_t* flip, const int32_t* holes,
                         int nholes, int b, int hin, int win, int size, frhip_stream_t stream);
/* alb.MotionBlur of the chain (utils/data_partial.py:139-140) on the uint8 HWC batch, before Resize: out = cv2.filter2D(in, -1,
 * kernel) per image -- correlation, anchor at the centre, BORDER_REFLECT_101, float accumulation, round-half-even + saturate.
 * kernels float [b][7][7]: the k x k line kernel (k = ksize[n] in {3, 5, 7}) zero-padded with its centre at (3, 3); ksize int32 [b],
 * 0 = image copied (ksize NULL: all copied).  lut (may be NULL): RandomGamma tables applied to the source bytes first.
 * out must not alias in. */
int frhip_motion_blur_u8(const uint8_t* in, uint8_t* out, const uint8_t* lut, const float* kernels, const int32_t* ksize,
                         int b, int h, int w, frhip_stream_t stream);
/* alb.ISONoise of the chain (utils/data_partial.py:142-143; albumentations functional.iso_noise) on the uint8 HWC batch:
 * RGB/255 -> HLS; hue += N(0, color_shift * 360 * intensity) wrapped to [0, 360]; L += Poisson(std(L) * intensity * 255) / 255 * (1 - L);
 * -> RGB * 255 truncated to uint8.  params float [b][2] = (color_shift, intensity), intensity <= 0: image copied.
 * Draws: lum_noise int32 [b][h][w] + color_noise float [b][h][w] (already scaled) when given (tests), else generated on the device
 * from seeds uint64 [b] (Philox-4x32-10, counter = pixel).  scratch: frhip_iso_noise_scratch_doubles(b) doubles. */
int frhip_iso_noise_scratch_doubles(int b);
int frhip_iso_noise_u8(const uint8_t* in, uint8_t* out, double* scratch, const float* params, const int32_t* lum_noise,
                       const float* color_noise, const uint64_t* seeds, int b, int h, int w, frhip_stream_t stream);

/* ---- optimizer: torch.optim.SGD(momentum, weight_decay).step() + torch.nn.utils.clip_grad_norm_ of the training step
 * (model/FR_PartialFC.py:153-160, :181-190) as multi-tensor kernels.  A chunk is a run of at most
 * FRHIP_SGD_CHUNK consecutive fp32 elements of one parameter (p), its gradient (g) and its momentum buffer (m, may be
 * NULL when momentum == 0); the chunk table lives in device memory, the (<= 8) parameter groups are passed by value.
 * d = g*coef + weight_decay*p;  m = momentum*m + d;  p -= lr*m   with coef = *clip_coef for groups with clip != 0. */
#define FRHIP_SGD_CHUNK 65536
#define FRHIP_SGD_MAX_GROUPS 8
typedef struct { float lr, weight_decay, momentum, clip; } frhip_sgd_group;
typedef struct { float* p; const float* g; float* m; uint32_t n; uint32_t group; } frhip_sgd_chunk;
/* coef_out[0] = min(1, max_norm / (||g|| + 1e-6)) over the gradients of every group with clip != 0, coef_out[1] = ||g||;
 * partial: nchunks floats of scratch */
int frhip_sgd_clip_coef(const frhip_sgd_chunk* chunks, int nchunks, const frhip_sgd_group* groups_host, int ngroups,
                        float max_norm, float* partial, float* coef_out, frhip_stream_t stream);
int frhip_sgd_multi(const frhip_sgd_chunk* chunks, int nchunks, const frhip_sgd_group* groups_host, int ngroups,
                    const float* clip_coef, frhip_stream_t stream);

/* torch.optim.AdamW(betas, eps, weight_decay) (amsgrad off), same table scheme (model/FR_PartialFC.py:153-160 'AdamW' branch;
 * nets/PartialFC.py:235-342 swaps exp_avg / exp_avg_sq rows in and out): bc1 = 1 - beta1^step, bc2 = 1 - beta2^step of the
 * step being taken, computed by the caller per group.
 * p *= 1 - lr*wd; m = b1*m + (1-b1)*g; v = b2*v + (1-b2)*g*g; p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps) */
typedef struct { float lr, beta1, beta2, eps, weight_decay, bc1, bc2, clip; } frhip_adamw_group;
typedef struct { float* p; const float* g; float* m; float* v; uint32_t n; uint32_t group; } frhip_adamw_chunk;
int frhip_adamw_clip_coef(const frhip_adamw_chunk* chunks, int nchunks, const frhip_adamw_group* groups_host, int ngroups,
                          float max_norm, float* partial, float* coef_out, frhip_stream_t stream);
int frhip_adamw_multi(const frhip_adamw_chunk* chunks, int nchunks, const frhip_adamw_group* groups_host, int ngroups,
                      const float* clip_coef, frhip_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
