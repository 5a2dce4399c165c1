/* centernet_uda_hip.h -- C ABI of libcenternet_uda_hip.so (MI355X / gfx950).
 *
 * The drop-in boundary for the CenterNet-UDA hot path.  Plain pointers, sizes
 * and a HIP stream; no torch types.  All tensors are contiguous fp32 NCHW in
 * device memory unless a comment says otherwise; index tensors are int64,
 * masks uint8 (the batch schema of datasets/coco.py:168-174,242-251).  All work
 * is enqueued on `stream` (a hipStream_t passed as void*; NULL = the default
 * stream) and nothing synchronises the host -- the reference likewise enqueues
 * on the current stream (libs/DCNv2/src/cuda/dcn_v2_cuda.cu:107,139).
 *
 * Every function returns 0 on success, CNUDA_ERR_INVALID_ARGUMENT (-1) for a
 * rejected argument, or a positive hipError_t for a failed launch;
 * cnuda_last_error() describes the most recent failure on the calling thread.
 * The reference raises C++ exceptions -> Python RuntimeError for the first
 * kind (dcn_v2_cuda.cu:60-64,80-84) and only printf's the second
 * (dcn_v2_im2col_cuda.cu:346-350); the Python shim in
 * centernet-uda_amd/hip_runtime raises RuntimeError for both.
 *
 * Scratch memory is caller-owned: each family has a *_workspace_bytes() query
 * and takes (workspace, workspace_bytes); nothing is allocated or freed inside
 * a call, so every entry point is legal inside hipGraph capture.
 */
#ifndef CENTERNET_UDA_HIP_H
#define CENTERNET_UDA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CNUDA_ERR_INVALID_ARGUMENT (-1)
#define CNUDA_ABI_VERSION 2

typedef void* cnuda_stream_t; /* hipStream_t */

int cnuda_abi_version(void);
const char* cnuda_last_error(void);

/* Matrix-pipe mode of the convolution / DCN implicit GEMMs (process-wide; no counterpart in the reference,
 * whose cuDNN / cuBLAS calls pick their own algorithm, backends/dla.py:26,34 and dcn_v2_cuda.cu:95-110):
 *   0  f32 MFMA: exact f32 products and f32 accumulation (default);
 *   1  every f32 operand is cut exactly into three bf16 pieces and the product is formed from the six leading
 *      piece products on the bf16 MFMA with f32 accumulation (error per product below 2^-23, one f32 rounding).
 * Initial value from the environment variable CNUDA_MATRIX_MODE.  Returns 0 or CNUDA_ERR_INVALID_ARGUMENT. */
int cnuda_set_matrix_mode(int mode);
int cnuda_get_matrix_mode(void);

/* Pack cache, no counterpart in the reference (cuDNN / cuBLAS keep their own operand layouts).  The implicit GEMMs
 * read the small weight operand from a zero-padded [Kp][Mp] image that used to be rebuilt by a tiny kernel on every
 * convolution call although it only changes when the weights do.  cnuda_pack_cache_attach hands the library ONE
 * caller-owned device arena (nothing is allocated inside the library; NULL detaches and forgets every slot).
 * cnuda_pack_stamp(token, version), called on the calling thread right before a convolution / DCN entry point,
 * names the weights that call will pack: `token` identifies their owner (0 = anonymous: never cached), `version`
 * changes whenever their values may have changed.  A slot is reused iff the same token packed the same source
 * buffer to the same image before and the version is unchanged; the stamp stays in force until the next
 * cnuda_pack_stamp on that thread (callers reset it to (0, 0) after the call).  cnuda_pack_cache_used: bytes taken;
 * cnuda_pack_cache_fills: how often a cached image was (re)written on a convolution's own call (steady state: 0 per call). */
int cnuda_pack_cache_attach(void* arena, size_t bytes);
int cnuda_pack_stamp(unsigned long long token, unsigned long long version);
size_t cnuda_pack_cache_used(void);
unsigned long long cnuda_pack_cache_fills(void);
/* Eviction: slots of modules that no longer exist are never returned one by one; when an image does not fit any more
 * the cache forgets EVERY slot at the start of the next stamped call (the live modules re-pack once) instead of
 * silently not caching for the rest of the process.  cnuda_pack_cache_resets: how often that happened. */
unsigned long long cnuda_pack_cache_resets(void);
/* After an optimizer step that rewrote [params, params + params_bytes) behind the callers' version counters: every
 * cached image whose source lies in that range and that was current in epoch `old_epoch` (the high 32 bits of the
 * version it was stamped with) is rebuilt by ONE launch on `stream` and re-stamped with `new_epoch` (low 32 bits
 * kept).  `table`: >= 96 bytes per cached image of device scratch that stays valid until the next call (the job
 * list; rewritten only when the set of images changes).  No reference counterpart. */
int cnuda_pack_refresh(const void* params, size_t params_bytes, unsigned long long old_epoch,
                       unsigned long long new_epoch, void* table, size_t table_bytes, cnuda_stream_t stream);

/* Measurement aid (bench.py roofline leg), not part of the reference's surface:
 * cnuda_prof_enable(n) pre-creates n hipEvent pairs; cnuda_prof_arm(tag) makes
 * the NEXT convolution / DCN main-kernel launch record a start/stop pair on its
 * own launch stream; cnuda_prof_collect synchronises on the recorded events and
 * returns (tag, milliseconds, kernel name) triples -- the name is written by the
 * launcher itself (the kernel template instance it selected for that call), in
 * `names` as cap strings of cnuda_prof_name_len() bytes (nullable).  An entry point
 * made of several timed kernels reports them under one tag, told apart by bits 24..
 * of the tag.  Costs nothing when not armed. */
int cnuda_prof_enable(int max_records);
int cnuda_prof_arm(int tag);
int cnuda_prof_collect(int* tags, float* ms, char* names, int cap);
int cnuda_prof_name_len(void);

/* Test aids (tests/test_zz_kernel_coverage.py, tests/test_gpu_dcn.py), not part of the reference's surface.
 * Launch log: while enabled, every kernel launch of the library is counted per kernel; cnuda_launch_log_collect
 * writes one line "<kernel symbol name, demangled, as a rocprofv3 kernel trace shows it>\t<launches so far>" per
 * distinct kernel, in first-launch order, into names[cap] and returns their count (callers diff two snapshots to
 * learn what ran in between).  enable(1) from the disabled state clears the counts.
 * cnuda_dcn_set_fused_min_tiles: cnuda_dcn_v2_backward runs its two data-gradient walks as ONE launch
 * (dcn_bwd_data_kernel + geometry records from dcn_prep_kernel) when the call has at least this many
 * (image, 256-pixel tile) pairs -- 512 by default: the 128 x 128 / 64 x 64 maps of the benched step -- and as two
 * kernels below it.  Tests set 1 to run small geometries (borders, odd widths, out-of-bounds samples) through the
 * one-launch form, and INT_MAX to force the two-kernel form; values < 1 restore the default.  Returns the previous
 * threshold.  Process-wide; results do not depend on it beyond the summation order inside grad_input.
 * cnuda_conv_set_halo_policy: which 3x3 / stride-1 / padding-1 convolutions take the halo-tile kernels (csrc/hconv.cuh):
 * level 0 none, 1 every eligible layer, 2 the 32-row GEMMs (default; initial value from CNUDA_HCONV), and only calls
 * with at least min_tiles 128-pixel tiles (default 128).  Negative level / min_tiles < 1 leave that setting unchanged.
 * Returns level | min_tiles << 8 after the change.  Tests use (1, 1) to run small shapes through every tile variant. */
int cnuda_launch_log_enable(int on);
int cnuda_launch_log_collect(char* names, size_t cap);
int cnuda_dcn_set_fused_min_tiles(int min_tiles);
/* cells of the data-gradient walk's LDS window beyond the undeformed 3x3 footprint (default 2; < 1 restores it).  Returns the
 * previous value.  Samples further out than this are strays (global atomics): a wider window for models whose offsets are
 * pixels, at fewer workgroups per CU.  Measurements and tests; results do not depend on it beyond summation order. */
int cnuda_dcn_set_scatter_margin(int margin);
/* columns of the data-gradient walk's pixel tile (16, 32 or 64; 256 pixels per tile; 0 = by the map width, the default).
 * Returns the previous value.  Measurements only; results do not depend on it beyond summation order. */
int cnuda_dcn_set_walk_tile(int tile_cols);
/* Offset regime of the next cnuda_dcn_v2_* calls (process-wide, the host layer sets it per call): bit 0 -> the data-gradient
 * walk's window takes a margin of 4 cells (many samples beyond +-2 px), bit 1 -> the forward takes the gathering loader
 * instead of the LDS-window kernel (many samples beyond +-3 px).  Returns the previous value.  Speed only: results do not
 * depend on it beyond summation order.  cnuda_dcn_offset_census adds to counts[0] / counts[1] (caller-zeroed) the number
 * of (pixel, tap) samples of `offset` [B][2*taps][HW] that leave +-2 px / +-3 px in either direction. */
int cnuda_dcn_set_offset_regime(int regime);
int cnuda_dcn_offset_census(const float* offset, int B, int taps, long long HW, unsigned* counts, cnuda_stream_t stream);
int cnuda_conv_set_halo_policy(int level, int min_tiles);
/* Split-K of the forward-type convolution GEMMs (forward, stride-1 and parity-class input gradient; csrc/igemm.cuh
 * igemm_fwd_*splitk_kernel): a call whose pixel x row tiles at the natural row tile number fewer than max_tiles (default
 * 128: half the chip) and whose K has at least 32 chunks cuts K over grid.y -- partial slabs in the workspace, a fixed-order
 * reduce with the usual epilogue -- instead of shrinking the row tile.  max_tiles 0 = never; < 0 restores the default.
 * Returns the previous value.  Speed only: results differ by summation order (blocked over the splits).  CNUDA_SPLITK=0
 * disables it for the process. */
int cnuda_conv_set_splitk_policy(int max_tiles);

/* ------------------------------------------------------------------------
 * Detection decode -- replaces backends/decode.py:35-76 (decode_detection),
 * :6-13 (_nms), :16-32 (_topk) and the gathers of utils/tensor.py:10-25.
 *
 *   heat [B,C,H,W]  scores (the callers pass clamp(sigmoid) probabilities,
 *                   uda/base.py:76-77, export.py:31-38)
 *   wh   [B,wh_ch,H,W]  wh_ch = 2, or 3 when rotated (3rd = angle logit)
 *   reg  [B,2,H,W] or NULL (NULL -> +0.5 cell centre, decode.py:49-51)
 *   dets [B,K,6]  = x1,y1,x2,y2,score,class         (rotated == 0)
 *        [B,K,7]  = cx,cy,w,h,angle_deg,score,class  (rotated != 0)
 *   inds [B,K] int64 spatial index y*W+x of every detection, or NULL
 * Order among equal scores (unspecified in torch.topk): score descending, then
 * class ascending, then spatial index ascending.  1 <= K <= min(H*W, 1024).
 * nms_size must be odd (reference default 3).
 * ---------------------------------------------------------------------- */
/* Stage 1 cuts a plane into up to `max_bands` (1, 2 or 4; default 2; < 1 restores it) bands of rows, one workgroup each, when
 * the planes alone would leave CUs idle (B * C * bands <= 512).  Returns the previous value.  Speed only: the result is the
 * same bit for bit (tests/test_gpu_decode.py runs every case with 1, 2 and 4). */
int cnuda_decode_set_max_bands(int max_bands);
/* threads of stage 2's workgroup: 256, 512 or 1024; 0 = the default (1,024).  Returns the previous value.  Measurements; the result does not depend on it. */
int cnuda_decode_set_stage2_threads(int threads);
size_t cnuda_decode_workspace_bytes(int B, int C, int H, int W, int K);
int cnuda_decode_detection(const float* heat, const float* wh, const float* reg,
                           float* dets, int64_t* inds,
                           int B, int C, int H, int W, int K, int wh_ch, int rotated, int nms_size,
                           void* workspace, size_t workspace_bytes, cnuda_stream_t stream);
/* keypoint branch (decode.py:69-74): out [B,K,J,2] = kps[b, 2j + {0,1}, inds[b,k]] + (xs, ys) with
 * xs = inds % W + reg_x, ys = inds / W + reg_y (reg NULL: + 0.5), the centres of decode.py:44-51 */
int cnuda_decode_keypoints(const float* kps, const float* reg, const int64_t* inds, float* out,
                           int B, int J, int K, int H, int W, cnuda_stream_t stream);
/* _nms alone: out = heat * (1 - ceil(maxpool_k(heat) - heat))  (decode.py:6-13) */
int cnuda_nms(const float* heat, float* out, int B, int C, int H, int W, int nms_size,
              cnuda_stream_t stream);

/* ------------------------------------------------------------------------
 * Modulated deformable convolution (DCNv2) -- replaces the native module
 * `_ext` of libs/DCNv2 (src/vision.cpp:4-8):
 *   dcn_v2_forward  (src/dcn_v2.h:10-46  -> cuda/dcn_v2_cuda.cu:42-172)
 *   dcn_v2_backward (src/dcn_v2.h:48-92  -> cuda/dcn_v2_cuda.cu:206-341)
 * Argument order follows the native entry points (input, weight, bias, offset,
 * mask, ...).  offset [B, 2*kh*kw*dg, Ho, Wo] with channel 2*tap = dy and
 * 2*tap+1 = dx (cuda/dcn_v2_im2col_cuda.cu:170-174); mask [B, kh*kw*dg, Ho, Wo].
 * A sample is taken iff -1 < y < H and -1 < x < W, corners outside the plane
 * read 0 (:37-48,180).
 * Scratch: everything lives in the caller's workspace (nothing is allocated inside
 * a call).  forward samples inside the implicit GEMM's loader and materialises no
 * `columns` buffer when the output channels fit one tile of the GEMM; layers whose
 * output channels span several tiles (small feature maps) sample the columns once
 * into the workspace and run a plain GEMM over them.  backward materialises the
 * column gradient dcol[B, kh*kw*C, Ho*Wo] in the workspace (one 1x1 implicit GEMM,
 * two streaming consumers) and a 16-byte geometry record per (pixel, tap).
 * backward writes (does not accumulate into) all five gradients.  grad_input is
 * accumulated in LDS windows with plain read-add-write; fp32 global atomics remain
 * only for the window flush (one per touched cell), for samples that leave their
 * tile's window and for in-instruction collisions -- far fewer than the reference's
 * one atomic per corner (:238-252), but grad_input is still bit-reproducible only up
 * to the order of those.  deformable_group > 1 (round 6): composed of deformable_group = 1 calls on contiguous copies
 * of each group's input channels and weights (offsets / mask and their gradients in place, through their batch strides),
 * the group outputs added in group order -- 64 -> 64 at 128 x 128, B = 32, dg = 2 with the groups' columns saved: 1.18 / 2.90 ms
 * forward / backward against 0.89 / 2.36 ms with one group (rounds 1-5: one thread per element and global atomics, 45 ms /
 * 1.5 s; profiles/r6_dcn_dg2.txt).  Width 1 keeps that plain path.
 * ---------------------------------------------------------------------- */
size_t cnuda_dcn_v2_workspace_bytes(int B, int C, int H, int W, int Cout, int kh, int kw,
                                    int sh, int sw, int ph, int pw, int dh, int dw, int dg);
int cnuda_dcn_v2_forward(const float* input, const float* weight, const float* bias,
                         const float* offset, const float* mask, float* output,
                         int B, int C, int H, int W, int Cout, int kh, int kw,
                         int sh, int sw, int ph, int pw, int dh, int dw, int dg,
                         void* workspace, size_t workspace_bytes, cnuda_stream_t stream);
int cnuda_dcn_v2_backward(const float* input, const float* weight, const float* bias,
                          const float* offset, const float* mask, const float* grad_output,
                          float* grad_input, float* grad_offset, float* grad_mask,
                          float* grad_weight, float* grad_bias,
                          int B, int C, int H, int W, int Cout, int kh, int kw,
                          int sh, int sw, int ph, int pw, int dh, int dw, int dg,
                          void* workspace, size_t workspace_bytes, cnuda_stream_t stream);

/* forward with a fused epilogue activation: output = act(dcn(...) + bias), act_slope < 0 none, 0 ReLU.  Used by
 * the BatchNorm-folded inference path (export.py), where the BatchNorm + ReLU of DeformConv (backends/dla.py:369-372)
 * live in the weights, the bias and this epilogue.  columns nullable as in forward_cols below. */
int cnuda_dcn_v2_forward_act(const float* input, const float* weight, const float* bias,
                             const float* offset, const float* mask, float* output, float* columns,
                             float act_slope,
                             int B, int C, int H, int W, int Cout, int kh, int kw,
                             int sh, int sw, int ph, int pw, int dh, int dw, int dg,
                             void* workspace, size_t workspace_bytes, cnuda_stream_t stream);
/* forward (+ saved columns) that also leaves the BatchNorm statistics of the output, as cnuda_conv2d_forward_stats does:
 * DeformConv = DCN + BatchNorm + ReLU (backends/dla.py:351-372).  cnuda_dcn_v2_stats_block: 0 = none for this geometry,
 * else pixels per block (blocks are numbered image-major and never straddle images), *rows = rows per block. */
int cnuda_dcn_v2_stats_block(int B, int C, int H, int W, int Cout, int kh, int kw, int sh, int sw, int ph, int pw,
                             int dh, int dw, int dg, int* rows);
/* stats_block / stats_rows: the layout the caller sized `stats` for (what cnuda_dcn_v2_stats_block answered).  The answer
 * depends on the kernel the call picks, and that on the offset regime (cnuda_dcn_set_offset_regime): a call whose own plan
 * disagrees with these two numbers fails instead of writing another layout into the caller's buffer. */
int cnuda_dcn_v2_forward_stats(const float* input, const float* weight, const float* bias, const float* offset,
                               const float* mask, float* output, float* columns, float* stats, int stats_block,
                               int stats_rows,
                               int B, int C, int H, int W, int Cout, int kh, int kw, int sh, int sw, int ph, int pw,
                               int dh, int dw, int dg, void* workspace, size_t workspace_bytes, cnuda_stream_t stream);

/* Round 6 -- offsets and mask read straight out of `om` [B, 3*kh*kw, Ho, Wo], the output of DCN's own offset convolution
 * (libs/DCNv2/dcn_v2.py:118-122: o1, o2, mask = chunk(out, 3); offset = cat(o1, o2); mask = sigmoid(mask)): channels
 * 0 .. 2T-1 ARE the offsets, channels 2T .. 3T-1 the mask, ALREADY sigmoid (cnuda_conv2d_forward_rowsig applies it in the
 * convolution's epilogue).  backward_om writes ONE tensor grad_om of the same shape: the offsets' gradient and the gradient of
 * the mask's LOGIT (g * m * (1 - m), multiplied where the kernel stores) -- exactly what the offset convolution's backward
 * consumes.  Replaces the split / concatenate / sigmoid passes of the Python layer (32 launches of a benched step).
 * deformable_group == 1; columns / stats / accumulate_input as in forward_stats / backward_acc. */
int cnuda_dcn_v2_forward_om(const float* input, const float* weight, const float* bias, const float* om, float* output,
                            float* columns, float* stats, int stats_block, int stats_rows,
                            float act_slope /* < 0 none, 0 ReLU: the BatchNorm-folded inference path, as forward_act */,
                            int B, int C, int H, int W, int Cout, int kh, int kw, int sh, int sw, int ph, int pw,
                            int dh, int dw, int dg, void* workspace, size_t workspace_bytes, cnuda_stream_t stream);
int cnuda_dcn_v2_backward_om(const float* input, const float* weight, const float* bias, const float* om,
                             const float* grad_output, const float* columns, float* grad_input, int accumulate_input,
                             float* grad_om, float* grad_weight, float* grad_bias,
                             int B, int C, int H, int W, int Cout, int kh, int kw, int sh, int sw, int ph, int pw,
                             int dh, int dw, int dg, void* workspace, size_t workspace_bytes, cnuda_stream_t stream);

/* Same two operations with the sampled column buffer kept between them (the
 * product's autograd path): forward_cols stores
 * columns[B, kh*kw*C, Ho*Wo] (rows in (tap, channel) order, mask already applied;
 * deformable_group > 1: the groups' buffers [dg][B, kh*kw*C/dg, Ho*Wo] one behind the other, the same bytes)
 * as a side output of the implicit GEMM, backward_cols computes grad_weight as a
 * plain GEMM grad_output x columns^T instead of re-sampling the input.  With
 * columns == NULL both behave exactly like the entry points above.  The buffer
 * is what the reference materialises inside every call
 * (cuda/dcn_v2_cuda.cu:89-102); here it is written once and read once. */
int cnuda_dcn_v2_forward_cols(const float* input, const float* weight, const float* bias,
                              const float* offset, const float* mask, float* output, float* columns,
                              int B, int C, int H, int W, int Cout, int kh, int kw,
                              int sh, int sw, int ph, int pw, int dh, int dw, int dg,
                              void* workspace, size_t workspace_bytes, cnuda_stream_t stream);
int cnuda_dcn_v2_backward_cols(const float* input, const float* weight, const float* bias,
                               const float* offset, const float* mask, const float* grad_output,
                               const float* columns,
                               float* grad_input, float* grad_offset, float* grad_mask,
                               float* grad_weight, float* grad_bias,
                               int B, int C, int H, int W, int Cout, int kh, int kw,
                               int sh, int sw, int ph, int pw, int dh, int dw, int dg,
                               void* workspace, size_t workspace_bytes, cnuda_stream_t stream);

/* backward_cols with accumulate_input != 0: grad_input is NOT cleared first -- the data-gradient walks add into it (window
 * flushes and strays are atomics either way), so a caller that already holds another consumer's share of the input's
 * gradient there (the offset / mask convolution's input gradient: backends/dla.py:263-270, x feeds both) saves the clear and
 * autograd's sum.  accumulate_input == 0: exactly backward_cols. */
int cnuda_dcn_v2_backward_acc(const float* input, const float* weight, const float* bias,
                              const float* offset, const float* mask, const float* grad_output,
                              const float* columns,
                              float* grad_input, int accumulate_input, float* grad_offset, float* grad_mask,
                              float* grad_weight, float* grad_bias,
                              int B, int C, int H, int W, int Cout, int kh, int kw,
                              int sh, int sw, int ph, int pw, int dh, int dw, int dg,
                              void* workspace, size_t workspace_bytes, cnuda_stream_t stream);

/* ------------------------------------------------------------------------
 * Dense convolution (groups 1, dilation 1) -- replaces torch.nn.Conv2d -> cuDNN
 * on the hot path: backends/dla.py:37-44,153-155,234-235,281-283,478-483 (DLA
 * trunk, roots, heads), libs/DCNv2/dcn_v2.py:104-110 (offset/mask conv),
 * backends/resnet.py:43-51 (heads), uda/adversarial_entropy_minimization.py:51-68
 * (discriminator).  weight is the state_dict layout [Cout, C, kh, kw].
 * forward: y = act(conv(x, w) + bias); bias may be NULL; act_slope < 0 -> no
 * activation, 0 -> ReLU, 0.2 -> LeakyReLU(0.2).  backward_weight writes
 * grad_weight (and grad_bias = channel sums of grad_y when non-NULL).
 * ---------------------------------------------------------------------- */
size_t cnuda_conv2d_workspace_bytes(int B, int C, int H, int W, int Cout, int kh, int kw,
                                    int sh, int sw, int ph, int pw);
int cnuda_conv2d_forward(const float* x, const float* weight, const float* bias, float* y,
                         int B, int C, int H, int W, int Cout, int kh, int kw,
                         int sh, int sw, int ph, int pw, float act_slope,
                         void* workspace, size_t workspace_bytes, cnuda_stream_t stream);
/* forward with a residual input in the epilogue: y = act(conv(x, w) + bias + residual), residual shaped like y
 * (nullable).  The BatchNorm-folded inference path (export.py) runs a BasicBlock's second convolution, its
 * BatchNorm, the skip connection and the ReLU (backends/dla.py:48-62) as this one launch. */
int cnuda_conv2d_forward_res(const float* x, const float* weight, const float* bias, const float* residual, float* y,
                             int B, int C, int H, int W, int Cout, int kh, int kw,
                             int sh, int sw, int ph, int pw, float act_slope,
                             void* workspace, size_t workspace_bytes, cnuda_stream_t stream);
/* forward with a sigmoid on the output channels >= sig_from (after the bias): the offset / mask convolution of a DCN layer
 * (libs/DCNv2/dcn_v2.py:104-122: channels 2T .. 3T-1 of its output are the modulation mask's logits) when the deformable
 * convolution reads offsets and mask out of this one tensor (cnuda_dcn_v2_forward_om).  cnuda_conv2d_rowsig_supported: 1 for
 * the geometries that have this epilogue (at most 32 output channels, C % 16 == 0, tensors below 2 GiB, f32 matrix mode). */
int cnuda_conv2d_rowsig_supported(int B, int C, int H, int W, int Cout, int kh, int kw, int sh, int sw, int ph, int pw);
int cnuda_conv2d_forward_rowsig(const float* x, const float* weight, const float* bias, float* y, int sig_from,
                                int B, int C, int H, int W, int Cout, int kh, int kw, int sh, int sw, int ph, int pw,
                                void* workspace, size_t workspace_bytes, cnuda_stream_t stream);
/* A 1x1 convolution (stride 1, no padding) over the CHANNEL CONCATENATION of n = 2 .. 4 tensors xs[i] [B, cs[i], H, W]
 * without the concatenation: DLA's Root is conv(torch.cat(children, 1)) (backends/dla.py:150-168; 17 concatenated buffers per
 * forward pass of DLA-34 and their slices in the backward pass).  weight [Cout, sum cs, 1, 1] as for the concatenated tensor.
 * The GEMMs' K axis IS the concatenated channel axis: the gather switches source per 16-deep chunk, the input-gradient
 * epilogue stores every source's rows into its own tensor (+ adds[i] + add2s[i], nullable: the other consumers' shares of that
 * tensor's gradient, as cnuda_conv2d_backward_data_add), the weight gradient reads each 64-column group from its source.
 * stats: nullable, cnuda_conv2d_forward_stats' layout for the concatenated geometry (cnuda_conv2d_stats_block(B, sum cs, H, W,
 * Cout, 1, 1, 1, 1, 0, 0)).  cnuda_conv2d_cat_supported: every cs[i] % 64 == 0, H * W % 4 == 0, tensors below 2 GiB, the
 * concatenated geometry's plan without a K split, f32 matrix mode; elsewhere the caller concatenates.  Workspace:
 * cnuda_conv2d_workspace_bytes of the concatenated geometry. */
int cnuda_conv2d_cat_supported(const int* cs, int n, int B, int H, int W, int Cout);
int cnuda_conv2d_cat_forward(const float* const* xs, const int* cs, int n, const float* weight, const float* bias /* nullable */,
                             float* y, float* stats, float act_slope /* < 0: none; with stats: none */,
                             int B, int H, int W, int Cout, void* workspace, size_t workspace_bytes, cnuda_stream_t stream);
int cnuda_conv2d_cat_backward_data(const float* grad_y, const float* weight, float* const* grad_xs, const float* const* adds,
                                   const float* const* add2s, const int* cs, int n, int B, int H, int W, int Cout,
                                   void* workspace, size_t workspace_bytes, cnuda_stream_t stream);
int cnuda_conv2d_cat_backward_weight(const float* const* xs, const int* cs, int n, const float* grad_y, float* grad_weight,
                                     int B, int H, int W, int Cout, void* workspace, size_t workspace_bytes,
                                     cnuda_stream_t stream);
/* forward whose output channels are interleaved in quads: y[B][Cout / 4][Ho * Wo][4] (channel m of pixel p at
 * ((m >> 2) * Ho*Wo + p) * 4 + (m & 3)); no bias, no activation.  The layout of the DCN column gradient (round 6): the
 * gradient walk of cnuda_dcn_v2_backward reads the four channels of a (pixel, tap) with one 16-byte load.  Replaces nothing
 * of the reference's (its dcn_v2_cuda.cu:262-275 GEMM writes [kh*kw*C][Ho*Wo] rows); internal to the backward, exported so
 * that the layout has a value test of its own.  cnuda_conv2d_rowquads_supported: Cout % 4 == 0, C % 16 == 0, tensors below
 * 2 GiB, no K split in the plan. */
int cnuda_conv2d_rowquads_supported(int B, int C, int H, int W, int Cout, int kh, int kw, int sh, int sw, int ph, int pw);
int cnuda_conv2d_forward_rowquads(const float* x, const float* weight, float* y,
                                  int B, int C, int H, int W, int Cout, int kh, int kw, int sh, int sw, int ph, int pw,
                                  void* workspace, size_t workspace_bytes, cnuda_stream_t stream);
/* forward that also leaves the statistics a train-mode BatchNorm of y needs (the layer behind almost every convolution of
 * DLA-34, backends/dla.py:37-62,150-168: the reference's cuDNN BatchNorm re-reads y for them; here the GEMM's epilogue sums
 * what it stores).  stats: [blocks][rows][2] floats = (sum, sum of squares) of y over one block of `cnuda_conv2d_stats_block`
 * consecutive pixels of the flattened (image, pixel) axis, per output channel (row); blocks = ceil(B*Ho*Wo / 128) * (128 /
 * block pixels).  cnuda_conv2d_stats_block returns 0 where this geometry's kernel cannot give them (then pass stats =
 * NULL and use cnuda_bn_train_forward), else the block size, and *rows.  Needs residual == NULL and act_slope < 0.
 * cnuda_bn_train_forward_stats (below) consumes them. */
int cnuda_conv2d_stats_block(int B, int C, int H, int W, int Cout, int kh, int kw, int sh, int sw, int ph, int pw, int* rows,
                             int* blocks_per_image /* > 0: blocks are pieces of output rows, this many per image (the 3- / 16-
                                                      channel layers' kernels) and `stats` holds B * blocks_per_image of them */);
int cnuda_conv2d_forward_stats(const float* x, const float* weight, const float* bias, const float* residual, float* y,
                               float* stats, int B, int C, int H, int W, int Cout, int kh, int kw,
                               int sh, int sw, int ph, int pw, float act_slope,
                               void* workspace, size_t workspace_bytes, cnuda_stream_t stream);
int cnuda_conv2d_backward_data(const float* grad_y, const float* weight, float* grad_x,
                               int B, int C, int H, int W, int Cout, int kh, int kw,
                               int sh, int sw, int ph, int pw,
                               void* workspace, size_t workspace_bytes, cnuda_stream_t stream);
/* Apply on load: x is the output of a convolution whose train-mode BatchNorm + ReLU has NOT been applied
 * (cnuda_bn_train_forward_stats with y == NULL left save_mean / save_invstd, [groups][C]); the kernel computes
 * max(x * (invstd * gamma) + (beta - mean * invstd * gamma), 0) while it stages x, per statistics group of imgs_per_group
 * images -- bit-identical to convolving the materialised activation, one pass over it less in each direction.  Replaces
 * the pass of reference backends/dla.py:19,40 (nn.BatchNorm2d + ReLU between two convolutions) for the geometries
 * cnuda_conv2d_norm_input_supported() returns 1 for; the other arguments as cnuda_conv2d_forward_stats /
 * cnuda_conv2d_backward_weight. */
int cnuda_conv2d_norm_input_supported(int B, int C, int H, int W, int Cout, int kh, int kw, int sh, int sw, int ph, int pw);
int cnuda_conv2d_forward_norm_input(const float* x, const float* mean, const float* invstd, const float* gamma,
                                    const float* beta, int imgs_per_group, const float* weight, const float* bias, float* y,
                                    float* stats, int B, int C, int H, int W, int Cout, int kh, int kw, int sh, int sw, int ph,
                                    int pw, float act_slope, void* workspace, size_t workspace_bytes, cnuda_stream_t stream);
int cnuda_conv2d_backward_weight_norm_input(const float* x, const float* mean, const float* invstd, const float* gamma,
                                            const float* beta, int imgs_per_group, const float* grad_y, float* grad_weight,
                                            float* grad_bias, int B, int C, int H, int W, int Cout, int kh, int kw, int sh,
                                            int sw, int ph, int pw, void* workspace, size_t workspace_bytes,
                                            cnuda_stream_t stream);
/* grad_x = input gradient + addend + addend2 (both nullable, shaped like grad_x, either may BE grad_x): where a tensor
 * feeds a convolution and something else (a skip connection, a concatenation, a DCN's sampling), the other consumers'
 * shares of its gradient are summed in the epilogue of this one instead of by passes of their own (what autograd's
 * accumulation does in the reference: torch's engine, one at::add per extra consumer; hip_runtime.fanout). */
int cnuda_conv2d_backward_data_add(const float* grad_y, const float* weight, const float* addend, const float* addend2,
                                   float* grad_x,
                                   int B, int C, int H, int W, int Cout, int kh, int kw,
                                   int sh, int sw, int ph, int pw,
                                   void* workspace, size_t workspace_bytes, cnuda_stream_t stream);
int cnuda_conv2d_backward_weight(const float* x, const float* grad_y, float* grad_weight, float* grad_bias,
                                 int B, int C, int H, int W, int Cout, int kh, int kw,
                                 int sh, int sw, int ph, int pw,
                                 void* workspace, size_t workspace_bytes, cnuda_stream_t stream);

/* ------------------------------------------------------------------------
 * BatchNorm2d fused with the residual add and ReLU that follow it in DLA-34
 * (backends/dla.py:48-62,150-168,277-287,351-372; nn.BatchNorm2d(momentum=0.1)).
 * x, y, residual: [B, C, HW].  residual may be NULL; relu: 0 none, 1 max(.,0), 2 ReLU6 = min(max(.,0),6)
 * (gradient strictly inside (0,6), torch's hardtanh backward).
 * train_forward updates running_mean/var in place (unbiased variance, may be
 * NULL) and saves mean / invstd for backward.  backward: grad_y is the gradient
 * w.r.t. y (post-activation); y is needed only when relu != 0; grad_residual
 * (nullable) receives the gradient flowing into the residual branch.
 * ---------------------------------------------------------------------- */
size_t cnuda_bn_workspace_bytes(int B, int C, long long HW);
int cnuda_bn_train_forward(const float* x, const float* gamma, const float* beta, const float* residual,
                           float* y, float* save_mean, float* save_invstd,
                           float* running_mean, float* running_var,
                           long long* num_batches_tracked /* nullable; += 1 like nn.BatchNorm2d.forward */,
                           float momentum, float eps, int relu,
                           int B, int C, long long HW,
                           int groups /* statistics groups: images [g*B/groups, (g+1)*B/groups) are normalised by their
                                         own batch statistics (save_mean / save_invstd are [groups][C]) and the running
                                         statistics take one momentum update per group, in group order -- one call on
                                         the concatenated source | target batch equals the reference's two forward
                                         calls (uda/entropy_minimization.py:18-19) */,
                           void* workspace, size_t workspace_bytes, cnuda_stream_t stream);
/* train_forward with sum(x) / sum(x^2) supplied by the kernel that produced x (cnuda_conv2d_forward_stats,
 * cnuda_dcn_v2_forward_stats): no statistics pass over x.  stats [blocks][rows][2], blocks numbered image-major; group g =
 * blocks [g, g + 1) * blocks_per_group (the caller checks that a statistics group is whole blocks). */
int cnuda_bn_train_forward_stats(const float* x, const float* stats, long long blocks_per_group, int rows, const float* gamma,
                                 const float* beta, const float* residual, float* y, float* save_mean, float* save_invstd,
                                 float* running_mean, float* running_var, long long* num_batches_tracked,
                                 float momentum, float eps, int relu, int B, int C, long long HW, int groups,
                                 void* workspace, size_t workspace_bytes, cnuda_stream_t stream);
int cnuda_bn_eval_forward(const float* x, const float* gamma, const float* beta,
                          const float* running_mean, const float* running_var, const float* residual,
                          float* y, float eps, int relu, int B, int C, long long HW, cnuda_stream_t stream);
int cnuda_bn_backward(const float* grad_y, const float* x, const float* y, const float* gamma,
                      const float* beta /* nullable.  Given (only without a residual): the activation's gate is
                                           recomputed from x with the forward's own multiply and add instead of read
                                           from y -- one tensor less per pass; y may then be NULL */,
                      const float* save_mean, const float* save_invstd,
                      float* grad_x, float* grad_residual, float* grad_gamma, float* grad_beta,
                      int relu, int B, int C, long long HW, int groups,
                      void* workspace, size_t workspace_bytes, cnuda_stream_t stream);

/* ------------------------------------------------------------------------
 * Spatial / elementwise pieces of the DLA graph.
 *   maxpool2d: window k, stride k, no padding  (nn.MaxPool2d(stride, stride), dla.py:202-203)
 *   dwconvt2d: depthwise ConvTranspose2d, weight [C,1,k,k], stride s, padding p, no bias
 *              (IDAUp.up, dla.py:385-388: k = 2f, s = f, p = f/2); grad_x / grad_w nullable
 *   add, act_backward (gx = gy * (y > 0 ? 1 : slope)), copy_channels (concat / slice),
 *   split_offset_mask: DCN.forward's chunk/cat/sigmoid (libs/DCNv2/dcn_v2.py:119-122)
 * ---------------------------------------------------------------------- */
int cnuda_maxpool2d_forward(const float* x, float* y, int B, int C, int H, int W, int k, cnuda_stream_t stream);
int cnuda_maxpool2d_backward(const float* x, const float* grad_y, float* grad_x,
                             int B, int C, int H, int W, int k, cnuda_stream_t stream);
/* ... with accumulate != 0: grad_x already holds another consumer's share of x's gradient (hip_runtime.fanout: a level input
 * that feeds the pooling shortcut and a strided convolution, backends/dla.py:199-203) and only the arg-max cell of every window
 * is touched (+= grad_y). */
int cnuda_maxpool2d_backward_acc(const float* x, const float* grad_y, float* grad_x, int accumulate, int B, int C, int H, int W,
                                 int k, cnuda_stream_t stream);
/* general window (kernel k, stride s, padding p with -inf, floor mode): torchvision's ResNet stem pool
 * nn.MaxPool2d(3, 2, 1), kept inside `base` by backends/resnet.py:27-30.  Backward is a deterministic gather. */
int cnuda_maxpool2d_window_forward(const float* x, float* y, int B, int C, int H, int W, int k, int s, int p,
                                   cnuda_stream_t stream);
int cnuda_maxpool2d_window_backward(const float* x, const float* grad_y, float* grad_x,
                                    int B, int C, int H, int W, int k, int s, int p, cnuda_stream_t stream);
int cnuda_dwconvt2d_forward(const float* x, const float* w, float* y,
                            int B, int C, int H, int W, int k, int s, int p, cnuda_stream_t stream);
/* y = dwconvt2d(x, w) + skip with skip of the output's shape (nullable): IDAUp's `up(project(x)) + layers[i-1]`
 * (dla.py:400-401) in one pass over the upsampled map, rounded as the separate add would round */
int cnuda_dwconvt2d_add_forward(const float* x, const float* w, const float* skip, float* y,
                                int B, int C, int H, int W, int k, int s, int p, cnuda_stream_t stream);
size_t cnuda_dwconvt2d_workspace_bytes(int B, int C, int k);   /* per-image partial weight gradients */
int cnuda_dwconvt2d_backward(const float* x, const float* w, const float* grad_y, float* grad_x, float* grad_w,
                             int B, int C, int H, int W, int k, int s, int p,
                             void* workspace, size_t workspace_bytes, cnuda_stream_t stream);
/* Depthwise convolution, weight [C,1,k,k] (k = 3 or 5), no bias -- torchvision MobileNetV2's
 * `nn.Conv2d(hidden, hidden, 3, stride, 1, groups=hidden, bias=False)` inside backends/mobilenetv2.py:31-36's hub
 * trunk.  backward writes grad_x and/or grad_w (either may be NULL); grad_w needs the workspace. */
size_t cnuda_dwconv2d_workspace_bytes(int B, int C, int k);
int cnuda_dwconv2d_forward(const float* x, const float* w, float* y, int B, int C, int H, int W, int k, int s, int p,
                           cnuda_stream_t stream);
int cnuda_dwconv2d_backward(const float* x, const float* w, const float* grad_y, float* grad_x, float* grad_w,
                            int B, int C, int H, int W, int k, int s, int p,
                            void* workspace, size_t workspace_bytes, cnuda_stream_t stream);
int cnuda_add(const float* a, const float* b, float* out, long long n, cnuda_stream_t stream);
int cnuda_act_backward(const float* grad_y, const float* y, float* grad_x, long long n, float slope,
                       cnuda_stream_t stream);
/* grad_hidden = act'(hidden) * conv1x1_input_gradient(grad_y, weight): the last layer of a detection head
 * (nn.Conv2d(head_conv, classes, 1), dla.py:474-483) has 1..8 output channels, so its input gradient is one pass over
 * the hidden map fused with the backward of the ReLU in front of it.  weight [Co, Ch] (the 1x1 kernel), hidden /
 * grad_hidden [B, Ch, HW] (HW % 4 == 0), grad_y [B, Co, HW]; channels summed in increasing order. */
int cnuda_conv1x1_backward_data_act(const float* grad_y, const float* weight, const float* hidden, float* grad_hidden,
                                    int B, int Co, int Ch, long long HW, float slope, cnuda_stream_t stream);
int cnuda_copy_channels(const float* src, float* dst, int B, int Cn, long long HW,
                        int Csrc, int src_off, int Cdst, int dst_off, cnuda_stream_t stream);
int cnuda_split_offset_mask(const float* om, float* offset, float* mask, int B, int taps, long long HW,
                            cnuda_stream_t stream);
int cnuda_split_offset_mask_backward(const float* grad_offset, const float* grad_mask, const float* mask,
                                     float* grad_om, int B, int taps, long long HW, cnuda_stream_t stream);

/* ------------------------------------------------------------------------
 * Losses.  Scalars live in device memory (no host sync; the reference's
 * `if num_pos == 0` host branch, losses/centernet.py:91, is taken on device).
 * `upstream` is a device pointer to the scalar gradient of the loss value.
 *   focal   : losses/centernet.py:69-95 on prob = clamp(sigmoid(logits)) (utils/tensor.py:5-7);
 *             out2 = {loss, num_pos}; prob receives the clamped probabilities (Q1)
 *   reg_l1  : losses/centernet.py:98-133 (ch 2, or 3 = rotated) and :192-223 (periodic != 0);
 *             masks `target` in place and, for the rotated non-periodic angle, replaces it by its
 *             sigmoid, as the reference does (Q2); out2 = {loss, mask.sum()+1e-4}
 *   softmax_loss kind 0: losses/entropy.py:10-28; kind 1: losses/max_square.py:6-14
 *   entropy_map: utils/image.py:121-124;  bce_const: losses/advent.py:10-18
 * ---------------------------------------------------------------------- */
size_t cnuda_loss_workspace_bytes(void);
int cnuda_focal_loss_forward(const float* logits, const float* gt, float* prob, float* out2, long long n,
                             float weight, void* workspace, size_t workspace_bytes, cnuda_stream_t stream);
int cnuda_focal_loss_backward(const float* logits, const float* gt, const float* out2, const float* upstream,
                              float* grad_logits, long long n, float weight, cnuda_stream_t stream);
int cnuda_reg_l1_forward(const float* feat, const uint8_t* mask, const int64_t* ind, float* target, float* out2,
                         int B, int M, int ch, long long HW, int periodic, float weight, float angle_weight,
                         cnuda_stream_t stream);
int cnuda_reg_l1_backward(const float* feat, const uint8_t* mask, const int64_t* ind, const float* target,
                          const float* out2, const float* upstream, float* grad_feat,
                          int B, int M, int ch, long long HW, int periodic, float weight, float angle_weight,
                          cnuda_stream_t stream);
/* KPSL1Loss (losses/centernet.py:136-189): feat [B,2J,HW], mask [B,M,2J] (kp_reg_mask, datasets/coco.py:183,226-227),
 * target [B,M,2J] masked in place; pairs [n_pairs][2] int32 = kps_weight_indices (NULL / 0: no distance term);
 * use_l1 selects the L1 pair distance, else sqrt(|.|^2 + 1e4) (:175-179); out2 = {loss, mask.sum() + 1e-4}. */
int cnuda_kps_l1_forward(const float* feat, const uint8_t* mask, const int64_t* ind, float* target,
                         const int32_t* pairs, float* out2, int B, int M, int J, long long HW, int n_pairs,
                         int use_l1, float weight, float distance_weight, cnuda_stream_t stream);
int cnuda_kps_l1_backward(const float* feat, const uint8_t* mask, const int64_t* ind, const float* target,
                          const int32_t* pairs, const float* out2, const float* upstream, float* grad_feat,
                          int B, int M, int J, long long HW, int n_pairs, int use_l1, float weight,
                          float distance_weight, cnuda_stream_t stream);
int cnuda_softmax_loss_forward(const float* logits, float* out1, int B, int C, long long HW, int kind,
                               void* workspace, size_t workspace_bytes, cnuda_stream_t stream);
int cnuda_softmax_loss_backward(const float* logits, const float* upstream, float* grad_logits,
                                int B, int C, long long HW, int kind, cnuda_stream_t stream);
int cnuda_entropy_map_forward(const float* logits, float* out, int B, int C, long long HW, cnuda_stream_t stream);
int cnuda_entropy_map_backward(const float* logits, const float* grad_out, float* grad_logits,
                               int B, int C, long long HW, cnuda_stream_t stream);
int cnuda_bce_const_forward(const float* logits, float label, float* out1, long long n, cnuda_stream_t stream);
int cnuda_bce_const_backward(const float* logits, float label, const float* upstream, float* grad_logits,
                             long long n, cnuda_stream_t stream);
/* x <- sigmoid(x) in place, y = clamp(x, 1e-4, 1-1e-4)  (utils/tensor.py:5-7) */
int cnuda_sigmoid_clamp_(float* x, float* y, long long n, cnuda_stream_t stream);
/* feat [B,ch,HW], ind [B,M] -> out [B,M,ch]  (utils/tensor.py:10-25) */
int cnuda_gather_feat(const float* feat, const int64_t* ind, float* out, int B, int M, int ch, long long HW,
                      cnuda_stream_t stream);

/* ------------------------------------------------------------------------
 * Optimizer -- torch.optim.Adam arithmetic (train.py:88-90) over a flat arena.
 * step is the 1-based step count used for bias correction.
 * ---------------------------------------------------------------------- */
int cnuda_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n,
                    float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                    cnuda_stream_t stream);

/* ------------------------------------------------------------------------
 * Target encoding for a batch (the step right before the hot path):
 * datasets/coco.py:191-221 (per-object loop: clip, gaussian radius, centre, ind / wh / reg / reg_mask /
 * gt_dets / gt_areas) and utils/image.py:8-57 (gaussian_radius, draw_umich_gaussian), all images in one launch.
 *   boxes [B,M,4] double (x1,y1,x2,y2 in output-map pixels, after augmentation / clip_out_of_image),
 *   classes [B,M] int32, counts [B] int32 (objects per image, <= M).  Every output is fully overwritten
 *   (zeros where the reference leaves np.zeros): hm [B,C,H,W] f32, reg_mask [B,M] u8, ind [B,M] i64,
 *   wh / reg [B,M,2] f32, gt_dets [B,M,6] f32, gt_areas [B,M] f32 (= w*h: annotations without "area").
 * ---------------------------------------------------------------------- */
int cnuda_encode_targets(const double* boxes, const int* classes, const int* counts,
                         float* hm, unsigned char* reg_mask, long long* ind, float* wh, float* reg,
                         float* gt_dets, float* gt_areas,
                         int B, int C, int H, int W, int M, cnuda_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CENTERNET_UDA_HIP_H */
