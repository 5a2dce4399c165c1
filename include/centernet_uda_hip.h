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
#define CNUDA_ABI_VERSION 1

typedef void* cnuda_stream_t; /* hipStream_t */

int cnuda_abi_version(void);
const char* cnuda_last_error(void);

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
size_t cnuda_decode_workspace_bytes(int B, int C, int H, int W, int K);
int cnuda_decode_detection(const float* heat, const float* wh, const float* reg,
                           float* dets, int64_t* inds,
                           int B, int C, int H, int W, int K, int wh_ch, int rotated, int nms_size,
                           void* workspace, size_t workspace_bytes, cnuda_stream_t stream);
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
 * read 0 (:37-48,180).  No `columns` buffer is materialised.
 * backward writes (does not accumulate into) all five gradients; grad_input is
 * summed with fp32 atomics like the reference (:238-252) and is therefore
 * bit-reproducible only up to summation order.
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

#ifdef __cplusplus
}
#endif
#endif /* CENTERNET_UDA_HIP_H */
