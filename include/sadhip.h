/*
 * sadhip.h -- C ABI of libsadhip.so, the MI355X (gfx950) implementation of the SAD
 * block-matching disparity path of conneroisu/steroscopic-hardware (Go package pkg/despair).
 *
 * This is the drop-in boundary: exactly the entry points a cgo binding for pkg/despair would
 * bind (INTEGRATION.md shows that binding).  Plain C types only: pointers, sizes, ints; no
 * structs by value, no C++/torch types.  Every entry point returns an int status (0 = ok) and
 * never aborts or throws (the reference panics instead: index out of range, divide by zero at
 * sad.go:92 -- SURVEY.md section 8b "Errors").
 *
 * There is NO CPU fallback in this library.  Without a usable gfx950 device every compute
 * entry point returns SADHIP_ERR_NO_DEVICE / SADHIP_ERR_HIP; the caller (the Go shim) decides
 * whether to fall back to its own pure-Go path.
 *
 * Semantics (bit-exact with pkg/despair/sad.go:55-94 + :205-244, Rect.Min == (0,0)):
 *   for every pixel (x,y):  hs = block/2,
 *     candidates d in [0, min(maxdisp, max(0, x - hs))]   (clamp quirk of sad.go:217-218 folded in)
 *     SAD(d) = sum over the window [max(x-hs,0), min(x+hs+1,W)) x [max(y-hs,0), min(y+hs+1,H))
 *              of |L[yy][xx] - R[yy][xx-d]|
 *     best   = lowest d attaining the minimum (strict '<', sad.go:79)
 *     out    = (uint8)(best * 255 / maxdisp)              (sad.go:91-93)
 *
 * Threading: a context may be used from any OS thread (every call selects the context's HIP
 * device itself -- goroutines migrate between threads); calls on one context are serialised
 * by an internal mutex.  Host pointers are never retained after a call returns (cgo rule).
 */
#ifndef SADHIP_H
#define SADHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SADHIP_ABI_VERSION 1

#if defined(__GNUC__)
#define SADHIP_API __attribute__((visibility("default")))
#else
#define SADHIP_API
#endif

/* status codes */
#define SADHIP_OK               0
#define SADHIP_ERR_NULL         1   /* a required pointer is NULL */
#define SADHIP_ERR_DIMS         2   /* w <= 0 or h <= 0, or larger than the context capacity */
#define SADHIP_ERR_STRIDE       3   /* a row stride is smaller than w */
#define SADHIP_ERR_BLOCK        4   /* block < 1 or block > SADHIP_MAX_BLOCK */
#define SADHIP_ERR_MAXDISP      5   /* maxdisp < 1 (the reference divides by zero) or > SADHIP_MAX_MAXDISP */
#define SADHIP_ERR_REGION       6   /* region not inside [0,w) x [0,h) or empty/inverted */
#define SADHIP_ERR_BATCH        7   /* n < 0 or n larger than the context's batch capacity */
#define SADHIP_ERR_NO_DEVICE    8   /* no HIP device / bad device ordinal */
#define SADHIP_ERR_HIP          9   /* a HIP runtime call failed; see sadhip_last_error() */
#define SADHIP_ERR_ARG         10   /* any other invalid argument */
#define SADHIP_ERR_NOMEM       11   /* host or device allocation failed */

#define SADHIP_MAX_BLOCK     255
#define SADHIP_MAX_MAXDISP  4096

/* sadhip_disparity_* "flags" argument */
#define SADHIP_FLAG_RAW        1u   /* store best d itself (clamped to 255) instead of best*255/maxdisp */
#define SADHIP_FLAG_FORCE_GENERIC 2u /* diagnostics: use the generic kernel even where the fast one applies */
/* Hardware-mirror modes: compute with the semantics of the reference's OTHER golden models instead
 * of pkg/despair's, raw d output, so the FPGA testbench fixtures can be regenerated at any size
 * (SURVEY.md section 8 f-4).  Mutually exclusive; block = the models' WIN, maxdisp = MAX_DISP. */
#define SADHIP_FLAG_MIRROR_SADC   4u /* hardware/sad.c:66-106 (interior only, xr<0 terms skipped) */
#define SADHIP_FLAG_MIRROR_TESTPY 8u /* hardware/test.py:262-294 compute_disparity_patch (zero padding) */

typedef struct sadhip_ctx sadhip_ctx;   /* opaque per-device context */

/* ---- library / device queries ------------------------------------------------------------ */
SADHIP_API int         sadhip_abi_version(void);
SADHIP_API int         sadhip_device_count(int* count);          /* replaces: nothing (reference is CPU-only) */
SADHIP_API const char* sadhip_status_string(int status);
SADHIP_API const char* sadhip_last_error(void);                  /* thread-local detail of the last failure */

/* ---- context ------------------------------------------------------------------------------
 * Replaces the worker pool of SetupConcurrentSAD (pkg/despair/sad.go:29-113): one context owns
 * the device buffers, pinned staging and HIP streams for frames up to max_w x max_h and batches
 * of up to max_batch frames.  Created when the Go shim's pipeline is set up, destroyed when its
 * input channel is closed (sad.go:106-110, pkg/camera/output.go:213-220). */
SADHIP_API int sadhip_ctx_create(int device, int max_w, int max_h, int max_batch, sadhip_ctx** out);
SADHIP_API int sadhip_ctx_destroy(sadhip_ctx* ctx);
SADHIP_API int sadhip_ctx_device(const sadhip_ctx* ctx, int* device);

/* ---- whole frame, host buffers (synchronous) ------------------------------------------------
 * Replaces RunSad (sad.go:119-169) and the OutputCamera frame loop
 * (pkg/camera/output.go:171-190): worker body sad.go:55-94 over Rect(0,0,w,h), assembled
 * WITHOUT the dropped-chunk hole of AssembleDisparityMap (sad.go:181-184).
 * left/right: w*h u8 gray, row strides lstride/rstride (image.Gray.Stride); out: ostride. */
SADHIP_API int sadhip_disparity_u8(sadhip_ctx* ctx,
                        const uint8_t* left, const uint8_t* right,
                        int w, int h, int lstride, int rstride,
                        int block, int maxdisp, unsigned flags,
                        uint8_t* out, int ostride);

/* ---- one region of a frame, host buffers (synchronous) --------------------------------------
 * Replaces one InputChunk -> OutputChunk trip through the worker (sad.go:47-101):
 * region = image.Rect(x0,y0,x1,y1); out_region is (y1-y0) rows of (x1-x0) bytes, row-major
 * within the region (OutputChunk.DisparityData, sad.go:48-50,91).  Only rows
 * [y0-hs, y1+hs) of the inputs are read/uploaded. */
SADHIP_API int sadhip_disparity_region_u8(sadhip_ctx* ctx,
                               const uint8_t* left, const uint8_t* right,
                               int w, int h, int lstride, int rstride,
                               int block, int maxdisp, unsigned flags,
                               int x0, int y0, int x1, int y1,
                               uint8_t* out_region);

/* ---- batch of independent frames, host buffers (synchronous; pipelined inside) --------------
 * Replaces the streaming loop of examples/run.stream.go:30-59 / BASELINE config 4: n independent
 * pairs, H2D / kernel / D2H double-buffered on the context's streams through pinned staging. */
SADHIP_API int sadhip_disparity_batch_u8(sadhip_ctx* ctx, int n,
                              const uint8_t* const* lefts, const uint8_t* const* rights,
                              int w, int h, int lstride, int rstride,
                              int block, int maxdisp, unsigned flags,
                              uint8_t* const* outs, int ostride);

/* ---- device-resident buffers (asynchronous on hip_stream) -----------------------------------
 * Same function for callers that already hold frames in HBM (bench.py, a capture pipeline
 * that DMA's frames to the GPU).  d_left/d_right/d_out are device pointers on the context's
 * device; n frames, frame f at base + f*frame_stride bytes.  hip_stream is a hipStream_t
 * (NULL = HIP's null stream, which is also PyTorch's default stream).  Returns after enqueueing;
 * no host synchronisation. */
SADHIP_API int sadhip_disparity_dev_u8(sadhip_ctx* ctx, int n,
                            const uint8_t* d_left, const uint8_t* d_right,
                            size_t left_frame_stride, size_t right_frame_stride,
                            int w, int h, int lstride, int rstride,
                            int block, int maxdisp, unsigned flags,
                            uint8_t* d_out, size_t out_frame_stride, int ostride,
                            void* hip_stream);

/* ---- single-window SADs (host buffers, synchronous) -----------------------------------------
 * Replaces SumAbsoluteDifferences (sad.go:205-244) for n queries at once, including its
 * independent clamping of the right window (sad.go:217-218) and the early row/column breaks
 * (sad.go:225-227,231-233).  queries = n x {leftX,leftY,rightX,rightY}; right image may have
 * its own size.  out_sad[i] = the int the Go function returns. */
SADHIP_API int sadhip_sad_windows_u8(sadhip_ctx* ctx,
                          const uint8_t* left, int lw, int lh, int lstride,
                          const uint8_t* right, int rw, int rh, int rstride,
                          int block, int n, const int32_t* queries, int32_t* out_sad);

/* ---- gray conversion feeding the path (host buffers, synchronous) ---------------------------
 * Replaces convertGenericToGray on *image.NRGBA (pkg/despair/gray.go:43-58) and the
 * color.GrayModel.Convert loops of pkg/camera/output.go:142-162:
 *   c16 = c8*0x101 * (a8*0x101) / 0xffff ;  Y = (19595 r + 38470 g + 7471 b + 2^15) >> 24.
 * pix: interleaved 8-bit, channels = 4 (non-premultiplied RGBA, PNG colour type 6) or
 * 3 (opaque RGB).  The all-zero bug of convertRGBAToGray (gray.go:20-40) is NOT reproduced. */
SADHIP_API int sadhip_gray_u8(sadhip_ctx* ctx, const uint8_t* pix, int channels,
                   int w, int h, int stride, uint8_t* gray, int gstride);

/* RGBA/RGB pair in, disparity out: gray conversion fused in front of the SAD kernel on device. */
SADHIP_API int sadhip_disparity_rgba_u8(sadhip_ctx* ctx,
                             const uint8_t* left_pix, const uint8_t* right_pix, int channels,
                             int w, int h, int lstride, int rstride,
                             int block, int maxdisp, unsigned flags,
                             uint8_t* out, int ostride);

/* ---- timing of the last synchronous call on this context (the "elapsed" log line of
 * pkg/camera/output.go:200-204): device time of the kernel(s) and wall time incl. copies. */
SADHIP_API int sadhip_last_timing(sadhip_ctx* ctx, float* kernel_ms, float* total_ms);

/* Which kernel a given parameter set dispatches to: 0 = generic, 1 = column-wave fast kernel.
 * (diagnostics / tests; needs no GPU) */
SADHIP_API int sadhip_kernel_kind(int w, int h, int block, int maxdisp, unsigned flags);

/* How the fast kernel would run a parameter set (diagnostics / tests; needs no GPU): writes one
 * line per pass, "dbase=<first candidate> count=<candidates> p=<columns per lane> nd=<chunk> nw=<waves>
 * packed=<0|1> extra=<0|1> w16=<0|1> lds=<bytes>", into buf (NUL-terminated, truncated to cap).
 * Returns the number of passes, 0 when the generic kernel would run, -1 on invalid parameters. */
SADHIP_API int sadhip_plan_describe(int w, int h, int block, int maxdisp, char* buf, int cap);

#ifdef __cplusplus
}
#endif
#endif /* SADHIP_H */
