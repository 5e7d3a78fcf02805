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
 * by an internal mutex.  Destroying a context while another thread is inside a call on it is a
 * use-after-free like in any C API: the bindings in INTEGRATION.md hold a reader lock across each
 * call.  Host pointers are never retained after a call returns, error returns included (cgo rule):
 * the host thread copies caller memory into / out of page-locked planes the context owns and only those
 * are handed to the copy engines (caller memory inside page-locked ranges this library knows,
 * sadhip_host_* below, is handed over as it is), and every host-buffer entry point waits for its
 * stream before it returns.
 * sadhip_last_error() is thread-local: read it on the OS thread that made the failing call (a Go
 * caller wraps call + read in runtime.LockOSThread).
 * The library reads no environment variable (tuning knobs exist only in the diagnostics build
 * lib/libsadhip_tune.so used by tools/).
 */
#ifndef SADHIP_H
#define SADHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SADHIP_ABI_VERSION 5   /* 2: flag validation, MIRROR_VERILOG, sadhip_last_kernel_kind, sadhip_pool_*;
                                * 3: sadhip_disparity_lr_u8 (left / right images of different sizes), sadhip_host_* (page-locked
                                *    host memory), sadhip_plan_describe_batch;
                                * 4: sadhip_pool_describe / sadhip_device_describe / sadhip_pool_host_alloc (NUMA placement), sadhip_host_register
                                *    refuses overlapping ranges, the generic kernel's work bound (SADHIP_GENERIC_MAX_WORK),
                                *    "nwh" in the plan lines;
                                * 5: sadhip_host_register accepts whole pages of a mapping of the caller's own only (never the program
                                *    break or a stack), sadhip_ctx_set_copy_spin_us / _get_copy_spin_us / sadhip_pool_set_copy_spin_us,
                                *    SADHIP_FLAG_HOST_GRAY for sadhip_disparity_rgba_u8 */

#if defined(__GNUC__)
#define SADHIP_API __attribute__((visibility("default")))
#else
#define SADHIP_API
#endif

/* status codes */
#define SADHIP_OK               0
#define SADHIP_ERR_NULL         1   /* a required pointer is NULL */
#define SADHIP_ERR_DIMS         2   /* w <= 0 or h <= 0, larger than the context capacity, or a right image too small for the call */
#define SADHIP_ERR_STRIDE       3   /* a row stride is smaller than w */
#define SADHIP_ERR_BLOCK        4   /* block < 1 or block > SADHIP_MAX_BLOCK */
#define SADHIP_ERR_MAXDISP      5   /* maxdisp < 1 (the reference divides by zero) or > SADHIP_MAX_MAXDISP */
#define SADHIP_ERR_REGION       6   /* region not inside [0,w) x [0,h) or empty/inverted */
#define SADHIP_ERR_BATCH        7   /* n < 0 (a batch may be of any length), or max_batch < 1 at context creation */
#define SADHIP_ERR_NO_DEVICE    8   /* no HIP device / bad device ordinal */
#define SADHIP_ERR_HIP          9   /* a HIP runtime call failed; see sadhip_last_error() */
#define SADHIP_ERR_ARG         10   /* any other invalid argument: unknown flag bits, two mirror modes at once, ... */
#define SADHIP_ERR_NOMEM       11   /* host or device allocation failed */

#define SADHIP_MAX_BLOCK     255
#define SADHIP_MAX_MAXDISP  4096
/* Work bound of the generic path (ABI 4).  Parameter sets the column-wave kernel does not cover (BlockSize > 31, a right image
 * smaller than the left one, ...) and SADHIP_FLAG_FORCE_GENERIC run a one-thread-per-pixel kernel whose cost per pixel is the
 * reference's own: (2*(block/2)+1)^2 * (maxdisp+1) byte differences.  Above this many per pixel the library runs the same
 * function with running sums instead (vertical column sums per thread, prefix sums across the workgroup): O(maxdisp) per
 * pixel whatever the block size -- 1920x1080, block 63, maxdisp 64 in milliseconds where the naive kernel needs 0.2 s, and no
 * legal call (block <= 255, maxdisp <= 4096) costs more than about 1e-4 ms per pixel-candidate.  sadhip_kernel_kind() /
 * sadhip_last_kernel_kind() report 5 for that form, 0 for the naive one. */
#define SADHIP_GENERIC_MAX_WORK 65536

/* sadhip_disparity_* "flags" argument.  Bits not defined here, or more than one MIRROR_* bit,
 * are rejected with SADHIP_ERR_ARG. */
#define SADHIP_FLAG_RAW        1u   /* store best d itself (clamped to 255) instead of best*255/maxdisp */
#define SADHIP_FLAG_FORCE_GENERIC 2u /* diagnostics: use the generic kernel even where the fast one applies */
/* Hardware-mirror modes: compute with the semantics of the reference's OTHER golden models instead
 * of pkg/despair's, raw d output, so the FPGA testbench fixtures can be regenerated at any size
 * (SURVEY.md section 8 f-4, a-10).  Mutually exclusive; block = the models' WIN, maxdisp = MAX_DISP. */
#define SADHIP_FLAG_MIRROR_SADC   4u /* hardware/sad.c:66-106 (interior only, xr<0 terms skipped) */
#define SADHIP_FLAG_MIRROR_TESTPY 8u /* hardware/test.py:262-294 compute_disparity_patch (zero padding) */
#define SADHIP_FLAG_MIRROR_VERILOG 16u /* the search the Verilog TESTBENCH golden values describe (hardware/mems/exp_disp_n.mem,
                                        * old_mems/exp_disp.mem, read by ez_tb.v:69-116): the right window slides to LARGER
                                        * columns, contiguously -- argmin_d sum |L[r][c] - R[r][c+d]| -- d in [0, maxdisp)
                                        * EXCLUSIVE (compute_SAD.v:206-215) and only while the window stays inside the image,
                                        * strict '<' (:201-204), interior pixels only (border 0); a WIN-row strip of the
                                        * testbenches gives its result at (x, y) = (col_index + WIN/2, WIN/2).
                                        * This is the INTENDED design, not the RTL as written: compute_max_disp shifts in
                                        * img_block_R[row][col_index + disp_idx + 1] (:213) where a contiguous window needs
                                        * column col_index + WIN + disp_idx, so from d = 1 on its window re-reads columns it
                                        * already holds, and it outputs best_disp - (WIN-1) (:219).  The mode reproduces the
                                        * committed golden files; fixtures regenerated with it describe the intended
                                        * sliding-window search the RTL is being brought up against. */

#define SADHIP_FLAG_HOST_GRAY 32u  /* sadhip_disparity_rgba_u8 ONLY (SADHIP_ERR_ARG elsewhere; ABI 5): the host threads that stage the two
                                    * decoded pictures into page-locked memory write LUMINANCE rows instead of copies of the pictures
                                    * -- the same integer formula as the device kernel, bit for bit (gray.go:52-55 on
                                    * color.NRGBA.RGBA()) -- so a quarter of the bytes cross the bus (1080p RGBA pair: 16.6 MB ->
                                    * 4.1 MB).  For callers whose pictures are in ordinary (pageable) memory and whose
                                    * cores are free; without the flag the pictures are uploaded as they are and converted on
                                    * the device (the default; the better choice for pictures in sadhip_host_alloc memory). */

typedef struct sadhip_ctx sadhip_ctx;   /* opaque per-device context */

/* ---- library / device queries ------------------------------------------------------------ */
SADHIP_API int         sadhip_abi_version(void);
SADHIP_API int         sadhip_device_count(int* count);          /* replaces: nothing (reference is CPU-only) */
SADHIP_API const char* sadhip_status_string(int status);
SADHIP_API const char* sadhip_last_error(void);                  /* thread-local detail of the last failure */

/* ---- context ------------------------------------------------------------------------------
 * Replaces the worker pool of SetupConcurrentSAD (pkg/despair/sad.go:29-113): one context owns
 * the device planes (three frame slots; each gets page-locked host planes of the same size when the
 * first frame in memory that is not page-locked arrives), scratch and HIP streams for frames up to
 * max_w x max_h.
 * max_batch (>= 1) is a sizing hint kept for ABI stability: batches of ANY length stream through
 * the slots.  Created when the Go shim's pipeline is set up, destroyed when its input channel
 * is closed (sad.go:106-110, pkg/camera/output.go:213-220). */
SADHIP_API int sadhip_ctx_create(int device, int max_w, int max_h, int max_batch, sadhip_ctx** out);
SADHIP_API int sadhip_ctx_destroy(sadhip_ctx* ctx);
SADHIP_API int sadhip_ctx_device(const sadhip_ctx* ctx, int* device);
/* Host threads that copy caller memory which is not page-locked into / out of the context's page-locked planes
 * (ABI 4): n = 1..32 INCLUDING the calling thread; the default is SADHIP_DEFAULT_COPY_THREADS.  1 = the calling
 * thread copies alone (no helper threads exist).  The helpers are started by the first copy of 512 KiB or more,
 * touch host memory only, poll for 150 us after a piece of work and sleep otherwise.  Takes effect from the next
 * call on the context; SADHIP_ERR_ARG outside 1..32. */
#define SADHIP_DEFAULT_COPY_THREADS 6
SADHIP_API int sadhip_ctx_set_copy_threads(sadhip_ctx* ctx, int n);
SADHIP_API int sadhip_ctx_get_copy_threads(sadhip_ctx* ctx, int* n);
/* How long such a thread (and the caller waiting for them) polls for the next piece of work before it sleeps on a condition
 * variable (ABI 5): us = 0..SADHIP_MAX_COPY_SPIN_US microseconds, default SADHIP_DEFAULT_COPY_SPIN_US.  The pieces of one call
 * -- left plane, right plane, result -- follow each other within that time, and waking a sleeping core costs 20-50 us per piece
 * (1080p, one call: 0 = +0.03..0.06 ms).  A host that calls once a second and wants no polling at all in its process (the
 * WebUI's OutputCamera) sets 0: the threads then sleep as soon as they have no work.  Takes effect at once. */
#define SADHIP_DEFAULT_COPY_SPIN_US 150
#define SADHIP_MAX_COPY_SPIN_US 10000
SADHIP_API int sadhip_ctx_set_copy_spin_us(sadhip_ctx* ctx, int us);
SADHIP_API int sadhip_ctx_get_copy_spin_us(sadhip_ctx* ctx, int* us);

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

/* ---- left and right images of DIFFERENT sizes, host buffers (synchronous) ---------------------
 * Every other disparity entry point takes ONE w x h: both images must hold at least w x h pixels (they
 * read exactly w bytes of h rows of each).  The reference has no such rule: pkg/camera/output.go:134-162
 * decodes left.png and right.png independently, the worker iterates over the LEFT image's rectangle and
 * SumAbsoluteDifferences clamps the left window to left.Rect, walks the right window with the same
 * extent and breaks where it leaves right.Rect.Max (sad.go:212-218,225-227,231-233).  This entry point
 * is that contract: region [x0,x1) x [y0,y1) of the left image (image.Rect(0,0,lw,lh) for RunSad,
 * InputChunk.Region for a worker chunk), out = (y1-y0) rows of (x1-x0) bytes, ostride apart.
 *   right at least lw x lh:  the ordinary job on its top-left lw x lh part (no right pixel beyond it is ever
 *                            addressed), column-wave kernel;
 *   right smaller:           window terms whose right pixel does not exist are dropped, as the reference's
 *                            breaks do (pixels whose whole right window is missing get SAD 0 at d = 0);
 *                            generic kernel (sadhip_last_kernel_kind() == 0).  Hardware-mirror flags are
 *                            rejected with SADHIP_ERR_DIMS there (those models have one image size).
 * Only right pixels that exist are read from the caller's memory. */
SADHIP_API int sadhip_disparity_lr_u8(sadhip_ctx* ctx,
                           const uint8_t* left, int lw, int lh, int lstride,
                           const uint8_t* right, int rw, int rh, int rstride,
                           int block, int maxdisp, unsigned flags,
                           int x0, int y0, int x1, int y1,
                           uint8_t* out, int ostride);

/* ---- batch of independent frames, host buffers (synchronous; pipelined inside) --------------
 * Replaces the streaming loop of examples/run.stream.go:30-59 / BASELINE config 4: n independent
 * pairs (any n >= 0), H2D / kernel / D2H pipelined over the context's three slots on one stream per direction (the host
 * thread stages frame i+1 while frame i is on the device), nothing is retained after return.
 * When every frame of the batch lies in page-locked memory this library knows (sadhip_host_alloc /
 * sadhip_host_register below), the batch runs as groups of frames per kernel launch with truly
 * asynchronous linear copies (frames that follow each other in memory travel as one copy): the
 * PCIe-bound rate of pkg/camera's frame stream instead of the host thread's staging rate. */
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

/* RGBA/RGB pair in, disparity out: ONE upload per image (staged in ~2 MB bands, or one linear copy
 * from page-locked memory this library knows), gray
 * conversion on the device into the context's planes, disparity kernel, one download -- no host
 * gray plane, no second H2D (what LoadPNG + RunSad cost separately). */
SADHIP_API int sadhip_disparity_rgba_u8(sadhip_ctx* ctx,
                             const uint8_t* left_pix, const uint8_t* right_pix, int channels,
                             int w, int h, int lstride, int rstride,
                             int block, int maxdisp, unsigned flags,
                             uint8_t* out, int ostride);

/* ---- timing of the last synchronous call on this context (the "elapsed" log line of
 * pkg/camera/output.go:200-204): device time of the kernel(s) and wall time incl. copies. */
SADHIP_API int sadhip_last_timing(sadhip_ctx* ctx, float* kernel_ms, float* total_ms);

/* Which kernel a given parameter set dispatches to (diagnostics / tests; needs no GPU):
 * 0 = generic, 1 = column-wave fast kernel, 2 / 3 / 4 = the sad.c / test.py / compute_SAD.v mirror
 * kernels, 5 = generic in its running-sum form (per-pixel work above SADHIP_GENERIC_MAX_WORK); -1 on invalid parameters or flags.  The answer is binding: a job planned for the
 * column-wave kernel either runs on it or fails with SADHIP_ERR_HIP -- there is no silent fallback. */
SADHIP_API int sadhip_kernel_kind(int w, int h, int block, int maxdisp, unsigned flags);

/* Which kernel the LAST successful disparity call of the calling thread actually launched (same
 * numbering; -1 before any call).  Tests and bench.py assert it equals sadhip_kernel_kind(). */
SADHIP_API int sadhip_last_kernel_kind(void);

/* How the fast kernel would run a parameter set (diagnostics / tests; needs no GPU): writes one
 * line per pass, "dbase=<first candidate> count=<candidates> p=<columns per lane> nd=<chunk> nw=<waves>
 * packed=<0|1> extra=<0|1> w16=<0|1> nwh=<trailing waves with chunks of nd/2: mixed-chunk plans, else 0> lds=<bytes>", into buf
 * (NUL-terminated, truncated to cap).
 * Returns the number of passes, 0 when the generic kernel would run, -1 on invalid parameters. */
SADHIP_API int sadhip_plan_describe(int w, int h, int block, int maxdisp, char* buf, int cap);

/* The same for a batch of n frames ON THE CALLING THREAD'S CURRENT HIP DEVICE, with the launch geometry
 * appended to every line: " tiles=<workgroups across a frame> strips=<strips per frame> rows=<output rows per
 * strip> wg_per_cu=<resident workgroups of this kernel per CU> device=<1: asked from the runtime, 0: no device,
 * one workgroup per CU assumed>".  Strip s of a frame computes the output rows [s*rows, min((s+1)*rows, h)):
 * the seams the full-size parity tests put their literal-oracle rows on. */
SADHIP_API int sadhip_plan_describe_batch(int w, int h, int block, int maxdisp, int n, char* buf, int cap);

/* ---- page-locked host memory ---------------------------------------------------------------------
 * The reference hands the worker Go heap memory (image.Gray.Pix); a cgo callee may not keep such
 * pointers, and memory the library did not lock itself is never handed to a copy engine (round 4: letting the
 * HIP runtime lock the caller's pageable pages in place was 0.13 ms per 1080p call faster and ended a process
 * with a GPU memory access fault on a host heap page three times in some twenty test-suite runs; DESIGN.md
 * section 6b).  So the host-buffer entry points stage pageable memory through the context's own page-locked
 * planes: about a third of the PCIe rate, and the host thread is busy while it happens.  A capture pipeline
 * (pkg/camera/serial.go fills a frame buffer per UART frame) or the C++ / Python host can keep its frames
 * in page-locked memory instead: sadhip_host_alloc returns hipHostMalloc'ed memory usable from every
 * device (a Go caller wraps it with unsafe.Slice as image.Gray.Pix -- C memory, not Go memory);
 * sadhip_host_register page-locks memory the caller already owns (C / C++ hosts only: NEVER Go heap
 * memory) until sadhip_host_unregister.  Every host-buffer entry point recognises pointers inside such
 * ranges and then uses asynchronous linear copies; results are identical either way.  Free / unregister a
 * range only when no call that was given pointers into it is still running (every entry point has finished
 * with the caller's memory when it returns).
 * Prefer sadhip_host_alloc where the host can choose its frame memory.  sadhip_host_register (hipHostRegister) locks the
 * CALLER's pages in place -- the mechanism the withdrawn pageable path relied on per call -- and is
 * therefore limited to memory no allocator shares or recycles (ABI 5).  CONTRACT, checked at the call against /proc/self/maps:
 *   - p is 4 KiB-aligned and bytes is a multiple of 4096: whole pages, so that no byte of another object shares a locked page;
 *   - every page is mapped readable + writable by a mapping of the caller's own (mmap: anonymous, shared memory, a driver's
 *     capture buffers); memory inside the program break ("[heap]": malloc's main arena returns pages to the system and takes
 *     them back under the caller) or a stack is refused -- as is anything that cannot be verified;
 *   - NOT checked, the caller's duty: the mapping stays mapped, at this address, until sadhip_host_unregister returns -- no
 *     munmap / mremap / madvise(MADV_DONTNEED / MADV_FREE) of any part of it, and no free() if it came from an allocator's
 *     own mmap (a large malloc / posix_memalign block is such a mapping only as long as the allocator says so: do not rely on it).
 *   Anything else returns SADHIP_ERR_ARG and registers nothing; sadhip_last_error() says which rule.
 * The known ranges are DISJOINT page by page: sadhip_host_register of memory that shares a 4 KiB page with a sadhip_host_alloc
 * block or an earlier registration (a sub-range, a super-range, a partial overlap) returns SADHIP_ERR_ARG and registers nothing. */
SADHIP_API int sadhip_host_alloc(size_t bytes, void** out);
SADHIP_API int sadhip_host_free(void* p);
SADHIP_API int sadhip_host_register(void* p, size_t bytes);
SADHIP_API int sadhip_host_unregister(void* p);

/* ---- multi-device pool (host buffers, synchronous) -------------------------------------------
 * Replaces, for a host with several GPUs, the streaming loop of examples/run.stream.go:30-59 at
 * BASELINE config 4's shape: one context and one host thread per entry of devices[] (an ordinal may
 * be listed more than once: several contexts on one GPU), frame i of a batch runs on entry
 * i mod n_devices, the two parameters are fanned out by the host, results land in outs[i].  There is
 * no device-to-device exchange on this path.  One batch at a time per pool; the calling thread
 * blocks until every device has finished (on failure too: no caller pointer stays in use). */
typedef struct sadhip_pool sadhip_pool;
SADHIP_API int sadhip_pool_create(const int* devices, int n_devices, int max_w, int max_h, sadhip_pool** out);
SADHIP_API int sadhip_pool_destroy(sadhip_pool* pool);
SADHIP_API int sadhip_pool_size(const sadhip_pool* pool, int* n_devices);
/* sadhip_ctx_set_copy_threads for every entry's context (n per entry, 1..32).  A pool of G devices runs G worker threads (one
 * per entry, each inside its device's calls) and, once an entry has seen a frame that is not in page-locked memory, n - 1 copy
 * helpers under it: eight devices at the default are 8 + 40 = 48 host threads; frames from sadhip_host_alloc /
 * sadhip_pool_host_alloc never start the helpers. */
SADHIP_API int sadhip_pool_set_copy_threads(sadhip_pool* pool, int n);
/* sadhip_ctx_set_copy_spin_us for every entry's context (ABI 5). */
SADHIP_API int sadhip_pool_set_copy_spin_us(sadhip_pool* pool, int us);
SADHIP_API int sadhip_pool_batch_u8(sadhip_pool* pool, int n,
                         const uint8_t* const* lefts, const uint8_t* const* rights,
                         int w, int h, int lstride, int rstride,
                         int block, int maxdisp, unsigned flags,
                         uint8_t* const* outs, int ostride);
/* kernel_ms_sum: device time of the kernels summed over the pool's devices; wall_ms: of the call */
SADHIP_API int sadhip_pool_last_timing(sadhip_pool* pool, float* kernel_ms_sum, float* wall_ms);
/* Host placement of the pool (ABI 4).  Each worker thread -- the thread that issues its device's copies and launches -- is
 * pinned at creation to the CPUs of the device's NUMA node: PCI bus id (hipDeviceGetPCIBusId) ->
 * /sys/bus/pci/devices/<id>/local_cpulist, cut down to the CPUs the process itself may use; best effort (a host without
 * that sysfs file, or an empty intersection, leaves the thread where the scheduler puts it).  Writes one line per entry,
 * "entry=<i> device=<ordinal> pci=<bus id> numa_node=<n or -1> pinned=<0|1> cpus=<list or ->", into buf (NUL-terminated,
 * truncated to cap).  The frames of a batch should come from memory of the same node: allocate (sadhip_host_alloc) and
 * first-touch device i's frames -- i, i + G, ... -- from a thread running on entry i's cpus; frames in one block
 * allocated by one thread cross the socket link for half the devices of a two-socket host. */
SADHIP_API int sadhip_pool_describe(const sadhip_pool* pool, char* buf, int cap);
/* Page-locked memory for the frames of pool entry `entry` (frames entry, entry + G, ... of a batch), allocated AND first-touched
 * by that entry's worker thread -- pinned to the CPUs next to its device, with that device selected as the thread's current one
 * (the runtime places page-locked memory by the allocating thread's current device); sadhip_host_alloc semantics otherwise
 * (usable from every device; free with sadhip_host_free from any thread).  Not while a batch is running on the pool (it waits for it). */
SADHIP_API int sadhip_pool_host_alloc(sadhip_pool* pool, int entry, size_t bytes, void** out);
/* The same lookup for a host that drives ONE context itself (the Go shim's case): "device=<ordinal> pci=<bus id>
 * numa_node=<n or -1> cpus=<the device's local CPUs within this process's affinity mask, or ->".  A caller that pins the
 * thread making its calls to those CPUs (Go: runtime.LockOSThread + unix.SchedSetaffinity) and allocates its page-locked
 * frames from there keeps its copies off the socket link of a two-socket host. */
SADHIP_API int sadhip_device_describe(int device, char* buf, int cap);

#ifdef __cplusplus
}
#endif
#endif /* SADHIP_H */
