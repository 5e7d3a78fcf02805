// sadhip_kernels.h -- launch interface between the C-ABI host code (sadhip.hip) and the
// gfx950 kernels (kernels_generic.hip, kernels_fast.hip).  Internal; not installed.
#pragma once
#include "../../include/sadhip.h"
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

namespace sadhip {

// One batched disparity job on device-resident frames.  Frame f of each plane lives at
// base + f * frame_stride bytes; rows are `stride` bytes apart.  The output rectangle is
// [x0,x1) x [y0,y1) of every frame, written at out + f*out_frame_stride + (y-oy)*ostride + (x-ox)
// (ox,oy = origin of the output buffer in image coordinates: (0,0) for whole frames,
// (x0,y0) for region-local output).
struct Job {
  const uint8_t* left;  size_t left_frame_stride;  int lstride;
  const uint8_t* right; size_t right_frame_stride; int rstride;
  uint8_t* out;         size_t out_frame_stride;   int ostride;
  int n;                 // frames
  int w, h;              // image size (the LEFT image's: windows clamp to it, sad.go:212-215)
  // Size of the right image when it is SMALLER than the left one in a dimension (0 = at least w / h).  The
  // reference walks the right window with the left window's extent and breaks where it leaves right.Rect.Max
  // (sad.go:225-227,231-233): terms whose right pixel lies outside the right image are dropped.  Only the generic
  // kernel implements that; fast_supported() is false for such a job.  A larger right image needs nothing: no
  // right pixel beyond (w, h) is ever addressed.
  int rw, rh;
  int block, maxdisp;    // BlockSize, MaxDisparity (pkg/despair/params.go:32-37)
  int x0, y0, x1, y1;    // output rectangle (image coordinates)
  int ox, oy;            // output buffer origin (image coordinates)
  int raw;               // 1: store best d (clamped to 255); 0: best*255/maxdisp
  // Rows of the input planes actually present in memory: plane row r holds image row r + row0
  // (row0 = 0 for whole frames; > 0 when only a band was uploaded for a region call).
  int row0;
};

// Which kernel the calling thread launched last (every launch_* records it on success); the C ABI
// reports it through sadhip_last_kernel_kind() so tests and the bench can assert which code ran.
enum { KIND_NONE = -1, KIND_GENERIC = 0, KIND_COLWAVE = 1, KIND_MIRROR_SADC = 2, KIND_MIRROR_TESTPY = 3, KIND_MIRROR_VERILOG = 4,
       KIND_GENERIC_SLIDING = 5 };
void note_launch(int kind);
int last_launch_kind();

// Stream-ordered scratch (the u32 / u64 key per pixel that travels between the passes of a multi-pass job, the running-sum
// kernel's keys): hipMallocFromPoolAsync / hipFreeAsync on the job's stream, from one pool per device that the library owns
// (release threshold 1 GiB: kernels_generic.hip).
hipError_t scratch_alloc(void** p, size_t bytes, hipStream_t s);
hipError_t scratch_free(void* p, hipStream_t s);

// Generic kernel: any block / maxdisp / size.  One thread per output pixel, closed-form
// restatement of sad.go:55-94 with u32 accumulators.  Always available.
hipError_t launch_generic(const Job& j, hipStream_t s);
// ... in its running-sum form (kernels_sliding.hip): O(1) work per (pixel, candidate) whatever BlockSize is.  launch_generic
// routes a job there when a pixel's naive work, (2h+1)^2 * (D+1) byte differences, exceeds SADHIP_GENERIC_MAX_WORK.
hipError_t launch_sliding(const Job& j, hipStream_t s);
bool generic_uses_sliding(int block, int maxdisp);

// Hardware-mirror golden models (mode 1: hardware/sad.c, mode 2: hardware/test.py, mode 3: the strip
// search of hardware/compute_SAD.v); raw d output.
hipError_t launch_mirror(const Job& j, int mode, hipStream_t s);

// Column-wave kernel (colwave_kernel.h, launched from kernels_fast.hip): fast_supported = it has a plan for the job.
// launch_fast plans once per call and runs the generic kernel itself for a job without a plan.
bool fast_supported(const Job& j);
hipError_t launch_fast(const Job& j, hipStream_t s);
// Text description of the passes launch_fast would run (empty when it would not): see sadhip_plan_describe;
// geometry: append tiles / strips / rows per strip / workgroups per CU of a launch of j.n frames on the current device.
int describe_fast(const Job& j, char* buf, int cap, bool geometry);

// Single-window SADs, literal clamps of sad.go:205-244.
hipError_t launch_sad_windows(const uint8_t* left, int lw, int lh, int lstride,
                              const uint8_t* right, int rw, int rh, int rstride,
                              int block, int n, const int32_t* queries, int32_t* out,
                              hipStream_t s);

// Gray conversion (gray.go:43-58 on NRGBA / opaque RGB), n planes.
hipError_t launch_gray(const uint8_t* pix, int channels, int w, int h, int stride,
                       uint8_t* gray, int gstride, hipStream_t s);

}  // namespace sadhip
