// kernels_sliding.hip -- the generic disparity function with RUNNING sums: any BlockSize (1..255), any MaxDisparity, any
// image size, a right image smaller than the left one included; O(1) work per (pixel, candidate) instead of the
// O(BlockSize^2) of k_disparity_generic (kernels_generic.hip), whose cost on a legal call has no useful bound
// (3840x2160, BlockSize 255, MaxDisparity 64: 3.5e13 byte differences in one launch).  launch_generic() routes a job here
// when a pixel's naive work, (2h+1)^2 * (D+1) byte differences, exceeds SADHIP_GENERIC_MAX_WORK (include/sadhip.h).
//
// Same function as every other kernel here (closed form of pkg/despair/sad.go:55-94 + :205-244):
//   candidates d in [0, min(D, max(0, x-h))], SAD over the window clipped to the LEFT image, terms whose right pixel lies
//   outside a smaller right image dropped (sad.go:225-227,231-233), first minimum (strict '<', sad.go:79), best*255/D.
//
// Mapping: a workgroup of 256 threads owns 512 adjacent columns (two per thread) of one strip of rows of one frame and
// walks the candidates d one after the other.  Per d: every thread slides the vertical sums of its two columns down the
// strip (+ |L-R| of the row entering the (2h+1)-row window, - |L-R| of the row leaving it); per output row the 512 sums
// are prefix-summed across the workgroup (wave scan + wave totals through LDS) and every output pixel takes
// prefix[x+h] - prefix[x-h-1]: its window SAD for d.  The running minimum (SAD << 16 | d, 64 bits) lives in a global
// scratch row per output row -- touched only by the thread that owns the column, so no atomics -- and the owning thread
// turns it into the output byte after the last d.  2h halo columns per side: 512 - 2h >= 258 output columns per tile.
// Not a speed-of-light kernel (the column-wave kernel is, for BlockSize <= 31): a BOUNDED one.
#include "sadhip_kernels.h"
#include <algorithm>

namespace sadhip {

namespace {

constexpr int SL_NT = 256, SL_E = 2, SL_SLOTS = SL_NT * SL_E;

struct SlArgs {
  Job j;
  int th;                         // output rows per strip
  int txo;                        // output columns per tile (SL_SLOTS - 2h)
  unsigned long long* keys;       // [frame][y - y0][x - x0]
  int kw, krows;
};

__device__ __forceinline__ unsigned wave_inclusive_scan(unsigned v, int lane)
{
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned t = __shfl_up(v, o, 64);
    if (lane >= o) v += t;
  }
  return v;
}

__global__ void __launch_bounds__(SL_NT) k_disparity_sliding(SlArgs a)
{
  __shared__ unsigned Q[SL_SLOTS + 1];            // Q[i + 1] = inclusive prefix sum of slot i; Q[0] = 0
  __shared__ unsigned wtot[SL_NT / 64];
  const Job& j = a.j;
  const int hs = j.block / 2, B1 = 2 * hs + 1;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int f = blockIdx.z;
  const int ox0 = j.x0 + (int)blockIdx.x * a.txo, ox1 = min(ox0 + a.txo, j.x1);     // output columns of this tile
  const int ys = j.y0 + (int)blockIdx.y * a.th, ye = min(ys + a.th, j.y1);          // output rows of this strip
  if (ox0 >= ox1 || ys >= ye) return;
  const int cx0 = ox0 - hs;                                                         // image column of slot 0
  const uint8_t* __restrict__ L = j.left + (size_t)f * j.left_frame_stride;
  const uint8_t* __restrict__ R = j.right + (size_t)f * j.right_frame_stride;
  const int rwid = j.rw > 0 ? j.rw : j.w;                                           // right pixels that exist
  const int rows_in = j.rh > 0 ? min(j.rh, j.h) : j.h;                              // rows where both images exist
  const int s0 = SL_E * tid;
  int col[SL_E];
  bool outp[SL_E];                                                                  // this slot is an output pixel of the tile
#pragma unroll
  for (int e = 0; e < SL_E; ++e) {
    col[e] = cx0 + s0 + e;
    outp[e] = s0 + e >= hs && s0 + e < SL_SLOTS - hs && col[e] >= ox0 && col[e] < ox1;
  }
  unsigned long long* kf = a.keys + (size_t)f * a.krows * a.kw;
  for (int y = ys; y < ye; ++y)
#pragma unroll
    for (int e = 0; e < SL_E; ++e)
      if (outp[e]) kf[(size_t)(y - j.y0) * a.kw + (col[e] - j.x0)] = ~0ull;
  if (tid == 0) Q[0] = 0;

  // |L[r][c] - R[r][c - d]| where both pixels exist, else 0 (clipped window / dropped term)
  auto diff = [&](int r, int c, int d) -> unsigned {
    const int rc = c - d;
    if (r < 0 || r >= rows_in || c < 0 || c >= j.w || rc < 0 || rc >= rwid) return 0u;
    const int l = L[(size_t)(r - j.row0) * j.lstride + c], q = R[(size_t)(r - j.row0) * j.rstride + rc];
    return (unsigned)(l >= q ? l - q : q - l);
  };

  const int dmax_tile = min(j.maxdisp, max(0, ox1 - 1 - hs));                       // larger d: valid for no pixel of this tile
  for (int d = 0; d <= dmax_tile; ++d) {
    unsigned C[SL_E];
#pragma unroll
    for (int e = 0; e < SL_E; ++e) C[e] = 0u;
    for (int r = ys - hs; r < ye + hs; ++r) {                                       // row r enters the window, row r - B1 leaves it
      const bool leaving = r - B1 >= ys - hs;                                       // (only rows this strip has added: the sums start at row ys - hs)
#pragma unroll
      for (int e = 0; e < SL_E; ++e) C[e] += diff(r, col[e], d) - (leaving ? diff(r - B1, col[e], d) : 0u);
      const int y = r - hs;                                                         // the output row whose window is now complete
      if (y < ys) continue;                                                         // (uniform: warm-up rows)
      const unsigned pair = C[0] + C[1];
      const unsigned incl = wave_inclusive_scan(pair, lane);
      if (lane == 63) wtot[wv] = incl;
      __syncthreads();
      unsigned off = 0;
      for (int k = 0; k < wv; ++k) off += wtot[k];
      Q[s0 + 1] = off + incl - C[1];
      Q[s0 + 2] = off + incl;
      __syncthreads();
#pragma unroll
      for (int e = 0; e < SL_E; ++e) {
        if (!outp[e]) continue;
        if (d != 0 && d > col[e] - hs) continue;                                    // candidate does not exist for this pixel (sad.go:64-67)
        const int s = s0 + e;
        const unsigned sad = Q[s + hs + 1] - Q[s - hs];
        const unsigned long long key = ((unsigned long long)sad << 16) | (unsigned)d;
        unsigned long long* kp = kf + (size_t)(y - j.y0) * a.kw + (col[e] - j.x0);
        if (key < *kp) *kp = key;                                                   // first minimum: lowest d among equal SADs
      }
    }
  }
  // keys -> output bytes (written and read by the same thread: no synchronisation needed)
  for (int y = ys; y < ye; ++y)
#pragma unroll
    for (int e = 0; e < SL_E; ++e) {
      if (!outp[e]) continue;
      const int best = (int)(kf[(size_t)(y - j.y0) * a.kw + (col[e] - j.x0)] & 0xffffull);
      const int v = j.raw ? min(best, 255) : (best * 255) / j.maxdisp;              // sad.go:91-93
      j.out[(size_t)f * j.out_frame_stride + (size_t)(y - j.oy) * j.ostride + (col[e] - j.ox)] = (uint8_t)v;
    }
}

}  // namespace

hipError_t launch_sliding(const Job& j, hipStream_t s)
{
  const int kw = j.x1 - j.x0, rows = j.y1 - j.y0;
  if (kw <= 0 || rows <= 0 || j.n <= 0) return hipSuccess;
  const int hs = j.block / 2;
  if (hs < 0 || 2 * hs >= SL_SLOTS - 2 || j.maxdisp > 0xffff) return hipErrorInvalidValue;
  SlArgs a{};
  a.txo = SL_SLOTS - 2 * hs;
  const int tiles = (kw + a.txo - 1) / a.txo;
  // frames per launch: the keys of one launch stay below 256 MB
  const size_t per_frame = (size_t)kw * rows * sizeof(unsigned long long);
  const int chunk = (int)std::max<size_t>(1, std::min<size_t>({(size_t)j.n, (size_t)32768, ((size_t)256 << 20) / per_frame}));
  // strips: enough workgroups to fill the chip, but every strip repeats 2h warm-up rows -- never shorter than that
  const long want = 768;
  const long max_strips = std::max<long>(1, rows / std::max(16, 2 * hs));
  const long per_layer = (long)tiles * std::min(chunk, j.n);
  long nstrips = std::min<long>(max_strips, std::max<long>(1, (want + per_layer - 1) / per_layer));
  nstrips = std::min<long>(nstrips, 65535);
  a.th = (int)((rows + nstrips - 1) / nstrips);
  nstrips = (rows + a.th - 1) / a.th;
  a.kw = kw; a.krows = rows;
  unsigned long long* keys = nullptr;
  hipError_t e = scratch_alloc((void**)&keys, per_frame * (size_t)chunk, s);
  if (e != hipSuccess) return e;
  a.keys = keys;
  for (int f0 = 0; f0 < j.n && e == hipSuccess; f0 += chunk) {
    a.j = j;
    a.j.n = std::min(chunk, j.n - f0);
    a.j.left += (size_t)f0 * j.left_frame_stride;
    a.j.right += (size_t)f0 * j.right_frame_stride;
    a.j.out += (size_t)f0 * j.out_frame_stride;
    hipLaunchKernelGGL(k_disparity_sliding, dim3((unsigned)tiles, (unsigned)nstrips, (unsigned)a.j.n), dim3(SL_NT), 0, s, a);
    e = hipGetLastError();
  }
  const hipError_t ef = scratch_free(keys, s);
  if (e == hipSuccess && ef == hipSuccess) note_launch(KIND_GENERIC_SLIDING);
  return e != hipSuccess ? e : ef;
}

}  // namespace sadhip
