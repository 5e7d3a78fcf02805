/*
 * despair_oracle.c -- CPU restatement of the reference's SAD block-matching
 * disparity path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity oracle.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it; nothing under
 * steroscopic-hardware_amd/ (the product) links, imports or calls it.
 *
 * Pinning status (see tests/test_oracle_golden.py, DESIGN.md section "Oracle"):
 *   - the Go package pkg/despair has NO tests of its own for this path
 *     (SURVEY.md section 8c), so the restatement is pinned by
 *       (1) the reference's hardware fixtures hardware/exp_disp.mem,
 *           hardware/mems/exp_disp_p.mem, mems/exp_disp_n.mem,
 *           old_mems/exp_disp.mem, old_mems/exp_sad.mem, winL/winR.mem on the
 *           regions where the Go semantics coincide with them,
 *       (2) outputs of the reference's own C program hardware/sad.c compiled
 *           unmodified into oracle/_ref/ (oracle/Makefile), and
 *       (3) outputs of the reference's own Python hardware/test.py imported in
 *           the build container (tests/golden/make_golden.py).
 *   - the literal restatement (oracle_region_literal) and the closed form
 *     (oracle_region_closed) are asserted equal on every test shape.
 *
 * Each function cites the reference file:line it follows.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#include <pthread.h>

#define ORACLE_API __attribute__((visibility("default")))

static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }

/* ------------------------------------------------------------------------ */
/* pkg/despair/sad.go:205-244  SumAbsoluteDifferences                        */
/* Images are image.Gray with Rect.Min == (0,0): Pix[y*Stride + x].          */
/* lw/lh = left.Rect.Max, rw/rh = right.Rect.Max.                            */
/* ------------------------------------------------------------------------ */
ORACLE_API int oracle_sum_absolute_differences(
    const uint8_t *left, int lw, int lh, int lstride,
    const uint8_t *right, int rw, int rh, int rstride,
    int leftX, int leftY, int rightX, int rightY, int blockSize)
{
    int halfSize = blockSize / 2;                       /* sad.go:209 */

    int leftMinY = imax(leftY - halfSize, 0);           /* sad.go:212 */
    int leftMaxY = imin(leftY + halfSize + 1, lh);      /* sad.go:213 */
    int leftMinX = imax(leftX - halfSize, 0);           /* sad.go:214 */
    int leftMaxX = imin(leftX + halfSize + 1, lw);      /* sad.go:215 */

    int rightMinY = imax(rightY - halfSize, 0);         /* sad.go:217 */
    int rightMinX = imax(rightX - halfSize, 0);         /* sad.go:218 */

    int sad = 0;
    for (int ly = leftMinY; ly < leftMaxY; ly++) {      /* sad.go:224 */
        if (rightMinY + (ly - leftMinY) >= rh) break;   /* sad.go:225-227 */
        long leftRowStart = (long)ly * lstride + leftMinX;                       /* :228 */
        long rightRowStart = (long)(rightMinY + (ly - leftMinY)) * rstride + rightMinX; /* :229 */
        for (int lx = leftMinX; lx < leftMaxX; lx++) {  /* sad.go:230 */
            if (rightMinX + (lx - leftMinX) >= rw) break; /* sad.go:231-233 */
            int diff = (int)left[leftRowStart + lx - leftMinX] -
                       (int)right[rightRowStart + (rightMinX + (lx - leftMinX)) - rightMinX]; /* :234-235 */
            if (diff < 0) diff = -diff;                 /* sad.go:236-238 */
            sad += diff;                                /* sad.go:239 */
        }
    }
    return sad;
}

/* ------------------------------------------------------------------------ */
/* pkg/despair/sad.go:55-94  worker body, literal: same loop order, skip     */
/* rule (:64-67), strict '<' (:79), early exit (:84-86), scaling (:91-93).   */
/* Region [x0,x1) x [y0,y1); output region-local row-major (sad.go:48-50,91).*/
/* raw != 0 stores bestDisparity itself (u8, for the hardware fixtures).     */
/* Returns 0, or -1 on maxDisparity <= 0 (the reference panics there).       */
/* ------------------------------------------------------------------------ */
/* Left and right image with their OWN sizes, as the reference's types allow  */
/* (two *image.Gray; pkg/camera/output.go:134-162 decodes them independently): */
/* the region and the left-edge skip rule refer to the LEFT image             */
/* (sad.go:55-57,64-67), SumAbsoluteDifferences clamps / breaks per image.     */
ORACLE_API int oracle_region_literal_lr(
    const uint8_t *left, int lw, int lh, int lstride,
    const uint8_t *right, int rw, int rh, int rstride,
    int blockSize, int maxDisparity,
    int x0, int y0, int x1, int y1, uint8_t *out, int raw)
{
    if (maxDisparity <= 0) return -1;
    int dx = x1 - x0, dy = y1 - y0;
    for (int y = 0; y < dy; y++) {                       /* sad.go:55 */
        int globalY = y0 + y;                            /* sad.go:56 */
        for (int x = 0; x < dx; x++) {                   /* sad.go:57 */
            int minSAD = INT32_MAX;                      /* sad.go:59 */
            int bestDisparity = 0;                       /* sad.go:60 */
            for (int d = 0; d <= maxDisparity; d++) {    /* sad.go:62 */
                if ((x0 + x) - d < 0) continue;          /* sad.go:64-67 */
                int sad = oracle_sum_absolute_differences(
                    left, lw, lh, lstride, right, rw, rh, rstride,
                    x0 + x, globalY, x0 + x - d, globalY, blockSize); /* :69-77 */
                if (sad < minSAD) {                      /* sad.go:79 */
                    minSAD = sad;
                    bestDisparity = d;
                    if (sad == 0) break;                 /* sad.go:84-86 */
                }
            }
            out[(long)y * dx + x] = raw ? (uint8_t)bestDisparity
                : (uint8_t)((bestDisparity * 255) / maxDisparity); /* sad.go:91-93 */
        }
    }
    return 0;
}

ORACLE_API int oracle_region_literal(
    const uint8_t *left, const uint8_t *right, int w, int h,
    int lstride, int rstride, int blockSize, int maxDisparity,
    int x0, int y0, int x1, int y1, uint8_t *out, int raw)
{
    return oracle_region_literal_lr(left, w, h, lstride, right, w, h, rstride,
                                    blockSize, maxDisparity, x0, y0, x1, y1, out, raw);
}

/* ------------------------------------------------------------------------ */
/* Closed form of the same function:               */
/*   dmax(x) = min(D, max(0, x - hs)); candidates d in [0, dmax(x)];          */
/*   window  = [max(x-hs,0), min(x+hs+1,W)) x [max(y-hs,0), min(y+hs+1,H)),   */
/*   L and R aligned (R sampled at xx-d); first minimum wins.                 */
/* Why: for d > x-hs the independent clamp at sad.go:218 pins the right       */
/* window to column 0, giving the SAD of d = x-hs again, never strictly       */
/* smaller (sad.go:79); for x < hs every candidate ties with d = 0.           */
/* This is what the HIP kernels implement; tests assert literal == closed.    */
/* ------------------------------------------------------------------------ */
ORACLE_API int oracle_region_closed(
    const uint8_t *left, const uint8_t *right, int w, int h,
    int lstride, int rstride, int blockSize, int maxDisparity,
    int x0, int y0, int x1, int y1, uint8_t *out, int raw)
{
    if (maxDisparity <= 0) return -1;
    int hs = blockSize / 2;
    int dx = x1 - x0;
    for (int gy = y0; gy < y1; gy++) {
        int wy0 = imax(gy - hs, 0), wy1 = imin(gy + hs + 1, h);
        for (int gx = x0; gx < x1; gx++) {
            int wx0 = imax(gx - hs, 0), wx1 = imin(gx + hs + 1, w);
            int dmax = imin(maxDisparity, imax(0, gx - hs));
            int best = 0, minSAD = INT32_MAX;
            for (int d = 0; d <= dmax; d++) {
                int sad = 0;
                for (int yy = wy0; yy < wy1; yy++) {
                    const uint8_t *lp = left + (long)yy * lstride;
                    const uint8_t *rp = right + (long)yy * rstride - d;
                    for (int xx = wx0; xx < wx1; xx++) {
                        int diff = (int)lp[xx] - (int)rp[xx];
                        sad += diff < 0 ? -diff : diff;
                    }
                }
                if (sad < minSAD) { minSAD = sad; best = d; }
            }
            out[(long)(gy - y0) * dx + (gx - x0)] = raw ? (uint8_t)best
                : (uint8_t)((best * 255) / maxDisparity);
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------ */
/* Multi-threaded literal restatement, chunked like the production caller:   */
/* pkg/camera/output.go:172-187 -- full-width row bands of                   */
/* max(1, H/(32*4)) rows, pulled by worker threads (sad.go:41-104).          */
/* Used only as bench.py's cpu_baseline ("port").                            */
/* ------------------------------------------------------------------------ */
typedef struct {
    const uint8_t *left, *right; int w, h, lstride, rstride, bs, md;
    int ya, yb; /* rows [ya,yb) to do */
    uint8_t *out; int chunk;
    int next; pthread_mutex_t mu;
} mt_job;

static void *mt_worker(void *p)
{
    mt_job *j = (mt_job *)p;
    for (;;) {
        pthread_mutex_lock(&j->mu);
        int y = j->next; j->next += j->chunk;
        pthread_mutex_unlock(&j->mu);
        if (y >= j->yb) break;
        int ye = imin(y + j->chunk, j->yb);
        oracle_region_literal(j->left, j->right, j->w, j->h, j->lstride, j->rstride,
                              j->bs, j->md, 0, y, j->w, ye,
                              j->out + (long)(y - j->ya) * j->w, 0);
    }
    return NULL;
}

ORACLE_API int oracle_rows_literal_mt(
    const uint8_t *left, const uint8_t *right, int w, int h,
    int lstride, int rstride, int blockSize, int maxDisparity,
    int ya, int yb, uint8_t *out, int nthreads)
{
    if (maxDisparity <= 0) return -1;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    mt_job j = { left, right, w, h, lstride, rstride, blockSize, maxDisparity,
                 ya, yb, out, imax(1, h / (32 * 4)), ya, PTHREAD_MUTEX_INITIALIZER };
    pthread_t th[256];
    for (int i = 0; i < nthreads; i++) pthread_create(&th[i], NULL, mt_worker, &j);
    for (int i = 0; i < nthreads; i++) pthread_join(th[i], NULL);
    return 0;
}

/* ------------------------------------------------------------------------ */
/* pkg/despair/gray.go:43-58 convertGenericToGray on a *image.NRGBA source   */
/* (8-bit RGBA PNG, colour type 6, e.g. testdata/L_00001.png):               */
/*   stdlib color.NRGBA.RGBA(): c16 = c8*0x101 * (a8*0x101) / 0xffff         */
/*   gray.go:53-55: uint8((19595*r + 38470*g + 7471*b + 1<<15) >> 24)        */
/* The same value color.GrayModel.Convert yields (pkg/camera/output.go:145). */
/* in: RGBA8 interleaved, non-premultiplied; stride in bytes.                */
/* ------------------------------------------------------------------------ */
ORACLE_API void oracle_gray_from_nrgba(const uint8_t *rgba, int w, int h, int stride,
                                       uint8_t *gray, int gstride)
{
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const uint8_t *p = rgba + (long)y * stride + 4 * x;
            uint32_t a = (uint32_t)p[3] * 0x101;
            uint32_t r = (uint32_t)p[0] * 0x101 * a / 0xffff;
            uint32_t g = (uint32_t)p[1] * 0x101 * a / 0xffff;
            uint32_t b = (uint32_t)p[2] * 0x101 * a / 0xffff;
            gray[(long)y * gstride + x] =
                (uint8_t)((19595u * r + 38470u * g + 7471u * b + (1u << 15)) >> 24);
        }
}

/* pkg/despair/gray.go:20-40 convertRGBAToGray -- the reference applies >>24  */
/* to 8-bit channels, so every pixel becomes 0.     */
/* Restated literally so the bug is on record; channels = 3 or 4.             */
ORACLE_API void oracle_gray_from_rgba8_refbug(const uint8_t *pix, int w, int h, int stride,
                                              int channels, uint8_t *gray, int gstride)
{
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const uint8_t *p = pix + (long)y * stride + channels * x;
            gray[(long)y * gstride + x] =
                (uint8_t)((19595u * p[0] + 38470u * p[1] + 7471u * p[2] + (1u << 15)) >> 24);
        }
}

/* Opaque RGB8 through the generic/GrayModel formula (what output.go:145 does */
/* for *image.RGBA too, since it goes through color.GrayModel.Convert).       */
ORACLE_API void oracle_gray_from_rgb8(const uint8_t *pix, int w, int h, int stride,
                                      int channels, uint8_t *gray, int gstride)
{
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const uint8_t *p = pix + (long)y * stride + channels * x;
            uint32_t r = p[0] * 0x101u, g = p[1] * 0x101u, b = p[2] * 0x101u;
            gray[(long)y * gstride + x] =
                (uint8_t)((19595u * r + 38470u * g + 7471u * b + (1u << 15)) >> 24);
        }
}

/* ------------------------------------------------------------------------ */
/* hardware/sad.c:66-106 semantics (C mirror): interior pixels only, border  */
/* output 0 (calloc :45); terms with xr < 0 skipped, candidate kept (:83-85);*/
/* strict '<' (:97-100); raw best_disp.                                      */
/* ------------------------------------------------------------------------ */
ORACLE_API void oracle_mirror_sadc(const uint8_t *L, const uint8_t *R, int w, int h,
                                   int win, int maxd, uint8_t *out)
{
    int m = win / 2;                                        /* sad.c:66 */
    memset(out, 0, (size_t)w * h);
    for (int y = m; y < h - m; y++)                         /* sad.c:68 */
        for (int x = m; x < w - m; x++) {                   /* sad.c:69 */
            int min_sad = INT_MAX, best = 0;
            for (int d = 0; d <= maxd; d++) {               /* sad.c:73 */
                int sad = 0;
                for (int dy = -m; dy <= m; dy++)
                    for (int dx = -m; dx <= m; dx++) {
                        int xl = x + dx, yl = y + dy, xr = xl - d;
                        if (xr < 0) continue;               /* sad.c:83-85 */
                        int a = L[yl * w + xl], b = R[yl * w + xr];
                        sad += a >= b ? a - b : b - a;
                    }
                if (sad < min_sad) { min_sad = sad; best = d; } /* sad.c:97-100 */
            }
            out[y * w + x] = (uint8_t)best;                 /* sad.c:104 */
        }
}

/* ------------------------------------------------------------------------ */
/* hardware/test.py:262-294 compute_disparity_patch semantics: zero-pad both */
/* images by m = win//2 (:264-266), shift padded R right by d with zero fill */
/* (:277-281), box-sum |L-R| (:282-289), strict '<' (:290-292), raw d.        */
/* ------------------------------------------------------------------------ */
ORACLE_API void oracle_mirror_testpy(const uint8_t *L, const uint8_t *R, int w, int h,
                                     int win, int maxd, uint8_t *out)
{
    int m = win / 2;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int best = 0; long bestc = INT32_MAX;
            for (int d = 0; d <= maxd; d++) {
                long s = 0;
                for (int dy = -m; dy <= m; dy++)
                    for (int dx = -m; dx <= m; dx++) {
                        int yy = y + dy, xl = x + dx;
                        /* padded coordinates: column xl+m in Lp, Rsh[:,c] = Rp[:,c-d] for c>=d else 0 */
                        int lv = (yy >= 0 && yy < h && xl >= 0 && xl < w) ? L[yy * w + xl] : 0;
                        int cp = xl + m;                 /* padded column */
                        int rv = 0;
                        if (cp - d >= 0) {               /* Rsh[:, d:] = Rp[:, :-d] */
                            int xr = cp - d - m;         /* unpadded R column */
                            rv = (yy >= 0 && yy < h && xr >= 0 && xr < w) ? R[yy * w + xr] : 0;
                        }
                        s += lv >= rv ? lv - rv : rv - lv;
                    }
                if (s < bestc) { bestc = s; best = d; }
            }
            out[y * w + x] = (uint8_t)best;
        }
}

/* ------------------------------------------------------------------------ */
/* hardware/compute_SAD.v:62-265 compute_max_disp as exercised by ez_tb.v:    */
/* one WIN x (cols) strip pair, argmin_d sum |L[r][c] - R[r][c+d]| over the   */
/* first WIN columns of L, d in [0, ndisp), strict '<' (:201-204).            */
/* Returns best d; *min_sad_out gets the minimum.  (Verilog sign convention:  */
/* the right window slides to LARGER columns.)  This is the search the        */
/* testbench GOLDEN VALUES describe (mems/exp_disp_n.mem = 12 12 15 0F,       */
/* old_mems/exp_disp.mem = 23 0E 29 31), i.e. the intended contiguous sliding */
/* window -- NOT the RTL as written, which shifts in                          */
/* img_block_R[row][col_index + disp_idx + 1] (compute_SAD.v:213; a           */
/* contiguous window needs column col_index + WIN + disp_idx) and outputs     */
/* best_disp - (WIN-1) (:219).                                                */
/* ------------------------------------------------------------------------ */
ORACLE_API int oracle_mirror_verilog_strip(const uint8_t *L, const uint8_t *R,
                                           int win, int cols, int ndisp, int *min_sad_out)
{
    int best = 0, bests = INT_MAX;
    for (int d = 0; d < ndisp; d++) {
        if (d + win > cols) break;
        int s = 0;
        for (int r = 0; r < win; r++)
            for (int c = 0; c < win; c++) {
                int a = L[r * cols + c], b = R[r * cols + c + d];
                s += a >= b ? a - b : b - a;
            }
        if (s < bests) { bests = s; best = d; }
    }
    if (min_sad_out) *min_sad_out = bests;
    return best;
}

/* The same search for every interior pixel of an image (what SADHIP_FLAG_MIRROR_VERILOG computes):  */
/* pixel (x,y) = the strip of rows [y-m, y+m] with col_index = x-m; candidates d in [0, ndisp) whose */
/* right window [x-m+d, x+m+d] stays inside the image (compute_SAD.v:206-215); border pixels 0.      */
ORACLE_API void oracle_mirror_verilog(const uint8_t *L, const uint8_t *R, int w, int h,
                                      int win, int ndisp, uint8_t *out)
{
    int m = win / 2;
    memset(out, 0, (size_t)w * h);
    for (int y = m; y < h - m; y++)
        for (int x = m; x < w - m; x++) {
            int best = 0, bests = INT_MAX;
            for (int d = 0; d < ndisp && x + m + d < w; d++) {
                int s = 0;
                for (int dy = -m; dy <= m; dy++)
                    for (int dx = -m; dx <= m; dx++) {
                        int a = L[(y + dy) * w + x + dx], b = R[(y + dy) * w + x + dx + d];   /* :213 */
                        s += a >= b ? a - b : b - a;
                    }
                if (s < bests) { bests = s; best = d; }                                    /* :201-204 */
            }
            out[y * w + x] = (uint8_t)(best > 255 ? 255 : best);
        }
}

/* ------------------------------------------------------------------------ */
/* Synthetic stereo pair generator of SURVEY.md section 8(d) (splitmix64).    */
/* Test/bench input only; the product never generates frames.                 */
/* ------------------------------------------------------------------------ */
static inline uint64_t splitmix64_at(uint64_t seed, uint64_t n) /* n-th draw, n >= 1 */
{
    uint64_t z = seed + n * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

ORACLE_API void oracle_synth_pair(uint64_t seed, int w, int h, uint8_t *L, uint8_t *R)
{
    uint64_t n = 0;
    for (long i = 0; i < (long)w * h; i++) L[i] = (uint8_t)(splitmix64_at(seed, ++n) & 0xFF);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            uint64_t a = splitmix64_at(seed, ++n), b = splitmix64_at(seed, ++n);
            int s = 3 + (x * 7) / w + (y * 5) / h;
            uint8_t v = (x + s < w) ? L[(long)y * w + x + s] : (uint8_t)(a & 0xFF);
            if (((b >> 8) & 15) == 0) v ^= (uint8_t)(b & 3);
            R[(long)y * w + x] = v;
        }
}
