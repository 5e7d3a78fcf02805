// kernels_fast.hip -- instantiations, launch geometry and launch table of the column-wave kernel
// (device code: colwave_kernel.h; planner: colwave_plan.{h,hip}).  Entry points of this file (sadhip_kernels.h):
// fast_supported, describe_fast, launch_fast.
#include "colwave_kernel.h"
#include "colwave_plan.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace sadhip {

namespace {

#ifdef SADHIP_NO_TR3
constexpr bool kTripleColumns = false;
#else
constexpr bool kTripleColumns = true;
#endif

// workgroups of `kernel` a CU holds (registers, LDS, waves), asked from the runtime once per instantiation, shape and
// thread; *asked (optional) = the runtime answered (false: no usable device, one workgroup per CU assumed)
template <typename Kernel>
static int wgs_per_cu(Kernel kernel, int threads, size_t lds, bool* asked = nullptr)
{
  struct Entry { int dev, threads; size_t lds; int n; bool asked; };
  thread_local std::vector<Entry> cache;                        // per kernel instantiation (template) and thread
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); if (asked) *asked = false; return 1; }
  for (const Entry& e : cache)
    if (e.dev == dev && e.threads == threads && e.lds == lds) { if (asked) *asked = e.asked; return e.n; }
  int n = 0;
  bool ok = true;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, threads, lds) != hipSuccess || n < 1) { (void)hipGetLastError(); n = 1; ok = false; }
  cache.push_back(Entry{dev, threads, lds, n, ok});
  if (asked) *asked = ok;
  return n;
}

// dry: run every host-side check (instantiation exists, per-wave shares fit, LDS fits) and return
// without launching -- what fast_supported() asks, so that a job the column-wave kernel cannot run
// is routed to the generic kernel at PLANNING time and never by a silent fallback after a failure.
// part: 0 = every tile of the job; 1 = only the tiles before the first one that holds the image's last two
// columns (the triple-column form, TR); 2 = only the tiles from that one on (the plain form next to it).
// geom (optional): fill in the launch geometry and return without launching.
template <int P, int ND, int H, bool PK, int WB = 1024, bool W16 = false, bool TR = false, bool MIX = false>
hipError_t launch_t(const Job& j, const Plan& pl, const PassIo& io, hipStream_t s, bool dry, int part = 0, PassGeom* geom = nullptr)
{
  using K = Cw<P, ND, H, PK, W16, TR>;
  if ((pl.nwh != 0) != MIX || pl.nwh < 0 || pl.nwh >= pl.nw || (MIX && (pl.nwh % 2 != 0 || pl.nd != ND))) return hipErrorInvalidValue;
  CwArgs a{};
  a.j = j;
  a.nw = pl.nw;
  a.nwa = pl.nw - pl.nwh;
  a.slots = a.nwa * ND + pl.nwh * MixChunk<P, ND, PK>::NDB;
  a.xw = pl.extra ? pl.nw - 1 : -1;
  a.kshg = PK ? 16 : 32 - __builtin_clz((unsigned)j.maxdisp);     // travelling keys: 2^kshg > maxdisp
  a.pixshift = 0;                                               // (unpacked: set below, once the slot count is known)
  a.dbase = io.dbase; a.kin = io.kin; a.kout = io.kout; a.kw = io.kw;
  a.ksh = PK ? 16 : 32 - __builtin_clz((unsigned)a.slots);        // unpacked: the field holds d - dbase <= slots (the odd candidate)
  if (!PK) a.pixshift = a.ksh;
  if (!PK && (a.ksh > 8 || ((long long)(2 * H + 1) * (2 * H + 1) * 255 << a.ksh) >= (1LL << 27))) return hipErrorInvalidValue;   // keys below the 2^27 penalty, pixels within 16 bits
  a.dpad = a.slots + K::XOFF;                                 // keeps every lane's R operands 16-byte aligned (u16 ring, chunks of 4: 8-byte)
  const int RW = K::TW + a.dpad + 1;
  // Row streaming and output are shared by the waves.  The wave that carries the odd candidate does
  // ~12 % more arithmetic per row than the others (9 candidates against 8) and every wave waits for it at
  // the row barrier -- with one or two workgroups per CU (P = 8) nothing else fills that wait.  Where the
  // shares still fit the registers reserved for them, that wave (the last one) takes no share at all:
  // its streaming + output work, about as many instructions as its extra candidate, goes to the others.
  int sharers = pl.nw;
  if (pl.extra && pl.nw > 1 && (K::TW + pl.nw - 2) / (pl.nw - 1) <= 64 * K::MAXL && (RW + pl.nw - 2) / (pl.nw - 1) <= 64 * K::MAXR)
    sharers = pl.nw - 1;
  a.shl = (K::TW + sharers - 1) / sharers;
  a.shr = (RW + sharers - 1) / sharers;
  if (a.shl > 64 * K::MAXL || a.shr > 64 * K::MAXR) return hipErrorInvalidValue;   // cannot happen for nw >= 4
  const int r = j.w % P;
  a.tx0 = r == 0 ? 0 : r - P;
  // tiles whose output range meets [x0, x1)
  int k0 = 0, k1 = 0;
  while (a.tx0 + (k0 + 1) * K::STEP + K::OL <= j.x0) ++k0;   // tile k0 ends after x0
  k1 = k0;
  while (a.tx0 + (k1 + 1) * K::STEP + K::OL < j.x1) ++k1;
  if (part != 0) {
    int ke = 0;                                               // first tile whose lanes hold column w-2 or w-1
    while (a.tx0 + ke * K::STEP + K::TW + 2 <= j.w) ++ke;
    if (part == 1) k1 = std::min(k1, ke - 1); else k0 = std::max(k0, ke);
    if (k1 < k0) return hipSuccess;                           // no tile of this kind in the job
  }
  const int ntiles = k1 - k0 + 1;
  a.tile0 = k0;
  const int rows = j.y1 - j.y0;
  static_assert(K::NR == 2 * H + 3 && K::NB == 2 && K::RBW == 72 * P && K::XOFF == ((W16 || K::SWZ) ? 7 : 3), "plan_lds_bytes mirrors Cw");
  Plan plk = pl; plk.p = P; plk.nd = ND; plk.pk = PK; plk.w16 = W16;
  const size_t lds = plan_lds_bytes(plk, H, j.maxdisp) + (size_t)env_int("SADHIP_TUNE_LDSPAD", 0);   // tuning aid: cap workgroups per CU
  if (lds > 160 * 1024 || lds < (size_t)(K::NR + 1) * (K::TW + RW) * sizeof(typename K::Px)) return hipErrorInvalidValue;
  if (((size_t)(K::TW + RW) * sizeof(typename K::Px)) % 16 != 0) return hipErrorInvalidValue;   // ring rows are whole 16-byte blocks (u16 ring with ND 4: even wave counts)
  if (dry) return hipSuccess;
  // Strips per frame.  Each strip repeats 2h warm-up rows (+ set-up), so strips should be tall; but the
  // chip must be filled, a launch that is a little more than one round of workgroups waits for its tail,
  // and many short workgroups balance better than few long ones (leftmost tiles mask keys, CUs differ).
  // choose_strips() minimises a small cost model over the possible strip counts, with the number of
  // workgroups of THIS kernel a CU holds asked from the runtime (registers, LDS, waves).
  const long base = (long)ntiles * j.n;
  long nstrips;
  bool asked = false;
  const int per_cu = wgs_per_cu(k_colwave<P, ND, H, PK, WB, W16, TR, MIX>, 64 * pl.nw, lds, &asked);
  // shortest strip: 8 rows (single small frames: 640x480 28.8 -> 26.7 us, 1280x720 B9 47.1 -> 38.5; no change at 1080p) -- and 4
  // rows for launches so small that even 4-row strips do not fill the chip's workgroup slots (round 5: one 640x480 pair, the
  // WebUI's own frame size, at the defaults 16 / 64 is 120 workgroups on 120 of 256 CUs with 8-row strips, each doing 8 output
  // rows in 28 row steps: 44.2 -> 34.0 us with 4-row strips; 640x480 7 / 32 22.5 -> 19.0, 9 / 64 27.2 -> 24.3, 320x240 42 -> 32;
  // launches that 4-row strips would push past one round keep 8 -- 1280x720 16 / 64 measured 5 % worse with 4;
  // profiles/r05_ab_minrows.txt)
  int min_rows = env_int("SADHIP_TUNE_MINROWS", 0);
  if (min_rows <= 0) min_rows = base * ((rows + 3) / 4) <= (long)cu_count() * per_cu ? 4 : 8;
  if (env_int("SADHIP_TUNE_LEGACY_STRIPS", 0)) {                // round 2's constants, for comparison
    const long want = env_int("SADHIP_TUNE_WGS", P == 4 ? 512 : 256);
    const long rows_target = env_int("SADHIP_TUNE_ROWS", 48 * H), wg_max = env_int("SADHIP_TUNE_WGMAX", 4096);
    nstrips = (want + base - 1) / base;
    if (base >= want) nstrips = std::max<long>(1, std::min<long>(rows / rows_target, wg_max / base));
  } else if (env_int("SADHIP_TUNE_STRIPS", 0) > 0) {
    nstrips = env_int("SADHIP_TUNE_STRIPS", 0);
  } else {
    nstrips = choose_strips(base, rows, 2 * H, cu_count(), per_cu, min_rows, PK && H > 5);
  }
  const long max_strips = (rows + min_rows - 1) / min_rows;
  if (nstrips > max_strips) nstrips = max_strips;
  if (nstrips < 1) nstrips = 1;
  a.th = (int)((rows + nstrips - 1) / nstrips);
  nstrips = (rows + a.th - 1) / a.th;
#ifdef SADHIP_TUNING
  a.prio_off = env_int("SADHIP_TUNE_PRIO", 1) ? 0 : 1;          // issue priority by progress (colwave_kernel.h, the row loop): on unless SADHIP_TUNE_PRIO=0
#endif
  if (geom) { geom->tiles = ntiles; geom->strips = (int)nstrips; geom->rows = a.th; geom->per_cu = per_cu; geom->device = asked; return hipSuccess; }
  if (env_int("SADHIP_TUNE_VERBOSE", 0))
    fprintf(stderr, "sadhip: P%d ND%d H%d nw%d: %ld strips of %d rows x %ld workgroups per layer, %d per CU, lds %zu\n", P, ND, H, pl.nw, nstrips, a.th,
            base, per_cu, lds);
  for (int f0 = 0; f0 < j.n; f0 += 32768) {
    CwArgs aa = a;
    aa.j.n = (j.n - f0 < 32768) ? j.n - f0 : 32768;
    aa.j.left += (size_t)f0 * j.left_frame_stride;
    aa.j.right += (size_t)f0 * j.right_frame_stride;
    aa.j.out += (size_t)f0 * j.out_frame_stride;
    const size_t koff = (size_t)f0 * (size_t)rows * (size_t)a.kw;      // the kernel indexes keys with its local frame number
    if (aa.kin) aa.kin += koff;
    if (aa.kout) aa.kout += koff;
    aa.nstrips = (int)nstrips;
#ifdef SADHIP_STAMPS
    const size_t nwg = (size_t)nstrips * aa.j.n * ntiles;
    aa.stamps = nullptr;
    if (env_int("SADHIP_TUNE_STAMPS", 0)) {
      if (hipMalloc((void**)&aa.stamps, nwg * 16 * 8 * sizeof(unsigned long long)) != hipSuccess) aa.stamps = nullptr;
      else (void)hipMemsetAsync(aa.stamps, 0, nwg * 16 * 8 * sizeof(unsigned long long), s);
    }
    aa.wgtl = nullptr;
    const char* wgtl_file = getenv("SADHIP_TUNE_WGTL_FILE");        // workgroup timeline of every launch, appended as CSV
    if (wgtl_file) {
      if (hipMalloc((void**)&aa.wgtl, nwg * 4 * sizeof(unsigned long long)) != hipSuccess) aa.wgtl = nullptr;
      else (void)hipMemsetAsync(aa.wgtl, 0, nwg * 4 * sizeof(unsigned long long), s);
    }
#endif
    hipLaunchKernelGGL((k_colwave<P, ND, H, PK, WB, W16, TR, MIX>), dim3((unsigned)(nstrips * aa.j.n), ntiles, 1), dim3(64 * pl.nw), lds, s, aa);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
#ifdef SADHIP_STAMPS
    if (aa.wgtl) {
      (void)hipStreamSynchronize(s);
      std::vector<unsigned long long> h(nwg * 4);
      (void)hipMemcpy(h.data(), aa.wgtl, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
      (void)hipFree(aa.wgtl);
      static int launch_no = 0;
      if (FILE* f = fopen(wgtl_file, "a")) {
        if (launch_no == 0) fprintf(f, "launch,p,nd,h,nw,strips,rows,frames,tiles,x,y,t0,t1,hw_id,xcc_id\n");
        const unsigned gx = (unsigned)(nstrips * aa.j.n);
        for (size_t g = 0; g < nwg; ++g)
          fprintf(f, "%d,%d,%d,%d,%d,%ld,%d,%d,%d,%u,%u,%llu,%llu,%llu,%llu\n", launch_no, P, ND, H, pl.nw, nstrips, a.th, aa.j.n, ntiles, (unsigned)(g % gx), (unsigned)(g / gx),
                  h[4 * g], h[4 * g + 1], h[4 * g + 2], h[4 * g + 3]);
        fclose(f);
      }
      ++launch_no;
    }
    if (aa.stamps) {                                      // average ticks per row step and phase, by wave class
      (void)hipStreamSynchronize(s);
      std::vector<unsigned long long> h(nwg * 16 * 8);
      (void)hipMemcpy(h.data(), aa.stamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
      (void)hipFree(aa.stamps);
      static const char* const nm[5] = {"update", "reduce", "land(+vmcnt wait)", "output", "barrier wait"};
      double sum[3][5] = {}, steps[3] = {};               // class 0: ordinary wave, 1: extra-candidate wave, 2: waves of tile 0
      for (size_t g = 0; g < nwg; ++g)
        for (int w = 0; w < pl.nw; ++w) {
          const unsigned long long* o = &h[(g * 16 + w) * 8];
          if (!o[5]) continue;
          const int c = o[7] == (unsigned long long)a.tile0 ? 2 : (o[6] ? 1 : 0);
          for (int k = 0; k < 5; ++k) sum[c][k] += (double)o[k];
          steps[c] += (double)o[5];
        }
      for (int c = 0; c < 3; ++c) {
        if (steps[c] == 0) continue;
        double tot = 0; for (int k = 0; k < 5; ++k) tot += sum[c][k];
        fprintf(stderr, "sadhip stamps [%s] P%d ND%d H%d nw%d: %.0f ticks/step:", c == 0 ? "wave" : c == 1 ? "extra-candidate wave" : "leftmost-tile waves",
                P, ND, H, pl.nw, tot / steps[c]);
        for (int k = 0; k < 5; ++k) fprintf(stderr, "  %s %.0f (%.0f%%)", nm[k], sum[c][k] / steps[c], 100.0 * sum[c][k] / tot);
        fprintf(stderr, "\n");
      }
    }
#endif
  }
  return hipSuccess;
}

}  // namespace

// plan exists AND every pass of it is a kernel this build instantiates and can launch
static bool passes_launchable(const Job& j, const std::vector<Pass>& passes);

// The job's passes when the column-wave kernel runs it, nullptr when the generic kernel does.  ONE planning per
// call: plan_job's memo (three compares for a caller that streams frames with the same parameters) and a dry run
// of every pass (a few integer checks, no HIP call).
static const std::vector<Pass>* plan_for(const Job& j)
{
  if (j.w < 1 || j.h < 1) return nullptr;
  if ((j.rw > 0 && j.rw < j.w) || (j.rh > 0 && j.rh < j.h)) return nullptr;   // smaller right image: dropped terms, generic kernel only
  const std::vector<Pass>& passes = plan_job(j.block / 2, j.maxdisp, j.w <= j.maxdisp + 2 * (j.block / 2) + 2);   // BlockSize 2..31
  return (!passes.empty() && passes_launchable(j, passes)) ? &passes : nullptr;
}

bool fast_supported(const Job& j) { return plan_for(j) != nullptr; }

static hipError_t launch_pass(const Job& j, const Plan& pl, const PassIo& io, hipStream_t s, bool dry = false, PassGeom* geom = nullptr);

int describe_fast(const Job& j, char* buf, int cap, bool geometry)
{
  if (buf && cap > 0) buf[0] = 0;
  const std::vector<Pass>* planned = plan_for(j);
  if (!planned) return 0;
  const std::vector<Pass>& passes = *planned;
  const int h = j.block / 2;
  int at = 0;
  for (size_t k = 0; k < passes.size(); ++k) {
    const Plan& pl = passes[k].pl;
    const int next = k + 1 < passes.size() ? passes[k + 1].dbase : j.maxdisp + 1;
    const size_t lds = plan_lds_bytes(pl, h, j.maxdisp);
    if (buf && at < cap)
      at += snprintf(buf + at, (size_t)(cap - at), "dbase=%d count=%d p=%d nd=%d nw=%d packed=%d extra=%d w16=%d nwh=%d lds=%zu",
                     passes[k].dbase, next - passes[k].dbase, pl.p, pl.nd, pl.nw, (int)pl.pk, (int)pl.extra, (int)pl.w16, pl.nwh, lds);
    if (geometry) {
      PassGeom g;
      if (launch_pass(j, pl, PassIo{passes[k].dbase, nullptr, nullptr, j.x1 - j.x0}, nullptr, false, &g) != hipSuccess) return -1;
      if (buf && at < cap)
        at += snprintf(buf + at, (size_t)(cap - at), " tiles=%d strips=%d rows=%d wg_per_cu=%d device=%d", g.tiles, g.strips, g.rows, g.per_cu, (int)g.device);
    }
    if (buf && at < cap) at += snprintf(buf + at, (size_t)(cap - at), "\n");
  }
  return (int)passes.size();
}

static hipError_t launch_pass(const Job& j, const Plan& pl, const PassIo& io, hipStream_t s, bool dry, PassGeom* geom)
{
  const int h = j.block / 2;
  if ((pl.w16 && pl.p != 8) || (pl.nwh && !(pl.w16 || (pl.pk && pl.p == 4)))) return hipErrorInvalidValue;
  if (pl.w16 && pl.pk && pl.nd == 8 && h == 7) return launch_t<8, 8, 7, true, 1024, true>(j, pl, io, s, dry, 0, geom);
  // unpacked on the u16 ring, 128-VGPR builds (up to 16 waves per workgroup, two or three workgroups per CU): chunks of 8 and of 4
#define SADHIP_W16U(HH) \
  if (pl.w16 && !pl.pk && pl.nd == 8 && pl.nwh && h == HH) return launch_t<8, 8, HH, false, 1024, true, false, true>(j, pl, io, s, dry, 0, geom); \
  if (pl.w16 && !pl.pk && pl.nd == 8 && h == HH) return launch_t<8, 8, HH, false, 1024, true>(j, pl, io, s, dry, 0, geom); \
  if (pl.w16 && !pl.pk && pl.nd == 4 && h == HH) return launch_t<8, 4, HH, false, 1024, true>(j, pl, io, s, dry, 0, geom)
  SADHIP_W16U(8); SADHIP_W16U(9); SADHIP_W16U(10); SADHIP_W16U(11); SADHIP_W16U(12); SADHIP_W16U(13); SADHIP_W16U(14); SADHIP_W16U(15);
  if (pl.w16) return hipErrorInvalidValue;
  // packed P = 4 with mixed chunks (4 waves of ND + 4 of ND - 4): BlockSize 9 / 15 in their two-form kernels
  if (pl.nwh && pl.pk && pl.p == 4) {
#define SADHIP_PMIX(NDD, HH, TRR) if (pl.nd == NDD && h == HH) return launch_t<4, NDD, HH, true, 1024, false, TRR, true>(j, pl, io, s, dry, 0, geom)
#define SADHIP_PMIX_LOW(NDD) SADHIP_PMIX(NDD, 1, false); SADHIP_PMIX(NDD, 2, false); SADHIP_PMIX(NDD, 3, false); SADHIP_PMIX(NDD, 4, kTripleColumns)
    SADHIP_PMIX_LOW(12); SADHIP_PMIX_LOW(16); SADHIP_PMIX_LOW(20); SADHIP_PMIX_LOW(24);
    SADHIP_PMIX(12, 5, false); SADHIP_PMIX(16, 5, false); SADHIP_PMIX(12, 6, false); SADHIP_PMIX(16, 6, false);
    SADHIP_PMIX(12, 7, kTripleColumns); SADHIP_PMIX(16, 7, kTripleColumns);
#undef SADHIP_PMIX_LOW
#undef SADHIP_PMIX
    return hipErrorInvalidValue;
  }
#define SADHIP_CASE(PP, NDD, HH, PKK) \
  if (pl.p == PP && pl.nd == NDD && h == HH && pl.pk == PKK) return launch_t<PP, NDD, HH, PKK>(j, pl, io, s, dry, 0, geom)
  // BlockSize 9: triple-column form for the tiles left of the image's last two columns, plain form for the rest
#define SADHIP_B9(NDD) \
  if (pl.p == 4 && pl.nd == NDD && h == 4 && pl.pk) { \
    const int trmode = kTripleColumns ? env_int("SADHIP_TUNE_TRMODE", 2) : 0;   /* 0 plain, 1 two launches split by tile, 2 one kernel that picks per workgroup */ \
    if (trmode == 0 || (trmode == 1 && !dry && j.n < env_int("SADHIP_TUNE_TRMIN", 48))) return launch_t<4, NDD, 4, true>(j, pl, io, s, dry, 0, geom); \
    if (trmode == 2) return launch_t<4, NDD, 4, true, 1024, false, true>(j, pl, io, s, dry, 0, geom); \
    const hipError_t e1 = launch_t<4, NDD, 4, true, 1024, false, true>(j, pl, io, s, dry, 1, geom); \
    return e1 != hipSuccess ? e1 : launch_t<4, NDD, 4, true>(j, pl, io, s, dry, 2, geom); \
  }
#define SADHIP_P4(NDD) SADHIP_CASE(4, NDD, 1, true); SADHIP_CASE(4, NDD, 2, true); SADHIP_CASE(4, NDD, 3, true); SADHIP_B9(NDD)
#define SADHIP_P8(NDD) SADHIP_CASE(8, NDD, 5, true); SADHIP_CASE(8, NDD, 6, true); SADHIP_CASE(8, NDD, 7, true)
  SADHIP_P4(4); SADHIP_P4(8); SADHIP_P4(12); SADHIP_P4(16); SADHIP_P4(20); SADHIP_P4(24);
  SADHIP_P8(4); SADHIP_P8(8); SADHIP_CASE(8, 12, 5, true);
  // packed P = 4 with two-lane halos (h = 5..7): wide disparity ranges as 8-wave workgroups, two or three per CU
  // BlockSize 15 (h = 7): its window is five pixel triples -- the same two-form kernel as BlockSize 9 (SADHIP_TUNE_TRMODE=0: plain form)
#define SADHIP_B15(NDD) \
  if (pl.p == 4 && pl.nd == NDD && h == 7 && pl.pk) { \
    if (!kTripleColumns || env_int("SADHIP_TUNE_TRMODE", 2) == 0) return launch_t<4, NDD, 7, true>(j, pl, io, s, dry, 0, geom); \
    return launch_t<4, NDD, 7, true, 1024, false, true>(j, pl, io, s, dry, 0, geom); \
  }
#define SADHIP_P4H(NDD) SADHIP_CASE(4, NDD, 5, true); SADHIP_CASE(4, NDD, 6, true); SADHIP_B15(NDD)
  SADHIP_P4H(4); SADHIP_P4H(8); SADHIP_P4H(12); SADHIP_P4H(16);
#ifdef SADHIP_TUNING
  SADHIP_CASE(8, 8, 4, true); SADHIP_CASE(8, 12, 4, true);     // forced for comparison (SADHIP_TUNE_P=8)
#endif
  // unpacked (window SAD does not fit 16 bits, or forced for comparison)
#define SADHIP_U(HH, NDD) \
  if (pl.p == 8 && pl.nd == NDD && h == HH && !pl.pk) \
    return (64 * pl.nw <= 512 && env_int("SADHIP_TUNE_WB", 512) == 512) ? launch_t<8, NDD, HH, false, 512>(j, pl, io, s, dry, 0, geom) : launch_t<8, NDD, HH, false, 1024>(j, pl, io, s, dry, 0, geom)
  SADHIP_U(8, 4); SADHIP_U(8, 8);
  SADHIP_U(7, 4); SADHIP_U(7, 8); SADHIP_U(6, 4); SADHIP_U(6, 8);   // D without an exact packed plan
  SADHIP_U(5, 4); SADHIP_U(5, 8); SADHIP_U(4, 4); SADHIP_U(4, 8); SADHIP_U(3, 4); SADHIP_U(3, 8);   // narrow frames
#define SADHIP_UH(HH) SADHIP_U(HH, 4); if (pl.p == 8 && pl.nd == 8 && h == HH && !pl.pk) return launch_t<8, 8, HH, false, 512>(j, pl, io, s, dry, 0, geom)
  SADHIP_UH(9); SADHIP_UH(10); SADHIP_UH(11); SADHIP_UH(12); SADHIP_UH(13); SADHIP_UH(14); SADHIP_UH(15);   // BlockSize 18..31: two-lane halos
#undef SADHIP_CASE
  return hipErrorInvalidValue;
}

static bool passes_launchable(const Job& j, const std::vector<Pass>& passes)
{
  Job jj = j;
  if (jj.n < 1) jj.n = 1;
  if (jj.x1 <= jj.x0) { jj.x0 = 0; jj.x1 = jj.w; }
  if (jj.y1 <= jj.y0) { jj.y0 = 0; jj.y1 = jj.h; }
  for (const Pass& p : passes)
    if (launch_pass(jj, p.pl, PassIo{p.dbase, nullptr, nullptr, jj.x1 - jj.x0}, nullptr, true) != hipSuccess) return false;
  return true;
}

// No fallback in here: the generic kernel runs only when fast_supported() says so up front; any
// launch error of a pass is returned to the caller (single-pass and multi-pass alike).
hipError_t launch_fast(const Job& j, hipStream_t s)
{
  if (j.n <= 0 || j.x1 <= j.x0 || j.y1 <= j.y0) return hipSuccess;
  const std::vector<Pass>* planned = plan_for(j);          // planned ONCE per call (round 2: fast_supported + here, each with a dry run)
  if (!planned) return launch_generic(j, s);                // routed at planning time; a launch error below is returned, never retried
  const std::vector<Pass>& passes = *planned;
  if (passes.size() == 1) {
    const hipError_t e = launch_pass(j, passes[0].pl, PassIo{0, nullptr, nullptr, 0}, s);
    if (e == hipSuccess) note_launch(KIND_COLWAVE);
    return e;
  }
  // several passes: a u32 key per output pixel travels from pass to pass in a stream-ordered
  // scratch buffer (<= 256 MB: frames are processed in chunks)
  const int kw = j.x1 - j.x0, rows = j.y1 - j.y0;
  const size_t per_frame = (size_t)kw * rows * sizeof(unsigned);
  const int chunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)j.n, ((size_t)256 << 20) / per_frame));
  unsigned* keys = nullptr;
  hipError_t e = scratch_alloc((void**)&keys, per_frame * chunk, s);
  if (e != hipSuccess) return e;
  for (int f0 = 0; f0 < j.n && e == hipSuccess; f0 += chunk) {
    Job jj = j;
    jj.n = std::min(chunk, j.n - f0);
    jj.left += (size_t)f0 * j.left_frame_stride;
    jj.right += (size_t)f0 * j.right_frame_stride;
    jj.out += (size_t)f0 * j.out_frame_stride;
    for (size_t k = 0; k < passes.size() && e == hipSuccess; ++k) {
      e = launch_pass(jj, passes[k].pl, PassIo{passes[k].dbase, k ? keys : nullptr, k + 1 < passes.size() ? keys : nullptr, kw}, s);
      if (env_int("SADHIP_TUNE_VERBOSE", 0))
        fprintf(stderr, "sadhip: pass %zu/%zu dbase %d p %d nd %d nw %d pk %d extra %d -> %s\n", k, passes.size(), passes[k].dbase,
                passes[k].pl.p, passes[k].pl.nd, passes[k].pl.nw, (int)passes[k].pl.pk, (int)passes[k].pl.extra, hipGetErrorString(e));
    }
  }
  const hipError_t ef = scratch_free(keys, s);
  if (e == hipSuccess && ef == hipSuccess) note_launch(KIND_COLWAVE);
  return e != hipSuccess ? e : ef;
}

}  // namespace sadhip