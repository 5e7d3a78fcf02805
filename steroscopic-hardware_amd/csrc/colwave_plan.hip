// colwave_plan.hip -- planner of the column-wave kernel: see colwave_plan.h.  Host code only.
#include "colwave_plan.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace sadhip {

// LDS bytes of one workgroup of a plan (the layout at the top of Cw::run): 2h+3 ring rows and the
// row of zeros, two row-best buffers, the output LUT.
size_t plan_lds_bytes(const Plan& pl, int h, int maxdisp)
{
  const bool swz = pl.p == 8 && !pl.w16 && pl.nd == 8;
  const size_t tw = 64 * (size_t)pl.p, rw = tw + (size_t)pl.slots() + ((pl.w16 || swz) ? 7 : 3) + 1, roww = tw + rw;
  return (size_t)(2 * h + 4) * roww * (pl.w16 ? 2 : 4) + (size_t)2 * 72 * pl.p * 4 + (size_t)((maxdisp + 1 + 15) & ~15);
}

#ifdef SADHIP_TUNING
int env_int(const char* name, int dflt)
{
  struct Entry { const char* name; int value; bool set; };
  static const auto table = [] {
    static const char* const names[] = {"SADHIP_TUNE_ND", "SADHIP_TUNE_NW", "SADHIP_TUNE_P", "SADHIP_TUNE_W16", "SADHIP_TUNE_PK",
                                        "SADHIP_TUNE_CHUNK", "SADHIP_TUNE_WGS", "SADHIP_TUNE_ROWS",
                                        "SADHIP_TUNE_WGMAX", "SADHIP_TUNE_LDSPAD", "SADHIP_TUNE_WB", "SADHIP_TUNE_VERBOSE",
                                        "SADHIP_TUNE_STAMPS", "SADHIP_TUNE_LEGACY_STRIPS", "SADHIP_TUNE_STRIPS", "SADHIP_TUNE_MINROWS", "SADHIP_TUNE_TRMIN", "SADHIP_TUNE_TRMODE",
                                        "SADHIP_TUNE_W16NW", "SADHIP_TUNE_W16ND4", "SADHIP_TUNE_MIX", "SADHIP_TUNE_TAILPCT", "SADHIP_TUNE_WARMPCT", "SADHIP_TUNE_TAIL2", "SADHIP_TUNE_PRIO"};
    std::vector<Entry> t;
    for (const char* n : names) {
      const char* e = getenv(n);
      t.push_back(Entry{n, e ? atoi(e) : 0, e != nullptr});
    }
    return t;
  }();
  for (const Entry& e : table)
    if (e.name == name || strcmp(e.name, name) == 0) return e.set ? e.value : dflt;
  const char* e = getenv(name);                                 // not in the table: read it live
  return e ? atoi(e) : dflt;
}
#endif

// Kernel variant for `count` consecutive candidates: columns per lane P, packed or not, chunk size
// ND, wave count NW and whether the last wave carries one extra candidate (NW*ND + 1 == count, the
// usual case: the UI's D is a multiple of 16).  Cost model: a workgroup advances at the pace of
// its most loaded SIMD, i.e. ceil(NW/4) waves of (per-candidate work * ND + per-wave overhead) per
// row.  exact: no chunk may run past the last candidate (packed h = 6, 7 masks keys with lane masks
// that cannot express it).
//   packed   (h <= 7): P = 4, ND in {4, 8, 12, 16} (h <= 4 also 20, 24; h = 5..7 with two-lane halos);
//                      P = 8 (h = 5..7, when P = 4 has no plan: exact plans of the lane-mask variants), ND in {4, 8, 12}
//   unpacked (h = 8; h = 6, 7 without an exact packed plan; SADHIP_TUNE_PK=0): P = 8, ND in {4, 8, 12}
// Measured on MI355X: for h = 5..7 the P = 4 kernels -- 4..8 waves per workgroup, two to four
// independent workgroups per CU -- beat the P = 8 ones (8..16 waves, one or two per CU) by 2-15 % although
// their windows span five lanes instead of three (7.0 against 6.6 instructions per 64 candidates).
static Plan plan_count_p(int h, bool pk, int count, bool exact, int maxdisp, int p);

Plan plan_count(int h, bool pk, int count, bool exact, int maxdisp)
{
  const int force_p = env_int("SADHIP_TUNE_P", 0);              // tuning aid: force the columns per lane
  if (pk && h <= 7 && force_p != 8) {
    const Plan pl = plan_count_p(h, pk, count, exact, maxdisp, 4);
    if (pl.nw || force_p == 4 || h <= 4) return pl;
  }
  Plan pl = plan_count_p(h, pk, count, exact, maxdisp, 8);
  // packed h = 5 beyond what P = 4 covers (more than 129 candidates): a P = 8 plan only where it fits exactly on whole SIMD
  // rounds (145 = 12 x 12 + 1); otherwise no plan here, and plan_job cuts the range into a full P = 4 pass + the rest
  // (round 4: 161 / 177 candidates ran as 14 / 15 waves of 12 with idle slots)
  if (pk && h == 5 && pl.nw && force_p != 8 && (pl.nw % 4 != 0 || pl.slots() + (pl.extra ? 1 : 0) != count)) pl.nw = 0;
  return pl;
}

static Plan plan_count_p(int h, bool pk, int count, bool exact, int maxdisp, int p)
{
  Plan best{0, 0, 0, false, false, false};
  const int force_nd = env_int("SADHIP_TUNE_ND", 0);            // tuning aids: force a chunk size / wave count
  const int force_nw = env_int("SADHIP_TUNE_NW", 0);
  long best_cost = 1L << 60;
  // chunk sizes: P = 4 up to 24; P = 8 up to 12 for the penalty variants (packed h = 5), 8 otherwise --
  // measured: the lane-mask variants spill at ND 12 (B15 D48: 47 vs 36 us) and the unpacked ones
  // lose their u16 ring (B16 D48: 41 vs 37 us)
  const int nd_max = p == 4 ? 24 : ((pk && h <= 5) ? 12 : 8);
  // nwh: the last nwh waves carry chunks of nd/2 (mixed-chunk plans: unpacked, u16 ring, nd 8, both groups multiples of 4
  // so that every SIMD gets the same mix -- 49 = 4*8 + 4*4 + 1, 81 = 8*8 + 4*4 + 1, 113 = 12*8 + 4*4 + 1)
  auto consider = [&](int nd, int nw, int nwh) {
      if (force_nw && force_nw != nw) return;
      if (p == 4 && nw > 8) return;                             // measured: 11-16 waves (one workgroup per CU) lose ~10 % to passes of 8 waves
      if (p == 4 && h > 4 && nd > 16) return;                   // two-lane halo kernels exist for ND 4, 8, 12, 16
      // u16 ring pixels where the u32 ring leaves room for one workgroup per CU only (instantiated for packed h = 7,
      // chunk size 8, and for unpacked h >= 8, chunk sizes 8 and 4 -- the latter with an even wave count: its ring
      // rows are whole 16-byte blocks only then; unpacked pixels are shifted by bitlen(slots) <= 8 bits: they always fit).
      // Chunks of 8 on the u16 ring: up to 8 waves for h = 8 (beyond, the u32 ring's 16-wave workgroups measured 1-3 %
      // faster: B16 D96 54.7 vs 56.6 us), up to 16 for the two-lane halo kernels, h > 8 (one pass of 11-16 waves instead of
      // two passes of 8: B21 D96 65.5 -> 59.5 us, D128 76.6 -> 72.8)
      const int w16_nw8 = env_int("SADHIP_TUNE_W16NW", h > 8 ? 16 : 8);
      const bool w16 = env_int("SADHIP_TUNE_W16", 1) != 0 && p == 8 &&
                       ((nd == 8 && (nw <= w16_nw8 || nwh) && ((pk && h == 7 && !nwh) || (!pk && h >= 8))) ||
                        (nd == 4 && nw % 2 == 0 && env_int("SADHIP_TUNE_W16ND4", 1) != 0 && !pk && h >= 8));
      // mixed chunks: unpacked 8 + 4 on the u16 ring; packed P = 4 (nd, nd - 4), 4 + 4 waves
      if (nwh && !((w16 && !pk && nd == 8) || (pk && p == 4 && nd >= 12 && nw == 8 && nwh == 4))) return;
      if (h > 8 && nd == 8 && nw > 8 && !w16) return;           // ... and so does ND 8 with two-lane halos on the u32 ring (256-VGPR builds: <= 8 waves)
      // LDS: (2h+4) ring rows of L[TW] + R[TW + slots + 4 (8)] pixels, two row buffers, the output LUT
      Plan cand{p, nd, nw, pk, false, w16};
      cand.nwh = nwh;
      const size_t lds_plan = plan_lds_bytes(cand, h, maxdisp);
      if (lds_plan > (size_t)160 * 1024) return;
      const int slots = cand.slots();
      if (slots + 1 < count) return;
      const bool extra = slots + 1 == count;
      if (exact && !extra && slots != count) return;
      if (nwh && !extra && slots != count) return;              // mixed chunks only where they fit exactly
      if (!extra && slots - nd >= count && nw > 4) return;      // a whole wave would be idle
      auto per_wave = [&](int ndw, bool x) { return (long)(pk ? 22 : 36) * ndw * (p / 4) + 75 + (x ? 40 * (p / 4) : 0); };
      // the busiest SIMD: waves go to the SIMDs round-robin (wave i -> SIMD i mod 4, tools/ubench/simd_map.hip); the last
      // wave carries the odd candidate
      long simd[4] = {0, 0, 0, 0};
      for (int i = 0; i < nw; ++i) simd[i % 4] += per_wave(i >= nw - nwh ? cand.ndh() : nd, extra && i == nw - 1);
      const long busiest = std::max(std::max(simd[0], simd[1]), std::max(simd[2], simd[3]));
      // waves a CU can hold of this plan (LDS; 128 VGPRs = 16 waves for the unpacked kernels): with one
      // 4-wave workgroup per CU nothing hides its latencies (measured: BlockSize 16, D 16 as 4 waves x 4
      // candidates on a u32 ring, one workgroup per CU, 26.7 us; as 4 x 8 on the u16 ring, three per CU, 20.5;
      // as 4 x 4 on the u16 ring, round 4: 13.8)
      const long resident = std::min<long>((long)((size_t)160 * 1024 / lds_plan) * nw, pk ? 20 : 16);
      const long pct = resident <= 4 ? 200 : resident <= 8 ? 118 : 100;
      // workgroups of more than 8 waves wait more at the row barrier (measured, unpacked: 12 x 4 candidates 36.9 us against
      // 7 x 8 with a quarter of the SIMD slots idle 33.4)
      const long big = (!pk && nw > 8) ? 112 : 100;
      const long cost = busiest * 64 * pct / 100 * big / 100 + nw;    // ties: fewer waves
      if (cost < best_cost) { best_cost = cost; best = cand; best.extra = extra; }
  };
  for (int nd = 4; nd <= nd_max; nd += 4) {
    if (force_nd && force_nd != nd) continue;
    for (int nw = 4; nw <= 16; ++nw) consider(nd, nw, 0);       // (two-wave workgroups for D <= 48 measured 25-35 % slower)
    if (nd == 8 && p == 8 && !pk && env_int("SADHIP_TUNE_MIX", 1) != 0)
      for (int nwa = 4; nwa <= 12; nwa += 4) consider(8, nwa + 4, 4);
    if (p == 4 && pk && nd >= 12 && env_int("SADHIP_TUNE_MIX", 1) != 0) consider(nd, 8, 4);   // 4 waves of nd + 4 of nd - 4
  }
  return best;
}

// The whole disparity range [0, D] as one pass, or -- when D+1 candidates do not fit the waves a
// workgroup may have -- as several full passes of `chunk` candidates plus a last pass for the rest
// (see CwArgs).
// Packed first; unpacked when the window SAD does not fit 16 bits (h = 8), when forced, or when a
// packed h = 6, 7 kernel has no exact plan.  Empty result: the generic kernel runs the job.
// narrow: the frame is at most D + 2h + 2 wide -- a right-clipped window of a packed variant could
// meet a missing candidate without its penalty column, so only the unpacked variants (every column
// penalised) may run.
static std::vector<Pass> plan_job_uncached(int h, int maxdisp, bool narrow);

const std::vector<Pass>& plan_job(int h, int maxdisp, bool narrow)
{
  thread_local int memo_h = -1, memo_d = -1, memo_n = -1;
  thread_local std::vector<Pass> memo;
  if (h != memo_h || maxdisp != memo_d || (int)narrow != memo_n) {
    memo = plan_job_uncached(h, maxdisp, narrow); memo_h = h; memo_d = maxdisp; memo_n = narrow;
  }
  return memo;
}

static std::vector<Pass> plan_job_uncached(int h, int maxdisp, bool narrow)
{
  std::vector<Pass> passes;
  if (h < 1 || h > 15 || maxdisp < 1) return passes;
  for (int variant = (h <= 7 && !narrow && env_int("SADHIP_TUNE_PK", 1) != 0) ? 0 : 1; variant < 2; ++variant) {
    const bool pk = variant == 0;
    if (!pk && h < 3) break;                                    // unpacked variants exist for h >= 3 only
    // (unpacked keys inside a pass are (SAD << bitlen(slots)) | (d - dbase) with at most 129 candidates per pass:
    //  31^2 * 255 << 8 stays below the 2^27 penalty for every block size, whatever MaxDisparity is -- round 4; before,
    //  the field held d itself and BlockSize 31 with D >= 512 went to the generic kernel)
    const bool exact = pk && h > 5;
    // full passes: 8 waves of ND 16 (P 4) / 16 waves of ND 8 (P 8); two-lane halos: 8 waves of ND 8
    // (round 4: the two-lane halo kernels too -- 16 waves of 8 on the u16 ring where D <= 255, else 8 waves of 8)
    const int chunk = env_int("SADHIP_TUNE_CHUNK", (h <= 8 || !pk) ? 128 : 64);
    passes.clear();
    int base = 0, rem = maxdisp + 1;
    while (rem > 0) {
      Plan pl = plan_count(h, pk, rem, exact, maxdisp);
      if (pl.nw) { passes.push_back(Pass{pl, base}); rem = 0; break; }
      if (rem <= chunk) break;                                  // no plan for the rest
      pl = plan_count(h, pk, chunk, true, maxdisp);
      if (!pl.nw) break;
      passes.push_back(Pass{pl, base});
      base += chunk; rem -= chunk;
    }
    if (rem == 0) return passes;
  }
  passes.clear();
  return passes;
}

// ---- strip count of a launch ---------------------------------------------------------------------
// base = workgroups per strip layer (tiles x frames), rows = output rows, h2 = 2h warm-up rows per strip,
// cus = compute units, per_cu = workgroups of this kernel a CU holds.  Cost of ns strips of th rows:
//   work per workgroup  = th + h2 + 4 rows (4: set-up, ring fill, first barrier)
//   k = ceil(base*ns / cus) workgroups on the busiest CU (round-robin dispatch)
//   k <= per_cu: one round; a workgroup with fewer neighbours runs faster -- measured single 1080p pairs:
//                1 / 2 / 5 resident workgroups take 0.42 / 0.55 / 1.0 of the full-occupancy time per row
//                -> duration factor 0.27 + 0.73 * k / per_cu
//   k >  per_cu: between base*ns / (cus*per_cu) rounds (perfect balance) and ceil(k / per_cu) (the busiest CU): their mean
//   + 0.15 of a workgroup's duration for the tail in both cases: workgroups are not equally long (the
//   leftmost tiles mask keys, CUs differ), and the longer they are the longer the chip waits for the last
//   one (measured: 128 1080p pairs as 1 / 4 strips 27.7 / 24.2 us per pair; config 5 with 16 pairs as
//   1 / 2 / 4 / 6 rounds 228 / 218 / 213 / 210)
long choose_strips(long base, long rows, int h2, int cus, int per_cu, int min_rows, bool uneven)
{
  const long max_strips = std::max<long>(1, (rows + min_rows - 1) / min_rows);
  const double slots = (double)cus * per_cu;
  double best = 1e300;
  long best_ns = 1;
  const int warm_pct = env_int("SADHIP_TUNE_WARMPCT", 100);
  for (long ns = 1; ns <= max_strips; ++ns) {
    const long th = (rows + ns - 1) / ns;
    if ((rows + th - 1) / th != ns) continue;                   // this strip height gives fewer strips: counted there
    const long w = base * ns, k = (w + cus - 1) / cus;
    const double whole = (double)((k + per_cu - 1) / per_cu);   // rounds on the busiest CU
    double t = (k <= per_cu ? 0.27 + 0.73 * (double)k / per_cu : 0.5 * ((double)w / slots + whole)) + 0.15;
    // tuning aid (round 4 A/B, tools/ab_strips4.sh): several rounds as ideal rounds + a tail of `SADHIP_TUNE_TAILPCT` % of one
    // workgroup's duration -- workgroups are handed out as slots free up, the busiest-CU round count overstates the wait
    const int tail_pct = env_int("SADHIP_TUNE_TAILPCT", 0);
    if (tail_pct > 0 && k > per_cu) t = (double)w / slots + tail_pct / 100.0;
    // a warm-up row costs less than an output row: no window sums, no argmin and (round 4) no leaving-row half of the
    // vertical update -- `warm_pct` % of a row (tuning aid SADHIP_TUNE_WARMPCT)
    const double cost = t * ((double)th + (double)h2 * warm_pct / 100.0 + 4.0);
    if (cost < best * 0.999) { best = cost; best_ns = ns; }     // ties: fewer strips
  }
  // Kernels that fit TWO workgroups per CU and whose workgroups differ in length by tile (`uneven`: the packed h = 6, 7 kernels -- the
  // leftmost tiles mask keys and, for h = 7, run the plain form of the row loop while the others run the triple-column one): twice
  // the model's strips on launches of more than one round.  Measured, not modelled (round 5).  First for every two-per-CU kernel
  // (profiles/r05_sweep_strips_bigwg.txt, r05_ab_strips_x2.txt: config 5 with 4 / 8 / 16 / 32 pairs -1.7 / -3.3 / -2.2 / -2.6 %,
  // B 13 D 128 -2.6 %, the defaults 16 / 64 -1 ... -2 %); then again with the issue priority by progress in the kernel (colwave_kernel.h),
  // which took the spread out of equal workgroups' durations (profiles/r05_ab_strips_x2_with_priority.txt): config 5 still -0.6 /
  // -2.2 / -2.9 / -1.4 %, B 13 D 128 -2.6 %, but the defaults (+0.6 %) and B 9 D 128 (+1.3 %), whose tiles are all alike, no longer
  // want it.  A tail term in the model that reproduces the first sweep's optima (ideal rounds + half a workgroup) picked worse
  // counts elsewhere (profiles/r05_ab_tail2_first.txt).  The one-per-CU kernels (12 and 16 waves) measure best at the model's count.
  if (per_cu == 2 && uneven && env_int("SADHIP_TUNE_TAIL2", 1) != 0) {
    const long th = (rows + best_ns - 1) / best_ns, ns0 = (rows + th - 1) / th;
    if (base * ns0 > (long)cus * per_cu && th / 2 >= std::max<long>(min_rows, 2L * h2)) best_ns = 2 * ns0;
  }
  return best_ns;
}

int cu_count()
{
  thread_local int dev_cached = -1, cus = 256;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return cus;
  if (dev != dev_cached) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cus = n;
    dev_cached = dev;
  }
  return cus;
}

}  // namespace sadhip
