// colwave_plan.h -- host-side planner of the column-wave kernel (colwave_kernel.h): which kernel variant runs a
// disparity range (columns per lane, chunk size, waves, packed or not), how a range wider than one workgroup is cut
// into passes, and how many strips a launch cuts a frame into.  No device code, no kernel template: the launch
// table in kernels_fast.hip maps a Plan onto an instantiation.
#pragma once
#include "sadhip_kernels.h"
#include <vector>

namespace sadhip {

// nw waves per workgroup; the LAST nwh of them own ndh() candidates each (mixed-chunk plans, 0 otherwise), the others nd
struct Plan {
  int p, nd, nw; bool pk, extra, w16; int nwh = 0;
  int ndh() const { return pk ? nd - 4 : nd / 2; }                   // chunk of the trailing waves (MixChunk in colwave_kernel.h)
  int slots() const { return (nw - nwh) * nd + nwh * ndh(); }        // candidates of the pass without the odd one
};

// LDS bytes of one workgroup of a plan (the layout at the top of Cw::run): 2h+3 ring rows and the
// row of zeros, two row-best buffers, the output LUT.
size_t plan_lds_bytes(const Plan& pl, int h, int maxdisp);

struct Pass { Plan pl; int dbase; };              // candidates [dbase, dbase + nw*nd (+1)) of the job
struct PassIo { int dbase; const unsigned* kin; unsigned* kout; int kw; };
// launch geometry of one pass, as launch_t would launch it on the current device (sadhip_plan_describe_batch)
struct PassGeom { int tiles = 0, strips = 0, rows = 0, per_cu = 0; bool device = false; };

// Tuning aids.  The product build (libsadhip.so) reads NO environment variable: env_int() is the
// constant default there, so a stray variable in the host process cannot change a plan.  The
// diagnostics build (-DSADHIP_TUNING -> lib/libsadhip_tune.so, loaded by tools/ only) reads
// SADHIP_TUNE_* once per process (a launch consults them a few hundred times while planning).
#ifdef SADHIP_TUNING
int env_int(const char* name, int dflt);
#else
constexpr int env_int(const char*, int dflt) { return dflt; }
#endif

// Kernel variant for `count` consecutive candidates (0 waves: none); see colwave_plan.hip.
Plan plan_count(int h, bool pk, int count, bool exact, int maxdisp);
// The passes of the whole range [0, maxdisp]; empty: the generic kernel runs the job.  One-entry memo per thread (a
// caller streams frames with the same parameters): the reference stays valid until the thread plans other parameters.
const std::vector<Pass>& plan_job(int h, int maxdisp, bool narrow);
// Strips per frame of a launch (cost model in colwave_plan.hip).
// uneven: the kernel's workgroups differ in length by tile (packed h = 6, 7: key masks / two forms of the row loop)
long choose_strips(long base, long rows, int h2, int cus, int per_cu, int min_rows, bool uneven);
int cu_count();

}  // namespace sadhip
