// colwave_kernel.h -- the "column-wave" SAD disparity kernel for gfx950 (MI355X): device code only.
// (Planner: colwave_plan.{h,hip}; instantiations, launch table and launch geometry: kernels_fast.hip.)
//
// Function computed (bit-exact with pkg/despair/sad.go:55-94 + :205-244, closed form of
// SURVEY.md section 8 a-1):   out(x,y) = scale(argmin_{d in [0,min(D,max(0,x-h))]} SAD(x,y,d)),
// first minimum wins, windows clipped to the image.
//
// Mapping (why it looks like this -- measured on MI355X, tools/ubench/*.hip):
//   * the path is integer-VALU bound: v_sad_*/min3/DPP/perm ops cost ~4.3 SIMD-cycles per
//     wave-instruction, plain v_add/v_sub ~2.5, v_qsad_pk_u16_u8 is quarter rate; one LDS dword per
//     lane costs as much as two such VALU ops.  So: every |L-R| is one SAD instruction with the
//     accumulate folded in, everything else is adds/subs, operands stay in registers.
//   * lane l of a wave owns P (4 or 8) adjacent columns; a wave owns ND consecutive disparities
//     for its 64*P columns; the NW waves of a workgroup split the disparity range of one tile.
//     NW is a multiple of 4 with equal work per wave whenever D allows: the four SIMDs of a CU
//     issue oldest-workgroup-first, so a workgroup whose waves sit 2,2,1,1 on them runs at the pace
//     of the doubly loaded SIMDs and a quarter of the CU idles (tools/ubench/block_bench.hip:
//     6 waves per workgroup cost 1.29x, 4 / 8 / 12 cost 1.00x).  The UI's disparity ranges are
//     multiples of 16, i.e. 16k+1 candidates: the last wave carries the odd one out.
//   * vertical: each lane keeps the column sums of its (column, disparity) pairs in registers and
//     slides them down the strip: + |L-R| of the row entering the window, - |L-R| of the row
//     leaving it.  Packed variant (window SAD < 65536): two adjacent disparities share a register
//     as u16 halves (v_sad_u8 -> low half, v_sad_hi_u8 -> high half); unpacked: u32 + v_sad_u32.
//   * horizontal: the (2h+1)-wide window sum = in-lane prefix sums + 2h values fetched from the
//     two neighbour lanes with v_add_u32_dpp wave_shr:1 / wave_shl:1 (no LDS).
//   * argmin: strict-'<' first minimum (sad.go:79) = unsigned min of keys (SAD << 16) | d
//     (v_min3_u32).  Packed: v_lshl_or_b32 / v_perm_b32 build the two keys of a register.
//     Unpacked: pixels are pre-scaled by 2^KSH and every window contains exactly one column whose
//     sum starts at d, so the window sum IS the key.
//   * waves merge their per-pixel minima with ds_min_u32 into a row buffer in LDS; after the one
//     barrier per row every wave maps its 1/NW share of the keys -> d -> best*255/D (LUT) and
//     stores u8.
//   * image rows stream through an LDS ring (2h+3 rows, one u32 per pixel); every wave loads its
//     1/NW share of the next row before its arithmetic and lands it in the ring afterwards, so the
//     load latency hides behind a whole row of VALU work and all waves stay symmetric (a dedicated
//     streaming wave made one SIMD the pacemaker, see above).
//     ND is a multiple of 4 and the R part of a ring row is offset so that every
//     lane's operands are 16-byte aligned: all operand reads are ds_read_b128, bank-conflict free
//     at a lane stride of 16 bytes (ds_read_b32 at that stride was a 4-way conflict: 72 % of LDS
//     cycles in the first version).  Variants whose u32 ring would leave room for one workgroup
//     per CU only keep u16 pixels instead (W16).
//   * missing candidates (d > x-h at the left image edge, d > D in a ragged last chunk) are removed
//     by the START VALUES of the column sums (penalties that survive the sliding updates), not by
//     masking keys row after row; packed h = 6, 7, where a penalty does not fit u16, masks keys
//     with 64-bit lane masks held in SGPRs (one v_cndmask_b32 per key, leftmost tile only).
//   * BlockSize 18..31: the window reaches into the lane after the neighbour (two DPP hops).
//   * D+1 beyond 16 waves' worth: several launches over candidate ranges, a u32 key per pixel
//     travelling between them (plan_job in colwave_plan.hip / launch_fast in kernels_fast.hip).
#pragma once
#include "sadhip_kernels.h"
#include <type_traits>

namespace sadhip {

namespace {

struct CwArgs {
  Job j;
  int nw;          // waves per workgroup (disparity chunks), 4..16
  int nwa;         // waves [0, nwa) own ND candidates each; mixed-chunk kernels (k_colwave<..., MIX>): waves [nwa, nw) own NDB each
                   // (MixChunk: ND/2 unpacked, ND-4 packed).  nwa == nw otherwise
  int slots;       // candidates of the pass without the odd one: nwa*ND + (nw - nwa)*NDB
  int xw;          // wave that also owns candidate `slots` (the 16k+1-th), or -1
  int ksh;         // bits of the disparity field of a key INSIDE a pass (packed: 16, the field holds d itself; unpacked:
                   // 2^ksh > the pass's slot count, the field holds d - dbase -- at most 8 bits, so the pre-shifted pixels fit
                   // the u16 ring whatever MaxDisparity is)
  int kshg;        // ... of a key that travels between passes / is turned into the output (unpacked: 2^kshg > maxdisp; packed: 16)
  int pixshift;    // left shift applied to pixels landing in LDS (unpacked: ksh; packed: 0)
  int dpad;        // columns of R kept left of the tile (== 3 mod 4, >= largest candidate)
  int tx0;         // origin of tile 0 (<= 0, == w mod P)
  int tile0;       // first tile index launched (blockIdx.y + tile0)
  int nstrips;     // strips per frame: blockIdx.x = strip + nstrips * frame
  int th;          // output rows per strip
  int shl, shr;    // per-wave share of a ring row: L elements (= output pixels), R elements
  // Disparity ranges wider than one workgroup can hold run as several passes (launches) over
  // candidates [dbase, dbase + nw*ND (+1)): a pass merges its per-pixel keys with the keys of the
  // passes before it (kin) and hands them on (kout); the last pass writes the u8 output.
  int dbase;
  const unsigned* kin;     // keys so far, or nullptr (first pass)
  unsigned* kout;          // where to leave the keys, or nullptr (last pass: write the output)
  int kw;                  // u32 per key row (= x1 - x0); key of (f, y, x) at (f*rows + y-y0)*kw + x-x0
#ifdef SADHIP_TUNING
  int prio_off;    // diagnostics build: 1 = no issue priority by progress (the hardware's oldest-first only), for the A/B
#endif
#ifdef SADHIP_STAMPS
  // stamps build only (make stamps): per-wave sums of shader-clock ticks spent in the five phases of a row
  // step (update, reduce, land, output, barrier wait) + the step count, at stamps[(workgroup*16 + wave)*8 + k]
  unsigned long long* stamps;
  // ... and (SADHIP_TUNE_WGTL_FILE) the workgroup timeline: per workgroup {start, end in REFCLK ticks (100 MHz, one clock for the
  // whole chip), HW_ID, XCC_ID} at wgtl[workgroup * 4 + k] -- which CU ran which workgroup when (round 5: where the resident waves
  // that config 5 lacks go missing)
  unsigned long long* wgtl;
#endif
};

constexpr unsigned KEY_INVALID = 0xffffffffu;
// |a - b| + acc in one instruction, in place.  Inline asm: left to itself hipcc canonicalises the
// C++ form into v_max_u32 + v_min_u32 + v_sub/v_add3 (3 ops) inside the unrolled update.
__device__ __forceinline__ void sad_acc(unsigned& acc, unsigned a, unsigned b)
{
  asm("v_sad_u32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
// c_i -= |l - r_i| for four disparities in ONE asm statement.  hipcc pads every asm boundary
// whose registers the next instruction touches with an s_nop, and on this chip an s_nop costs a
// full issue slot (tools/ubench/block_bench.hip: 46 such pads were 24 % of the update block);
// four per statement leaves one pad per four candidates and keeps the sad -> sub pairs apart.
__device__ __forceinline__ void sad_sub4(unsigned& c0, unsigned& c1, unsigned& c2, unsigned& c3,
                                         unsigned l, unsigned r0, unsigned r1, unsigned r2, unsigned r3)
{
  unsigned t0, t1, t2, t3;
  asm("v_sad_u32 %4, %8, %9, 0\n\t"
      "v_sad_u32 %5, %8, %10, 0\n\t"
      "v_sad_u32 %6, %8, %11, 0\n\t"
      "v_sad_u32 %7, %8, %12, 0\n\t"
      "v_sub_u32 %0, %0, %4\n\t"
      "v_sub_u32 %1, %1, %5\n\t"
      "v_sub_u32 %2, %2, %6\n\t"
      "v_sub_u32 %3, %3, %7"
      : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
      : "v"(l), "v"(r0), "v"(r1), "v"(r2), "v"(r3));
}
// c_i -= |l_i - r_i| for four columns of one disparity.
__device__ __forceinline__ void sad_sub4x(unsigned& c0, unsigned& c1, unsigned& c2, unsigned& c3,
                                          unsigned l0, unsigned l1, unsigned l2, unsigned l3,
                                          unsigned r0, unsigned r1, unsigned r2, unsigned r3)
{
  unsigned t0, t1, t2, t3;
  asm("v_sad_u32 %4, %8, %12, 0\n\t"
      "v_sad_u32 %5, %9, %13, 0\n\t"
      "v_sad_u32 %6, %10, %14, 0\n\t"
      "v_sad_u32 %7, %11, %15, 0\n\t"
      "v_sub_u32 %0, %0, %4\n\t"
      "v_sub_u32 %1, %1, %5\n\t"
      "v_sub_u32 %2, %2, %6\n\t"
      "v_sub_u32 %3, %3, %7"
      : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
      : "v"(l0), "v"(l1), "v"(l2), "v"(l3), "v"(r0), "v"(r1), "v"(r2), "v"(r3));
}
// Byte-lane forms of the two helpers above (triple-column operands, Cw::TR3: three pixel bytes per word).
__device__ __forceinline__ void sad8_acc(unsigned& acc, unsigned a, unsigned b)
{
  asm("v_sad_u8 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void sad8_sub4x(unsigned& c0, unsigned& c1, unsigned& c2, unsigned& c3,
                                           unsigned l0, unsigned l1, unsigned l2, unsigned l3,
                                           unsigned r0, unsigned r1, unsigned r2, unsigned r3)
{
  unsigned t0, t1, t2, t3;
  asm("v_sad_u8 %4, %8, %12, 0\n\t"
      "v_sad_u8 %5, %9, %13, 0\n\t"
      "v_sad_u8 %6, %10, %14, 0\n\t"
      "v_sad_u8 %7, %11, %15, 0\n\t"
      "v_sub_u32 %0, %0, %4\n\t"
      "v_sub_u32 %1, %1, %5\n\t"
      "v_sub_u32 %2, %2, %6\n\t"
      "v_sub_u32 %3, %3, %7"
      : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
      : "v"(l0), "v"(l1), "v"(l2), "v"(l3), "v"(r0), "v"(r1), "v"(r2), "v"(r3));
}
// ---- packed variant: one register holds the column sums of two adjacent disparities as u16 halves.
// Pixels in LDS are plain zero-extended bytes, so v_sad_u8 / v_sad_hi_u8 (|a-b| over 4 byte lanes,
// result added to the low / high half of the accumulator) are exact single-pixel differences.
// Two registers (four candidates) per asm statement, independent chains interleaved.
__device__ __forceinline__ void pk_sad_in2(unsigned& c0, unsigned& c1, unsigned l,
                                           unsigned re0, unsigned ro0, unsigned re1, unsigned ro1)
{
  asm("v_sad_u8 %0, %2, %3, %0\n\t"
      "v_sad_u8 %1, %2, %5, %1\n\t"
      "v_sad_hi_u8 %0, %2, %4, %0\n\t"
      "v_sad_hi_u8 %1, %2, %6, %1"
      : "+v"(c0), "+v"(c1) : "v"(l), "v"(re0), "v"(ro0), "v"(re1), "v"(ro1));
}
__device__ __forceinline__ void pk_sad_out2(unsigned& c0, unsigned& c1, unsigned l,
                                            unsigned re0, unsigned ro0, unsigned re1, unsigned ro1)
{
  unsigned t0, t1;
  asm("v_sad_u8 %2, %4, %5, 0\n\t"
      "v_sad_u8 %3, %4, %7, 0\n\t"
      "v_sad_hi_u8 %2, %4, %6, %2\n\t"
      "v_sad_hi_u8 %3, %4, %8, %3\n\t"
      "v_sub_u32 %0, %0, %2\n\t"
      "v_sub_u32 %1, %1, %3"
      : "+v"(c0), "+v"(c1), "=&v"(t0), "=&v"(t1) : "v"(l), "v"(re0), "v"(ro0), "v"(re1), "v"(ro1));
}
// Same for four registers (eight candidates): the temporaries of consecutive statements share
// registers and hipcc pads that write-after-write across asm boundaries with an s_nop (a full issue
// slot here), so fewer, larger statements.
__device__ __forceinline__ void pk_sad_out4(unsigned& c0, unsigned& c1, unsigned& c2, unsigned& c3, unsigned l,
                                            unsigned re0, unsigned ro0, unsigned re1, unsigned ro1,
                                            unsigned re2, unsigned ro2, unsigned re3, unsigned ro3)
{
  unsigned t0, t1, t2, t3;
  asm("v_sad_u8 %4, %8, %9, 0\n\t"
      "v_sad_u8 %5, %8, %11, 0\n\t"
      "v_sad_u8 %6, %8, %13, 0\n\t"
      "v_sad_u8 %7, %8, %15, 0\n\t"
      "v_sad_hi_u8 %4, %8, %10, %4\n\t"
      "v_sad_hi_u8 %5, %8, %12, %5\n\t"
      "v_sad_hi_u8 %6, %8, %14, %6\n\t"
      "v_sad_hi_u8 %7, %8, %16, %7\n\t"
      "v_sub_u32 %0, %0, %4\n\t"
      "v_sub_u32 %1, %1, %5\n\t"
      "v_sub_u32 %2, %2, %6\n\t"
      "v_sub_u32 %3, %3, %7"
      : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
      : "v"(l), "v"(re0), "v"(ro0), "v"(re1), "v"(ro1), "v"(re2), "v"(ro2), "v"(re3), "v"(ro3));
}
__device__ __forceinline__ unsigned from_left(unsigned v)     // lane l <- lane l-1 (lane 0: 0)
{
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, true);    // wave_shr:1 bound_ctrl:0
}
__device__ __forceinline__ unsigned from_right(unsigned v)    // lane l <- lane l+1 (lane 63: 0)
{
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, true);    // wave_shl:1 bound_ctrl:0
}

// key where the lane's bit of a 64-bit lane mask (SGPR pair) is set, else KEY_INVALID: one
// v_cndmask_b32 with the mask as its scalar operand -- no v_cmp, no VCC round trip.
__device__ __forceinline__ unsigned sel_lanes(unsigned long long mask, unsigned key)
{
  unsigned r;
  asm("v_cndmask_b32 %0, -1, %1, %2" : "=v"(r) : "v"(key), "s"(mask));
  return r;
}
// Workgroup barrier that waits for this wave's LDS operations only.  __syncthreads() also emits
// s_waitcnt vmcnt(0): every wave would then sit at every row barrier until the output bytes it has
// just stored are acknowledged (a full memory round trip per row).  Nothing depends on them;
// hipcc places the vmcnt wait of the row loads at their use by itself.  "memory" keeps the
// compiler from moving LDS accesses across.
__device__ __forceinline__ void wg_barrier()
{
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int P, int ND, int H, bool PK, bool W16 = false, bool TR = false>
struct Cw {                                         // P columns per lane: 4 (packed, h <= 7; two-lane halos from h = 5) or 8 (h <= 15)
  // PK: column sums of disparity pairs packed as u16 halves (window SAD < 65536, i.e. h <= 7)
  static_assert(!PK || (H <= 7 && ND % 4 == 0), "packed sums need (2h+1)^2 * 255 < 65536");
  static constexpr int NC = PK ? ND / 2 : ND;       // accumulator registers per column
  static_assert(P == 4 || P == 8, "operand reads are whole ds_read_b128");
  static_assert(H >= 1 && H <= 2 * P - 1, "halo must fit in the two neighbour lanes on each side");
  // two-lane halos (P < h): the unpacked variants, and packed P = 4 for h = 5..7 (see plan_count)
  static_assert(H <= P || !PK || P == 4, "packed two-lane halos: P = 4 only");
  static_assert(ND % 4 == 0, "ND must keep every lane's R operands 16-byte aligned");
  static constexpr int TW = 64 * P;                 // columns per tile
  static constexpr int OL = P * ((H + P - 1) / P);  // first output column inside the tile
  static constexpr int STEP = TW - 2 * OL;          // output columns per tile
  static constexpr int B1 = 2 * H + 1;              // window width
  static constexpr int NB = 2;                      // row-best buffers: one being filled, one being drained
  // A row buffer holds the key of column p of lane l at p*RBP + l: the P ds_min_u32 of a wave then
  // touch consecutive dwords (at l*P + p all 64 lanes hit 4 (P = 8) or 8 banks: 16-/8-way conflicts,
  // two thirds of the LDS cycles of the P = 8 variants), and pitch 72 keeps the output pass, which
  // walks the pixels in image order, at two-way.
  static constexpr int RBP = 72;
  static constexpr int RBW = P * RBP;               // u32 per row buffer
  static constexpr int NR = 2 * H + 3;              // ring rows: window + the leaving row + the landing row
  static constexpr int NRV = P + ND;                // R values a lane loads per row (last one unused)
  static constexpr int MAXL = P / 4;                // ring-row L elements per lane (>= 4 waves share a row)
  // W16: ring pixels are u16 (two per dword) instead of u32 -- half the LDS per workgroup, one
  // v_and / v_lshrrev more per operand.  For the variants whose ring (2h+4 rows of 1100 pixels)
  // otherwise leaves room for one workgroup per CU: P = 8, h >= 7.
  // ND = 8: every operand block is whole 16-byte reads.  ND = 4 (round 4; 16k+1 candidates with k odd -- D = 16, 48, 80, ... --
  // plan as 4-candidate chunks instead of rounding up to 8): a lane's 12 R pixels are 24 bytes that start on an 8-byte
  // boundary (on a 16-byte one for every other wave), read as three ds_read_b64; ring rows stay 16-byte multiples for
  // even wave counts (plan_count_p / launch_t check).
  static_assert(!W16 || (P == 8 && (ND == 8 || ND == 4)), "u16 ring: P = 8, chunks of 8 (16-byte reads) or 4 (8-byte reads of the R operands)");
  static constexpr bool R_A16 = !W16 || ND % 8 == 0;   // a lane's R operand block starts 16-byte aligned
  using Px = typename std::conditional<W16, unsigned short, unsigned>::type;
  // SWZ: quad-swizzled ring rows for the u32 ring of the P = 8, ND = 8 variants.  A lane's operands
  // are 8 (L) / 16 (R) consecutive pixels = 32-byte lane stride: as ds_read_b128 that is a 2-way bank
  // conflict (lanes l and l+8 of one 16-lane service group hit the same banks; 45 % of the LDS cycles
  // of config 5 were conflict cycles, profiles/r02_c5.md).  Each part (L, R) of a ring row is
  // therefore stored as two halves: quads (4 pixels) with even index in the first half, odd ones in
  // the second, so that every ds_read_b128 of a lane's operand set sees a 16-byte lane stride again.
  static constexpr bool SWZ = (P == 8) && !W16 && (ND == 8);
  static constexpr int XOFF = (W16 || SWZ) ? 7 : 3;  // dpad = nw*ND + XOFF keeps every lane's R operands 16-byte aligned (SWZ: on an even quad)
  // physical position of logical pixel e within a part of len pixels (len % 8 == 0), and its inverse
  static __device__ __forceinline__ int phys(int e, int len)
  {
    return SWZ ? ((((e >> 3) << 2) | (e & 3)) + ((e >> 2) & 1) * (len >> 1)) : e;
  }
  static __device__ __forceinline__ int logical(int f, int len)
  {
    if constexpr (!SWZ) return f;
    const int hl = len >> 1, hi = f >= hl ? 1 : 0, t = f - hi * hl;
    return ((t >> 2) << 3) | (hi << 2) | (t & 3);
  }
  struct Src { const Px* a; const Px* b; };           // operand block of a lane: even quads at a, odd quads at b (SWZ); else all at a
  static constexpr int MAXR = (TW + 4 * ND + 8 + 255) / 256;   // ... R elements per lane
  // packed h = 6, 7: 2*(2h+1)^2*255 does not fit u16, so missing candidates are removed by masking keys (lane masks)
  // instead of penalised start values.  The triple-column form of h = 7 carries NEITHER: k_colwave runs it only in
  // workgroups where every candidate exists for every output pixel (not the leftmost tiles), so it holds no mask
  // state at all (with it the two-form kernel spilled 24 SGPRs into VGPR lanes and ran scalar work on the vector ALU).
  static constexpr bool NEED_MASK = PK && H > 5 && !TR;
  // TR3: "triple-column" form of the packed P = 4 variants whose window width is a multiple of three
  // (BlockSize 9: h = 4; BlockSize 15: h = 7, five triples reaching two lanes either way; BlockSize 3: h = 1, not used).  v_sad_u8 sums FOUR byte lanes, and the plain form
  // uses one of them.  Here a ring word of column c holds the pixels c, c+1, c+2 as its three low
  // bytes, so the same instruction accumulates T(x, d) = the column sums of x, x+1, x+2 together, and
  // the (2h+1)-wide window of pixel x is T(x-h) + T(x-h+3) + ... : for h = 4 two additions per column
  // (seven of eight as DPP adds pulling from a neighbour lane) instead of the prefix sums + suffix
  // differences + two DPP adds of the plain form -- 8 instructions per register instead of 13; for
  // h = 1 the window IS T(x-1).  Same sums, same keys, same penalties per window (init_sum below).
  // At the right image edge the form is wrong: the register of column w-2 (w-1) would also sum |0 - R| of
  // the columns w (w, w+1), which a clipped window of the reference does not contain -- R(x-d) is a real
  // pixel there, only L is outside the image.  (At the left edge both are outside: |0 - 0|.)  So a
  // workgroup whose tile holds the columns w-2, w-1 runs the plain form (k_colwave picks per workgroup);
  // TR is set for the other workgroups of the BlockSize-9 kernels only.  (BlockSize 3, whose window IS one
  // triple, measured slower in this form: more load traffic per streamed pixel for no arithmetic saved
  // that its small horizontal pass had not saved already.)
  static constexpr bool TR3 = TR;
  static_assert(!TR || (PK && P == 4 && !W16 && (H == 4 || H == 7)), "triple-column form: packed P = 4, h = 4 or 7 (BlockSize 9, 15)");
  static constexpr int NM = ND + P;                 // lane masks: one per (candidate - column), see run()
  // Mask k is "lanes >= max(0, floor((T0 + k) / P))" for a wave-uniform T0: consecutive k share their
  // mask P at a time, so a wave holds only NMB distinct masks plus one carry bit per residue k % P, and
  // the mask of a compile-time k is a scalar select between two neighbours (SALU: no vector issue slot,
  // no VGPR).  Round 2's first form kept all NM masks in SGPR pairs: 32-40 SGPRs, 13-25 of them parked
  // in VGPR lanes by the compiler.
  static constexpr int NMB = (NM - 1) / P + 2;
  struct Masks {
    unsigned long long mb[NMB];
    unsigned carry;                                   // bit r: (r + T0 mod P) >= P
    __device__ __forceinline__ unsigned long long at(int k) const
    {
      const int i = k / P;
      unsigned long long a = mb[i], b = mb[i + 1];
      asm volatile("" : "+s"(a), "+s"(b));            // two scalars selected HERE: not "mb[i + bit]" (array goes to scratch), not hoisted out of the row loop (40 SGPRs again)
      return ((carry >> (k % P)) & 1u) ? b : a;
    }
  };

  // N ring pixels starting at a 16-byte aligned element -> one u32 register each.
  // (u16 ring, A16 = false: N ring pixels starting at an 8-byte aligned element, ds_read_b64 each)
  template <int N, bool A16 = true>
  static __device__ __forceinline__ void load_px(Src src, unsigned (&out)[N])
  {
    const uint4* s4 = reinterpret_cast<const uint4*>(src.a);
    if constexpr (W16) {
      // the mask lives in a VGPR: v_and_b32 with the 32-bit literal 0xffff is a slow-class instruction
      // (4.2 cycles), with a register operand it is fast-class (2.3 in a run) -- tools/ubench/valu_rates.hip
      unsigned lo16 = 0xffffu;
      asm("" : "+v"(lo16));
      if constexpr (A16 && N % 8 == 0) {
#pragma unroll
        for (int m = 0; m < N / 8; ++m) {
          const uint4 v = s4[m];
          out[8 * m] = v.x & lo16; out[8 * m + 1] = v.x >> 16; out[8 * m + 2] = v.y & lo16; out[8 * m + 3] = v.y >> 16;
          out[8 * m + 4] = v.z & lo16; out[8 * m + 5] = v.z >> 16; out[8 * m + 6] = v.w & lo16; out[8 * m + 7] = v.w >> 16;
        }
      } else {
        static_assert(N % 4 == 0, "");
        const uint2* s2 = reinterpret_cast<const uint2*>(src.a);
#pragma unroll
        for (int m = 0; m < N / 4; ++m) {
          const uint2 v = s2[m];
          out[4 * m] = v.x & lo16; out[4 * m + 1] = v.x >> 16; out[4 * m + 2] = v.y & lo16; out[4 * m + 3] = v.y >> 16;
        }
      }
    } else {
      static_assert(N % 4 == 0, "");
      const uint4* t4 = reinterpret_cast<const uint4*>(src.b);
#pragma unroll
      for (int m = 0; m < N / 4; ++m) {
        const uint4 v = SWZ ? ((m & 1) ? t4[m >> 1] : s4[m >> 1]) : s4[m];
        out[4 * m] = v.x; out[4 * m + 1] = v.y; out[4 * m + 2] = v.z; out[4 * m + 3] = v.w;
      }
    }
  }

  // ---- vertical slide: C += |L-R|(entering row) ; C -= |L-R|(leaving row) -----------------------
  // lin/lout: this lane's P L values; rin/rout: its NRV R values, value m = column x0 - (dc+ND-1) + m.
  // leave = false (the first 2h+1 steps of a strip: the window is not full yet, the "leaving row" is the row of zeros):
  // only the entering row is accumulated -- 40 % of a warm-up step, which short strips (single frames: 12 rows + 8 warm-up
  // rows per strip) repeat for every strip.
  static __device__ __forceinline__ void update_columns(unsigned (&C)[P][NC], Src lin, Src rin, Src lout, Src rout, const bool leave)
  {
    {
      unsigned li[P], ri[NRV];
      load_px(lin, li);
      load_px<NRV, R_A16>(rin, ri);
      if constexpr (PK) {
#pragma unroll
        for (int p = 0; p < P; ++p)
#pragma unroll
          for (int r = 0; r < NC; r += 2)                   // candidates q = 2r .. 2r+3
            pk_sad_in2(C[p][r], C[p][r + 1], li[p], ri[p + ND - 1 - 2 * r], ri[p + ND - 2 - 2 * r],
                       ri[p + ND - 3 - 2 * r], ri[p + ND - 4 - 2 * r]);
      } else {
#pragma unroll
        for (int p = 0; p < P; ++p)
#pragma unroll
          for (int q = 0; q < ND; ++q) sad_acc(C[p][q], li[p], ri[p + ND - 1 - q]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);                    // keep the two operand sets from overlapping
    if (leave) {
      unsigned lo[P], ro[NRV];
      load_px(lout, lo);
      load_px<NRV, R_A16>(rout, ro);
      if constexpr (PK) {
#pragma unroll
        for (int p = 0; p < P; ++p) {
#pragma unroll
          for (int r = 0; r + 4 <= NC; r += 4)
            pk_sad_out4(C[p][r], C[p][r + 1], C[p][r + 2], C[p][r + 3], lo[p],
                        ro[p + ND - 1 - 2 * r], ro[p + ND - 2 - 2 * r], ro[p + ND - 3 - 2 * r], ro[p + ND - 4 - 2 * r],
                        ro[p + ND - 5 - 2 * r], ro[p + ND - 6 - 2 * r], ro[p + ND - 7 - 2 * r], ro[p + ND - 8 - 2 * r]);
          if constexpr (NC % 4 != 0) {
            constexpr int r = NC - 2;
            pk_sad_out2(C[p][r], C[p][r + 1], lo[p], ro[p + ND - 1 - 2 * r], ro[p + ND - 2 - 2 * r],
                        ro[p + ND - 3 - 2 * r], ro[p + ND - 4 - 2 * r]);
          }
        }
      } else {
#pragma unroll
        for (int p = 0; p < P; ++p)
#pragma unroll
          for (int q = 0; q < ND; q += 4)
            sad_sub4(C[p][q], C[p][q + 1], C[p][q + 2], C[p][q + 3], lo[p], ro[p + ND - 1 - q],
                     ro[p + ND - 2 - q], ro[p + ND - 3 - q], ro[p + ND - 4 - q]);
      }
    }
  }

  // The wave that owns the odd candidate dx = nw*ND (16k+1 candidates, see plan_job) slides its
  // column sums here: lin/rxin point at this lane's L pixels and at the 16-byte aligned block whose
  // element XOFF is R column x0 - dx.  Pixels are plain bytes (packed) or pre-shifted (unpacked):
  // v_sad_u32 accumulates the right thing in both.
  static __device__ __forceinline__ void extra_update(unsigned (&CX)[P][1], Src lin, Src rxin, Src lout, Src rxout, const bool leave)
  {
    constexpr int NX = XOFF == 7 ? 16 : P + 4;            // elements XOFF .. XOFF+P-1 are used
    {
      unsigned li[P], ri[NX];
      load_px(lin, li);
      load_px(rxin, ri);
#pragma unroll
      for (int p = 0; p < P; ++p) {
        if constexpr (TR3) sad8_acc(CX[p][0], li[p], ri[p + XOFF]);
        else sad_acc(CX[p][0], li[p], ri[p + XOFF]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (leave) {
      unsigned lo[P], ro[NX];
      load_px(lout, lo);
      load_px(rxout, ro);
#pragma unroll
      for (int p = 0; p < P; p += 4) {
        if constexpr (TR3)
          sad8_sub4x(CX[p][0], CX[p + 1][0], CX[p + 2][0], CX[p + 3][0], lo[p], lo[p + 1], lo[p + 2], lo[p + 3],
                     ro[p + XOFF], ro[p + XOFF + 1], ro[p + XOFF + 2], ro[p + XOFF + 3]);
        else
          sad_sub4x(CX[p][0], CX[p + 1][0], CX[p + 2][0], CX[p + 3][0], lo[p], lo[p + 1], lo[p + 2], lo[p + 3],
                    ro[p + XOFF], ro[p + XOFF + 1], ro[p + XOFF + 2], ro[p + XOFF + 3]);
      }
    }
  }

  // ---- horizontal window sums + key minimum for one row --------------------------------------
  // Per disparity q: stage A = in-lane prefix sums ps[] and the suffix sums the right neighbour
  // needs; stage B = window sums (DPP adds pulling from both neighbour lanes) folded into best[].
  // A DPP operand must be written >= 2 instructions before the DPP op reads it, or the compiler
  // pads with s_nop (a full issue slot each here): stage A ends with the suffix subtractions the
  // DPP adds do not read first, and the stages are pinned with sched_barrier.  (Software-pipelining
  // A(q+1) in front of B(q) saved a few pads but cost 13-18 VGPRs in the P = 8 variants and was
  // no faster anywhere.)
  struct Pre { unsigned ps[P + 1]; unsigned suf[P + 1]; };   // suf[k] = sum of the lane's last k columns

  template <int W>
  static __device__ __forceinline__ void stage_a(const unsigned (&C)[P][W], int q, Pre& r)
  {
    r.ps[0] = 0;
    if constexpr (TR3) {                                   // no prefix sums: ps[k + 1] carries the triple sum T of column k as it is
#pragma unroll
      for (int k = 0; k < P; ++k) r.ps[k + 1] = C[k][q];
      return;
    }
#pragma unroll
    for (int k = 0; k < P; ++k) r.ps[k + 1] = r.ps[k] + C[k][q];
#pragma unroll
    for (int k = 1; k <= H && k <= P; ++k) r.suf[k] = r.ps[P] - r.ps[P - k];
  }

  // Unpacked: the window sum is the key (bias trick).  Packed: the window sums of disparities
  // dbase+2q (low half) and dbase+2q+1 (high half) become two keys (SAD << 16) | d.
  // MASK (packed h = 6, 7 only; every other variant removes missing candidates through the start
  // values of the column sums, see run()): 0 = every candidate of the chunk exists for every pixel;
  // 2 = candidate j of the wave exists for column p of lane l iff bit l of M[j - p + P-1] is set
  // (d <= x-h near the left image edge).  plan_for() gives these variants exact plans only (no chunk
  // runs past MaxDisparity); the extra candidate's empty high half is dropped here (qn = 1).
  // JOFS = index of the array's first candidate within the wave (0; ND for the extra candidate).
  // LOW_ONLY = the register holds one real candidate (the extra one, low half): no key for the high half.
  template <int MASK, int JOFS, bool LOW_ONLY = false>
  static __device__ __forceinline__ void stage_b(const Pre& r, int q, const Masks& M,
                                                 unsigned (&best)[P], unsigned dbase, int qn)
  {
    // window sum of column p = own columns + last (h-p) columns of the left lane + first (p+h-P+1)
    // columns of the right lane.  The DPP adds are issued as two sweeps over p (all "left" pulls,
    // then all "right" pulls): a DPP instruction that reads a VGPR written by the instruction just
    // before it costs wait states, so the two adds of one column are kept P instructions apart.
    unsigned s[P];
    if constexpr (TR3) {
      // window of column p = T(p-h) + T(p-h+3) + ... ((2h+1)/3 terms); T(i) of this lane for 0 <= i < P, of the
      // left lane for i < 0 (T(i+P) there), of the right lane for i >= P.  Own-lane terms first, then the
      // left pulls, then (second sweep, as in the plain form) the right pulls: every pull is one DPP add.
      constexpr int M3 = (2 * H + 1) / 3;
      if constexpr (H == 4 && P == 4) {
        // the four windows by hand: T0 + T3 serves two of them (once as itself, once pulled from the left
        // lane as a whole: left(T0) + left(T3) = left(T0 + T3)) -- 7 instructions instead of 8
        const unsigned t0 = r.ps[1], t1 = r.ps[2], t2 = r.ps[3], t3 = r.ps[4];
        const unsigned u = t0 + t3;
        s[1] = u + from_left(t1);
        s[2] = t1 + from_left(t2);
        s[3] = t2 + from_left(t3);
        s[0] = t2 + from_left(u);                              // (three instructions after u: no DPP hazard pad)
#pragma unroll
        for (int p = 0; p < P; ++p) asm("" : "+v"(s[p]));
        s[2] += from_right(t0);
        s[3] += from_right(t1);
      } else if constexpr (H == 7 && P == 4) {
        // BlockSize 15: window of pixel p = T(p-7) + T(p-4) + T(p-1) + T(p+2) + T(p+5), i.e. (lane-relative)
        //   p = 0:  L2.T1  L1.T0  L1.T3  T2     R1.T1          L1 / L2 = one / two lanes to the left, R1 / R2 to the right
        //   p = 1:  L2.T2  L1.T1  T0     T3     R1.T2
        //   p = 2:  L2.T3  L1.T2  T1     R1.T0  R1.T3
        //   p = 3:  L1.T0  L1.T3  T2     R1.T1  R2.T0
        // A term two lanes away travels as part of a partial sum built in the lane between (x0, x1, x2: "what my right
        // neighbour's pixel needs from me and from my left neighbour"), T0 + T3 serves three pixels (its own, and as a
        // whole the neighbours' on both sides) and T2 + R1.T1 two: 13 instructions per register, 12 of them DPP, against
        // the 20 of the plain form's prefix / suffix sums and two-hop pulls.  Every partial sum is a subset of ONE pixel's
        // window, so no u16 half can overflow (15 * 15 * 255 < 65536).
        const unsigned t0 = r.ps[1], t1 = r.ps[2], t2 = r.ps[3], t3 = r.ps[4];
        unsigned a = t0 + t3;
        unsigned x1 = t1 + from_left(t2);
        unsigned x2 = t2 + from_left(t3);
        unsigned y = t2 + from_right(t1);
        unsigned z = from_right(t0);                               // R1.T0, pulled once more below: R2.T0
        asm("" : "+v"(a), "+v"(x1), "+v"(x2), "+v"(y), "+v"(z));
        unsigned x0 = a + from_left(t1);
        s[1] = a + from_right(t2);
        s[2] = t1 + from_right(a);
        s[3] = y + from_left(a);
        asm("" : "+v"(x0), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]));
        s[0] = y + from_left(x0);
        s[1] += from_left(x1);
        s[2] += from_left(x2);
        s[3] += from_right(z);
      } else {
#pragma unroll
      for (int p = 0; p < P; ++p) {
        bool have = false;
#pragma unroll
        for (int j = 0; j < M3; ++j) {
          const int i = p - H + 3 * j;
          if (i >= 0 && i < P) { s[p] = have ? s[p] + r.ps[i + 1] : r.ps[i + 1]; have = true; }
        }
#pragma unroll
        for (int j = 0; j < M3; ++j) {
          const int i = p - H + 3 * j;
          if (i < 0) { s[p] = have ? s[p] + from_left(r.ps[i + P + 1]) : from_left(r.ps[i + P + 1]); have = true; }
        }
      }
#pragma unroll
      for (int p = 0; p < P; ++p) asm("" : "+v"(s[p]));
#pragma unroll
      for (int p = 0; p < P; ++p)
#pragma unroll
        for (int j = 0; j < M3; ++j) {
          const int i = p - H + 3 * j;
          if (i >= P) s[p] += from_right(r.ps[i - P + 1]);
        }
      }
    } else if constexpr (H == P && P == 8) {
      // BlockSize 16 / 17 (the reference's default parameters): every window holds the whole lane, the last 8-p columns of
      // the left lane and the first p+1 of the right one.  suffix[8-p] of the left lane = its total - its prefix[p], so
      //   W(p) = (total + left.total) + right.prefix[p+1] - left.prefix[p]:
      // no suffix sums at all (7 subtractions per candidate less) and one DPP add for the shared base -- 23 instructions
      // per candidate instead of 30.  Exact modulo 2^32 like every other sum here (penalised columns included).
      // The two subtractions are written as  n = left.prefix[p] - base ;  W = right.prefix[p+1] - n  in inline asm:
      // measured on this chip with this toolchain (tools/ubench/dpp_sub.hip, profiles/ubench_dpp_sub.txt),
      // `v_subrev_u32_dpp D, a, b <shift>` gives shift(b) - a in 64 of 64 lanes, NOT b - shift(a) as hipcc assumes when it
      // folds `base - dpp(x)` into it -- every key came out wrong; `v_sub_u32_dpp D, a, b` gives shift(a) - b, as the ISA
      // text says.  Either way the lane-shifted operand is the minuend.
      // (s_nop 1: a DPP source needs two wait states after the VALU instruction that wrote it, and the compiler does
      // not look into asm.)
      unsigned base = r.ps[P] + from_left(r.ps[P]);
      asm("" : "+v"(base));
      s[0] = base + from_right(r.ps[1]);
      unsigned n[P];
#define SADHIP_DPPSUB(SHIFT) \
      "s_nop 1\n\t" \
      "v_sub_u32_dpp %0, %7, %14 " SHIFT " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
      "v_sub_u32_dpp %1, %8, %15 " SHIFT " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
      "v_sub_u32_dpp %2, %9, %16 " SHIFT " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
      "v_sub_u32_dpp %3, %10, %17 " SHIFT " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
      "v_sub_u32_dpp %4, %11, %18 " SHIFT " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
      "v_sub_u32_dpp %5, %12, %19 " SHIFT " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
      "v_sub_u32_dpp %6, %13, %20 " SHIFT " row_mask:0xf bank_mask:0xf bound_ctrl:1"
      asm(SADHIP_DPPSUB("wave_shr:1")                            // n[p] = left.prefix[p] - base
          : "=&v"(n[1]), "=&v"(n[2]), "=&v"(n[3]), "=&v"(n[4]), "=&v"(n[5]), "=&v"(n[6]), "=&v"(n[7])
          : "v"(r.ps[1]), "v"(r.ps[2]), "v"(r.ps[3]), "v"(r.ps[4]), "v"(r.ps[5]), "v"(r.ps[6]), "v"(r.ps[7]),
            "v"(base), "v"(base), "v"(base), "v"(base), "v"(base), "v"(base), "v"(base));
      asm(SADHIP_DPPSUB("wave_shl:1")                            // W(p) = right.prefix[p+1] - n[p]
          : "=&v"(s[1]), "=&v"(s[2]), "=&v"(s[3]), "=&v"(s[4]), "=&v"(s[5]), "=&v"(s[6]), "=&v"(s[7])
          : "v"(r.ps[2]), "v"(r.ps[3]), "v"(r.ps[4]), "v"(r.ps[5]), "v"(r.ps[6]), "v"(r.ps[7]), "v"(r.ps[8]),
            "v"(n[1]), "v"(n[2]), "v"(n[3]), "v"(n[4]), "v"(n[5]), "v"(n[6]), "v"(n[7]));
#undef SADHIP_DPPSUB
    } else if constexpr (H <= P) {
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const int a = p - H, b = p + H;
        const int lo = a > 0 ? a : 0, hi = b < P - 1 ? b : P - 1;
        s[p] = lo == 0 ? r.ps[hi + 1] : r.ps[hi + 1] - r.ps[lo];
        if (a < 0) s[p] += from_left(r.suf[-a]);
      }
#pragma unroll
      for (int p = 0; p < P; ++p) asm("" : "+v"(s[p]));      // v_add_u32_dpp twice, not v_mov_dpp x2 + v_add3
#pragma unroll
      for (int p = 0; p < P; ++p)
        if (p + H > P - 1) s[p] += from_right(r.ps[p + H - P + 1]);
    } else {
      // P < h < 2P (BlockSize 19..31): the window covers the whole lane, up to the whole neighbour
      // lane on each side, and H-p-P (left) / p+H-2P+1 (right) columns of the lane after that:
      // the neighbours' totals once, the second hop as v_mov_dpp of the partial sum followed by a
      // DPP add one lane further.
      constexpr int N2 = H - P;                              // most columns needed from a lane two away
      unsigned t2l[N2 + 1], t2r[N2 + 1];
#pragma unroll
      for (int k = 1; k <= N2; ++k) { t2l[k] = from_left(r.suf[k]); t2r[k] = from_right(r.ps[k]); }
      if constexpr (2 * P - H - 1 <= 0) {
        // h >= 2P-1 (P = 4, h = 7; P = 8, h = 15): every window holds BOTH neighbour lanes whole -- their
        // totals join the lane's own once per candidate instead of once per column
        unsigned u3 = r.ps[P] + from_left(r.ps[P]);
        asm("" : "+v"(u3));
        u3 += from_right(r.ps[P]);
#pragma unroll
        for (int p = 0; p < P; ++p) {
          const int nl = H - p;
          s[p] = nl > P ? u3 + from_left(t2l[nl - P]) : u3;
        }
#pragma unroll
        for (int p = 0; p < P; ++p) asm("" : "+v"(s[p]));
#pragma unroll
        for (int p = 0; p < P; ++p) {
          const int nr = p + H - P + 1;
          if (nr > P) s[p] += from_right(t2r[nr - P]);
        }
      } else {
      const unsigned tl = from_left(r.ps[P]), tr = from_right(r.ps[P]);
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const int nl = H - p;                                // columns left of the lane (>= H-P+1 > 0)
        if (nl >= P) { s[p] = r.ps[P] + tl; if (nl > P) s[p] += from_left(t2l[nl - P]); }
        else s[p] = r.ps[P] + from_left(r.suf[nl]);
      }
#pragma unroll
      for (int p = 0; p < P; ++p) asm("" : "+v"(s[p]));
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const int nr = p + H - P + 1;                        // columns right of the lane (> 0)
        if (nr >= P) { s[p] += tr; if (nr > P) s[p] += from_right(t2r[nr - P]); }
        else s[p] += from_right(r.ps[nr]);
      }
      }
    }
#pragma unroll
    for (int p = 0; p < P; ++p) {
      if constexpr (PK) {
        unsigned ke = (s[p] << 16) | (dbase + 2 * q);             // v_lshl_or_b32
        if constexpr (LOW_ONLY) {
          if (MASK == 2) {
            const int ie = JOFS + 2 * q - p + P - 1;
            ke = sel_lanes(M.at(ie < NM ? ie : NM - 1), ke);
          }
          best[p] = ke < best[p] ? ke : best[p];
          continue;
        }
        unsigned ko = __builtin_amdgcn_perm(s[p], dbase + 2 * q + 1, 0x07060100u);   // (s & 0xffff0000) | d: one v_perm_b32
        if (MASK == 2) {
          const int ie = JOFS + 2 * q - p + P - 1, io = ie + 1;
          ke = sel_lanes(M.at(ie < NM ? ie : NM - 1), ke);
          ko = (JOFS != 0 && 2 * q + 1 >= qn) ? KEY_INVALID : sel_lanes(M.at(io < NM ? io : NM - 1), ko);
        }
        const unsigned m = ke < ko ? ke : ko;
        best[p] = m < best[p] ? m : best[p];                      // v_min3_u32
      } else {
        unsigned k = s[p];
        if (MASK == 2) {
          const int i = JOFS + q - p + P - 1;
          k = sel_lanes(M.at(i < NM ? i : NM - 1), k);
        }
        best[p] = k < best[p] ? k : best[p];
      }
    }
  }

  template <int MASK, int JOFS, int W>
  static __device__ __forceinline__ void reduce_row(const unsigned (&C)[P][W], const Masks& M,
                                                    unsigned (&best)[P], unsigned dbase, int qn)
  {
#pragma unroll
    for (int q = 0; q < W; ++q) {
      Pre cur;
      stage_a(C, q, cur);
      __builtin_amdgcn_sched_barrier(0);
      stage_b<MASK, JOFS, PK && W == 1>(cur, q, M, best, dbase, qn);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // wv0 / dcl0: first wave of this wave's group and the group's first candidate within the pass (0, 0 unless the kernel
  // mixes two chunk sizes: there the waves [nwa, nw) run Cw<P, NDB, ...>::run(a, nwa, nwa*ND) on the SAME ring, row
  // buffers and barriers -- every quantity of the LDS layout below depends on a.slots / a.nw, none on ND)
  // PRIO: issue priority by progress (the row loop's head).  Off in the mixed-chunk kernels: their waves are of two kinds, the
  // heavier chunk's waves are the older ones and the arbiter's oldest-first is what keeps them -- the row step's critical path --
  // ahead; with the priority updates the 12- and 16-wave mixed kernels measured 3-6 % SLOWER (B 16 D 80 46.6 -> 49.6 us, D 112
  // 58.5 -> 62.2; packed B 13 D 80 37.2 -> 38.6), every uniform kernel faster or equal.
  template <bool PRIO = true>
  static __device__ __forceinline__ void run(const CwArgs& a, const int wv0 = 0, const int dcl0 = 0)
  {
    extern __shared__ __attribute__((aligned(16))) unsigned smem[];
    const Job& j = a.j;
    const int nt = blockDim.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dcl = dcl0 + (wv - wv0) * ND;               // first disparity of this wave within the pass
    const int dc = a.dbase + dcl;                         // ... and absolute
    const int RW = TW + a.dpad + 1;                       // R values per ring row (16 bytes' worth aligned)
    const int ROWW = TW + RW;                             // pixels per ring row: L[TW] then R[RW]
    Px* ring = reinterpret_cast<Px*>(smem);               // (NR + 1) rows; row NR stays zero
    unsigned* rowbest = smem + (size_t)(NR + 1) * ROWW * sizeof(Px) / 4;   // [NB][RBW]  (ROWW * sizeof(Px) is a multiple of 16)
    unsigned char* lut = reinterpret_cast<unsigned char*>(rowbest + NB * RBW);  // [maxdisp+1]

    // Grid = (strips x frames, tiles): workgroups are dispatched x-fastest, so ALL workgroups of the
    // leftmost tile (the ones that mask keys in the packed h = 6, 7 variants and run longest) start in
    // the first round instead of trailing the launch.
    const int tile = blockIdx.y + a.tile0;
    const int tx = a.tx0 + tile * STEP;                   // image column of lane 0, p 0
    const int f = blockIdx.x / a.nstrips;
    const int strip = blockIdx.x - f * a.nstrips;
    const uint8_t* __restrict__ Lg = j.left + (size_t)f * j.left_frame_stride;
    const uint8_t* __restrict__ Rg = j.right + (size_t)f * j.right_frame_stride;
    uint8_t* __restrict__ Og = j.out + (size_t)f * j.out_frame_stride;

    const int ys = j.y0 + strip * a.th;                   // strip rows [ys, ye)
    const int ye = min(ys + a.th, j.y1);
    // output columns of this tile (tile 0 also owns everything left of its interior)
    const int ox0 = max(tile == 0 ? j.x0 : tx + OL, j.x0);
    const int ox1 = min(tx + OL + STEP, j.x1);
    if (ys >= ye || ox0 >= ox1) return;
#ifdef SADHIP_STAMPS
    const unsigned long long wg_t0 = a.wgtl ? __builtin_amdgcn_s_memrealtime() : 0ull;
#endif

    // ---- one-time setup: zero row, row-best buffers, output LUT, first ring row ----------------
    // Ring row of image row yy: L columns [tx, tx+TW) then R columns [tx-dbase-dpad, tx-dbase+TW), one u32 per
    // pixel (shifted by ksh in the unpacked variant), zero outside the image.
    for (int i = tid; i < ROWW; i += nt) ring[NR * ROWW + i] = 0;
    for (int i = tid; i < NB * RBW; i += nt) rowbest[i] = KEY_INVALID;
    for (int i = tid; i <= j.maxdisp; i += nt)
      lut[i] = (unsigned char)(j.raw ? min(i, 255) : (i * 255) / j.maxdisp);     // sad.go:91-93
    {
      const int yy = ys - H;                              // step 0's entering row -> slot 0
      for (int i = tid; i < ROWW; i += nt) {                // i = physical position in the ring row
        const bool isR = i >= TW;
        const int col = isR ? tx - a.dbase - a.dpad + logical(i - TW, RW) : tx + logical(i, TW);
        unsigned v = 0u;
        if (yy >= 0 && yy < j.h && col >= 0 && col < j.w)
          v = (unsigned)(isR ? Rg + (size_t)(yy - j.row0) * j.rstride : Lg + (size_t)(yy - j.row0) * j.lstride)[col];
        if constexpr (TR3) {                                 // + the two pixels to the right in bytes 1 and 2
#pragma unroll
          for (int k = 1; k < 3; ++k)
            if (yy >= 0 && yy < j.h && col + k >= 0 && col + k < j.w)
              v |= (unsigned)(isR ? Rg + (size_t)(yy - j.row0) * j.rstride : Lg + (size_t)(yy - j.row0) * j.lstride)[col + k] << (8 * k);
        }
        ring[i] = (Px)(v << a.pixshift);
      }
    }
    __syncthreads();

    // ---- this wave's share of the row streaming and of the output -------------------------------
    // L positions (= output pixels) [wv*shl, (wv+1)*shl), R positions [wv*shr, (wv+1)*shr) of a ring
    // row (physical positions: see SWZ); lane l takes positions l, l+64, ...  lcol / rcol: image column
    // of the pixel stored there, NOT_MINE otherwise.
    // The packed variants fetch rows with raw buffer loads: resource = the row, vector offset = the
    // column, range-checked by the hardware against the row width -- columns left or right of the
    // image (and NOT_MINE, as unsigned) read as 0 without branches or address arithmetic.
    constexpr int NOT_MINE = (int)0x80000000;
    int il0, ir0;
    int lcol[MAXL], rcol[MAXR];
    // The unpacked variants run at their 128-register limit: held across the row loop, these
    // loop-invariant per-lane values were spilled to scratch in the set-up code and reloaded in every
    // step (14-21 VGPRs, 60-88 bytes of scratch per lane in round 1).  There (RECOMPUTE_SHARE) the
    // columns are re-derived in every step from a lane id the compiler cannot see through (one add
    // each) and "this position is mine" is a wave-uniform 64-bit lane mask per position, held in SGPRs
    // and applied as the branch condition itself (s_and_saveexec, no vector instruction).
    constexpr bool RECOMPUTE_SHARE = !PK;
    unsigned long long mineL[MAXL], mineR[MAXR];
    auto row_share = [&](int ln) {                        // ln = lane
      il0 = wv * a.shl + ln; ir0 = wv * a.shr + ln;
#pragma unroll
      for (int v = 0; v < MAXL; ++v) {
        const int i = il0 + 64 * v;
        const int col = tx + logical(i, TW);
        lcol[v] = (RECOMPUTE_SHARE || (ln + 64 * v < a.shl && i < TW)) ? col : NOT_MINE;
      }
#pragma unroll
      for (int v = 0; v < MAXR; ++v) {
        const int i = ir0 + 64 * v;
        const int col = tx - a.dbase - a.dpad + logical(i, RW);
        rcol[v] = (RECOMPUTE_SHARE || (ln + 64 * v < a.shr && i < RW)) ? col : NOT_MINE;
      }
    };
    row_share(lane);
#pragma unroll
    for (int v = 0; v < MAXL; ++v) mineL[v] = __builtin_amdgcn_ballot_w64(lane + 64 * v < a.shl && il0 + 64 * v < TW);
#pragma unroll
    for (int v = 0; v < MAXR; ++v) mineR[v] = __builtin_amdgcn_ballot_w64(lane + 64 * v < a.shr && ir0 + 64 * v < RW);
    // TR3: a pixel triple is one unaligned dword load; a wave that streams a position left of column 0 takes the
    // slower path of load_row / land_row for all its positions (wave-uniform choice, fixed for the workgroup's life)
    bool whole_words = false;
    if constexpr (TR3) {
      bool ok = true;
#pragma unroll
      for (int v = 0; v < MAXL; ++v) ok = ok && (lcol[v] == NOT_MINE || lcol[v] >= 0);      // (<= w-5: see load_row)
#pragma unroll
      for (int v = 0; v < MAXR; ++v) ok = ok && (rcol[v] == NOT_MINE || rcol[v] >= 0);
      whole_words = __builtin_amdgcn_ballot_w64(!ok) == 0ull;
    }
    auto mine_l = [&](int v) -> bool { if constexpr (RECOMPUTE_SHARE) return __builtin_amdgcn_inverse_ballot_w64(mineL[v]); else return lcol[v] != NOT_MINE; };
    auto mine_r = [&](int v) -> bool { if constexpr (RECOMPUTE_SHARE) return __builtin_amdgcn_inverse_ballot_w64(mineR[v]); else return rcol[v] != NOT_MINE; };
    unsigned pfl[MAXL], pfr[MAXR];                        // the row in flight (raw bytes)
    auto load_row = [&](int yy) {
      if (yy >= 0 && yy < j.h) {
        // TR3 && h = 7: the strides are pinned to scalar registers here.  That kernel runs at the SGPR limit and the
        // compiler otherwise keeps the two (sign-extended) strides in VGPRs across the loop: the row pointers are then
        // computed on the vector ALU, the buffer resource looks divergent and EVERY row load becomes a waterfall loop
        // (4 v_readfirstlane + 2 v_cmp + saveexec per load, ~70 vector instructions per row step in all).
        // (The unpacked kernels, at their 128-VGPR / SGPR limits, do the same: pinned, their per-lane byte loads take a
        // scalar row base + a 32-bit lane offset instead of a 64-bit vector address each.)
        constexpr bool PIN_STRIDES = (TR3 && H > 5) || !PK;
        const int ls = PIN_STRIDES ? __builtin_amdgcn_readfirstlane(j.lstride) : j.lstride;
        const int rs = PIN_STRIDES ? __builtin_amdgcn_readfirstlane(j.rstride) : j.rstride;
        const uint8_t* lrow = Lg + (size_t)(yy - j.row0) * ls;
        const uint8_t* rrow = Rg + (size_t)(yy - j.row0) * rs;
        if constexpr (PK) {
          // one resource per row (scalar arithmetic): base = the row, num_records = its width
          const __amdgpu_buffer_rsrc_t rsL = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(lrow), 0, j.w, 0x00020000);
          const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(rrow), 0, j.w, 0x00020000);
          if constexpr (TR3) {
            // one (unaligned) dword per pixel triple.  Workgroups in this form never reach the row's right end:
            // their last lane column is <= w-3 (k_colwave) and == w-1 (mod 4) like every tile's (tile origins
            // == w mod 4), i.e. <= w-5, so the dword of every used position ends at column w-2 or before (the R
            // part's spare last position may lie further right: it is never an operand).
            // At the left end a dword that starts at column -1 or -2 is partly outside and would
            // read as 0 whole, while it holds the pixels 0, 1: a wave that streams such positions loads the
            // dword of column 0 for them and shifts when it lands it (columns <= -3: all outside, 0 is right).
#pragma unroll
            for (int v = 0; v < MAXL; ++v)
              pfl[v] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rsL, whole_words ? lcol[v] : max(lcol[v], min(lcol[v] + 2, 0)), 0, 0);
#pragma unroll
            for (int v = 0; v < MAXR; ++v)
              pfr[v] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rsR, whole_words ? rcol[v] : max(rcol[v], min(rcol[v] + 2, 0)), 0, 0);
          } else {
#pragma unroll
            for (int v = 0; v < MAXL; ++v) pfl[v] = __builtin_amdgcn_raw_buffer_load_b8(rsL, lcol[v], 0, 0);
#pragma unroll
            for (int v = 0; v < MAXR; ++v) pfr[v] = __builtin_amdgcn_raw_buffer_load_b8(rsR, rcol[v], 0, 0);
          }
        } else {
          // the unpacked variants run at their register limit; there the plain loads allocate better
          // (B16 D64: 35.4 us against 39.9 with buffer loads)
#pragma unroll
          for (int v = 0; v < MAXL; ++v) pfl[v] = (unsigned)lcol[v] < (unsigned)j.w ? (unsigned)lrow[(unsigned)lcol[v]] : 0u;
#pragma unroll
          for (int v = 0; v < MAXR; ++v) pfr[v] = (unsigned)rcol[v] < (unsigned)j.w ? (unsigned)rrow[(unsigned)rcol[v]] : 0u;
        }
      } else {
#pragma unroll
        for (int v = 0; v < MAXL; ++v) pfl[v] = 0u;
#pragma unroll
        for (int v = 0; v < MAXR; ++v) pfr[v] = 0u;
      }
    };
    auto land_row = [&](Px* dst) {
      if constexpr (TR3) {
        // the triple of column c from the dword loaded at column ld >= c: moved up by ld - c (0..2) bytes
        auto place = [&](unsigned dw, int col) -> unsigned {
          if (whole_words) return dw & 0xffffffu;
          const int up = max(col, min(col + 2, 0)) - col;
          return (dw << (8 * up)) & 0xffffffu;
        };
#pragma unroll
        for (int v = 0; v < MAXL; ++v)
          if (mine_l(v)) dst[il0 + 64 * v] = (Px)place(pfl[v], lcol[v]);
#pragma unroll
        for (int v = 0; v < MAXR; ++v)
          if (mine_r(v)) dst[TW + ir0 + 64 * v] = (Px)place(pfr[v], rcol[v]);
        return;
      }
#pragma unroll
      for (int v = 0; v < MAXL; ++v)
        if (mine_l(v)) dst[il0 + 64 * v] = (Px)(PK ? pfl[v] : pfl[v] << a.pixshift);
#pragma unroll
      for (int v = 0; v < MAXR; ++v)
        if (mine_r(v)) dst[TW + ir0 + 64 * v] = (Px)(PK ? pfr[v] : pfr[v] << a.pixshift);
    };
    const unsigned kmask = PK ? 0xffffu : (1u << a.ksh) - 1u, gmask = PK ? 0xffffu : (1u << a.kshg) - 1u;   // (packed: 16-bit field, a constant)
    auto finalize = [&](int t) {                          // output row of step t, this wave's pixels
      const int y = ys + t - 2 * H;
      unsigned* rb = rowbest + (t & 1) * RBW;
      uint8_t* orow = Og + (size_t)(y - j.oy) * j.ostride;  // uniform base + 32-bit lane offset
      const size_t krow = ((size_t)f * (j.y1 - j.y0) + (y - j.y0)) * a.kw;
#pragma unroll
      for (int v = 0; v < MAXL; ++v) {
        if (mine_l(v)) {
          const int x = lcol[v], i = x - tx;                // i = column within the tile (>= 0)
          const int ri = (i & (P - 1)) * RBP + (i >> (P == 8 ? 3 : 2));
          unsigned k = rb[ri];
          rb[ri] = KEY_INVALID;
          if (x >= ox0 && x < ox1) {
            if constexpr (!PK) {
              // unpacked: the pass's key (SAD << ksh) | (d - dbase) becomes the travelling key (SAD << kshg) | d; a key
              // with a penalty in it (a candidate that does not exist for this pixel: >= 2^27) becomes "no candidate"
              k = k >= (1u << 27) ? KEY_INVALID : (((k >> a.ksh) << a.kshg) | ((k & kmask) + (unsigned)a.dbase));
            }
            if (a.kin) k = min(k, a.kin[krow + (unsigned)(x - j.x0)]);
            if (a.kout) {
              a.kout[krow + (unsigned)(x - j.x0)] = k;
            } else {
              // x <= h: d = 0 is the only candidate (sad.go:66-69)
              const unsigned dk = k & gmask;
              const unsigned d = x <= H ? 0u : min(dk, (unsigned)j.maxdisp);
              orow[(unsigned)(x - j.ox)] = lut[d];
            }
          }
        }
      }
    };

    // ---- per-lane state ------------------------------------------------------------------------
    const int x0 = tx + P * lane;                         // first column of this lane
    const bool lane_valid = x0 < j.w;                     // (w - tx) % P == 0: all-or-nothing
    const bool xtra = wv == a.xw;                         // this wave also owns candidate dx = the pass's slot count
    const int dx = a.dbase + a.slots;
    // Candidates d > x-h do not exist near the left image edge (sad.go:66-69).  Instead of masking
    // their keys in every row (which made the leftmost tile's workgroups the critical path of the
    // whole launch), their column sums start with a penalty that outlives the sliding updates:
    // a pixel's candidate d is missing exactly when its window's first column c0 = x-h is < d.
    //   unpacked: every column c < d starts at 2^27 (17 columns * 2^27 + any key < 2^32);
    //   packed:   u16 sums cannot hold more than one penalty per window, so only columns
    //             c < d with c == d-1 (mod 2h+1) start at PEN = (2h+1)^2*255: every full window
    //             holds exactly one column of that class, and it is < d iff c0 < d; a penalised sum
    //             is >= the worst d = 0 sum and <= 2*PEN, which fits 16 bits for h <= 5.
    // Windows clipped on the left (x <= h) only have d = 0 and are forced to lut[0] at the output;
    // packed windows clipped on the right can meet missing candidates only when w <= D + 2h + 2:
    // plan_job() gives those frames the unpacked variant.  Packed h = 6, 7 (NEED_MASK) masks
    // keys instead: candidate j of the wave is valid for column p of lane l iff
    // tx + P*l + p - h >= dc + j, i.e. l >= ceil((dc + h - tx + (j - p)) / P): one 64-bit lane mask
    // per value of j - p, kept in SGPRs and applied with one v_cndmask_b32 per key.
    constexpr unsigned PEN = PK ? (unsigned)(B1 * B1 * 255) : (1u << 27);
    constexpr bool use_pen = !(PK && H > 5);          // (packed h = 6, 7: key masks, or -- triple form -- nothing to remove)
    // Lane masks (Masks above): NMB distinct 64-bit masks in SGPR pairs + the carry bits, held across the
    // row loop; the mask of a key is a scalar select at its use.  (Measured alternatives, DESIGN.md
    // section 6b: all ND + P masks ready-made cost 32-40 SGPRs, 13-25 of them parked in VGPR lanes; a
    // vector compare per key instead of a mask costs 8 % on config 5.)
    Masks M;
    {
      const int T0 = dc + H - tx;                         // mask k: lanes >= max(0, floor((T0 + k) / P))
      const int q0 = T0 >= 0 ? T0 / P : -((-T0 + P - 1) / P);
      const int c = T0 - q0 * P;                          // 0 .. P-1
#pragma unroll
      for (int i = 0; i < NMB; ++i) {
        const int l0 = q0 + i > 0 ? q0 + i : 0;
        M.mb[i] = (NEED_MASK && l0 < 64) ? (~0ull << l0) : 0ull;
      }
      M.carry = 0u;
#pragma unroll
      for (int r = 0; r < P; ++r) M.carry |= (r + c >= P) ? (1u << r) : 0u;
      M.carry = (unsigned)__builtin_amdgcn_readfirstlane((int)M.carry);
    }
    // Candidates d > D (a chunk that runs past MaxDisparity) exist for no pixel: same trick with the
    // class c == w-1 (mod 2h+1), 0 <= c < w, which every window has exactly once -- also the windows
    // clipped at the right image edge (they end in column w-1).  Columns >= w must start at 0: they
    // never accumulate and their neighbours' windows sum over them.
    auto init_sum = [&](int col, int d) -> unsigned {     // start value of the column sum of (col, d)
      const bool in = col >= 0 && col < j.w;
      if constexpr (PK) {
        if (!use_pen || !in) return 0u;
        // TR3: the register of column c sums the columns c, c+1, c+2, and a window is the registers of
        // c0, c0+3, ... (c0 = x-h): the penalised class is three consecutive columns of every 2h+1, so that
        // again exactly one register of a window is in it (the residues of c0, c0+3, ... mod 2h+1 fall into
        // the (2h+1)/3 different thirds) -- and it is < d whenever c0 is.
        constexpr int CW = TR3 ? 3 : 1;
        if (d > j.maxdisp) return (j.w - 1 - col) % B1 < CW ? PEN : 0u;
        return (col < d && (d - 1 - col) % B1 < CW) ? PEN : 0u;
      } else {
        const bool biased = ((unsigned)(col + B1 * 4096) % (unsigned)B1) == 0u;
        return (biased ? (unsigned)(d - a.dbase) : 0u) + ((in && (col < d || d > j.maxdisp)) ? PEN : 0u);   // the key's field is d - dbase
      }
    };
    unsigned C[P][NC], CX[P][1];
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int col = x0 + p;
#pragma unroll
      for (int q = 0; q < NC; ++q) {
        if constexpr (PK) C[p][q] = init_sum(col, dc + 2 * q) | (init_sum(col, dc + 2 * q + 1) << 16);
        else C[p][q] = init_sum(col, dc + q);
      }
      // the extra register's high half is candidate dx+1 > D: never updated, starts penalised
      if constexpr (PK) CX[p][0] = init_sum(col, dx) | (init_sum(col, dx + 1) << 16);
      else CX[p][0] = init_sum(col, dx);
    }
    // packed h = 6, 7: does this wave have to mask keys?  (wave-uniform)
    const int qn = min(ND, j.maxdisp - dc + 1);           // chunk runs past MaxDisparity: first qn only
    // some pixel of this tile has d <= x-h < chunk end (exact plans: qn == ND)
    const bool key_mask = !use_pen && max(ox0, 0) - H < min(dc + ND - 1 + (xtra ? 1 : 0), j.maxdisp);

    // this lane's operands inside a ring row (pixel index): L at lofs, R value m at rofs + m
    // (SWZ: logical quad 2*lane + c + j of a part lives at quad lane + (c + j) / 2 of half (c + j) & 1;
    //  c = (dpad - dcl - (ND-1)) / 4 = 2 * (nw - wv) is even, so even j read half 0 and odd j half 1)
    const int lofs = SWZ ? 4 * lane : P * lane;
    const int rofs = TW + (SWZ ? 4 * (a.nw - wv + lane) : (a.dpad - dcl - (ND - 1)) + P * lane);   // 16-byte aligned by construction (dpad = nw*ND + XOFF)
    const int rxofs = TW + (SWZ ? 4 * lane : (a.dpad - a.slots - XOFF) + P * lane);               // extra candidate: R value of column p at +XOFF+p
    const int lhalf = SWZ ? TW / 2 : 0, rhalf = SWZ ? RW / 2 : 0;

    // ---- main loop: step t brings in image row ys-H+t (ring slot t % NR) ------------------------
    //   load    this wave's share of row t+1 (global -> registers; lands after the arithmetic)
    //   update  column sums: + row t, - row t-(2h+1)
    //   reduce  window sums -> per-pixel minimum of this wave's candidates -> ds_min into row buffer t&1
    //   land    row t+1 in the slot of the row that left the window one step ago
    //   output  this wave's pixels of row buffer (t-1)&1 (complete since the last barrier)
    //   barrier
    const int nsteps = (ye - ys) + 2 * H;
    int slot_in = 0;
#ifdef SADHIP_STAMPS
    unsigned long long st_sum[5] = {0, 0, 0, 0, 0}, st_prev = 0;
#define SADHIP_STAMP(K) do { if (a.stamps) { __builtin_amdgcn_sched_barrier(0); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
                             st_sum[K] += now_ - st_prev; st_prev = now_; __builtin_amdgcn_sched_barrier(0); } } while (0)
#define SADHIP_STAMP0() do { if (a.stamps) st_prev = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SADHIP_STAMP(K) do {} while (0)
#define SADHIP_STAMP0() do {} while (0)
#endif
    asm volatile("; SADHIP_ROWLOOP form=%c0 (tools/valu_budget.py: 3 = triple-column, 1 = plain)" :: "n"(TR3 ? 3 : 1));
    for (int t = 0; t < nsteps; ++t) {
      // Issue priority by progress: a workgroup's waves issue with priority 3 in the first quarter of its strip, 2 / 1 / 0 in the
      // others (s_setprio, once per 8 row steps).  The arbiter takes the highest priority first and the OLDEST wave among equals:
      // left alone, the older of the workgroups that share a CU wins every tie -- measured with the workgroup timeline of the stamps
      // build (profiles/r05_wg_timeline.txt): the first-dispatched workgroups of a launch finish in 0.6-0.7 of the mean duration, the
      // later ones in 1.2 -- so the workgroups of a launch end spread over a whole duration, a launch of k rounds ends with 0.4 of a
      // workgroup's duration of emptying chip (config 5 with 16 pairs: 12 % of the slot-time; 1080p B 9 D 64 with 128 pairs: 10 %;
      // the 16-wave kernels, one workgroup per CU, 1 %), and short launches pay it in full.  With progress as priority the
      // workgroups of a CU stay within a quarter strip of each other and the rounds of a launch end together: 1080p B 9 D 64 with
      // 16 / 48 / 128 pairs -8 / -4.5 / -3.5 %, a single pair -5 %, the defaults 16 / 64 with 16 / 64 pairs -8 / -3 %, B 15, B 31, B 3,
      // 640x480 -6 ... -8 %, config 5 -1 ... -2 %, launches of 32 rounds and the one-per-CU kernels unchanged
      // (profiles/r05_ab_prio.txt).  Results do not depend on it.
      // (quarters of the strip's own step count, from the loop bound: no argument, no register held for it)
#ifdef SADHIP_TUNING
      if (!a.prio_off)
#endif
      if (PRIO && (t & 7) == 0) {
        const int half = nsteps >> 1, quarter = nsteps >> 2;
        if (t < quarter) __builtin_amdgcn_s_setprio(3);
        else if (t < half) __builtin_amdgcn_s_setprio(2);
        else if (t < half + quarter) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
      }
      if constexpr (RECOMPUTE_SHARE) {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        row_share(ln);
      }
      SADHIP_STAMP0();
      // (also in the last step: one harmless row more keeps loads and landings unconditional)
      load_row(ys - H + t + 1);
      const int so = slot_in + 2;                         // row t-(2h+1): NR - (2h+1) = 2 slots ahead
      const int slot_out = t > 2 * H ? (so >= NR ? so - NR : so) : NR;   // window not full yet: zero row
      const int sn = slot_in + 1 == NR ? 0 : slot_in + 1;
      const Px* rin = ring + slot_in * ROWW;
      const Px* rout = ring + slot_out * ROWW;
      unsigned best[P];
#pragma unroll
      for (int p = 0; p < P; ++p) best[p] = KEY_INVALID;
      const bool out_row = t >= 2 * H;
      const bool leave = t > 2 * H;                       // a row leaves the window (before: slot_out is the row of zeros)
      if (lane_valid) update_columns(C, Src{rin + lofs, rin + lofs + lhalf}, Src{rin + rofs, rin + rofs + rhalf},
                                     Src{rout + lofs, rout + lofs + lhalf}, Src{rout + rofs, rout + rofs + rhalf}, leave);
      SADHIP_STAMP(0);
      if (out_row) {
        if (NEED_MASK && key_mask) {                 // leftmost tiles only
          asm volatile("; SADHIP_MASKED_BEGIN (tools/valu_budget.py: not on the common path)");
          reduce_row<2, 0>(C, M, best, (unsigned)dc, qn);
          asm volatile("; SADHIP_MASKED_END");
        }
        else reduce_row<0, 0>(C, M, best, (unsigned)dc, qn);
      }
      if (xtra) {                                         // the odd candidate dx (wave-uniform branch)
        asm volatile("; SADHIP_XTRA_BEGIN (tools/valu_budget.py counts this path separately)");
        __builtin_amdgcn_sched_barrier(0);
        if (lane_valid)
          extra_update(CX, Src{rin + lofs, rin + lofs + lhalf}, Src{rin + rxofs, rin + rxofs + rhalf},
                       Src{rout + lofs, rout + lofs + lhalf}, Src{rout + rxofs, rout + rxofs + rhalf}, leave);
        if (out_row) {
          if constexpr (NEED_MASK) reduce_row<2, ND>(CX, M, best, (unsigned)dx, 1);   // also drops the empty high half
          else reduce_row<0, 0>(CX, M, best, (unsigned)dx, 1);
        }
        asm volatile("; SADHIP_XTRA_END");
      }
      if (out_row) {
        unsigned* rb = rowbest + (t & 1) * RBW + lane;
#pragma unroll
        for (int p = 0; p < P; ++p) atomicMin(rb + p * RBP, best[p]); // ds_min_u32
      }
      // One unconditional wait for the row loads (issued a whole update + reduce ago).  Left to
      // itself hipcc waits inside each lane-masked store block of land_row and, unable to prove that
      // one of them always runs, again at the loop head -- where vmcnt(0) also waits for the output
      // bytes stored at the end of the previous step (a memory round trip per row and wave).
      SADHIP_STAMP(1);
      __builtin_amdgcn_s_waitcnt(0x0F70);                 // vmcnt(0); expcnt / lgkmcnt untouched
      land_row(ring + sn * ROWW);
      SADHIP_STAMP(2);
      if (t - 1 >= 2 * H) finalize(t - 1);
      SADHIP_STAMP(3);
      wg_barrier();
      SADHIP_STAMP(4);
      slot_in = sn;
    }
    finalize(nsteps - 1);
#ifdef SADHIP_STAMPS
    if (a.wgtl && tid == 0) {
      unsigned long long* o = a.wgtl + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4;
      o[0] = wg_t0; o[1] = __builtin_amdgcn_s_memrealtime();
      o[2] = (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_REG_HW_ID, all 32 bits
      o[3] = (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 20);     // HW_REG_XCC_ID
    }
    if (a.stamps && lane == 0) {
      unsigned long long* o = a.stamps + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 16 + wv) * 8;
      for (int k = 0; k < 5; ++k) o[k] = st_sum[k];
      o[5] = (unsigned long long)nsteps;
      o[6] = xtra ? 1 : 0;
      o[7] = (unsigned long long)tile;
    }
#endif
  }
};


// MIX (round 4): a workgroup of nwa waves with ND candidates each followed by nw - nwa waves with a smaller chunk NDB, for
// disparity ranges that are no multiple of ND per SIMD -- unpacked on the u16 ring 8 + 4: 49 = 4*8 + 4*4 + 1, 81 = 8*8 + 4*4 + 1,
// 113 = 12*8 + 4*4 + 1 (the UI's MaxDisparity 48, 80, 112); packed P = 4 (ND, ND - 4): 81 = 4*12 + 4*8 + 1, 113 = 4*16 + 4*12 + 1,
// 145 = 4*20 + 4*16 + 1, 177 = 4*24 + 4*20 + 1.  The dispatcher places the waves of a workgroup on the four SIMDs round-robin
// (tools/ubench/simd_map.hip), so with nwa and nw - nwa multiples of 4 every SIMD gets the same mix and no SIMD paces
// the row barrier; uniform plans of these ranges either leave a quarter of the chip idle (7 waves of 8) or need 12-wave
// workgroups of 4 (measured 10 % slower than 7 x 8).  Both groups run on one ring, one pair of row buffers and the same
// barriers; the groups differ only in their column-sum registers and operand offsets.
template <int P, int ND, bool PK> struct MixChunk { static constexpr int NDB = PK ? ND - 4 : ND / 2; };

// One wave group of a kernel: picks the form (TR: per workgroup, see k_colwave) and runs it.
template <int P, int ND, int H, bool PK, bool W16, bool TR, bool PRIO = true>
__device__ __forceinline__ void run_group(const CwArgs& a, int wv0, int dcl0)
{
  if constexpr (TR) {
    // per workgroup: the tile with the image's last two columns runs the plain form (Cw::TR3), and for h = 6, 7 so does
    // every tile with an output pixel x whose window starts left of a candidate of this pass (x - h < d: the plain form
    // masks those keys; the triple form carries no masks)
    using K = Cw<P, ND, H, PK, W16, true>;
    const int tile = (int)blockIdx.y + a.tile0;
    const int tx = a.tx0 + tile * K::STEP;
    bool triple = tx + K::TW + 2 <= a.j.w;
    if constexpr (PK && H > 5) {
      const int ox0 = max(tile == 0 ? a.j.x0 : tx + K::OL, a.j.x0);
      triple = triple && !(max(ox0, 0) - H < min(a.dbase + a.slots, a.j.maxdisp));
    }
    if (triple) K::template run<PRIO>(a, wv0, dcl0); else Cw<P, ND, H, PK, W16, false>::template run<PRIO>(a, wv0, dcl0);
  } else {
    Cw<P, ND, H, PK, W16, false>::template run<PRIO>(a, wv0, dcl0);
  }
}

// WB = workgroup-size bound (VGPR budget = 512 / (WB / 256) per lane).
// TR: the kernel holds both forms of the row loop (Cw::TR3): the triple-column form for the workgroups whose tile does
// not hold the image's last two columns, the plain form for the others.
template <int P, int ND, int H, bool PK, int WB, bool W16, bool TR = false, bool MIX = false>
__global__ void __launch_bounds__(WB) k_colwave(CwArgs a)
{
  if constexpr (MIX) {
    static_assert((W16 && !PK && ND == 8 && !TR) || (PK && P == 4 && !W16 && ND >= 8),
                  "mixed chunks: unpacked on the u16 ring (8 + 4), or packed P = 4 (ND, ND - 4)");
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (wv < a.nwa) run_group<P, ND, H, PK, W16, TR, false>(a, 0, 0);
    else run_group<P, MixChunk<P, ND, PK>::NDB, H, PK, W16, TR, false>(a, a.nwa, a.nwa * ND);
  } else {
    run_group<P, ND, H, PK, W16, TR>(a, 0, 0);
  }
}

}  // namespace

}  // namespace sadhip
