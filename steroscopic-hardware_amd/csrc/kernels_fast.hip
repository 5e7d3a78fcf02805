// kernels_fast.hip -- "column-wave" SAD disparity kernel for gfx950 (MI355X).
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
//   * lane l of a wave owns P (4 or 8) adjacent columns; a compute wave owns ND consecutive
//     disparities for its 64*P columns; the NW compute waves of a workgroup split the disparity
//     range of one tile; one more wave (the helper) streams rows and writes output.
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
//   * compute waves merge their per-pixel minima with ds_min_u32 into a row buffer in LDS; after
//     the one barrier per row the helper wave maps key -> d -> best*255/D (LUT) and stores u8.
//   * image rows stream through an LDS ring (2h+3 rows, one u32 per pixel), loaded by the helper
//     two rows ahead.  ND is a multiple of 4 and the R part of a ring row is offset so that every
//     lane's operands are 16-byte aligned: all operand reads are ds_read_b128, bank-conflict free
//     at a lane stride of 16 bytes (ds_read_b32 at that stride was a 4-way conflict: 72 % of LDS
//     cycles in the first version).
#include "sadhip_kernels.h"
#include <cstdlib>

namespace sadhip {

namespace {

struct CwArgs {
  Job j;
  int nw;          // waves per workgroup (disparity chunks)
  int ksh;         // bits of the disparity field of a key (unpacked: 2^ksh > maxdisp; packed: 16)
  int pixshift;    // left shift applied to pixels landing in LDS (unpacked: ksh; packed: 0)
  int dpad;        // columns of R kept left of the tile: nw*ND - 1 (== 3 mod 4)
  int tx0;         // origin of tile 0 (<= 0, == w mod P)
  int tile0;       // first tile index launched (blockIdx.x + tile0)
  int th;          // output rows per strip
  int dbg;         // tuning aid (SADHIP_TUNE_DBG): 1 skip column update, 2 skip row reduce, 4 skip finalize, 8 skip prefetch
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
__device__ __forceinline__ unsigned from_left(unsigned v)     // lane l <- lane l-1 (lane 0: 0)
{
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, true);    // wave_shr:1 bound_ctrl:0
}
__device__ __forceinline__ unsigned from_right(unsigned v)    // lane l <- lane l+1 (lane 63: 0)
{
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, true);    // wave_shl:1 bound_ctrl:0
}

// Workgroup barrier that waits for this wave's LDS operations only.  __syncthreads() also emits
// s_waitcnt vmcnt(0): the helper wave would then sit at every row barrier until the global loads
// it has just issued for a LATER row have returned (a full memory latency per row, with every
// compute wave waiting behind it).  The loads are consumed a step later; hipcc places the
// vmcnt wait at that use by itself.  "memory" keeps the compiler from moving LDS accesses across.
__device__ __forceinline__ void wg_barrier()
{
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int P, int ND, int H, bool PK>
struct Cw {                                         // P columns per lane: 4 (h <= 4) or 8 (h <= 8)
  // PK: column sums of disparity pairs packed as u16 halves (window SAD < 65536, i.e. h <= 7)
  static_assert(!PK || (H <= 7 && ND % 4 == 0), "packed sums need (2h+1)^2 * 255 < 65536");
  static constexpr int NC = PK ? ND / 2 : ND;       // accumulator registers per column
  static_assert(P == 4 || P == 8, "operand reads are whole ds_read_b128");
  static_assert(H >= 1 && H <= P, "halo must fit in the neighbour lane");
  static_assert(ND % 4 == 0, "ND must keep every lane's R operands 16-byte aligned");
  static constexpr int TW = 64 * P;                 // columns per tile
  static constexpr int OL = P * ((H + P - 1) / P);  // first output column inside the tile
  static constexpr int STEP = TW - 2 * OL;          // output columns per tile
  static constexpr int B1 = 2 * H + 1;              // window width
  static constexpr int RB = 1;                      // rows per barrier (a "super-step")
  static constexpr int NB = 2 * RB;                 // row-best buffers: RB being filled + RB being drained
  static constexpr int NR = 2 * H + 1 + 2 * RB;     // ring rows: window + leaving rows of a super-step + RB landing
  static constexpr int NRV = P + ND;                // R values a lane loads per row (last one unused)

  // ---- vertical slide: C += |L-R|(entering row) ; C -= |L-R|(leaving row) -----------------------
  // lin/lout: this lane's 4 L values; rin/rout: its NRV R values, value m = column x0 - (dc+ND-1) + m.
  static __device__ __forceinline__ void update_columns(unsigned (&C)[P][NC],
                                                        const uint4* lin, const uint4* rin,
                                                        const uint4* lout, const uint4* rout)
  {
    {
      unsigned li[P], ri[NRV];
#pragma unroll
      for (int m = 0; m < P / 4; ++m) {
        const uint4 l4 = lin[m];
        li[4 * m] = l4.x; li[4 * m + 1] = l4.y; li[4 * m + 2] = l4.z; li[4 * m + 3] = l4.w;
      }
#pragma unroll
      for (int m = 0; m < NRV / 4; ++m) {
        const uint4 r4 = rin[m];
        ri[4 * m] = r4.x; ri[4 * m + 1] = r4.y; ri[4 * m + 2] = r4.z; ri[4 * m + 3] = r4.w;
      }
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
    {
      unsigned lo[P], ro[NRV];
#pragma unroll
      for (int m = 0; m < P / 4; ++m) {
        const uint4 l4 = lout[m];
        lo[4 * m] = l4.x; lo[4 * m + 1] = l4.y; lo[4 * m + 2] = l4.z; lo[4 * m + 3] = l4.w;
      }
#pragma unroll
      for (int m = 0; m < NRV / 4; ++m) {
        const uint4 r4 = rout[m];
        ro[4 * m] = r4.x; ro[4 * m + 1] = r4.y; ro[4 * m + 2] = r4.z; ro[4 * m + 3] = r4.w;
      }
      if constexpr (PK) {
#pragma unroll
        for (int p = 0; p < P; ++p)
#pragma unroll
          for (int r = 0; r < NC; r += 2)
            pk_sad_out2(C[p][r], C[p][r + 1], lo[p], ro[p + ND - 1 - 2 * r], ro[p + ND - 2 - 2 * r],
                        ro[p + ND - 3 - 2 * r], ro[p + ND - 4 - 2 * r]);
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

  // ---- horizontal window sums + key minimum for one row --------------------------------------
  // Per disparity q: stage A = in-lane prefix sums ps[] and the suffix sums the right neighbour
  // needs; stage B = window sums (DPP adds pulling from both neighbour lanes) folded into best[].
  // A DPP operand must be written >= 2 instructions before the DPP op reads it, or the compiler
  // pads with s_nop (a full issue slot each here), so the stages are software-pipelined:
  // A(q+1) is issued between A(q) and B(q), pinned with sched_barrier.
  struct Pre { unsigned ps[P + 1]; unsigned suf[P + 1]; };   // suf[k] = sum of the lane's last k columns

  static __device__ __forceinline__ void stage_a(const unsigned (&C)[P][NC], int q, Pre& r)
  {
    r.ps[0] = 0;
#pragma unroll
    for (int k = 0; k < P; ++k) r.ps[k + 1] = r.ps[k] + C[k][q];
#pragma unroll
    for (int k = 1; k <= H && k <= P; ++k) r.suf[k] = r.ps[P] - r.ps[P - k];
  }

  // Unpacked: the window sum is the key (bias trick).  Packed: the window sums of disparities
  // dbase+2q (low half) and dbase+2q+1 (high half) become two keys (SAD << 16) | d.
  // MASK: 0 = every candidate of the chunk is valid for every pixel; 1 = the chunk runs past
  // MaxDisparity (wave-uniform: only the first qn candidates exist); 2 = per-pixel limits (the
  // leftmost columns of the image, where d <= x-h cuts the range).
  template <int MASK>
  static __device__ __forceinline__ void stage_b(const Pre& r, int q, const int (&jlim)[P], unsigned (&best)[P],
                                                 unsigned dbase, int qn)
  {
    // window sum of column p = own columns + last (h-p) columns of the left lane + first (p+h-P+1)
    // columns of the right lane.  The DPP adds are issued as two sweeps over p (all "left" pulls,
    // then all "right" pulls): a DPP instruction that reads a VGPR written by the instruction just
    // before it costs wait states, so the two adds of one column are kept P instructions apart.
    unsigned s[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int a = p - H, b = p + H;
      const int lo = a > 0 ? a : 0, hi = b < P - 1 ? b : P - 1;
      s[p] = lo == 0 ? r.ps[hi + 1] : r.ps[hi + 1] - r.ps[lo];
      if (a < 0) s[p] += from_left(r.suf[-a]);
    }
#pragma unroll
    for (int p = 0; p < P; ++p) asm("" : "+v"(s[p]));        // v_add_u32_dpp twice, not v_mov_dpp x2 + v_add3
#pragma unroll
    for (int p = 0; p < P; ++p)
      if (p + H > P - 1) s[p] += from_right(r.ps[p + H - P + 1]);
#pragma unroll
    for (int p = 0; p < P; ++p) {
      if constexpr (PK) {
        unsigned ke = (s[p] << 16) | (dbase + 2 * q);             // v_lshl_or_b32
        unsigned ko = __builtin_amdgcn_perm(s[p], dbase + 2 * q + 1, 0x07060100u);   // (s & 0xffff0000) | d: one v_perm_b32
        if (MASK == 2) {
          ke = (2 * q > jlim[p]) ? KEY_INVALID : ke;
          ko = (2 * q + 1 > jlim[p]) ? KEY_INVALID : ko;
        }
        if (MASK == 1 && 2 * q + 1 >= qn) ko = KEY_INVALID;       // wave-uniform: scalar branch
        const unsigned m = ke < ko ? ke : ko;
        best[p] = m < best[p] ? m : best[p];                      // v_min3_u32
      } else {
        unsigned k = s[p];
        if (MASK == 2) k = (q > jlim[p]) ? KEY_INVALID : k;       // d > min(D, x-h): not a candidate
        best[p] = k < best[p] ? k : best[p];
      }
    }
  }

  template <int MASK>
  static __device__ __forceinline__ void reduce_row(const unsigned (&C)[P][NC], const int (&jlim)[P],
                                                    unsigned (&best)[P], unsigned dbase, int qn)
  {
    constexpr int CPR = PK ? 2 : 1;                            // candidates per accumulator register
    Pre cur, nxt;
    stage_a(C, 0, cur);
#pragma unroll
    for (int q = 0; q < NC; ++q) {
      if (MASK != 1 || CPR * q < qn) {                         // else: the rest of the chunk does not exist
        if (q + 1 < NC) stage_a(C, q + 1, nxt);
        __builtin_amdgcn_sched_barrier(0);
        stage_b<MASK>(cur, q, jlim, best, dbase, qn);
        __builtin_amdgcn_sched_barrier(0);
        cur = nxt;
      }
    }
  }

  static __device__ void run(const CwArgs& a)
  {
    extern __shared__ __attribute__((aligned(16))) unsigned smem[];
    const Job& j = a.j;
    const int nt = blockDim.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dc = wv * ND;                               // first disparity of this wave
    const int RW = TW + a.dpad + 1;                       // R values per ring row (multiple of 4)
    const int ROWW = TW + RW;                             // u32 per ring row: L[TW] then R[RW]
    unsigned* ring = smem;                                // (NR + 1) rows; row NR stays zero
    unsigned* rowbest = smem + (NR + 1) * ROWW;           // [NB][TW]
    unsigned char* lut = reinterpret_cast<unsigned char*>(rowbest + NB * TW);  // [maxdisp+1]

    const int tile = blockIdx.x + a.tile0;
    const int tx = a.tx0 + tile * STEP;                   // image column of lane 0, p 0
    const int f = blockIdx.z;
    const uint8_t* __restrict__ Lg = j.left + (size_t)f * j.left_frame_stride;
    const uint8_t* __restrict__ Rg = j.right + (size_t)f * j.right_frame_stride;
    uint8_t* __restrict__ Og = j.out + (size_t)f * j.out_frame_stride;

    const int ys = j.y0 + blockIdx.y * a.th;              // strip rows [ys, ye)
    const int ye = min(ys + a.th, j.y1);
    // output columns of this tile (tile 0 also owns everything left of its interior)
    const int ox0 = max(tile == 0 ? j.x0 : tx + OL, j.x0);
    const int ox1 = min(tx + OL + STEP, j.x1);
    if (ys >= ye || ox0 >= ox1) return;

    // ---- one-time setup: zero row, row-best buffers, output LUT ------------------------------
    for (int i = tid; i < ROWW; i += nt) ring[NR * ROWW + i] = 0;
    for (int i = tid; i < NB * TW; i += nt) rowbest[i] = KEY_INVALID;
    for (int i = tid; i <= j.maxdisp; i += nt)
      lut[i] = (unsigned char)(j.raw ? min(i, 255) : (i * 255) / j.maxdisp);     // sad.go:91-93

    // ---- ring rows: L columns [tx, tx+TW) then R columns [tx-dpad, tx+TW), u32 pixels shifted by
    //      ksh, zero outside the image.
    const int nvals = ROWW;
    auto fetch = [&](int yy, int i) -> unsigned {         // element i of the ring row of image row yy
      const bool isR = i >= TW;
      const int col = isR ? tx - a.dpad + (i - TW) : tx + i;
      if (yy < 0 || yy >= j.h || col < 0 || col >= j.w) return 0u;
      const uint8_t* base = isR ? Rg + (size_t)(yy - j.row0) * j.rstride : Lg + (size_t)(yy - j.row0) * j.lstride;
      return (unsigned)base[col];
    };
    // Step t handles the entering row ys-H+t, which lives in ring slot t % NR.  RB steps form a
    // super-step with ONE barrier (a barrier per row cost 28 % in tools/ubench/block_bench.hip:
    // six waves on four SIMDs never finish a row together).
    for (int k = 0; k < RB; ++k) {                        // rows of super-step 0 resident before the loop
      unsigned* dst = ring + k * ROWW;
      for (int i = tid; i < nvals; i += nt) dst[i] = fetch(ys - H + k, i) << a.pixshift;
    }
    __syncthreads();

    const int nsteps = (ye - ys) + 2 * H;
    // Roles: waves [0, nw) compute; wave nw is the helper that streams rows into the ring and
    // turns finished row buffers into output, so the compute waves' loop is VALU + LDS reads only.
    // Every wave executes exactly one barrier per super-step.
    if (wv == a.nw) {
      if (a.dbg & 64) return;                             // tuning aid (with 16): no helper at all
      helper_loop(a, ring, rowbest, lut, Lg, Rg, Og, ROWW, nvals, tx, ys, nsteps, ox0, ox1, lane);
      return;
    }

    // ---- per-lane state of a compute wave ------------------------------------------------------
    const int x0 = tx + P * lane;                         // first column of this lane
    const bool lane_valid = x0 < j.w;                     // (w - tx) % P == 0: all-or-nothing
    unsigned C[P][NC];
    int jlim[P];
    // Candidates d > x-h do not exist near the left image edge (sad.go:66-69).  Instead of masking
    // their keys in every row (which made the leftmost tile's workgroups the critical path of the
    // whole launch), their column sums start with a penalty that outlives the sliding updates:
    // a pixel's candidate d is missing exactly when its window's first column c0 = x-h is < d.
    //   unpacked: every column c < d starts at 2^27 (17 columns * 2^27 + any key < 2^32);
    //   packed:   u16 sums cannot hold more than one penalty per window, so only columns
    //             c < d with c == d-1 (mod 2h+1) start at PEN = (2h+1)^2*255: every full window
    //             holds exactly one column of that class, and it is < d iff c0 < d; a penalised sum
    //             is >= the worst d = 0 sum and <= 2*PEN, which fits 16 bits for h <= 5.
    // Windows clipped on the left (x <= h) only have d = 0 and are forced to lut[0] by the helper;
    // windows clipped on the right can meet missing candidates only when w <= D + 2h + 1: those
    // frames, and packed h = 6, 7, keep the per-key masks (reduce_row<2>).
    constexpr bool PEN_OK = !PK || H <= 5;
    constexpr unsigned PEN = PK ? (unsigned)(B1 * B1 * 255) : (1u << 27);
    const bool use_pen = PEN_OK && (!PK || j.w > j.maxdisp + B1 + 1) && !(a.dbg & 256);
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int col = x0 + p;
      const bool biased = !PK && ((unsigned)(col + B1 * 4096) % (unsigned)B1) == 0u;
      jlim[p] = min(j.maxdisp, max(0, col - H)) - dc;     // candidate q valid iff q <= jlim[p]
#pragma unroll
      for (int q = 0; q < NC; ++q) {
        if constexpr (PK) {
          unsigned v = 0u;
          if (use_pen && col >= 0) {
            const int de = dc + 2 * q, dd = dc + 2 * q + 1;
            if (col < de && (de - 1 - col) % B1 == 0) v |= PEN;
            if (col < dd && (dd - 1 - col) % B1 == 0) v |= PEN << 16;
          }
          C[p][q] = v;
        } else {
          unsigned v = biased ? (unsigned)(dc + q) : 0u;
          if (use_pen && col >= 0 && col < dc + q) v += PEN;
          C[p][q] = v;
        }
      }
    }
    // which candidates of this wave's chunk exist?  (all three are wave-uniform)
    const int qn = min(ND, j.maxdisp - dc + 1);           // chunk runs past MaxDisparity: first qn only
    const bool tail_mask = qn < ND;
    const bool lane_mask = !use_pen && !(a.dbg & 128) &&
                           max(ox0, 0) - H < min(dc + ND - 1, j.maxdisp);   // some pixel has d <= x-h < chunk end

    // this lane's operands inside a ring row (u32 index): L at lofs, R value m at rofs + m
    const int lofs = P * lane;
    const int rofs = TW + (a.dpad - dc - (ND - 1)) + P * lane;        // multiple of 4 by construction

    int slot_in = 0;                                      // row ys-H+t lives in slot t % NR
    for (int t0 = 0; t0 < nsteps; t0 += RB) {
#pragma unroll
      for (int k = 0; k < RB; ++k) {
        const int t = t0 + k;
        if (t < nsteps) {
          // the leaving row (B1 rows earlier) sits NR-B1 = 2*RB slots ahead; before the window is
          // full: the zero row
          const int so = slot_in + 2 * RB;
          const int slot_out = t > 2 * H ? (so >= NR ? so - NR : so) : NR;
          const unsigned* rin = ring + slot_in * ROWW;
          const unsigned* rout = ring + slot_out * ROWW;
          if (lane_valid && !(a.dbg & 1))
            update_columns(C, reinterpret_cast<const uint4*>(rin + lofs), reinterpret_cast<const uint4*>(rin + rofs),
                           reinterpret_cast<const uint4*>(rout + lofs), reinterpret_cast<const uint4*>(rout + rofs));

          if (t >= 2 * H && !(a.dbg & 2)) {
            unsigned best[P];
#pragma unroll
            for (int p = 0; p < P; ++p) best[p] = KEY_INVALID;
            if (lane_mask) {
#pragma unroll
              for (int p = 0; p < P; ++p) asm volatile("" : "+v"(jlim[p]));   // keep the compares in the loop
              reduce_row<2>(C, jlim, best, (unsigned)dc, qn);
            } else if (tail_mask) {
              reduce_row<1>(C, jlim, best, (unsigned)dc, qn);
            } else {
              reduce_row<0>(C, jlim, best, (unsigned)dc, qn);
            }
            unsigned* rb = rowbest + (t % NB) * TW + lofs;
#pragma unroll
            for (int p = 0; p < P; ++p)
              if (!(a.dbg & 32)) atomicMin(rb + p, best[p]);                  // ds_min_u32
              else asm volatile("" :: "v"(best[p]));
          }
          slot_in = slot_in + 1 == NR ? 0 : slot_in + 1;
        }
      }
      if (!(a.dbg & 16)) wg_barrier();
    }
  }

  // ---- helper wave: row streaming + output --------------------------------------------------------
  // Step t (between barrier t-1 and barrier t): issue the global loads of row ys-H+t+2 (two rows
  // ahead, so their latency spans a whole step), finalize the output row of step t-1 from its
  // (now complete) row buffer and reset it, then land row ys-H+t+1 (loaded during step t-1) in the
  // ring slot that held the row which left the window two steps ago.
  static __device__ __forceinline__ void helper_loop(const CwArgs& a, unsigned* ring, unsigned* rowbest,
                                                  const unsigned char* lut,
                                                  const uint8_t* __restrict__ Lg, const uint8_t* __restrict__ Rg,
                                                  uint8_t* __restrict__ Og, int ROWW, int nvals, int tx, int ys,
                                                  int nsteps, int ox0, int ox1, int lane)
  {
    const Job& j = a.j;
    constexpr int MAXV = (2 * TW + 15 * ND + 63) / 64;   // ring-row elements per helper lane (15 waves worst case)
    int pcol[MAXV];                                       // image column, or -1: always zero
    const int nL = TW;
#pragma unroll
    for (int v = 0; v < MAXV; ++v) {
      const int i = lane + 64 * v;
      const int col = i >= nL ? tx - a.dpad + (i - nL) : tx + i;
      pcol[v] = (i < nvals && col >= 0 && col < j.w) ? col : -1;
    }
    auto load_row = [&](int yy, unsigned (&pf)[MAXV]) {   // raw bytes of image row yy (0 outside the image)
      if (yy >= 0 && yy < j.h && !(a.dbg & 8)) {
        const uint8_t* lrow = Lg + (size_t)(yy - j.row0) * j.lstride;
        const uint8_t* rrow = Rg + (size_t)(yy - j.row0) * j.rstride;
#pragma unroll
        for (int v = 0; v < MAXV; ++v)                    // slots below TW/64 read L, the rest R (compile time)
          pf[v] = pcol[v] >= 0 ? (unsigned)(64 * v >= nL ? rrow : lrow)[pcol[v]] : 0u;
      } else {
#pragma unroll
        for (int v = 0; v < MAXV; ++v) pf[v] = 0u;
      }
    };
    const unsigned kmask = (1u << a.ksh) - 1u;
    auto finalize = [&](int t) {                          // output row of step t: 4 pixels per lane per pass
      const int y = ys + t - 2 * H;
      uint4* rb = reinterpret_cast<uint4*>(rowbest + (t % NB) * TW);
      uint8_t* orow = Og + (size_t)(y - j.oy) * j.ostride - j.ox;
#pragma unroll
      for (int g = 0; g < TW / 256; ++g) {
        const int i = 4 * (lane + 64 * g);
        const uint4 k = rb[lane + 64 * g];
        rb[lane + 64 * g] = make_uint4(KEY_INVALID, KEY_INVALID, KEY_INVALID, KEY_INVALID);
        unsigned char v0 = lut[k.x & kmask], v1 = lut[k.y & kmask], v2 = lut[k.z & kmask], v3 = lut[k.w & kmask];
        const int x = tx + i;
        if (x <= H) {                                       // x <= h: d = 0 is the only candidate (sad.go:66-69)
          v0 = lut[0];
          if (x + 1 <= H) v1 = lut[0];
          if (x + 2 <= H) v2 = lut[0];
          if (x + 3 <= H) v3 = lut[0];
        }
        if (x >= ox0 && x < ox1) orow[x] = v0;
        if (x + 1 >= ox0 && x + 1 < ox1) orow[x + 1] = v1;
        if (x + 2 >= ox0 && x + 2 < ox1) orow[x + 2] = v2;
        if (x + 3 >= ox0 && x + 3 < ox1) orow[x + 3] = v3;
      }
    };
    // Super-step s (steps t0 = s*RB .. t0+RB-1, between barrier s-1 and barrier s):
    //   land   the rows of super-step s+1 (loaded during super-step s-1) into their ring slots,
    //   load   the rows of super-step s+2 into the other register set,
    //   drain  the row buffers of super-step s-1 (complete since barrier s-1) to the output.
    // Order matters: hipcc cannot count the conditional loads / stores, so its wait before the
    // first use of a landed register is vmcnt(0); landing first makes that wait cover only
    // operations issued a whole super-step earlier.
    unsigned pfa[RB][MAXV], pfb[RB][MAXV];
#pragma unroll
    for (int k = 0; k < RB; ++k) load_row(ys - H + RB + k, pfa[k]);       // rows of super-step 1
    auto sstep = [&](int t0, unsigned (&land)[RB][MAXV], unsigned (&fill)[RB][MAXV]) {
#pragma unroll
      for (int k = 0; k < RB; ++k) {
        const int t = t0 + RB + k;                        // step whose row is landed now
        if (t < nsteps) {
          unsigned* dst = ring + (t % NR) * ROWW;
#pragma unroll
          for (int v = 0; v < MAXV; ++v) {
            const int i = lane + 64 * v;
            if (i < nvals) dst[i] = land[k][v] << a.pixshift;
          }
        }
      }
#pragma unroll
      for (int k = 0; k < RB; ++k)
        if (t0 + 2 * RB + k < nsteps) load_row(ys - H + t0 + 2 * RB + k, fill[k]);
#pragma unroll
      for (int k = 0; k < RB; ++k) {
        const int t = t0 - RB + k;                        // step whose output row is drained now
        if (t >= 2 * H && !(a.dbg & 4)) finalize(t);
      }
      if (!(a.dbg & 16)) wg_barrier();
    };
    int t0 = 0;
    for (; t0 + RB < nsteps; t0 += 2 * RB) {
      sstep(t0, pfa, pfb);
      sstep(t0 + RB, pfb, pfa);
    }
    if (t0 < nsteps) { sstep(t0, pfa, pfb); t0 += RB; }
    // rows of the last super-step (complete since the last barrier)
    for (int t = t0 - RB; t < nsteps; ++t)
      if (t >= 2 * H && !(a.dbg & 4)) finalize(t);
  }
};


// WB = workgroup-size bound: 1024 threads -> 128 VGPRs.  The unpacked P = 8 kernel keeps 96 column
// sums: 512 threads (256 VGPRs, no spills) when the workgroup fits, else 768 (168 VGPRs).
template <int P, int ND, int H, bool PK, int WB>
__global__ void __launch_bounds__(WB) k_colwave(CwArgs a) { Cw<P, ND, H, PK>::run(a); }

struct Plan { int p, nd, nw; bool pk; };

int env_int(const char* name, int dflt)
{
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

// Kernel variant for (h, D): columns per lane P, packed or not, chunk size ND and wave count NW
// covering the D+1 candidates with the fewest idle slots (ties: fewer waves).
//   packed   (h <= 7): P = 4 (h <= 4) with ND in {24, 12, 8}, or P = 8 with ND = 12; up to 15 waves
//   unpacked (h == 8, or SADHIP_TUNE_PK=0): P = 4 ND {12, 8} up to 15 waves; P = 8 ND 12 up to 11 waves
Plan plan_for(int h, int maxdisp)
{
  Plan best{0, 0, 0, false};
  if (h < 1 || h > 8 || maxdisp < 1) return best;
  const bool pk = h <= 7 && env_int("SADHIP_TUNE_PK", 1) != 0;
  int p = h <= 4 ? 4 : 8;
  const int fp = env_int("SADHIP_TUNE_P", 0);                   // tuning aid: force 4 or 8
  if ((fp == 4 || fp == 8) && h <= fp) p = fp;
  const int force_nd = env_int("SADHIP_TUNE_ND", 0);            // tuning aid: force a chunk size
  const int cands = maxdisp + 1;
  const int max_waves = (pk || p == 4) ? 15 : 11;               // + 1 helper wave per workgroup
  int best_waste = 1 << 30;
  for (int nd : {12, 24, 8}) {                               // ties: 12 (measured fastest)
    if (force_nd && force_nd != nd) continue;
    if (nd == 24 && !(pk && p == 4)) continue;
    if (nd == 8 && p == 8) continue;
    const int nw = (cands + nd - 1) / nd;
    if (nw > max_waves) continue;
    const int waste = nw * nd - cands;
    if (waste < best_waste) { best_waste = waste; best = Plan{p, nd, nw, pk}; }
  }
  return best;
}

template <int P, int ND, int H, bool PK, int WB = 1024>
hipError_t launch_t(const Job& j, const Plan& pl, hipStream_t s)
{
  using K = Cw<P, ND, H, PK>;
  CwArgs a{};
  a.j = j;
  a.nw = pl.nw;
  a.ksh = PK ? 16 : 32 - __builtin_clz((unsigned)j.maxdisp);  // unpacked: 2^ksh > maxdisp
  a.pixshift = PK ? 0 : a.ksh;
  a.dbg = env_int("SADHIP_TUNE_DBG", 0);
  a.dpad = pl.nw * ND - 1;                                    // == 3 (mod 4): lane operands 16-B aligned
  const int r = j.w % P;
  a.tx0 = r == 0 ? 0 : r - P;
  // tiles whose output range meets [x0, x1)
  int k0 = 0, k1 = 0;
  while (a.tx0 + (k0 + 1) * K::STEP + K::OL <= j.x0) ++k0;   // tile k0 ends after x0
  k1 = k0;
  while (a.tx0 + (k1 + 1) * K::STEP + K::OL < j.x1) ++k1;
  const int ntiles = k1 - k0 + 1;
  a.tile0 = k0;
  const int rows = j.y1 - j.y0;
  // strip height.  About 512 workgroups are resident at once; a launch runs fastest when strips
  // are tall (each strip repeats 2h warm-up rows): split rows only until one full round exists.
  const long want = env_int("SADHIP_TUNE_WGS", 512);          // tuning aid: target workgroup count
  long nstrips = (want + (long)ntiles * j.n - 1) / ((long)ntiles * j.n);
  const long max_strips = (rows + 15) / 16;
  if (nstrips > max_strips) nstrips = max_strips;
  if (nstrips < 1) nstrips = 1;
  a.th = (int)((rows + nstrips - 1) / nstrips);
  nstrips = (rows + a.th - 1) / a.th;
  const int ROWW = 2 * K::TW + a.dpad + 1;
  const size_t lds = (size_t)((K::NR + 1) * ROWW + K::NB * K::TW) * 4 + (size_t)((j.maxdisp + 1 + 15) & ~15)
                     + (size_t)env_int("SADHIP_TUNE_LDSPAD", 0);   // tuning aid: cap workgroups per CU
  for (int f0 = 0; f0 < j.n; f0 += 32768) {
    CwArgs aa = a;
    aa.j.n = (j.n - f0 < 32768) ? j.n - f0 : 32768;
    aa.j.left += (size_t)f0 * j.left_frame_stride;
    aa.j.right += (size_t)f0 * j.right_frame_stride;
    aa.j.out += (size_t)f0 * j.out_frame_stride;
    hipLaunchKernelGGL((k_colwave<P, ND, H, PK, WB>), dim3(ntiles, (unsigned)nstrips, aa.j.n), dim3(64 * (pl.nw + 1)), lds, s, aa);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

}  // namespace

bool fast_supported(const Job& j)
{
  if (j.w < 1 || j.h < 1) return false;
  return plan_for(j.block / 2, j.maxdisp).nw != 0;  // BlockSize 2..17, candidates fit the wave budget
}

hipError_t launch_fast(const Job& j, hipStream_t s)
{
  if (j.n <= 0 || j.x1 <= j.x0 || j.y1 <= j.y0) return hipSuccess;
  const int h = j.block / 2;
  const Plan pl = plan_for(h, j.maxdisp);
#define SADHIP_CASE(PP, NDD, HH, PKK) \
  if (pl.p == PP && pl.nd == NDD && h == HH && pl.pk == PKK) return launch_t<PP, NDD, HH, PKK>(j, pl, s)
  // packed
  SADHIP_CASE(4, 24, 1, true); SADHIP_CASE(4, 24, 2, true); SADHIP_CASE(4, 24, 3, true); SADHIP_CASE(4, 24, 4, true);
  SADHIP_CASE(4, 12, 1, true); SADHIP_CASE(4, 12, 2, true); SADHIP_CASE(4, 12, 3, true); SADHIP_CASE(4, 12, 4, true);
  SADHIP_CASE(4, 8, 1, true);  SADHIP_CASE(4, 8, 2, true);  SADHIP_CASE(4, 8, 3, true);  SADHIP_CASE(4, 8, 4, true);
  SADHIP_CASE(8, 12, 4, true); SADHIP_CASE(8, 12, 5, true); SADHIP_CASE(8, 12, 6, true); SADHIP_CASE(8, 12, 7, true);
  // unpacked (window SAD does not fit 16 bits, or forced for comparison)
  if (pl.p == 8 && pl.nd == 12 && h == 8 && !pl.pk)
    return 64 * (pl.nw + 1) <= 512 ? launch_t<8, 12, 8, false, 512>(j, pl, s) : launch_t<8, 12, 8, false, 768>(j, pl, s);
  SADHIP_CASE(4, 12, 3, false); SADHIP_CASE(4, 12, 4, false);
#undef SADHIP_CASE
  return launch_generic(j, s);
}

}  // namespace sadhip
