// Persistent NT GEMM with loader and consumer waves: conv1 / conv2 forward, conv2 / conv1 dgrad and
// the linear layer's forward and dgrad at training batches (derl/models.py:94-124 and its autograd
// backward).  C[m][n] = epi(sum_k A(m,k) * Wp[n][k]) with the implicit A matrix of igemm.hpp.
//
// The LDS-tiled kernels of igemm.hip / igemm_pix.hip start a workgroup per output tile: with K =
// 512 / 576 (and 64..576 for a dgrad pixel, whose invalid taps are skipped) a tile is 2..18 K steps,
// and every tile pays its own memory round trip before the first MFMA and its own epilogue after
// the last.  Here 512 workgroups (two per CU) stay resident and walk their tiles as ONE stream of
// K steps through a ring of LDS stages:
//   * LOADER waves issue nothing but LDS-DMA fills (global_load_lds_dwordx4, 1 KiB pieces of 8 rows
//     x 128 bytes) in the SGPR-base form: a uniform 64-bit base (tensor + the step's offset, scalar
//     adds) plus one per-lane 32-bit offset that changes only with the tile -- no vector ALU
//     instruction per fill (beside busy matrix pipes every VALU instruction waits for a gap);
//   * CONSUMER waves do the fragment reads, the MFMAs and the epilogues; ONE s_barrier per K step is
//     the whole hand-off: the loaders arrive once the step's fill has landed (their own counted
//     vmcnt), the consumers once they are done with the previous step, whose slot the loaders then
//     refill.  The ring keeps running across tile boundaries, so the next tile's operands are in
//     flight while this tile's last MFMAs and its epilogue run;
//   * the DMA writes LDS linearly, so the bank swizzle of the 128-byte rows is applied to the
//     SOURCE address (16-byte chunk c of row R sits in slot c ^ ((R >> 1) & 7)): a lane reads 4
//     consecutive k of its row with one conflict-free ds_read_b128 (3 reads feed 8 MFMAs);
//   * MODE 0 (forward): a tile is BM consecutive output pixels, the K steps come from a host-built
//     table (the stride-2 layer walks its taps grouped by parity: L2 reuse of the input);
//     MODE 1 (dgrad): a tile is one input pixel of BM images, so the set of valid taps is uniform
//     (a rectangle computed arithmetically) and the others are skipped; the workgroups of an XCD
//     walk the pixels of the same image groups (shared taps from L2); MODE 2 (plain rows, the
//     linear layer): the same walk with the column tile in the pixel's place.
// Shapes, measurements and what did NOT work: DESIGN.md section 3.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "igemm_dev.hpp"

namespace dx {
namespace ntp {
constexpr int kBK = 32;  // K elements per stage

// Tile shape of one instantiation.  BM x BN tile, consumer waves of (32 TM) x (32 TN), NLOAD loader waves,
// RING stages of (BM + BN) rows x 128 bytes, WGS workgroups per CU.
template <int BM_, int BN_, int TM_, int TN_, int RING_, int NLOAD_, int WGS_>
struct Shape {
  static constexpr int BM = BM_, BN = BN_, TM = TM_, TN = TN_, RING = RING_, NLOAD = NLOAD_, WGS = WGS_;
  static constexpr int WN = BN / (32 * TN), NCONS = (BM / (32 * TM)) * WN, THREADS = 64 * (NCONS + NLOAD);
  static constexpr int STAGE_FLOATS = (BM + BN) * kBK;
  static constexpr int APIECES = BM / 8 / NLOAD, WPIECES = BN / 8 / NLOAD;  // per loader wave and stage
  static constexpr int LDS_BYTES = RING * STAGE_FLOATS * 4;
  static_assert(BM % (32 * TM) == 0 && BN % (32 * TN) == 0 && (BM / 8) % NLOAD == 0 && (BN / 8) % NLOAD == 0, "tile shape");
};
using ShapeS = Shape<128, 64, 2, 1, 3, 4, 2>;   // 64-column stages: 4 + 4 waves, two workgroups per CU
using ShapeF = Shape<128, 64, 2, 1, 3, 8, 2>;   // forward: the same tile with eight loaders (66 registers: 6 waves per SIMD fit; -2 %)
using ShapeL = Shape<128, 128, 2, 2, 2, 4, 2>;  // 128-column stages: 4 + 4 waves of 64 x 64, two workgroups per CU

// between rollout and training sizes (256..511 tiles of 64 rows): 64 x 64 tiles on four consumer waves
// of 32 x 32 and eight loaders (a wave gets one fill instruction through per ~900 cycles): 5-10 %
// ahead of the latency kernels there, behind them below 256 tiles
using ShapeX = Shape<64, 64, 1, 1, 3, 8, 2>;
}  // namespace ntp
namespace {
using ntp::ShapeS;
using ntp::ShapeL;
using ntp::ShapeF;
using ntp::ShapeX;
using ntp::kBK;

using f4 = __attribute__((ext_vector_type(4))) float;

struct NtpArgs {
  NTArgs nt;
  int ntiles;         // MODE 0: row tiles
  int nimg, ngroups;  // MODE 1: images, groups of 128 images
  int tiles_per_xcd;  // MODE 1: ceil(ngroups / 8) * pixels
  // MODE 2: rows_inner = the walk of an XCD's tiles has the row groups innermost (u -> column tile
  // u / G, row group u % G, G = groups_per_xcd): the XCD's resident workgroups then cover ALL of its
  // row groups x a few column tiles, so its rows (G x 128 x K floats: 2 MB of dhid at minibatch 8192)
  // stay in its L2 for the whole launch and every 64-column slice of W passes through once.  Column
  // tiles innermost (the MODE 1 walk) cycled all of W (6.4 MB > the 4 MB L2) once per row group:
  // fc_dgrad fetched 484 MB for 126 MB of operands.
  int rows_inner, groups_per_xcd;
  FastDiv div_groups;
  // MODE 2 forward with few tiles: K in `kparts` parts (1, 2 or 7 of the linear layer's 98 steps) --
  // the "column tile" index is column tile * kparts + part (tile walk and cursor unchanged: the part is the cursor's run `ta`, PA elements
  // apart in a row of A and of W), part kp writes its partial sums to out + kp * slab_bytes (part 0
  // adds the bias); the launcher sums the parts afterwards
  int kparts;
  FastDiv div_kparts;
  long long slab_bytes;
  // MODE 1: an XCD walks a CONTIGUOUS eighth of the (image group, pixel) tiles instead of whole groups
  // (group g on XCD g % 8): with a group count that is not a multiple of 8 some XCDs owned one group
  // more than the others (config 5's shard, 20 groups: 3 against 2 -- conv1_dgrad 173 us where the
  // balanced walk takes 145)
  int blocked;
  int TA, TB, PA, PB;  // run grid and the element pitch of a step in ta / tb (see Cursor)
  // MODE 0: the K steps of a tile in the order they are walked (element offsets into the input
  // window and into a row of Wp).  The order is free (any permutation of the K axis); the
  // stride-2 layer walks its taps grouped by (kh % 2, kw % 2): the four taps that read the same
  // input pixel for neighbouring output pixels follow each other, so the pixel is still in L2
  // when it is wanted again (kernel-row-major order: 0.72 GB of fetches for 0.42 GB of input).
  int nstep;
  int step_ao[24], step_ko[24];
  int diag;           // DX_NTP_DIAG bits: 1 = stamps, 2 = cycles at the wait + barrier; WRONG RESULTS: 4 = no fills
                      // after the first two, 8 = no barrier, 16 = no epilogue loads / stores, 32 = every tile reads the rows of the first
};

// Epilogue accesses in the SGPR-base form too: a uniform 64-bit base (tensor + the row's offset, scalar
// adds) plus ONE per-lane 32-bit byte offset per column block, instead of a 64-bit per-lane
// address per element (two VALU instructions and two registers each).
// this lane's index in its wave, as a value the compiler neither hoists nor merges with another call's
__device__ __forceinline__ int fresh_lane() {
  int l = static_cast<int>(__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)));
  asm volatile("" : "+v"(l));
  return l;
}

__device__ __forceinline__ float load_at(const float *tensor, long long uniform_bytes, uint32_t lane_bytes) {
  asm volatile("" : "+v"(lane_bytes));
  return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(tensor) + uniform_bytes + lane_bytes);
}
__device__ __forceinline__ void store_at(float *tensor, long long uniform_bytes, uint32_t lane_bytes, float v) {
  asm volatile("" : "+v"(lane_bytes));
  *reinterpret_cast<float *>(reinterpret_cast<char *>(tensor) + uniform_bytes + lane_bytes) = v;
}

// Position in the stream of K steps: tile `i` of this workgroup, tap (ta, tb), offset q inside the
// tap's run.  The runs of a row form a TA x TB grid (forward: TA = kernel rows, TB = 1; dgrad: one
// run per kernel tap, tap (ta, tb) reads pixel (y - ta, x - tb)), so the runs that lie inside the
// image are a rectangle [ta_lo, ta_hi] x [tb_lo, tb_hi] and every offset is arithmetic: no table
// look-ups (scalar loads) on the per-step path.
struct Cursor {
  int i;        // tile counter of this workgroup
  int valid;    // tile exists
  int tile;     // MODE 0: row tile; MODE 1: image group; MODE 2: row block
  int pix;      // MODE 1: pixel; MODE 2: column tile
  int pixoff;   // MODE 1: element offset of the pixel inside an image
  int ta, tb, q;
  int ta_hi, tb_lo, tb_hi;
};

template <int MODE, int BM>
__device__ __forceinline__ void open_tile(Cursor &c, const NtpArgs &p) {
  const Gather &g = p.nt.g;
  if (MODE == 0) {
    c.tile = blockIdx.x + c.i * gridDim.x;
    c.valid = c.tile < p.ntiles;
    c.pix = 0; c.pixoff = 0;
    c.ta = 0; c.ta_hi = p.TA - 1; c.tb_lo = 0; c.tb_hi = 0;
  } else {
    const int xcd = blockIdx.x & 7, per = gridDim.x >> 3;
    const uint32_t u = (blockIdx.x >> 3) + c.i * per;
    if ((MODE == 1 || MODE == 2) && p.blocked) {  // tile t of the group-major list, this XCD's contiguous eighth
      const uint32_t t = xcd * static_cast<uint32_t>(p.tiles_per_xcd) + u;
      const uint32_t gb = fdiv(t, g.div_img);
      c.pix = t - gb * g.OHW;
      c.tile = gb;
    } else {
      uint32_t gl = fdiv(u, g.div_img);
      c.pix = u - gl * g.OHW;
      if (MODE == 2 && p.rows_inner) {
        c.pix = fdiv(u, p.div_groups);
        gl = u - c.pix * p.groups_per_xcd;
      }
      c.tile = gl * 8 + xcd;
    }
    c.valid = u < static_cast<uint32_t>(p.tiles_per_xcd) && c.tile < p.ngroups;
    if (MODE == 2) {  // plain rows: the "pixel" is the column tile (x the K part), one run of K elements
      const int kp = c.pix - static_cast<int>(fdiv(c.pix, p.div_kparts)) * p.kparts;
      c.ta = kp; c.ta_hi = kp; c.tb_lo = 0; c.tb_hi = 0; c.pixoff = 0;
    } else {
      const int y = fdiv(c.pix, g.div_row), x = c.pix - y * g.OW;  // unit stride (checked on the host)
      c.ta = max(0, y - g.H + 1); c.ta_hi = min(p.TA - 1, y);
      c.tb_lo = max(0, x - g.W + 1); c.tb_hi = min(p.TB - 1, x);
      c.pixoff = (y * g.W + x) * g.C;
    }
  }
  c.tb = c.tb_lo;
  c.q = 0;
}

// -> true when the step just left was the last of its tile
template <int MODE, int BM>
__device__ __forceinline__ bool advance(Cursor &c, const NtpArgs &p) {
  if (MODE == 0) {  // q counts the steps of the table
    if (++c.q < p.nstep) return false;
  } else {
    c.q += kBK;
    if (c.q < p.nt.g.seglen) return false;
    c.q = 0;
    if (++c.tb <= c.tb_hi) return false;
    c.tb = c.tb_lo;
    if (++c.ta <= c.ta_hi) return false;
  }
  ++c.i;
  open_tile<MODE, BM>(c, p);
  return true;
}

// (the shape is spelled out as integers: with a class parameter in __launch_bounds__ hipcc emits no
// host stub for the instantiations)
// TAG = the network stage: one instantiation (= one profiler row) per stage
template <int TAG, int MODE, int EPI, int BM, int BN, int TM, int TN, int RING, int NLOAD, int WGS>
__global__ __launch_bounds__(64 * ((BM / (32 * TM)) * (BN / (32 * TN)) + NLOAD),  // second argument: waves per SIMD
                             ((BM / (32 * TM)) * (BN / (32 * TN)) + NLOAD) * WGS / 4) void ntp_kernel(const NtpArgs p,
                                                                                               unsigned long long *stamps) {
  using S = ntp::Shape<BM, BN, TM, TN, RING, NLOAD, WGS>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const unsigned long long t_entry = (kDiag && stamps) ? __builtin_amdgcn_s_memrealtime() : 0;
  const NTArgs &a = p.nt;
  const Gather &g = a.g;
  // The lane index is formed where each role needs it (mbcnt: two instructions), not carried from
  // threadIdx.x across the role branch: at the 128-register cap of the 128x128 shape (conv1's data
  // gradient, the dominant kernel) that one live value and its copy were spilled -- a 12-byte private
  // segment for a kernel that otherwise needs none.
  const int wave8 = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
  Cursor first;
  first.i = 0;
  open_tile<MODE, BM>(first, p);
  if (!first.valid) return;  // uniform: this workgroup has no tile

  if (wave8 >= S::NCONS) {
    // ================= loader waves: fills only =================
    // loader l fills A pieces l, l + NLOAD, ... (8 rows each) and W pieces l, l + NLOAD, ... of every stage
    const int wave = wave8 - S::NCONS;
    const int lane = fresh_lane();
    const int lrow = lane >> 3;
    // Addresses are a UNIFORM 64-bit base (tensor + the step's offset: scalar adds) plus a per-lane
    // 32-bit byte offset that changes only with the tile: the fill then issues no vector ALU
    // instruction at all (global_load_lds with an SGPR base).  A loader shares its SIMD with
    // consumers that keep the matrix pipe full; every VALU instruction it issued waited for a gap
    // there (measured: ~800 cycles per fill instruction with 64-bit per-lane address adds).
    uint32_t wv[S::WPIECES];
#pragma unroll
    for (int j = 0; j < S::WPIECES; ++j) {
      const int n = (wave + S::NLOAD * j) * 8 + lrow;
      wv[j] = (static_cast<uint32_t>(n) * a.K + 4 * ((lane & 7) ^ ((n >> 1) & 7))) * 4u;
    }
    uint32_t av[S::APIECES];
    Cursor ld = first;  // where the NEXT fill goes
#define DX_NTP_ROWS()                                                                            \
  _Pragma("unroll") for (int q_ = 0; q_ < S::APIECES; ++q_) {                                    \
    const int R = (wave + S::NLOAD * q_) * 8 + lrow;                                             \
    const int chunk = (lane & 7) ^ ((R >> 1) & 7);                                               \
    uint32_t base;                                                                               \
    if (MODE == 0) {                                                                             \
      const uint32_t m = ld.tile * BM + R;                                                       \
      const uint32_t img = fdiv(m, g.div_img), rem = m - img * g.OHW;                            \
      const uint32_t oy = fdiv(rem, g.div_row), ox = rem - oy * g.OW;                            \
      base = img * static_cast<uint32_t>(g.img_stride) + (oy * g.sy * g.W + ox * g.sx) * g.C;    \
    } else {                                                                                     \
      base = static_cast<uint32_t>(ld.tile * BM + R) * static_cast<uint32_t>(g.img_stride);      \
    }                                                                                            \
    av[q_] = (base + 4 * chunk) * 4u;                                                            \
  }
#define DX_NTP_FILL(SLOT)                                                                        \
  {                                                                                              \
    float *dst_ = smem + (SLOT) * S::STAGE_FLOATS;                                               \
    const char *abase_ = static_cast<const char *>(g.src) +                                      \
        4LL * (MODE == 0 ? p.step_ao[ld.q] : ld.pixoff + ld.ta * p.PA + ld.tb * p.PB + ld.q);    \
    const char *wbase_ = reinterpret_cast<const char *>(a.Wp) +                                  \
        4LL * (MODE == 0 ? p.step_ko[ld.q] : (ld.ta * p.TB + ld.tb) * g.seglen + ld.q) +         \
        (MODE == 2 ? 4LL * fdiv(ld.pix, p.div_kparts) * S::BN * a.K : 0LL);                      \
    _Pragma("unroll") for (int q_ = 0; q_ < S::APIECES; ++q_)                                    \
        dma_piece(abase_, av[q_], dst_ + (wave + S::NLOAD * q_) * 256);                         \
    _Pragma("unroll") for (int j_ = 0; j_ < S::WPIECES; ++j_)                                    \
        dma_piece(wbase_, wv[j_], dst_ + (BM / 8 + wave + S::NLOAD * j_) * 256);                \
    const int before_ = ld.tile;                                                                 \
    if (advance<MODE, BM>(ld, p) && ld.valid && (MODE == 0 || ld.tile != before_)) { DX_NTP_ROWS() } \
    ++issued;                                                                                    \
  }
    DX_NTP_ROWS()
    int issued = 0, slot = 0;  // slot of the next fill
    for (int r = 0; r < S::RING - 1; ++r)
      if (r == 0 || ld.valid) {
        DX_NTP_FILL(slot)
        ++slot;
      }
    slot = S::RING - 1;
    unsigned long long l_vm = 0, l_bar = 0, l_all = (kDiag && stamps) ? __builtin_amdgcn_s_memtime() : 0;
    for (int gstep = 0; gstep < issued; ++gstep) {
      // hand step gstep over: its fill is the oldest in flight, at most RING - 2 younger fills behind it
      const unsigned long long l0 = (kDiag && stamps && (p.diag & 2)) ? __builtin_amdgcn_s_memtime() : 0;
      if (S::RING == 3 && issued - gstep >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(S::APIECES + S::WPIECES) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned long long l1 = (kDiag && stamps && (p.diag & 2)) ? __builtin_amdgcn_s_memtime() : 0;
      __builtin_amdgcn_s_barrier();  // consumers are done with step gstep - 1: its slot is free
      asm volatile("" ::: "memory");
      if (kDiag && stamps && (p.diag & 2)) { l_vm += l1 - l0; l_bar += __builtin_amdgcn_s_memtime() - l1; }
      if (ld.valid && !(kDiag && (p.diag & 4))) {
        DX_NTP_FILL(slot)
        slot = slot == S::RING - 1 ? 0 : slot + 1;
      } else if (ld.valid) {  // diagnostic: walk the cursor without filling
        advance<MODE, BM>(ld, p);
        ++issued;
      }
    }
#undef DX_NTP_FILL
#undef DX_NTP_ROWS
    if (kDiag && stamps && lane == 0) {  // loader stamps behind the consumers': total, vmcnt wait, barrier wait
      unsigned long long *o = stamps + static_cast<long long>(gridDim.x) * S::NCONS * 7 +
                              (static_cast<long long>(blockIdx.x) * S::NLOAD + wave) * 3;
      o[0] = __builtin_amdgcn_s_memtime() - l_all; o[1] = l_vm; o[2] = l_bar;
    }
    return;
  }

  // ================= consumer waves: fragment reads, MFMAs, epilogues =================
  const int wave = wave8;
  const int lane = fresh_lane();
  const int hi = lane >> 5, l31 = lane & 31;
  const int wm = wave / S::WN, wn = wave % S::WN;  // waves of (32 TM) x (32 TN)
  const char *lds = reinterpret_cast<const char *>(smem);
  const unsigned x = (l31 >> 1) & 7;
  unsigned aoff[4], boff[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const unsigned slot = ((2 * q + hi) ^ x) * 16;
    aoff[q] = (wm * 32 * TM + l31) * 128 + slot;
    boff[q] = (BM + wn * 32 * TN + l31) * 128 + slot;
  }
  f32x16 acc[TM][TN];
#pragma unroll
  for (int t = 0; t < TM; ++t)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][j][r] = 0.f;
  Cursor cs = first;  // the step being multiplied
  int slot = 0;
  float bias[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    bias[j] = (EPI == EPI_BIAS_RELU) ? a.bias[wn * 32 * TN + 32 * j + l31] : 0.f;
    // a use of the bias in front of the loop: otherwise the compiler waits for this load at its
    // first use, inside the epilogue
    asm volatile("" ::"v"(bias[j]));
  }

  const unsigned long long t_loop = (kDiag && stamps) ? __builtin_amdgcn_s_memrealtime() : 0;
  const unsigned long long c_loop = (kDiag && stamps) ? __builtin_amdgcn_s_memtime() : 0;
  unsigned long long c_wait = 0;
  int nsteps = 0, ntiles = 0;
  while (cs.valid) {
    const unsigned long long c_w0 = (kDiag && stamps && (p.diag & 2)) ? __builtin_amdgcn_s_memtime() : 0;
    __builtin_amdgcn_s_barrier();  // the loaders arrive once this step's fill has landed
    asm volatile("" ::: "memory");
    if (kDiag && stamps && (p.diag & 2)) c_wait += __builtin_amdgcn_s_memtime() - c_w0;
    ++nsteps;
    if (kDiag && (p.diag & 64)) {  // diagnostic: loaders alone (no fragment reads, no MFMAs)
      slot = slot == S::RING - 1 ? 0 : slot + 1;
      advance<MODE, BM>(cs, p);
      continue;
    }
    const char *base = lds + slot * S::STAGE_FLOATS * 4;
    f4 af[2][TM], bf[2][TN];
#pragma unroll
    for (int t = 0; t < TM; ++t) af[0][t] = *reinterpret_cast<const f4 *>(base + aoff[0] + t * 32 * 128);
#pragma unroll
    for (int j = 0; j < TN; ++j) bf[0][j] = *reinterpret_cast<const f4 *>(base + boff[0] + j * 32 * 128);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = q & 1, nx = c ^ 1;
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c][0][0], bf[c][0][0], acc[0][0], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (q + 1 < 4) {  // the next group's fragments, requested in the first MFMA's shadow
#pragma unroll
        for (int t = 0; t < TM; ++t) af[nx][t] = *reinterpret_cast<const f4 *>(base + aoff[q + 1] + t * 32 * 128);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[nx][j] = *reinterpret_cast<const f4 *>(base + boff[q + 1] + j * 32 * 128);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            if (e + t + j > 0)
              acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c][t][e], bf[c][j][e], acc[t][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    slot = slot == S::RING - 1 ? 0 : slot + 1;

    const int tile = cs.tile, pix = cs.pix;
    if (!advance<MODE, BM>(cs, p)) continue;
    // ---- epilogue of (tile, pix): C/D layout column = l31, row = (r & 3) + 8 (r >> 2) + 4 hi.
    // Tiles are never ragged (the launchers require whole tiles): a guarded store or load here is
    // a branch, and the compiler then drains vmcnt at the loop header.
    // element (t, r) of column block j: uniform part (tile, wave, t, r) + lane part (hi, l31)
    const long long rowbytes = 4LL * (MODE == 0 ? a.ldc : (a.om.enabled ? a.om.OUT_H * a.om.OUT_W : (MODE == 2 ? static_cast<int>(fdiv(g.OHW, p.div_kparts)) : g.OHW)) * a.ldc);
    const long long tile0 = static_cast<long long>(tile * BM + wm * 32 * TM) * rowbytes;  // uniform
    if (kDiag && (p.diag & 16)) {  // diagnostic: no epilogue traffic
    } else if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const uint32_t lane_off = static_cast<uint32_t>(4 * hi) * static_cast<uint32_t>(rowbytes) +
                                  4u * (wn * 32 * TN + 32 * j + l31);
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            store_at(a.out, tile0 + (32 * t + (r & 3) + 8 * (r >> 2)) * rowbytes, lane_off, fmaxf(acc[t][j][r] + bias[j], 0.f));
      }
    } else {
      const OutMap &om = a.om;
      const int oy = fdiv(pix, g.div_row), ox = pix - oy * g.OW;
      long long pix0[TN];  // uniform: byte offset of this row's output pixel and column block inside an image
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int nb = wn * 32 * TN + 32 * j;  // first column of this 32-wide block
        if (om.enabled) {  // column block -> (py, px) of the osy x osx output pixels of this row
          const int gq = nb / om.chan;
          const int py = gq / om.osx, px = gq - py * om.osx;
          pix0[j] = 4LL * (((oy * om.osy + py) * om.OUT_W + ox * om.osx + px) * a.ldc + (nb - gq * om.chan));
        } else if (MODE == 2) {  // column tile pix / kparts of K part pix % kparts
          const int cp = static_cast<int>(fdiv(pix, p.div_kparts));
          pix0[j] = 4LL * (cp * a.ldc + nb) + (pix - cp * p.kparts) * p.slab_bytes;
        } else {
          pix0[j] = 4LL * (pix * a.ldc + nb);
        }
      }
      const uint32_t lane_off = static_cast<uint32_t>(4 * hi) * static_cast<uint32_t>(rowbytes) + 4u * l31;
      if (EPI == EPI_BIAS) {  // plain rows forward: the bias of this column tile, no mask
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int cpb = static_cast<int>(fdiv(pix, p.div_kparts));
          const float bj = (pix - cpb * p.kparts) ? 0.f : a.bias[cpb * S::BN + wn * 32 * TN + 32 * j + l31];
#pragma unroll
          for (int t = 0; t < TM; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              store_at(a.out, tile0 + pix0[j] + (32 * t + (r & 3) + 8 * (r >> 2)) * rowbytes, lane_off, acc[t][j][r] + bj);
        }
      } else {
      // every mask load ahead of the first store (a load behind a store waits for it).  With two
      // column blocks the first block's masks are packed into bits before the second block's loads
      // go out: 32 registers instead of 64 keep the kernel at two workgroups per CU.
      uint32_t keep[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        float mk[TM][16];
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            mk[t][r] = load_at(a.mask_src, tile0 + pix0[j] + (32 * t + (r & 3) + 8 * (r >> 2)) * rowbytes, lane_off);
        keep[j] = 0;
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) keep[j] |= (mk[t][r] > 0.f ? 1u : 0u) << (16 * t + r);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            store_at(a.out, tile0 + pix0[j] + (32 * t + (r & 3) + 8 * (r >> 2)) * rowbytes, lane_off,
                     ((keep[j] >> (16 * t + r)) & 1u) ? acc[t][j][r] : 0.f);
      }
    }
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][j][r] = 0.f;
    ++ntiles;
  }
  if (kDiag && stamps && lane == 0) {  // DX_NTP_DIAG: 100 MHz ticks (entry, loop start, exit), loop cycles, wait cycles, steps, tiles
    unsigned long long *o = stamps + (static_cast<long long>(blockIdx.x) * S::NCONS + wave) * 7;
    o[0] = t_entry; o[1] = t_loop; o[2] = __builtin_amdgcn_s_memrealtime(); o[3] = __builtin_amdgcn_s_memtime() - c_loop;
    o[4] = c_wait; o[5] = nsteps; o[6] = ntiles;
  }
}

bool ntp_on() {  // DX_NTP=0: these stages on the per-tile kernels
  const int v = DX_ENV("DX_NTP", 1);
  return v != 0;
}

// DX_NTP_MIN_TILES: fewest tiles a stage must have to take these kernels (below it a workgroup per
// CU is not reached and the per-tile kernels' finer tiles win)
int ntp_min_tiles() {
  const int v = DX_ENV("DX_NTP_MIN_TILES", 256);
  return v;
}

bool rows_inner_on() {  // DX_NTP_ROWS_INNER=0: the linear layer's tiles walked column tiles innermost
  const int v = DX_ENV("DX_NTP_ROWS_INNER", 1);
  return v != 0;
}

bool ntp_small_on() {  // DX_NTP_SMALL=0: rollout-sized forward stages on the latency kernels
  const int v = DX_ENV("DX_NTP_SMALL", 1);
  return v != 0;
}

int ntp_workgroups(int per_cu) {  // every CU full (diag build: DX_NTP_NWG forces a count)
  int v = 0;
#if DX_DIAG
  v = DX_ENV("DX_NTP_NWG", 0);
#endif
  const int n = v > 0 ? v : 256 * per_cu;
  return n < 8 ? 8 : n / 8 * 8;
}

template <int TAG, int MODE, int EPI, class S>
int launch_as(const NtpArgs &p, hipStream_t stream) {
  DX_LDS_OPT_IN((ntp_kernel<TAG, MODE, EPI, S::BM, S::BN, S::TM, S::TN, S::RING, S::NLOAD, S::WGS>), S::LDS_BYTES);
  const int grid = ntp_workgroups(S::WGS);
#if DX_DIAG
  const int diag = DX_ENV("DX_NTP_DIAG", 0);
#else
  constexpr int diag = 0;
#endif
  if (!diag) {
    hipLaunchKernelGGL((ntp_kernel<TAG, MODE, EPI, S::BM, S::BN, S::TM, S::TN, S::RING, S::NLOAD, S::WGS>), dim3(grid), dim3(S::THREADS), S::LDS_BYTES, stream, p, nullptr);
    DX_LAUNCH_CHECK();
    return DX_OK;
  }
#if DX_DIAG
  // diagnostic: in-kernel stamps, summarised on stderr (synchronous; never on the product path)
  NtpArgs pd = p;
  pd.diag = diag;
  unsigned long long *dev = nullptr;
  const size_t count = static_cast<size_t>(grid) * S::NCONS * 7, lcount = static_cast<size_t>(grid) * S::NLOAD * 3;
  DX_HIP(hipMalloc(&dev, (count + lcount) * 8));
  DX_HIP(hipMemsetAsync(dev, 0, (count + lcount) * 8, stream));
  hipLaunchKernelGGL((ntp_kernel<TAG, MODE, EPI, S::BM, S::BN, S::TM, S::TN, S::RING, S::NLOAD, S::WGS>), dim3(grid), dim3(S::THREADS), S::LDS_BYTES, stream, pd, dev);
  DX_LAUNCH_CHECK();
  DX_HIP(hipStreamSynchronize(stream));
  std::vector<unsigned long long> h(count + lcount);
  DX_HIP(hipMemcpy(h.data(), dev, (count + lcount) * 8, hipMemcpyDeviceToHost));
  DX_HIP(hipFree(dev));
  unsigned long long first = ~0ull, last = 0;
  std::vector<double> pro, loop, cps, wfrac, steps;
  for (size_t i = 0; i < count; i += 7) {
    if (h[i + 5] == 0) continue;
    first = std::min(first, h[i]); last = std::max(last, h[i + 2]);
    pro.push_back((h[i + 1] - h[i]) * 0.01); loop.push_back((h[i + 2] - h[i + 1]) * 0.01);
    cps.push_back(static_cast<double>(h[i + 3]) / h[i + 5]); wfrac.push_back(static_cast<double>(h[i + 4]) / h[i + 3]);
    steps.push_back(static_cast<double>(h[i + 5]));
  }
  std::vector<double> lvm, lbar;
  for (size_t i = count; i < count + lcount; i += 3) {
    if (h[i] == 0) continue;
    lvm.push_back(static_cast<double>(h[i + 1]) / h[i]); lbar.push_back(static_cast<double>(h[i + 2]) / h[i]);
  }
  auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  auto mx = [](const std::vector<double> &v) { return *std::max_element(v.begin(), v.end()); };
  const int ideal = S::TM * S::TN * 16 * 64 * (S::NCONS * S::WGS / 4);  // MFMA cycles of a step on a SIMD shared by NCONS WGS / 4 waves
  fprintf(stderr, "[ntp mode %d %dx%d M=%d K=%d grid=%d] span %.1f us | per wave: prologue %.2f us, loop median %.1f max %.1f us, "
          "steps median %.0f max %.0f, %.0f cycles/step (ideal %d), wait+barrier %.1f %% of the loop\n", MODE, S::BM, S::BN,
          p.nt.M, p.nt.K, grid, (last - first) * 0.01, med(pro), med(loop), mx(loop), med(steps), mx(steps), med(cps), ideal,
          100 * med(wfrac));
  if (!lvm.empty())
    fprintf(stderr, "        loaders: %.1f %% of their time waiting for a fill to land, %.1f %% at the barrier\n", 100 * med(lvm), 100 * med(lbar));
#endif
  return DX_OK;
}

// `rows`: GEMM rows (forward) or images (dgrad) in whole tiles, `tiles`: enough of them to fill the
// chip, `images`: images of the gathered tensor (the loaders address it with 32-bit lane offsets:
// it has to stay below 4 GiB)
template <class S>
bool shape_fits(const NTArgs &a, long long rows, long long tiles, long long images) {
  const Gather &g = a.g;
  if (4LL * images * g.img_stride >= (1LL << 32)) return false;
  return a.N == S::BN && !g.idx && a.ksplit == 1 && g.seglen % kBK == 0 && g.nseg <= kMaxSeg &&
         a.K == g.nseg * g.seglen && rows % S::BM == 0 && tiles >= ntp_min_tiles();
}

}  // namespace

// Forward conv stage (bias + ReLU), 64 output channels.  DX_ENOSUP = not covered: the caller keeps
// its own kernel.
template <class S>
int launch_fwd_as(const NTArgs &a, NtpArgs &p, int stage, hipStream_t stream) {
  p.ntiles = a.M / S::BM;
  if (stage == ST_CONV1_FWD) return launch_as<ST_CONV1_FWD, 0, EPI_BIAS_RELU, S>(p, stream);
  if (stage == ST_CONV2_FWD) return launch_as<ST_CONV2_FWD, 0, EPI_BIAS_RELU, S>(p, stream);
  return DX_ENOSUP;
}

int launch_ntp_fwd(const NTArgs &a, int stage, hipStream_t stream) {
  const Gather &g = a.g;
  if (!ntp_on() || g.check || a.om.enabled || a.ldc != ShapeS::BN) return DX_ENOSUP;
  // 128-row tiles when they fill the chip, 64-row tiles for rollout-sized batches
  const long long images = (a.M + g.OHW - 1) / g.OHW;
  const bool big = shape_fits<ShapeS>(a, a.M, a.M / ShapeS::BM, images);
  if (!big && !(ntp_small_on() && shape_fits<ShapeX>(a, a.M, a.M / ShapeX::BM, images))) return DX_ENOSUP;
  DX_REQUIRE(aligned(g.src, 16) && aligned(a.Wp, 16) && g.C % 4 == 0, "ntp: operands must be 16-byte aligned");
  NtpArgs p;
  p.nt = a;
  p.nimg = p.ngroups = p.tiles_per_xcd = p.diag = p.rows_inner = 0;
  p.kparts = 1; p.div_kparts = make_fastdiv(1); p.slab_bytes = 0; p.blocked = 0;
  p.groups_per_xcd = 1; p.div_groups = make_fastdiv(1);
  p.TA = g.nseg; p.TB = 1; p.PA = g.nseg > 1 ? g.seg_off[1] : 0; p.PB = 0;
  const int per_run = g.seglen / kBK, taps_per_run = g.seglen / g.C, steps_per_tap = g.C / kBK;
  p.nstep = g.nseg * per_run;
  if (p.nstep > 24 || g.C % kBK || g.seglen % g.C) return DX_ENOSUP;
  for (int s2 = 0; s2 < g.nseg; ++s2)
    if (g.seg_off[s2] != s2 * p.PA) return DX_ENOSUP;
  int n = 0;
  // stride 2: taps grouped by parity class (see NtpArgs); otherwise kernel-row-major
  const bool by_parity = g.sy == 2 && g.sx == 2 && g.nseg % 2 == 0 && taps_per_run % 2 == 0;
  for (int cls = 0; cls < (by_parity ? 4 : 1); ++cls)
    for (int kh = 0; kh < g.nseg; ++kh)
      for (int kw = 0; kw < taps_per_run; ++kw) {
        if (by_parity && (kh % 2 != cls / 2 || kw % 2 != cls % 2)) continue;
        for (int h = 0; h < steps_per_tap; ++h, ++n) {
          p.step_ao[n] = g.seg_off[kh] + kw * g.C + h * kBK;
          p.step_ko[n] = kh * g.seglen + kw * g.C + h * kBK;
        }
      }
  if (n != p.nstep) return fail(DX_EINVAL, "ntp: step table has %d of %d steps", n, p.nstep);
  return big ? launch_fwd_as<ShapeF>(a, p, stage, stream) : launch_fwd_as<ShapeX>(a, p, stage, stream);
}

// dgrad stage tiled as one pixel x BM images (ReLU mask from the kept activation): 64 columns on
// 128-image tiles, 128 columns (the stride-2 layer's four parity classes) on 256-image tiles
int launch_ntp_pix(const NTArgs &a, int nimg, int TA, int TB, hipStream_t stream) {
  const Gather &g = a.g;
  const OutMap &om = a.om;
  if (!ntp_on() || TA * TB != g.nseg || TB < 2 || g.sy != 1 || g.sx != 1 || !g.check || !a.mask_src ||
      static_cast<long long>(nimg) * g.OHW != a.M)
    return DX_ENOSUP;
  for (int s2 = 0; s2 < g.nseg; ++s2)  // tap (ta, tb) reads pixel (y - ta, x - tb)
    if (g.seg_dy[s2] != -(s2 / TB) || g.seg_dx[s2] != -(s2 % TB) ||
        g.seg_off[s2] != (s2 / TB) * g.seg_off[TB] + (s2 % TB) * g.seg_off[1])
      return DX_ENOSUP;
  if (om.enabled) {  // every output pixel of every row exists, columns in whole 32-wide channel blocks
    if (om.OHW != g.OHW || om.OW != g.OW || om.chan % 32 || a.ldc != om.chan || a.N != om.osy * om.osx * om.chan ||
        om.OUT_H != (g.OHW / g.OW) * om.osy || om.OUT_W != g.OW * om.osx)
      return DX_ENOSUP;
  } else if (a.ldc != a.N) {
    return DX_ENOSUP;
  }
  const bool large = a.N == ShapeL::BN;
  if (nimg / (large ? ShapeL::BM : ShapeS::BM) < 8) return DX_ENOSUP;  // an image group per XCD at least
  if (!(large ? shape_fits<ShapeL>(a, nimg, 1LL * nimg / ShapeL::BM * g.OHW, nimg)
              : shape_fits<ShapeS>(a, nimg, 1LL * nimg / ShapeS::BM * g.OHW, nimg)))
    return DX_ENOSUP;
  DX_REQUIRE(aligned(g.src, 16) && aligned(a.Wp, 16) && g.C % 4 == 0, "ntp: operands must be 16-byte aligned");
  NtpArgs p;
  p.nt = a;
  p.ntiles = p.diag = p.rows_inner = 0;
  p.kparts = 1; p.div_kparts = make_fastdiv(1); p.slab_bytes = 0;
  const bool blocked_on = DX_ENV("DX_NTP_BLOCKED", 1) != 0;
  p.blocked = blocked_on ? 1 : 0;
  p.groups_per_xcd = 1; p.div_groups = make_fastdiv(1);
  p.TA = TA; p.TB = TB; p.PA = g.seg_off[TB]; p.PB = g.seg_off[1];
  p.nimg = nimg;
  p.ngroups = nimg / (large ? ShapeL::BM : ShapeS::BM);
  p.tiles_per_xcd = p.blocked ? static_cast<int>((1LL * p.ngroups * g.OHW + 7) / 8) : cdiv(p.ngroups, 8) * g.OHW;
  return large ? launch_as<ST_CONV1_DGRAD, 1, EPI_MASK, ShapeL>(p, stream) : launch_as<ST_CONV2_DGRAD, 1, EPI_MASK, ShapeS>(p, stream);
}

// Plain rows (a linear layer), N in whole 64-column tiles: the masked dgrad out[m][n] = mask[m][n] > 0 ?
// sum_k A[m][k] W[n][k] : 0 (mask given; N = 49 pixels x 64 channels of the conv stack) or the
// forward out[m][n] = bias[n] + sum_k A[m][k] W[n][k] (bias given).  The tile walk is the dgrad one
// with the column tile in the pixel's place (MODE 2).
// DX_NTP_FC_FWD_MIN_TILES: fewest 128x64 tiles for the linear layer's FORWARD to take the ring kernel.
// Its alternative, nt_dma.hip, has 128x128 tiles: at 2,048-2,560 rows (a 64-env shard's minibatch,
// config 5's shard) that is 64-80 workgroups walking 98 K steps each on a 256-CU chip -- 197 us for
// 2,560 rows where the 8,192-row minibatch takes 202 us; 128-160 ring-kernel tiles use twice the CUs.
int ntp_fc_fwd_min_tiles() {
  const int v = DX_ENV("DX_NTP_FC_FWD_MIN_TILES", 128);
  return v;
}

int launch_ntp_rows(const float *A, int lda, const float *W, const float *mask, const float *bias, float *out, int M,
                    int N, int K, float *ksplit_slabs, long long ksplit_capacity, hipStream_t stream) {
  const int min_tiles = bias != nullptr && ntp_fc_fwd_min_tiles() < ntp_min_tiles() ? ntp_fc_fwd_min_tiles() : ntp_min_tiles();
  // Forward with few tiles: every tile is a chain of K / 32 = 98 dependent steps on a chip that is a
  // quarter to half full.  K in `kparts` parts multiplies the tiles and shortens the chains; the parts
  // meet in one small reduction.  Two halves where that still leaves at most ONE tile per CU (<= 16 row
  // groups of 128): 2,048 rows 119 -> 85 us incl. the reduction, 1,152 rows 200 -> 75 us; at 2,560 /
  // 3,072 rows the doubled tiles share CUs and it loses (118 -> 131, 121 -> 137 us).  Seven parts (the
  // other divisor of the 98 steps) for <= 8 row groups, where halves would fill a quarter of the chip.
  // DX_NTP_FC_FWD_KSPLIT=0: off; =2: halves only; =7: seven parts wherever the scratch allows (experiments).
  const int ksplit_mode = DX_ENV("DX_NTP_FC_FWD_KSPLIT", 1);
  const long long whole_tiles = (M % ShapeS::BM || N % ShapeS::BN) ? 0 : 1LL * (M / ShapeS::BM) * (N / ShapeS::BN);
  int kparts = 1;
  if (ksplit_mode != 0 && bias != nullptr && whole_tiles > 0 && ksplit_slabs && whole_tiles < ntp_min_tiles()) {
    const int groups = M / ShapeS::BM, cols = N / ShapeS::BN, steps = K / kBK;
    if (ksplit_mode == 7 && steps % 7 == 0 && 7LL * M * N <= ksplit_capacity) kparts = 7;  // experiment: wherever it fits
    else if (ksplit_mode == 1 && groups <= 8 && steps % 7 == 0 && 7LL * M * N <= ksplit_capacity) kparts = 7;
    else if (steps % 2 == 0 && cdiv(groups, 8) * 2 * cols <= 32 && 2LL * M * N <= ksplit_capacity) kparts = 2;
    // 17-20 row groups (config 5's shard: 2,560 rows): seven parts again -- 2,304 rows 116 -> 93 us,
    // 2,560 rows 119 -> 107 us incl. the reduction; from 24 groups on it loses (3,072 rows 122 -> 128 us)
    else if (ksplit_mode == 1 && groups <= 20 && steps % 7 == 0 && 7LL * M * N <= ksplit_capacity) kparts = 7;
  }
  const bool ksplit = kparts > 1;
  if (!ntp_on() || N % ShapeS::BN || K % kBK || lda < K || lda % 4 || M % ShapeS::BM || M / ShapeS::BM < 8 ||
      kparts * whole_tiles < min_tiles || (mask != nullptr) == (bias != nullptr) ||
      // 32-bit per-lane byte offsets: rows of A ((tile * BM + R) * lda * 4) and of out / mask
      4LL * M * N >= (1LL << 32) || 4LL * M * lda >= (1LL << 32))
    return DX_ENOSUP;
  DX_REQUIRE(A && W && out && aligned(A, 16) && aligned(W, 16), "ntp_rows: bad operands");
  NtpArgs p;
  std::memset(&p, 0, sizeof(p));
  p.kparts = 1; p.div_kparts = make_fastdiv(1);
  const int gn = N / ShapeS::BN;
  Gather &g = p.nt.g;
  g.src = A; g.img_stride = lda; g.H = g.W = 1; g.C = K;
  g.OHW = gn; g.OW = gn; g.div_img = make_fastdiv(gn); g.div_row = make_fastdiv(gn);
  g.sy = g.sx = 1; g.nseg = 1; g.seglen = K;
  p.nt.Wp = W; p.nt.mask_src = mask; p.nt.bias = bias; p.nt.out = out; p.nt.ldc = ShapeS::BN;
  p.nt.M = M; p.nt.N = ShapeS::BN; p.nt.K = K; p.nt.ksplit = 1;
  p.TA = p.TB = 1;
  p.nimg = M;
  p.ngroups = M / ShapeS::BM;
  p.tiles_per_xcd = cdiv(p.ngroups, 8) * gn;
  // the dgrad only (measured, profiles/r03_pmc_traffic.json: fc_dgrad FETCH 484 -> 263 MB; the forward,
  // whose row blocks (128 x 3136 floats) are the bigger operand, went 155 -> 218 MB and keeps the old walk);
  // every XCD must own the same number of row groups
  p.rows_inner = mask != nullptr && p.ngroups % 8 == 0 && rows_inner_on();
  // a row-group count that is not a multiple of 8: contiguous eighths of the tile list (see NtpArgs::blocked)
  const bool blocked_on = DX_ENV("DX_NTP_BLOCKED", 1) != 0;
  p.blocked = (blocked_on && !p.rows_inner && p.ngroups % 8 != 0) ? 1 : 0;
  if (p.blocked) p.tiles_per_xcd = static_cast<int>((1LL * p.ngroups * gn + 7) / 8);
  p.groups_per_xcd = p.ngroups / 8 > 0 ? p.ngroups / 8 : 1;
  p.div_groups = make_fastdiv(p.groups_per_xcd);
  if (mask) return launch_as<ST_FC_DGRAD, 2, EPI_MASK, ShapeS>(p, stream);
  if (ksplit) {
    p.kparts = kparts; p.div_kparts = make_fastdiv(kparts);
    p.slab_bytes = 4LL * M * N;
    g.OHW = kparts * gn; g.OW = kparts * gn; g.div_img = make_fastdiv(kparts * gn); g.div_row = make_fastdiv(kparts * gn);
    g.seglen = K / kparts;
    p.PA = K / kparts;
    p.tiles_per_xcd = p.blocked ? static_cast<int>((1LL * kparts * p.ngroups * gn + 7) / 8) : cdiv(p.ngroups, 8) * kparts * gn;
    p.nt.out = ksplit_slabs;
    if (int rc = launch_as<ST_FC_FWD, 2, EPI_BIAS, ShapeF>(p, stream)) return rc;
    const PermuteJob sum{ksplit_slabs, out, 1LL * M * N, 1, 1, 1, 1, 0, 0, 0, 0, kparts, 1LL * M * N, 0};
    return launch_permute_reduce(&sum, 1, stream);
  }
  return launch_as<ST_FC_FWD, 2, EPI_BIAS, ShapeF>(p, stream);
}

}  // namespace dx
