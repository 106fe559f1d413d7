// Direct (image-resident) weight gradients of the 4x4/2 and 3x3/1 convolutions on fp32 MFMA.
//
//   dW[oc][kh][kw][ic] = sum over (img, oy, ox) of dY[img][oy][ox][oc] * X[img][oy*S+kh][ox*S+kw][ic]
//
// (the autograd backward of derl/models.py:103-108's second and third conv, triggered by
// derl/alg/common.py:70).  The implicit-GEMM wgrad (igemm_tn_kernel) streams an im2col view of X:
// every input pixel is fetched once per kernel tap that touches it, by DIFFERENT workgroups (the K
// blocks of one M slice), and with ~150 workgroups per XCD each streaming 24 KB per step the 4 MB
// L2 cannot hold the sibling's rows until the sibling asks for them -- measured 1.11 GB (conv2) and
// 1.43 GB (conv1) of HBM-side fetches per launch for 0.27 / 0.59 GB of unique bytes
// (profiles/r01_pmc_traffic.json).  Here the reuse is inside the workgroup: ONE image's input
// (9x9x64 or 20x20x32 floats) and output gradient are copied into LDS once -- both are contiguous
// NHWC blocks, so the copy is a flat stream of 1 KiB LDS-DMA pieces (global_load_lds_dwordx4: no
// staging registers, no ds_write), double-buffered: image i + 1 lands while image i multiplies --
// and every tap's operand is read from LDS with a per-lane address.  HBM traffic = the unique
// bytes; ONE barrier per image instead of two per 32-row K step.
//
// MFMA mapping (v_mfma_f32_32x32x2_f32, exact fp32 fma chain): the contraction index is the
// output pixel, two pixels per instruction (lanes 0-31 pixel 2j, lanes 32-63 pixel 2j+1; an odd
// pixel count is padded with a zero gradient row).  One instruction accumulates a 32 (oc) x 32
// (columns) tile, where a column is a (tap, input channel) pair.  On this part every non-MFMA
// instruction of a wave costs matrix-pipe time (measured: the loop ran at 1 / (1 + 0.02 x
// instructions per MFMA) of the MFMA rate whatever was done about latency), so the operand reads
// are as wide as the layout allows: ONE ds_read_b128 per lane fetches 4 consecutive channels of
// one tap, the 32 lanes of a pixel covering IC/4 lanes x (128/IC) taps = 128 columns = the B
// operands of FOUR MFMAs (column n of tile j = tap T[n / (IC/4)], channel 4 (n % (IC/4)) + j;
// the taps of a read are chosen so that its 16-lane bank groups stay conflict-free), and one
// ds_read_b32 fetches the gradient fragment shared by all of a wave's tiles.  The waves of a
// workgroup split the (oc tile, column tile) set evenly: conv2 4 waves x 9 tiles (oc half; taps
// 4g..4g+3 by two b128 reads + half of tap 8 by one b32 read), conv1 8 waves x 4 tiles (oc half;
// kernel row g by one b128 read).  Which share a wave owns is a template parameter of the loop
// body.  The pixel -> LDS address map is a small table built once (byte offsets of the pixel's
// input window and gradient row): the loop carries no address arithmetic but one add per read,
// and its loads run one pair (table: two pairs) ahead of the MFMAs.  Accumulators stay in
// registers over all images of the workgroup (persistent loop), then go to that workgroup's slab
// [oc][kh][kw][ic] with 16-byte stores -- the layout the implicit-GEMM kernel writes, so gradient
// finalisation is unchanged (deterministic, no atomics).
#include "igemm_dev.hpp"
#include <cstdlib>

namespace dx {
namespace {

using f4 = __attribute__((ext_vector_type(4))) float;
using u2 = __attribute__((ext_vector_type(2))) unsigned;

template <int IC, int OC, int IH, int IW, int OH, int OW, int NW>
struct WdLayout {
  static constexpr int OHW = OH * OW, XN = IH * IW * IC, GN = OHW * OC;
  static constexpr int XV = XN / 4, TOTAL = XV + GN / 4;  // 16-byte units (IC, OC multiples of 32)
  static constexpr int PIECES = (TOTAL + 63) / 64;         // 1 KiB LDS-DMA pieces (64 lanes x 16 B)
  static constexpr int BUF = PIECES * 256;                 // floats per image buffer
  static constexpr int NP = (OHW + 1) / 2;                 // pixel pairs per image
  static constexpr int TBL = 2 * (NP + 2);                 // table entries per buffer (read-ahead)
  // floats: [buffer 0][buffer 1][zero row OC][table 2 x TBL x 2][reduction scratch 64 NW]
  static constexpr int ZERO = 2 * BUF, TABLE = ZERO + OC, RED = TABLE + 4 * TBL, END = RED + 64 * NW;
};

// Geometry is a template parameter (IH x IW input, OH x OW output): other input sizes take the
// implicit-GEMM kernel (wgrad_direct_supported).
template <int GRP, int IC, int OC, int KH, int KW, int S, int IH, int IW, int OH, int OW, int NW>
__device__ __forceinline__ void conv_wgrad_direct_body(const WgradDirectArgs &a, float *smem) {
  using L = WdLayout<IC, OC, IH, IW, OH, OW, NW>;
  constexpr int OCT = OC / 32, ICT = IC / 32, TAPS = KH * KW;
  constexpr int COMBOS = TAPS * ICT;  // (tap, ic tile) pairs of one oc tile
  constexpr int GROUPS = NW / OCT;    // waves that share an oc tile
  constexpr int CPW = COMBOS / GROUPS;
  static_assert(OC == 64 && OCT * GROUPS == NW && COMBOS % GROUPS == 0, "tile set does not split over the waves");
  static_assert(OW >= 2 && (OH - 1) * S + KH <= IH && (OW - 1) * S + KW <= IW, "geometry");
  constexpr int OHW = L::OHW, XN = L::XN, GN = L::GN, NP = L::NP;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int och = wave % OCT;
  const int hi = lane >> 5, l31 = lane & 31;
  const char *lds = reinterpret_cast<const char *>(smem);

  // pixel table, one per image buffer: entry q = {byte offset of pixel q's input window, byte
  // offset of its gradient row}; pixels past the image (the pad of an odd image and the
  // read-ahead) get the last pixel's window and the zero row
  for (int e = tid; e < 2 * L::TBL; e += 64 * NW) {
    const int buf = e / L::TBL, q = e % L::TBL;
    const int qc = q < OHW ? q : OHW - 1;
    const unsigned x = (buf * L::BUF + ((qc / OW) * (S * IW) + (qc % OW) * S) * IC) * 4;
    const unsigned g = (q < OHW ? buf * L::BUF + XN + q * OC : L::ZERO) * 4;
    reinterpret_cast<u2 *>(smem + L::TABLE)[e] = u2{x, g};
  }
  if (tid < OC) smem[L::ZERO + tid] = 0.f;

  f32x16 acc[CPW];
#pragma unroll
  for (int c = 0; c < CPW; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  float bias_acc = 0.f;

  // Column tiles of this wave.  NB128 b128 reads of TPR taps each (4 tiles per read) and, for
  // conv2, one b32 tile (tap 8, ic half GRP).  Lane n of a pixel reads tap slot n / LPT, channels
  // 4 (n % LPT) .. + 3.
  constexpr int LPT = IC / 4, TPR = 32 / LPT;
  constexpr bool HAS32 = (TAPS % TPR) != 0;                  // conv2: 9 taps = 4 pairs + 1
  constexpr int NB128 = (CPW - (HAS32 ? 1 : 0)) / 4;
  static_assert(NB128 * 4 + (HAS32 ? 1 : 0) == CPW, "tiles per wave");
  // tap of slot s of read r: conv1 (TPR 4) kernel row GRP in the order kw 0, 2, 1, 3 (their
  // offsets are 0, 0, 32, 32 dwords mod 64: the bank pattern a 16-lane group needs); conv2 (TPR 2)
  // taps 4 GRP + 2r + s (all offsets are multiples of 64 dwords)
  auto tap_of = [](int r, int slot) {
    return TPR == 4 ? (GRP * NB128 + r) * 4 + (slot == 1 ? 2 : slot == 2 ? 1 : slot) : (GRP * NB128 + r) * 2 + slot;
  };
  auto tap_off = [](int tap) { return ((tap / KW) * IW + (tap % KW)) * IC * 4; };  // bytes
  const unsigned glane = (och * 32 + l31) * 4;
  unsigned xl[NB128];
#pragma unroll
  for (int r = 0; r < NB128; ++r) {
    unsigned off = 0;
#pragma unroll
    for (int slot = 0; slot < TPR; ++slot)
      if (l31 / LPT == slot) off = tap_off(tap_of(r, slot));
    xl[r] = off + (l31 % LPT) * 16;
  }
  const unsigned x32 = tap_off(TAPS - 1) + (GRP * 32 + l31) * 4;  // HAS32: tap 8, ic half GRP
  // operands of the pixel whose table entry is E (the b128 results stay whole 4-register tuples:
  // an MFMA reads its operand straight out of the tuple)
  struct Ops {
    float g, x32;
    f4 v[NB128];
  };
#define DX_WD_LOAD(E, O)                                                                         \
  {                                                                                              \
    O.g = *reinterpret_cast<const float *>(lds + (E[1] + glane));                                \
    _Pragma("unroll") for (int r = 0; r < NB128; ++r)                                            \
        O.v[r] = *reinterpret_cast<const f4 *>(lds + (E[0] + xl[r]));                            \
    if (HAS32) O.x32 = *reinterpret_cast<const float *>(lds + (E[0] + x32));                     \
  }
#define DX_WD_MMA(O, C) \
  acc[C] = __builtin_amdgcn_mfma_f32_32x32x2f32(O.g, (C) < 4 * NB128 ? O.v[(C) / 4][(C) % 4] : O.x32, acc[C], 0, 0, 0);
  // one pair: the first MFMA goes out, in its shadow the operands of the next pair and the table
  // entry two pairs ahead are requested, then the remaining MFMAs
  // (sched_group_barrier masks: 0x8 MFMA, 0x100 DS read)
#define DX_WD_STEP(OC_, ECUR, ON_, ENEXT, JNEXT2)                                                \
  {                                                                                              \
    DX_WD_MMA(OC_, 0)                                                                            \
    DX_WD_LOAD(ENEXT, ON_)                                                                       \
    ECUR = tbl[2 * (JNEXT2)];                                                                    \
    _Pragma("unroll") for (int c = 1; c < CPW; ++c) DX_WD_MMA(OC_, c)                            \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                           \
    __builtin_amdgcn_sched_group_barrier(0x100, NB128 + 3, 0);                                   \
    __builtin_amdgcn_sched_group_barrier(0x008, CPW - 1, 0);                                     \
  }
  // wave w copies pieces w, w + NW, ... of an image: lane l of piece p brings 16-byte unit
  // 64p + l of the flat image [input | gradient]; the lanes past the end of the last piece stay
  // inactive (nothing is written behind the image).  Every fill is a uniform base plus the lane's
  // constant offset (dma_piece: no address VALU between the MFMAs)
#define DX_WD_COPY(IMG, DST)                                                                     \
  {                                                                                              \
    const f4 *xs_ = reinterpret_cast<const f4 *>(a.x + static_cast<long long>(IMG) * XN);        \
    const f4 *gs_ = reinterpret_cast<const f4 *>(a.g + static_cast<long long>(IMG) * GN);        \
    for (int p = wave_u; p < L::PIECES; p += NW) {                                               \
      const int u0_ = p * 64; /* uniform: first 16-byte unit of the piece */                      \
      float *d_ = (DST) + p * 256;                                                               \
      if (u0_ + 64 <= L::XV) {                                                                   \
        dma_piece(xs_ + u0_, lane * 16u, d_);                                                    \
      } else if (u0_ >= L::XV) {                                                                 \
        if (u0_ + 64 <= L::TOTAL || lane < L::TOTAL - u0_) dma_piece(gs_ + (u0_ - L::XV), lane * 16u, d_); \
      } else { /* the piece that straddles [input | gradient]: two masked fills of the same KiB */ \
        if (lane < L::XV - u0_) dma_piece(xs_ + u0_, lane * 16u, d_);                            \
        else if (lane < L::TOTAL - u0_) dma_piece(gs_ - (L::XV - u0_), lane * 16u, d_);          \
      }                                                                                          \
    }                                                                                            \
  }

  int cur = 0;
  if (static_cast<int>(blockIdx.x) < a.B) DX_WD_COPY(blockIdx.x, smem)
  for (int img = blockIdx.x; img < a.B; img += gridDim.x, cur ^= 1) {
    // this wave's pieces of image `img` have landed; after the barrier so have everyone's (and
    // the table), and every wave is done reading the other buffer, which the next copy reuses
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (img + static_cast<int>(gridDim.x) < a.B && !(kDiag && (a.diag & 1))) DX_WD_COPY(img + gridDim.x, smem + (cur ^ 1) * L::BUF)
    // bias gradient = column sums of the output gradient: thread (oc = lane, rows wave + NW t)
    const float *Gs = smem + cur * L::BUF + XN;
    for (int r = wave; r < OHW; r += NW) bias_acc += Gs[r * OC + lane];

    const u2 *tbl = reinterpret_cast<const u2 *>(smem + L::TABLE) + cur * L::TBL + hi;  // + 2j
    u2 e0 = tbl[0], e1 = tbl[2];
    Ops o0, o1;
    DX_WD_LOAD(e0, o0)
#pragma unroll 1
    for (int j = 0; j + 1 < NP; j += 2) {
      DX_WD_STEP(o0, e0, o1, e1, j + 2)
      DX_WD_STEP(o1, e1, o0, e0, j + 3)
    }
    if (NP & 1) {  // compile-time: the last pair of an odd count sits in set 0
#pragma unroll
      for (int c = 0; c < CPW; ++c) DX_WD_MMA(o0, c)
    }
  }
#undef DX_WD_COPY
#undef DX_WD_STEP
#undef DX_WD_MMA
#undef DX_WD_LOAD

  // accumulators -> this workgroup's slab [oc][tap][ic]
  constexpr int K = TAPS * IC;
  float *slab = a.slab + static_cast<long long>(blockIdx.x) * OC * K;
  if (kDiag && (a.diag & 2) && acc[0][0] != 12345.678f) return;
  // tile (r, j), column n = tap T[n / LPT], channel 4 (n % LPT) + j: the four tiles of a read
  // hold four consecutive channels -> one 16-byte store per accumulator register
#pragma unroll
  for (int r = 0; r < NB128; ++r) {
    int tap = 0;
#pragma unroll
    for (int slot = 0; slot < TPR; ++slot)
      if (l31 / LPT == slot) tap = tap_of(r, slot);
    float *col = slab + tap * IC + 4 * (l31 % LPT);
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
      const int n = och * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * hi;
      *reinterpret_cast<f4 *>(col + static_cast<long long>(n) * K) =
          f4{acc[4 * r][rr], acc[4 * r + 1][rr], acc[4 * r + 2][rr], acc[4 * r + 3][rr]};
    }
  }
  if (HAS32) {
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
      const int n = och * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * hi;
      slab[static_cast<long long>(n) * K + (TAPS - 1) * IC + GRP * 32 + l31] = acc[CPW - 1][rr];
    }
  }
  float *red = smem + L::RED;
  red[tid] = bias_acc;
  __syncthreads();
  if (tid < OC) {
    float sum = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) sum += red[tid + 64 * w];
    a.bias_slab[static_cast<long long>(blockIdx.x) * OC + tid] = sum;
  }
}

// The waves that share an oc tile split the (tap, ic tile) set; which share a wave owns is
// wave-uniform, so the body is instantiated per share and selected by one scalar branch.
template <int TAG, int IC, int OC, int KH, int KW, int S, int IH, int IW, int OH, int OW, int NW>
__global__ __launch_bounds__(64 * NW, 2) void conv_wgrad_direct_kernel(const WgradDirectArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int OCT = OC / 32, GROUPS = NW / OCT;
  static_assert(GROUPS == 2 || GROUPS == 4, "two or four shares per oc tile");
  const int grp = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6)) / OCT;
#define DX_WD_BODY(G) conv_wgrad_direct_body<(G) < GROUPS ? (G) : 0, IC, OC, KH, KW, S, IH, IW, OH, OW, NW>(a, smem)
  if (grp == 0) DX_WD_BODY(0);
  else if (grp == 1) DX_WD_BODY(1);
  else if (GROUPS == 4 && grp == 2) DX_WD_BODY(2);
  else if (GROUPS == 4) DX_WD_BODY(3);
#undef DX_WD_BODY
}

template <int TAG, int IC, int OC, int KH, int KW, int S, int IH, int IW, int OH, int OW, int NW>
int launch_as(const WgradDirectArgs &a, int nwg, hipStream_t stream) {
  constexpr int lds = WdLayout<IC, OC, IH, IW, OH, OW, NW>::END * 4;
  static_assert(lds <= 160 * 1024, "LDS budget");
  auto kernel = conv_wgrad_direct_kernel<TAG, IC, OC, KH, KW, S, IH, IW, OH, OW, NW>;
  DX_LDS_OPT_IN(kernel, lds);
  hipLaunchKernelGGL(kernel, dim3(nwg), dim3(64 * NW), lds, stream, a);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

}  // namespace

// the instantiated geometries: the conv stack of an 84 x 84 observation (20x20x32 -> 9x9x64 ->
// 7x7x64).  conv2: 4 waves, 70 KB of LDS, two workgroups per CU; conv1: 8 waves, 146 KB, one.
bool wgrad_direct_supported(int IH, int IW, int IC, int OH, int OW, int OC, int KH, int KW, int S) {
  const bool c1 = IH == 20 && IW == 20 && IC == 32 && OH == 9 && OW == 9 && OC == 64 && KH == 4 && KW == 4 && S == 2;
  const bool c2 = IH == 9 && IW == 9 && IC == 64 && OH == 7 && OW == 7 && OC == 64 && KH == 3 && KW == 3 && S == 1;
  return c1 || c2;
}

// Workgroups (= slabs) of a launch.  conv2: two 4-wave workgroups per CU; below 2,048 images one per
// CU (four images each at 1,024): the same kernel time and half the slabs for the finalize pass.
int wgrad_direct_workgroups(int stage, long long batch) {
#if DX_DIAG
  const int forced = DX_ENV("DX_WD_NWG", 0);  // timing experiments
  if (forced > 0 && stage == ST_CONV2_WGRAD) return forced;
#endif
  return stage == ST_CONV1_WGRAD || batch < 2048 ? 256 : 512;
}

int launch_wgrad_direct(const WgradDirectArgs &a_in, int stage, int nwg, hipStream_t stream) {
  WgradDirectArgs a = a_in;
#if DX_DIAG
  a.diag = DX_ENV("DX_WD_DIAG", 0);
#else
  a.diag = 0;
#endif
  DX_REQUIRE(a.x && a.g && a.slab && a.bias_slab && a.B > 0 && nwg > 0 && nwg <= a.B,
             "wgrad_direct: bad arguments (B=%d, workgroups=%d)", a.B, nwg);
  DX_REQUIRE(aligned(a.x, 16) && aligned(a.g, 16), "wgrad_direct: activations must be 16-byte aligned");
  switch (stage) {
    case ST_CONV1_WGRAD:
      DX_REQUIRE(wgrad_direct_supported(a.IH, a.IW, 32, a.OH, a.OW, 64, 4, 4, 2), "wgrad_direct: conv1 geometry");
      return launch_as<ST_CONV1_WGRAD, 32, 64, 4, 4, 2, 20, 20, 9, 9, 8>(a, nwg, stream);
    case ST_CONV2_WGRAD:
      DX_REQUIRE(wgrad_direct_supported(a.IH, a.IW, 64, a.OH, a.OW, 64, 3, 3, 1), "wgrad_direct: conv2 geometry");
      return launch_as<ST_CONV2_WGRAD, 64, 64, 3, 3, 1, 9, 9, 7, 7, 4>(a, nwg, stream);
    default:
      return fail(DX_EINVAL, "wgrad_direct: unknown stage %d", stage);
  }
}

}  // namespace dx
