// Weight gradients of the 4x4/2 and 3x3/1 convolutions on the bf16 matrix cores at fp32 accuracy.
//
//   dW[oc][kh][kw][ic] = sum over (img, oy, ox) of dY[img][oy][ox][oc] * X[img][oy*S+kh][ox*S+kw][ic]
//
// (the autograd backward of derl/models.py:103-108's second and third conv, triggered by
// derl/alg/common.py:70).  Same contract as wgrad_direct.hip (image-resident, persistent workgroups, one
// slab [oc][kh][kw][ic] + one bias row per workgroup, deterministic), other arithmetic: BOTH operands are
// fp32 values, each is split EXACTLY into three bf16 terms (hi + mid + lo) when it is copied into LDS, and
// x g = the six largest of the nine exact products (convstack.hip's rule: the dropped three are below 2^-23
// of the product), v_mfma_f32_16x16x32_bf16 accumulating in fp32.  Six MFMAs of 16 cycles do the work of
// sixteen fp32 MFMAs of 16 cycles: 2.7x less matrix time than v_mfma_f32_32x32x2_f32.
//
// The contraction index is the output pixel.  Both operands therefore want 8 consecutive PIXELS of one
// channel per lane, while the images are pixel-major ([pixel][channel], as they come from memory): the
// fragments are read with ds_read_b64_tr_b16, which hands every 16-lane group a 4-row x 16-column block
// column-major -- each lane supplies the address of ITS row, so a "row" can be the input pixel of any tap
// (oy*S+kh, ox*S+kw): the im2col matrix is never materialised.  Slot k = 8 g + 4 r + q of a 32-pixel K step
// holds pixel 32 ks + 16 r + 4 g + q (the same permutation on both operands): a 32-lane half then reads 8
// CONSECUTIVE pixels, and with convstack.hip's pixel / row pitches (80 / 20 x 80 + 16 bytes for the 32-channel
// image, 160 / 9 x 160 + 192 for the 64-channel one, 160 for the gradient rows) every read is free of bank
// conflicts (tools/ubench/tr_pitch_search.py).  Pixels past the image read a ZERO gradient row.
//
//   conv1 (81 pixels = 3 K steps): wave w owns taps 2 w, 2 w + 1 (4 column tiles of 16) x 4 oc tiles.
//   conv2 (49 pixels = 2 K steps): wave w owns oc tiles 2 (w & 1) .. + 1 x column tiles 9 (w >> 1) .. + 8.
// The next image's fp32 rows travel in registers while this image multiplies; accumulators stay in
// registers over all images of the workgroup.
#include "bf16_split.hpp"
#include "igemm_dev.hpp"
#include <cstdlib>

namespace dx {
namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using s16x4 = __attribute__((ext_vector_type(4))) short;
using lds_s16x4 = __attribute__((address_space(3))) s16x4;

template <int L>
struct B6Geom;
template <>
struct B6Geom<1> {
  static constexpr int IC = 32, KH = 4, KW = 4, S = 2, IH = 20, IW = 20, OH = 9, OW = 9, KS = 3;
  static constexpr int PX = 80, RP = IW * PX + 16;
  static constexpr int OCT = 4, CT = 4;  // tiles per wave: oc x column
};
template <>
struct B6Geom<2> {
  static constexpr int IC = 64, KH = 3, KW = 3, S = 1, IH = 9, IW = 9, OH = 7, OW = 7, KS = 2;
  static constexpr int PX = 160, RP = IW * PX + 192;
  static constexpr int OCT = 2, CT = 9;
};
constexpr int kOC = 64, kPG = 160;  // output channels; byte pitch of a gradient row (64 bf16 + pad)

template <int L>
struct B6Layout {
  using G = B6Geom<L>;
  static constexpr int OHW = G::OH * G::OW, KK = G::KH * G::KW * G::IC;
  static constexpr int XPLANE = G::IH * G::RP, GROWS = 32 * G::KS, GPLANE = GROWS * kPG;
  static constexpr int oX = 0, oG = 3 * XPLANE, END = oG + 3 * GPLANE;  // bytes
  static constexpr int NX4 = G::IH * G::IW * G::IC / 4, NG4 = OHW * kOC / 4;  // float4 pieces of an image
  static constexpr int XR = (NX4 + 511) / 512, GR = (NG4 + 511) / 512;        // ... per lane
  static_assert(XPLANE % 16 == 0 && GPLANE % 16 == 0 && END <= 160 * 1024, "LDS layout");
  static_assert(512 * 16 <= END, "the bias reduction reuses the image's LDS");
};

__device__ __forceinline__ bf16x8 b6_frag(s16x4 lo, s16x4 hi) {
  const __attribute__((ext_vector_type(8))) short v = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ s16x4 b6_tr(const uint8_t *smem, int off) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(smem + off));  // (flat -> LDS address space)
}
// the three planes of one 16 x 32 operand fragment: rows (pixels) at byte offsets r0 / r1 (K slots 4 r + q)
__device__ __forceinline__ void b6_read(const uint8_t *smem, int r0, int r1, int plane, bf16x8 (&f)[3]) {
#pragma unroll
  for (int pl = 0; pl < 3; ++pl) f[pl] = b6_frag(b6_tr(smem, r0 + pl * plane), b6_tr(smem, r1 + pl * plane));
}
// g x = the six largest exact products, smallest first (planes 0 hi, 1 mid, 2 lo)
__device__ __forceinline__ f32x4 b6_mac(f32x4 acc, const bf16x8 (&g)[3], const bf16x8 (&x)[3]) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(g[2], x[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(g[0], x[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(g[1], x[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(g[1], x[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(g[0], x[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(g[0], x[0], acc, 0, 0, 0);
  return acc;
}

template <int L>
__global__ __launch_bounds__(512) void conv_wgrad_b6_kernel(const WgradDirectArgs a) {
  using G = B6Geom<L>;
  using Y = B6Layout<L>;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nimg = (a.B - static_cast<int>(blockIdx.x) + static_cast<int>(gridDim.x) - 1) / static_cast<int>(gridDim.x);

  // zero gradient rows past the image (never overwritten)
  for (int i = tid; i < 3 * (Y::GROWS - Y::OHW) * (kPG / 16); i += 512) {
    const int pl = i / ((Y::GROWS - Y::OHW) * (kPG / 16)), rem = i % ((Y::GROWS - Y::OHW) * (kPG / 16));
    *reinterpret_cast<u32x4 *>(smem + Y::oG + pl * Y::GPLANE + Y::OHW * kPG + 16 * rem) = u32x4{0u, 0u, 0u, 0u};
  }

  // ---- this lane's share of an image's rows: float4 pieces tid + 512 u; LDS byte offsets in plane 0 ----
  // conv1's 32-channel image: a pixel is 8 lanes x 8 bytes and the 80-byte pixel pitch (conflict-free for the transposed
  // reads at stride 2) puts two CONSECUTIVE pixels of a 16-lane write group 16 bytes into each other mod 128 (16 % of the
  // kernel's LDS cycles were bank conflicts, round-6 counters).  Pixels 4 apart sit 320 = 64 (mod 128) bytes apart: lane
  // groups take the pieces with pixel bits 0 and 2 exchanged (the same 64-piece block of the image: the loads stay coalesced).
  const bool exchange = L == 1 && !(kDiag && (a.diag & 4));  // (diag flavour, DX_WB6_DIAG=4: the plain order, an A/B switch)
  auto piece_of = [&](int v) { return exchange ? ((v & ~0x28) | ((v & 0x08) << 2) | ((v & 0x20) >> 2)) : v; };
  static_assert(L != 1 || Y::NX4 % 64 == 0, "the exchange stays inside whole 64-piece blocks");
  int xdst[Y::XR], gdst[Y::GR];
#pragma unroll
  for (int u = 0; u < Y::XR; ++u) {
    const int v = piece_of(min(tid + 512 * u, Y::NX4 - 1)), pix = v / (G::IC / 4), c4 = v % (G::IC / 4);
    xdst[u] = Y::oX + (pix / G::IW) * G::RP + (pix % G::IW) * G::PX + 8 * c4;
  }
#pragma unroll
  for (int u = 0; u < Y::GR; ++u) {
    const int v = min(tid + 512 * u, Y::NG4 - 1);
    gdst[u] = Y::oG + (v / 16) * kPG + 8 * (v % 16);
  }
  f32x4 xr[Y::XR], gr[Y::GR];
  auto fetch = [&](int slot) {  // slot = blockIdx.x + t gridDim.x; the image: from the last one down when `descending`
    const int img = a.descending ? a.B - 1 - slot : slot;
    const f32x4 *xs = reinterpret_cast<const f32x4 *>(a.x) + static_cast<long long>(img) * Y::NX4;
    const f32x4 *gs = reinterpret_cast<const f32x4 *>(a.g) + static_cast<long long>(img) * Y::NG4;
#pragma unroll
    for (int u = 0; u < Y::GR; ++u) gr[u] = gs[min(tid + 512 * u, Y::NG4 - 1)];
#pragma unroll
    for (int u = 0; u < Y::XR; ++u) xr[u] = xs[piece_of(min(tid + 512 * u, Y::NX4 - 1))];
  };

  // ---- operand addresses (bytes): lane (g, q, p4) reads row 4 r + q of its group's block, columns 4 p4 .. ----
  const int g4 = lane >> 4, q = (lane >> 2) & 3, p4 = lane & 3;
  int grow[G::KS][2], xrow[G::KS][2];
#pragma unroll
  for (int ks = 0; ks < G::KS; ++ks)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int p = 32 * ks + 16 * r + 4 * g4 + q;  // the pixel in K slot 8 g + 4 r + q
      grow[ks][r] = Y::oG + p * kPG + 8 * p4;
      const int pc = min(p, Y::OHW - 1), py = pc / G::OW, px = pc - py * G::OW;  // (past the image: any valid window, x 0)
      xrow[ks][r] = Y::oX + G::S * py * G::RP + G::S * px * G::PX + 8 * p4;
    }

  f32x4 acc[G::OCT][G::CT];
#pragma unroll
  for (int i = 0; i < G::OCT; ++i)
#pragma unroll
    for (int j = 0; j < G::CT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};  // column sums of the gradient rows this lane copies (channels 4 (tid % 16) ..)

  // this wave's tiles: oc tile index i -> channels 16 (oc_first + i); column tile j -> (tap, first input channel)
  const int oc_first = L == 1 ? 0 : 2 * (wave & 1);
  const int ct_first = L == 1 ? 4 * wave : 9 * (wave >> 1);
  constexpr int CPT = G::IC / 16;  // column tiles per tap

  fetch(blockIdx.x);
  for (int t = 0; t < nimg; ++t) {
    if (t > 0) __syncthreads();  // every wave is done with the previous image
#pragma unroll
    for (int u = 0; u < Y::GR; ++u)
      if (tid + 512 * u < Y::NG4) {
        const Split4 s = split4(gr[u]);
        *reinterpret_cast<uint2 *>(smem + gdst[u]) = s.hi;
        *reinterpret_cast<uint2 *>(smem + gdst[u] + Y::GPLANE) = s.mid;
        *reinterpret_cast<uint2 *>(smem + gdst[u] + 2 * Y::GPLANE) = s.lo;
        bsum += gr[u];
      }
#pragma unroll
    for (int u = 0; u < Y::XR; ++u)
      if (tid + 512 * u < Y::NX4) {
        const Split4 s = split4(xr[u]);
        *reinterpret_cast<uint2 *>(smem + xdst[u]) = s.hi;
        *reinterpret_cast<uint2 *>(smem + xdst[u] + Y::XPLANE) = s.mid;
        *reinterpret_cast<uint2 *>(smem + xdst[u] + 2 * Y::XPLANE) = s.lo;
      }
    if (t + 1 < nimg) fetch(blockIdx.x + (t + 1) * gridDim.x);
    __syncthreads();

    // (K step, column tile) units in order; one scheduling region per unit: its 6 OCT MFMAs with the six transposed
    // reads of the NEXT unit's input fragment interleaved, one read behind each of the first MFMAs (left to the
    // compiler the reads sat in front of the MFMAs that use them and were waited for at once; as a burst they hold
    // the wave's in-order stream while the LDS takes them: convstack_train.hip)
    auto xoff = [&](int unit) {
      const int ks = unit / G::CT, ct = ct_first + unit % G::CT, tap = ct / CPT, c0 = 16 * (ct % CPT);
      return int2{xrow[ks][0] + (tap / G::KW) * G::RP + (tap % G::KW) * G::PX + 2 * c0,
                  xrow[ks][1] + (tap / G::KW) * G::RP + (tap % G::KW) * G::PX + 2 * c0};
    };
    bf16x8 xf[3];
    {
      const int2 o = xoff(0);
      b6_read(smem, o.x, o.y, Y::XPLANE, xf);
    }
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks) {
      bf16x8 gf[G::OCT][3];
#pragma unroll
      for (int i = 0; i < G::OCT; ++i) b6_read(smem, grow[ks][0] + 32 * (oc_first + i), grow[ks][1] + 32 * (oc_first + i), Y::GPLANE, gf[i]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < G::CT; ++j) {
        const int unit = ks * G::CT + j;
        bf16x8 xfn[3];
        const bool more = unit + 1 < G::KS * G::CT;  // (compile-time after unrolling)
        if (more) {
          const int2 o = xoff(unit + 1);
          b6_read(smem, o.x, o.y, Y::XPLANE, xfn);
        }
#pragma unroll
        for (int i = 0; i < G::OCT; ++i) acc[i][j] = b6_mac(acc[i][j], gf[i], xf);
        if (more) {
#pragma unroll
          for (int r = 0; r < 6; ++r) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
          __builtin_amdgcn_sched_group_barrier(0x008, 6 * G::OCT - 6, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (more) {
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) xf[pl] = xfn[pl];
        }
      }
    }
  }

  // ---- slab [oc][kh][kw][ic]: D[i = 4 (lane >> 4) + r][j = lane & 15] = (channel 16 tile + i, column c0 + j) ----
  float *slab = a.slab + static_cast<long long>(blockIdx.x) * kOC * Y::KK;
#pragma unroll
  for (int i = 0; i < G::OCT; ++i)
#pragma unroll
    for (int j = 0; j < G::CT; ++j) {
      const int ct = ct_first + j, tap = ct / CPT, c0 = 16 * (ct % CPT);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        slab[(16 * (oc_first + i) + 4 * (lane >> 4) + r) * Y::KK + tap * G::IC + c0 + (lane & 15)] = acc[i][j][r];
    }
  // ---- bias gradient: the 32 lanes with the same tid % 16 hold partial sums of the same four channels ----
  __syncthreads();
  reinterpret_cast<f32x4 *>(smem)[tid] = bsum;
  __syncthreads();
  if (tid < kOC) {
    float s = 0.f;
    for (int m = 0; m < 32; ++m) s += reinterpret_cast<const float *>(smem)[(16 * m + (tid >> 2)) * 4 + (tid & 3)];
    a.bias_slab[static_cast<long long>(blockIdx.x) * kOC + tid] = s;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// conv2's weight gradient with the contraction STREAMED over the workgroup's images (round 6).  An image has 49 output
// pixels: as two K steps of 32 per image (above) 15 of the 64 slots multiply zeros -- executed / useful 1.31, at a stage
// that runs at the package power limit.  Every lane supplies the address of ITS pixel to the transposed reads, so a K step
// need not stop at an image: here the workgroup's images form ONE run of pixels f = 49 n + p (n = the workgroup's n-th
// image), K step k covers f = 32 k .. 32 k + 31 whatever images those fall into, and LDS holds TWO images (slots n & 1:
// 68 KB each; conv1's 143 KB image does not allow that).  A step runs as soon as its last pixel's image is in LDS
// (1 or 2 steps per image: 49 / 32 = 1.53 on average instead of 2), the run's last step is padded with a zero gradient
// row.  Same tiles per wave, same fragments, same six products; the lane's row addresses are formed per step (two
// divisions by constants) instead of once per launch.
// ---------------------------------------------------------------------------------------------------------------------
struct B6Stream {
  using G = B6Geom<2>;
  static constexpr int OHW = 49, KK = 9 * 64;
  // The workgroup's n-th image sits (n & 7) x 160 bytes into its slot's planes: a pixel is 160 bytes, the pitches make the
  // bank phase advance by 5 x 32 bytes per pixel ALONG an image (tools/ubench/tr_pitch_search.py), and 49 pixels = 1 (mod 8)
  // -- with that rotation the 8 consecutive pixels of the run a 32-lane half reads stay conflict-free across an image's
  // end (without it 14 % of the reads were 2-way conflicted: the next image starts where pixel 48 sits, mod 256 bytes).
  static constexpr int ROT = 7 * kPG, ZROW = OHW + 7;                      // rotation room; the zero gradient row
  static constexpr int XPLANE = G::IH * G::RP + ROT, GPLANE = (ZROW + 1) * kPG;
  static constexpr int oX = 0, oG = 3 * XPLANE, SLOT = (oG + 3 * GPLANE + 255) / 256 * 256, END = 2 * SLOT;
  static constexpr int NX4 = G::IH * G::IW * G::IC / 4, NG4 = OHW * kOC / 4;
  static constexpr int XR = (NX4 + 511) / 512, GR = (NG4 + 511) / 512;
  static_assert(XPLANE % 16 == 0 && GPLANE % 16 == 0 && END <= 160 * 1024 && 512 * 16 <= END, "LDS layout");
};

__global__ __launch_bounds__(512) void conv2_wgrad_stream_kernel(const WgradDirectArgs a) {
  using G = B6Geom<2>;
  using Y = B6Stream;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nimg = (a.B - static_cast<int>(blockIdx.x) + static_cast<int>(gridDim.x) - 1) / static_cast<int>(gridDim.x);

  if (tid < 60) {  // the zero gradient row of each slot's planes: 2 slots x 3 planes x 160 bytes
    const int sl = tid / 30, rem = tid % 30;
    *reinterpret_cast<u32x4 *>(smem + sl * Y::SLOT + Y::oG + (rem / 10) * Y::GPLANE + Y::ZROW * kPG + 16 * (rem % 10)) = u32x4{0u, 0u, 0u, 0u};
  }

  // ---- this lane's share of an image's rows: float4 pieces tid + 512 u; byte offsets in plane 0 of a slot ----
  int xdst[Y::XR], gdst[Y::GR];
#pragma unroll
  for (int u = 0; u < Y::XR; ++u) {
    const int v = min(tid + 512 * u, Y::NX4 - 1), pix = v / (G::IC / 4), c4 = v % (G::IC / 4);
    xdst[u] = Y::oX + (pix / G::IW) * G::RP + (pix % G::IW) * G::PX + 8 * c4;
  }
#pragma unroll
  for (int u = 0; u < Y::GR; ++u) {
    const int v = min(tid + 512 * u, Y::NG4 - 1);
    gdst[u] = Y::oG + (v / 16) * kPG + 8 * (v % 16);
  }
  f32x4 xr[Y::XR], gr[Y::GR];
  auto fetch = [&](int n) {  // the workgroup's n-th image: from the last one down when `descending`
    const int slot = static_cast<int>(blockIdx.x) + n * static_cast<int>(gridDim.x);
    const int img = a.descending ? a.B - 1 - slot : slot;
    const f32x4 *xs = reinterpret_cast<const f32x4 *>(a.x) + static_cast<long long>(img) * Y::NX4;
    const f32x4 *gs = reinterpret_cast<const f32x4 *>(a.g) + static_cast<long long>(img) * Y::NG4;
#pragma unroll
    for (int u = 0; u < Y::GR; ++u) gr[u] = gs[min(tid + 512 * u, Y::NG4 - 1)];
#pragma unroll
    for (int u = 0; u < Y::XR; ++u) xr[u] = xs[min(tid + 512 * u, Y::NX4 - 1)];
  };

  const int g4 = lane >> 4, q = (lane >> 2) & 3, p4 = lane & 3;
  constexpr int OCT = G::OCT, CT = G::CT, CPT = G::IC / 16;
  f32x4 acc[OCT][CT];
#pragma unroll
  for (int i = 0; i < OCT; ++i)
#pragma unroll
    for (int j = 0; j < CT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
  const int oc_first = 2 * (wave & 1), ct_first = 9 * (wave >> 1);

  // K step k of the run: the lane's two rows (K slots 4 r + q: pixel 32 k + 16 r + 4 g + q of the run), then the wave's
  // tiles -- (column tile) units in order, the next unit's input fragment read behind this unit's first MFMAs
  auto kstep = [&](int k) {
    int grow[2], xrow[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int f = 32 * k + 16 * r + 4 * g4 + q, nf = f / Y::OHW;
      const bool in = nf < nimg;  // (past the run: the zero gradient row x pixel 0 of the LAST image -- real, finite data;
                                  // anything else of the slots may be pad bytes or never-written LDS, and NaN x 0 is NaN)
      const int n = in ? nf : nimg - 1, p = in ? f - nf * Y::OHW : 0;
      const int base = (n & 1) * Y::SLOT, rot = (n & 7) * kPG, py = p / G::OW, px = p - py * G::OW;
      grow[r] = base + Y::oG + (in ? rot + p * kPG : Y::ZROW * kPG) + 8 * p4;
      xrow[r] = base + Y::oX + rot + py * G::RP + px * G::PX + 8 * p4;
    }
    auto xoff = [&](int j) {
      const int ct = ct_first + j, tap = ct / CPT, c0 = 16 * (ct % CPT);
      return int2{xrow[0] + (tap / G::KW) * G::RP + (tap % G::KW) * G::PX + 2 * c0,
                  xrow[1] + (tap / G::KW) * G::RP + (tap % G::KW) * G::PX + 2 * c0};
    };
    bf16x8 gf[OCT][3], xf[3];
#pragma unroll
    for (int i = 0; i < OCT; ++i) b6_read(smem, grow[0] + 32 * (oc_first + i), grow[1] + 32 * (oc_first + i), Y::GPLANE, gf[i]);
    {
      const int2 o = xoff(0);
      b6_read(smem, o.x, o.y, Y::XPLANE, xf);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < CT; ++j) {
      bf16x8 xfn[3];
      const bool more = j + 1 < CT;  // (compile-time after unrolling)
      if (more) {
        const int2 o = xoff(j + 1);
        b6_read(smem, o.x, o.y, Y::XPLANE, xfn);
      }
#pragma unroll
      for (int i = 0; i < OCT; ++i) acc[i][j] = b6_mac(acc[i][j], gf[i], xf);
      if (more) {
#pragma unroll
        for (int r = 0; r < 6; ++r) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 6 * OCT - 6, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (more) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) xf[pl] = xfn[pl];
      }
    }
  };

  fetch(0);
  int k_done = 0;
  for (int t = 0; t < nimg; ++t) {
    __syncthreads();  // every wave is done with the steps run so far: none of them still reads image t - 2's slot
    uint8_t *slot = smem + (t & 1) * Y::SLOT + (t & 7) * kPG;  // (the image's rotation inside its slot)
#pragma unroll
    for (int u = 0; u < Y::GR; ++u)
      if (tid + 512 * u < Y::NG4) {
        const Split4 s = split4(gr[u]);
        *reinterpret_cast<uint2 *>(slot + gdst[u]) = s.hi;
        *reinterpret_cast<uint2 *>(slot + gdst[u] + Y::GPLANE) = s.mid;
        *reinterpret_cast<uint2 *>(slot + gdst[u] + 2 * Y::GPLANE) = s.lo;
        bsum += gr[u];
      }
#pragma unroll
    for (int u = 0; u < Y::XR; ++u)
      if (tid + 512 * u < Y::NX4) {
        const Split4 s = split4(xr[u]);
        *reinterpret_cast<uint2 *>(slot + xdst[u]) = s.hi;
        *reinterpret_cast<uint2 *>(slot + xdst[u] + Y::XPLANE) = s.mid;
        *reinterpret_cast<uint2 *>(slot + xdst[u] + 2 * Y::XPLANE) = s.lo;
      }
    if (t + 1 < nimg) fetch(t + 1);
    __syncthreads();
    // the steps whose 32 pixels are all in LDS now (the last image: the padded last step too)
    const int k_end = t + 1 < nimg ? (Y::OHW * (t + 1)) / 32 : (Y::OHW * nimg + 31) / 32;
    for (int k = k_done; k < k_end; ++k) kstep(k);
    k_done = k_end;
  }

  // ---- slab [oc][kh][kw][ic] and the bias row: as conv_wgrad_b6_kernel<2> ----
  float *slab = a.slab + static_cast<long long>(blockIdx.x) * kOC * Y::KK;
#pragma unroll
  for (int i = 0; i < OCT; ++i)
#pragma unroll
    for (int j = 0; j < CT; ++j) {
      const int ct = ct_first + j, tap = ct / CPT, c0 = 16 * (ct % CPT);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        slab[(16 * (oc_first + i) + 4 * (lane >> 4) + r) * Y::KK + tap * G::IC + c0 + (lane & 15)] = acc[i][j][r];
    }
  __syncthreads();
  reinterpret_cast<f32x4 *>(smem)[tid] = bsum;
  __syncthreads();
  if (tid < kOC) {
    float s = 0.f;
    for (int m = 0; m < 32; ++m) s += reinterpret_cast<const float *>(smem)[(16 * m + (tid >> 2)) * 4 + (tid & 3)];
    a.bias_slab[static_cast<long long>(blockIdx.x) * kOC + tid] = s;
  }
}

// DX_WGRAD_B6_STREAM=0: conv2's weight gradient image by image (two K steps per image, the second half empty)
bool wgrad_b6_stream_on() { return DX_ENV("DX_WGRAD_B6_STREAM", 1) != 0; }

template <int L>
int launch_b6(const WgradDirectArgs &a, int nwg, hipStream_t stream) {
  constexpr int lds = B6Layout<L>::END;
  auto kernel = conv_wgrad_b6_kernel<L>;
  DX_LDS_OPT_IN(kernel, lds);
  hipLaunchKernelGGL(kernel, dim3(nwg), dim3(512), lds, stream, a);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

}  // namespace

// DX_WGRAD_B6=0: the fp32-MFMA kernels of wgrad_direct.hip
bool wgrad_b6_on() {
  return DX_ENV("DX_WGRAD_B6", 1) != 0;
}

// conv1 (stage ST_CONV1_WGRAD) / conv2 weight gradient of an 84 x 84 observation's conv stack; nwg <= one per CU
int launch_wgrad_b6(const WgradDirectArgs &a_in, int stage, int nwg, hipStream_t stream) {
  WgradDirectArgs a = a_in;
#if DX_DIAG
  a.diag = DX_ENV("DX_WB6_DIAG", 0);
#endif
  DX_REQUIRE(a.x && a.g && a.slab && a.bias_slab && a.B > 0 && nwg > 0 && nwg <= a.B, "wgrad_b6: bad arguments (B=%d, workgroups=%d)",
             a.B, nwg);
  DX_REQUIRE(aligned(a.x, 16) && aligned(a.g, 16), "wgrad_b6: activations must be 16-byte aligned");
  if (stage == ST_CONV1_WGRAD) {
    DX_REQUIRE(a.IH == 20 && a.IW == 20 && a.OH == 9 && a.OW == 9, "wgrad_b6: conv1 geometry");
    return launch_b6<1>(a, nwg, stream);
  }
  DX_REQUIRE(stage == ST_CONV2_WGRAD && a.IH == 9 && a.IW == 9 && a.OH == 7 && a.OW == 7, "wgrad_b6: conv2 geometry");
  if (wgrad_b6_stream_on()) {
    DX_LDS_OPT_IN(conv2_wgrad_stream_kernel, B6Stream::END);
    hipLaunchKernelGGL(conv2_wgrad_stream_kernel, dim3(nwg), dim3(512), B6Stream::END, stream, a);
    DX_LAUNCH_CHECK();
    return DX_OK;
  }
  return launch_b6<2>(a, nwg, stream);
}

}  // namespace dx
