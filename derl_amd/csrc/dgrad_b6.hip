// Data gradients of the 4x4/2 and 3x3/1 convolutions on the bf16 matrix cores at fp32 accuracy,
// image-resident: one workgroup per CU walks its share of the minibatch's images.
//
//   conv2: dY1[iy][ix][ic] = relu'(Y1) * sum over (kh, kw, oc) of dY2[iy-kh][ix-kw][oc] * W2[oc][ic][kh][kw]
//   conv1: dY0[2y'+py][2x'+px][ic] = relu'(Y0) * sum over (a, b, oc) of dY1[y'-a][x'-b][oc] * W1[oc][ic][py+2a][px+2b]
//
// (the autograd backward of derl/models.py:103-108's second and third conv w.r.t. their inputs, triggered by
// derl/alg/common.py:70; relu' = the mask of the PREVIOUS layer's ReLU, folded into the store as in the
// layer-by-layer kernels).  Arithmetic as in convstack.hip / wgrad_b6.hip: the output gradient (fp32) is split
// exactly into three bf16 planes when it is copied into LDS, the weights are pre-split by dx_cnn_pack
// (launch_dgrad_b6_pack: fragment order, one contiguous KB per wave load), and x w = the six largest of the
// nine exact products on v_mfma_f32_16x16x32_bf16 with fp32 accumulation.
//
// The gradient image sits in LDS with a ZERO border (11 x 11 pixels for both layers), so a tap is a
// compile-time byte offset from the lane's pixel and no tap needs a bounds test; pixel / row pitches
// (160 / 11 x 160 + 96 resp. + 192 bytes) keep the ds_read_b128 operand reads free of bank conflicts
// (tools/ubench/b128_pitch_search.py).  The weights never touch LDS: a wave is the only reader of its slice
// and keeps it in registers for the whole launch (96 / 108 VGPRs).  Products are formed as D[ic][pixel]: a
// lane ends up with FOUR consecutive input channels of its pixel -- one 16-byte load of the mask source, one
// 16-byte store.
//   conv1: wave = (parity class (py, px), ic tile of 16): 7 tiles of 16 pixels (10 x 10 per class) x K = 4 taps
//          x 64 oc = 8 steps of 32; no K split, no exchange.
//   conv2: wave = (ic tile, K half): 6 tiles of 16 pixels (9 x 9) x 9 of the 18 steps; the halves swap three
//          tiles each way through LDS and finish three each (convstack.hip's conv1 scheme).
// Two LDS images: the next image's rows are split and stored while this image multiplies -- one barrier
// per image.
#include "bf16_split.hpp"
#include "igemm_dev.hpp"
#include <cstdlib>

namespace dx {
namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

constexpr int kPX = 160;                       // bytes per pixel of the gradient image (64 bf16 + pad)
constexpr int kPad1 = 96, kPad2 = 192;         // row pads of the two layers' images
constexpr int kGW = 11;                        // pixels per row / rows of the bordered image (both layers)

template <int L>
struct DgGeom;
template <>
struct DgGeom<1> {  // conv1: 9 x 9 x 64 gradient, border 1; 4 classes of 10 x 10 input pixels, 32 channels
  static constexpr int RP = kGW * kPX + kPad1, PLANE = kGW * RP, BORDER = 1, GH = 9, GW = 9;
  static constexpr int NT = 7, NS = 8, MW = 10, MPIX = 100, OUT_C = 32;
  static constexpr int XBYTES = 0;
};
template <>
struct DgGeom<2> {  // conv2: 7 x 7 x 64 gradient, border 2; 9 x 9 input pixels, 64 channels
  static constexpr int RP = kGW * kPX + kPad2, PLANE = kGW * RP, BORDER = 2, GH = 7, GW = 7;
  static constexpr int NT = 6, NS = 9, MW = 9, MPIX = 81, OUT_C = 64;
  static constexpr int XBYTES = 8 * 3 * 64 * 16;  // the K halves' exchange: three accumulator tiles per wave
};

template <int L>
struct DgLayout {
  using G = DgGeom<L>;
  static constexpr int IMG = 3 * G::PLANE;            // bytes of one image (three planes)
  static constexpr int oX = 2 * IMG, END = oX + G::XBYTES;
  static constexpr int NG4 = G::GH * G::GW * 64 / 4;  // float4 pieces of a gradient image
  static constexpr int GR = (NG4 + 511) / 512;        // ... per lane
  static_assert(G::PLANE % 16 == 0 && END <= 160 * 1024, "LDS layout");
};

// byte offset of K step S (32 of the 64 output channels of one tap) from a lane's pixel
template <int L, int KH2, int S>
__device__ __forceinline__ constexpr int dg_step_offset() {
  using G = DgGeom<L>;
  constexpr int g = (L == 2 ? 9 * KH2 : 0) + S, tap = g >> 1, half = g & 1;
  constexpr int ty = L == 1 ? tap >> 1 : tap / 3, tx = L == 1 ? tap & 1 : tap % 3;
  return -ty * G::RP - tx * kPX + 64 * half;
}

__device__ __forceinline__ bf16x8 dg_bf(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }
#define DX_DG_T(a, b) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dg_bf(w[a]), dg_bf(x[b]), acc, 0, 0, 0);
__device__ __forceinline__ f32x4 dg_mac_first(f32x4 acc, const u32x4 (&w)[3], const u32x4 (&x)[3]) {
  DX_DG_T(2, 0)
  return acc;
}
__device__ __forceinline__ f32x4 dg_mac_rest(f32x4 acc, const u32x4 (&w)[3], const u32x4 (&x)[3]) {
  DX_DG_T(0, 2) DX_DG_T(1, 1) DX_DG_T(1, 0) DX_DG_T(0, 1) DX_DG_T(0, 0)
  return acc;
}
#undef DX_DG_T

template <int L, int KH2, int S>
__device__ __forceinline__ void dg_load(const uint8_t *img, int pb, u32x4 (&x)[3]) {
  constexpr int off = dg_step_offset<L, KH2, S>();
#pragma unroll
  for (int pl = 0; pl < 3; ++pl) x[pl] = *reinterpret_cast<const u32x4 *>(img + pb + off + pl * DgGeom<L>::PLANE);
}

// A (K step, tile) unit whose tap reads nothing but the zero border for EVERY pixel of the tile is not computed:
// a tile is 16 consecutive pixels of the raster (two or three image rows), a tap shifts them by whole rows / columns.
//   conv2 (9 x 9 pixels, gradient pixel (iy - kh, ix - kw) inside 7 x 7): tile 0 (rows 0-1) skips kh = 2, tile 4 (rows
//          7-8) kh = 0, tile 5 (the one pixel (8, 8)) everything but tap (2, 2): 28 of the 108 (K half, step, tile)
//          units, a quarter of the busier K half.
//   conv1 (10 x 10 pixels per parity class, gradient pixel (y' - a, x' - b) inside 9 x 9): tile 6 (row 9) skips a = 0.
template <int L, int KH2, int S, int T>
__device__ __forceinline__ constexpr bool dg_unit_active() {
  using G = DgGeom<L>;
  constexpr int g = (L == 2 ? 9 * KH2 : 0) + S, tap = g >> 1;
  constexpr int ty = L == 1 ? tap >> 1 : tap / 3, tx = L == 1 ? tap & 1 : tap % 3;  // rows / columns the tap shifts by
  for (int p = 16 * T; p < 16 * T + 16 && p < G::MPIX; ++p) {                        // any pixel of the tile with its source inside?
    const int y = p / G::MW - ty, x = p % G::MW - tx;
    if (y >= 0 && y < G::GH && x >= 0 && x < G::GW) return true;
  }
  return false;
}
template <int L, int KH2>
struct DgUnits {  // the active units in (step, tile) order
  using G = DgGeom<L>;
  int n = 0;
  int s[G::NS * G::NT] = {}, t[G::NS * G::NT] = {};
  template <int S, int T>
  constexpr void add() {
    if (dg_unit_active<L, KH2, S, T>()) { s[n] = S; t[n] = T; ++n; }
    if constexpr (T + 1 < G::NT) add<S, T + 1>();
    else if constexpr (S + 1 < G::NS) add<S + 1, 0>();
  }
  constexpr DgUnits() { add<0, 0>(); }
};

// units in order, the fragments of the unit two ahead read behind this unit's first MFMA
// (hipcc otherwise sinks every read to just before its use and waits for it at once)
template <int L, int KH2, int I>
__device__ __forceinline__ void dg_units(const uint8_t *img, const int (&pb)[DgGeom<L>::NT], const u32x4 (&w)[DgGeom<L>::NS][3],
                                         f32x4 (&acc)[DgGeom<L>::NT], u32x4 (&x0)[3], u32x4 (&x1)[3]) {
  constexpr DgUnits<L, KH2> units{};
  constexpr int S = units.s[I], T = units.t[I];
  u32x4 x2[3];
  // one scheduling region per unit: its six MFMAs with the three fragment reads of the unit two ahead INTERLEAVED, one
  // read behind each of the first three MFMAs (as a burst behind the first MFMA the reads held the wave's in-order stream
  // while the LDS took them and no MFMA issued meanwhile: convstack_train.hip)
  if constexpr (I + 2 < units.n) dg_load<L, KH2, units.s[I + 2]>(img, pb[units.t[I + 2]], x2);
  acc[T] = dg_mac_first(acc[T], w[S], x0);
  acc[T] = dg_mac_rest(acc[T], w[S], x0);
  if constexpr (I + 2 < units.n) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
  }
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (I + 1 < units.n) dg_units<L, KH2, I + 1>(img, pb, w, acc, x1, x2);
}

template <int L, int KH2>
__device__ __forceinline__ void dg_loop(const uint8_t *img, const int (&pb)[DgGeom<L>::NT], const u32x4 (&w)[DgGeom<L>::NS][3],
                                        f32x4 (&acc)[DgGeom<L>::NT]) {
  constexpr DgUnits<L, KH2> units{};
  static_assert(units.n >= 2, "at least two units");
  u32x4 x0[3], x1[3];
  dg_load<L, KH2, units.s[0]>(img, pb[units.t[0]], x0);
  dg_load<L, KH2, units.s[1]>(img, pb[units.t[1]], x1);
  dg_units<L, KH2, 0>(img, pb, w, acc, x0, x1);
}

__device__ __forceinline__ void dg_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct DgradB6Args {
  const float *g;        // the layer's output gradient (B, GH, GW, 64) fp32 NHWC
  const uint16_t *Wf;    // the weights in fragment order (launch_dgrad_b6_pack)
  const float *mask_src; // the layer's INPUT activation (post-ReLU): its sign is the mask
  float *out;            // the input gradient, same shape as mask_src
  int B;
  int descending;        // walk the minibatch from its last image down (igemm.hpp: bwd_descending)
};

template <int L>
__global__ __launch_bounds__(512) void conv_dgrad_b6_kernel(const DgradB6Args a) {
  using G = DgGeom<L>;
  using Y = DgLayout<L>;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n16 = lane & 15, kq = lane >> 4;
  const int nimg = (a.B - static_cast<int>(blockIdx.x) + static_cast<int>(gridDim.x) - 1) / static_cast<int>(gridDim.x);

  // both images zero: the borders stay zero, the interiors are overwritten per image
  for (int i = tid; i < 2 * Y::IMG / 16; i += 512) reinterpret_cast<u32x4 *>(smem)[i] = u32x4{0u, 0u, 0u, 0u};

  // ---- this wave's weight fragments: resident for the whole launch ----
  u32x4 w[G::NS][3];
#pragma unroll
  for (int s = 0; s < G::NS; ++s)
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
      w[s][pl] = *reinterpret_cast<const u32x4 *>(a.Wf + ((wave * G::NS + s) * 3 + pl) * 512 + lane * 8);

  // ---- this lane's share of a gradient image: float4 pieces tid + 512 u -> pixel (v / 16), channels 4 (v % 16) .. ----
  int gdst[Y::GR];
#pragma unroll
  for (int u = 0; u < Y::GR; ++u) {
    const int v = min(tid + 512 * u, Y::NG4 - 1), pix = v / 16;
    gdst[u] = (pix / G::GW + G::BORDER) * G::RP + (pix % G::GW + G::BORDER) * kPX + 8 * (v % 16);
  }
  f32x4 gr[Y::GR];
  auto image_of = [&](int slot) { return a.descending ? a.B - 1 - slot : slot; };
  auto fetch = [&](int slot) {
    const f32x4 *gs = reinterpret_cast<const f32x4 *>(a.g) + static_cast<long long>(image_of(slot)) * Y::NG4;
#pragma unroll
    for (int u = 0; u < Y::GR; ++u) gr[u] = gs[min(tid + 512 * u, Y::NG4 - 1)];
  };
  auto stage = [&](int buf) {  // split + store this lane's pieces into image `buf`
    uint8_t *dst = smem + buf * Y::IMG;
#pragma unroll
    for (int u = 0; u < Y::GR; ++u)
      if (tid + 512 * u < Y::NG4) {
        const Split4 s = split4(gr[u]);
        *reinterpret_cast<uint2 *>(dst + gdst[u]) = s.hi;
        *reinterpret_cast<uint2 *>(dst + gdst[u] + G::PLANE) = s.mid;
        *reinterpret_cast<uint2 *>(dst + gdst[u] + 2 * G::PLANE) = s.lo;
      }
  };

  // ---- this lane's pixels: byte offset of pixel 16 mt + n16 (k group kq) in plane 0, and its place in the output ----
  const int cls = wave >> 1, py = cls >> 1, px = cls & 1;     // conv1: parity class
  const int nt = L == 1 ? (wave & 1) : (wave & 3);            // output-channel tile of 16
  const int kh2 = L == 1 ? 0 : wave >> 2;                     // conv2: K half
  int pb[G::NT], oidx[G::NT];
#pragma unroll
  for (int mt = 0; mt < G::NT; ++mt) {
    const int p = min(16 * mt + n16, G::MPIX - 1), y = p / G::MW, x = p - y * G::MW;
    pb[mt] = (y + G::BORDER) * G::RP + (x + G::BORDER) * kPX + 16 * kq;
    // conv1: input pixel (2 y + py, 2 x + px) of the 20 x 20 image; conv2: pixel p of the 9 x 9 image
    oidx[mt] = (L == 1 ? (2 * y + py) * 20 + 2 * x + px : p) * G::OUT_C + 16 * nt + 4 * kq;
  }
  constexpr int OUT_IMG = (L == 1 ? 400 : 81) * G::OUT_C;  // floats per output image
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  fetch(blockIdx.x);
  __syncthreads();  // the zero fill is complete
  stage(0);
  if (nimg > 1) fetch(blockIdx.x + gridDim.x);
  for (int t = 0; t < nimg; ++t) {
    const int img = image_of(blockIdx.x + t * gridDim.x);
    dg_lds_barrier();  // image t is in LDS (and every wave is done with image t - 1: its buffer may be overwritten)
    if (t + 1 < nimg) {
      stage((t + 1) & 1);
      if (t + 2 < nimg) fetch(blockIdx.x + (t + 2) * gridDim.x);
    }
    const uint8_t *image = smem + (t & 1) * Y::IMG;
    const float *msrc = a.mask_src + static_cast<long long>(img) * OUT_IMG;
    float *out = a.out + static_cast<long long>(img) * OUT_IMG;
    f32x4 acc[G::NT];
#pragma unroll
    for (int mt = 0; mt < G::NT; ++mt) acc[mt] = zero4;
    if constexpr (L == 1) {
      f32x4 m[G::NT];  // the mask source of this lane's outputs, in flight under the MFMAs
#pragma unroll
      for (int mt = 0; mt < G::NT; ++mt) m[mt] = *reinterpret_cast<const f32x4 *>(msrc + oidx[mt]);
      __builtin_amdgcn_sched_barrier(0);
      dg_loop<1, 0>(image, pb, w, acc);
#pragma unroll
      for (int mt = 0; mt < G::NT; ++mt) {
        if (16 * mt + n16 >= G::MPIX) continue;
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = m[mt][j] > 0.f ? acc[mt][j] : 0.f;
        *reinterpret_cast<f32x4 *>(out + oidx[mt]) = v;
      }
    } else {
      f32x4 m[3];  // the mask source of the three tiles this wave finishes (constant indices: no scratch)
#pragma unroll
      for (int i = 0; i < 3; ++i) m[i] = *reinterpret_cast<const f32x4 *>(msrc + (kh2 == 0 ? oidx[i] : oidx[3 + i]));
      __builtin_amdgcn_sched_barrier(0);
      if (kh2 == 0) dg_loop<2, 0>(image, pb, w, acc);
      else dg_loop<2, 1>(image, pb, w, acc);
      // the K halves swap: half 0 finishes tiles 0-2, half 1 tiles 3-5; each hands the other's three over
      f32x4 *red = reinterpret_cast<f32x4 *>(smem + Y::oX);
#pragma unroll
      for (int i = 0; i < 3; ++i) red[(wave * 3 + i) * 64 + lane] = kh2 == 0 ? acc[3 + i] : acc[i];
      dg_lds_barrier();
      const int partner = wave ^ 4;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const f32x4 theirs = red[(partner * 3 + i) * 64 + lane];
        const f32x4 mine = kh2 == 0 ? acc[i] : acc[3 + i];
        const int mt = (kh2 == 0 ? 0 : 3) + i;
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = m[i][j] > 0.f ? mine[j] + theirs[j] : 0.f;
        if (16 * mt + n16 < G::MPIX) *reinterpret_cast<f32x4 *>(out + (kh2 == 0 ? oidx[i] : oidx[3 + i])) = v;
      }
      // (the exchange scratch is rewritten only after the next image's barrier at the loop top)
    }
  }
}

// fp32 weight mirror rows [N][K] -> the three bf16 planes in fragment order [wave][step][plane][lane][8]:
// lane (n16 = row of the wave's 16-row tile, kq) holds k = 32 (step) + 8 kq .. + 7.
//   conv1: four parity packs [32 ic][256] (dx_cnn_pack: pk_c1d[p]); wave = 2 p + ic tile, 8 steps.
//   conv2: one pack [64 ic][576] (pk_c2d); wave = ic tile + 4 K half, 9 steps from step 9 (K half).
__global__ __launch_bounds__(256) void dgrad_b6_pack_kernel(const float *c1d0, const float *c1d1, const float *c1d2, const float *c1d3,
                                                            const float *c2d, uint16_t *Wf1, uint16_t *Wf2) {
  constexpr int P1 = 8 * 8 * 64, P2 = 8 * 9 * 64;  // (wave, step, lane) triples of the two layers
  int q = blockIdx.x * 256 + threadIdx.x;
  const int lane = q & 63, n16 = lane & 15, kq = lane >> 4;
  const float *src;
  uint16_t *dst;
  int plane_stride;  // uint16 elements between the planes of one (wave, step)
  if (q < P1) {
    const int s = (q >> 6) & 7, wave = q >> 9, p = wave >> 1, ict = wave & 1;
    const float *pack = p == 0 ? c1d0 : p == 1 ? c1d1 : p == 2 ? c1d2 : c1d3;
    src = pack + (16 * ict + n16) * 256 + 32 * s + 8 * kq;
    dst = Wf1 + ((wave * 8 + s) * 3) * 512 + lane * 8;
    plane_stride = 512;
  } else if (q < P1 + P2) {
    q -= P1;
    const int r = q >> 6, s = r % 9, wave = r / 9, nt = wave & 3, kh2 = wave >> 2;
    src = c2d + (16 * nt + n16) * 576 + 32 * (9 * kh2 + s) + 8 * kq;
    dst = Wf2 + ((wave * 9 + s) * 3) * 512 + lane * 8;
    plane_stride = 512;
  } else {
    return;
  }
  const Split4 a = split4(*reinterpret_cast<const f32x4 *>(src)), b = split4(*reinterpret_cast<const f32x4 *>(src + 4));
  *reinterpret_cast<u32x4 *>(dst) = u32x4{a.hi.x, a.hi.y, b.hi.x, b.hi.y};
  *reinterpret_cast<u32x4 *>(dst + plane_stride) = u32x4{a.mid.x, a.mid.y, b.mid.x, b.mid.y};
  *reinterpret_cast<u32x4 *>(dst + 2 * plane_stride) = u32x4{a.lo.x, a.lo.y, b.lo.x, b.lo.y};
}

template <int L>
int launch_dg(const DgradB6Args &a, int nwg, hipStream_t stream) {
  constexpr int lds = DgLayout<L>::END;
  auto kernel = conv_dgrad_b6_kernel<L>;
  DX_LDS_OPT_IN(kernel, lds);
  hipLaunchKernelGGL(kernel, dim3(nwg), dim3(512), lds, stream, a);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

}  // namespace

// DX_DGRAD_B6=0: the fp32-MFMA kernels (ntp.hip / igemm_pix)
bool dgrad_b6_on() {
  return DX_ENV("DX_DGRAD_B6", 1) != 0;
}
// uint16 elements of the fragment-ordered dgrad weights of layer 1 / 2
long long dgrad_b6_pack_elems(int layer) { return (layer == 1 ? 8LL * 8 : 8LL * 9) * 3 * 512; }

int launch_dgrad_b6_pack(const float *const c1d[4], const float *c2d, uint16_t *Wf1, uint16_t *Wf2, hipStream_t stream) {
  DX_REQUIRE(c1d && c1d[0] && c1d[1] && c1d[2] && c1d[3] && c2d && Wf1 && Wf2, "dgrad_b6_pack: bad arguments");
  DX_REQUIRE(aligned(c1d[0], 16) && aligned(c1d[1], 16) && aligned(c1d[2], 16) && aligned(c1d[3], 16) && aligned(c2d, 16) &&
                 aligned(Wf1, 16) && aligned(Wf2, 16), "dgrad_b6_pack: operands must be 16-byte aligned");
  constexpr int total = 8 * 8 * 64 + 8 * 9 * 64;
  hipLaunchKernelGGL(dgrad_b6_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, c1d[0], c1d[1], c1d[2], c1d[3], c2d, Wf1, Wf2);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

// layer 1: dY0 (B, 20, 20, 32) from dY1 (B, 9, 9, 64), masked by Y0; layer 2: dY1 from dY2 (B, 7, 7, 64), masked by Y1
int launch_dgrad_b6(int layer, const float *g, const uint16_t *Wf, const float *mask_src, float *out, int B, hipStream_t stream,
                    bool descending) {
  DX_REQUIRE((layer == 1 || layer == 2) && g && Wf && mask_src && out && B > 0, "dgrad_b6: bad arguments");
  DX_REQUIRE(aligned(g, 16) && aligned(Wf, 16) && aligned(mask_src, 16) && aligned(out, 16), "dgrad_b6: operands must be 16-byte aligned");
  int cus = 0;
  if (int rc = device_cus(&cus)) return rc;
  const DgradB6Args a{g, Wf, mask_src, out, B, descending ? 1 : 0};
  const int nwg = B < cus ? B : cus;
  return layer == 1 ? launch_dg<1>(a, nwg, stream) : launch_dg<2>(a, nwg, stream);
}

}  // namespace dx
