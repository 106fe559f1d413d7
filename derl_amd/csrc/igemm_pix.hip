// Data-gradient GEMMs of the strided convolutions with the zero taps SKIPPED.
//
// A dgrad row is an input pixel; its K axis is one run per kernel tap, and taps that fall outside
// the output-gradient image are zeros (conv2: 40 % of all taps, conv1: 19 %).  The generic kernel
// (igemm.hip) tiles 64 consecutive pixels of an image, whose valid taps differ, so it multiplies
// the zeros.  Here a 64-row tile is ONE pixel position of 64 different images:
//   * the set of valid taps is uniform over the workgroup, so invalid taps are never loaded or
//     multiplied -- no masking anywhere (rows past the batch read a clamped image and are not
//     stored);
//   * workgroups are numbered so that the ones an XCD receives (blockIdx round-robins over the
//     8 XCDs) walk all pixels of ONE group of 64 images: the taps shared by neighbouring pixels
//     are re-read from that XCD's L2, not from HBM.
// Same MFMA tile machinery as igemm_nt_kernel (LDS rows of 32 k + 4 pad, register prefetch).  128
// images x 128 columns per workgroup (64x64 per wave) measured 647 vs 508 us on conv1: occupancy wins.
#include "igemm_dev.hpp"
#include <cstdlib>

namespace dx {
namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int TAG, int BM, int BN, int WM, int WN, int EPI>
__global__ __launch_bounds__(256) void igemm_nt_pix_kernel(const NTArgs a, int nimg, int ngroups) {
  constexpr int BK = 32, LD = BK + 4, TPR = BK / 4, RPP = 256 / TPR;  // 32 tile rows per pass
  constexpr int NWN = BN / WN, TM = WM / 32, TN = WN / 32;
  static_assert((BM / WM) * NWN == 4, "four waves");
  constexpr int APASS = BM / RPP, BPASS = BN / RPP;
  __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * LD];
  float *As = smem;
  float *Bs = smem + BM * LD;
  const Gather &g = a.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave / NWN) * WM, wn0 = (wave % NWN) * WN;
  const int l8 = tid % TPR, lr = tid / TPR;

  // workgroup -> (image group, pixel)
  const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
  const int gh = seq / g.OHW, pix = seq - gh * g.OHW;
  const int grp = gh * 8 + xcd;
  if (grp >= ngroups) return;  // uniform: the grid is padded to a multiple of 8 groups
  const int oy = pix / g.OW, ox = pix - oy * g.OW;
  const int y0 = oy * g.sy, x0 = ox * g.sx;
  uint32_t tapmask = 0;  // uniform
  for (int s = 0; s < g.nseg; ++s) {
    const int yy = y0 + g.seg_dy[s], xx = x0 + g.seg_dx[s];
    const bool in = !g.check || (static_cast<unsigned>(yy) < static_cast<unsigned>(g.H) &&
                                 static_cast<unsigned>(xx) < static_cast<unsigned>(g.W));
    tapmask |= (in ? 1u : 0u) << s;
  }
  const long long pixoff = static_cast<long long>(y0 * g.W + x0) * g.C + 4 * l8;
  const float *arow[APASS];
#pragma unroll
  for (int p = 0; p < APASS; ++p) {
    const int img = min(grp * BM + p * RPP + lr, nimg - 1);  // BM / 32 passes of 32 images
    arow[p] = static_cast<const float *>(g.src) + static_cast<long long>(img) * g.img_stride + pixoff;
  }
  const float *wrow[BPASS];
#pragma unroll
  for (int p = 0; p < BPASS; ++p) wrow[p] = a.Wp + static_cast<long long>(p * RPP + lr) * a.K + 4 * l8;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // (seg, q): position of the K step being fetched; only valid taps are visited
  int seg = tapmask ? __builtin_ctz(tapmask) : g.nseg, q = 0;
  f32x4 araw[APASS], braw[BPASS];
  auto fetch = [&]() {
    const long long so = static_cast<long long>(g.seg_off[seg]) + q;
    const int kt = seg * g.seglen + q;
#pragma unroll
    for (int p = 0; p < APASS; ++p) araw[p] = *reinterpret_cast<const f32x4 *>(arow[p] + so);
#pragma unroll
    for (int p = 0; p < BPASS; ++p) braw[p] = *reinterpret_cast<const f32x4 *>(wrow[p] + kt);
  };
  if (seg < g.nseg) fetch();
  while (seg < g.nseg) {
    __syncthreads();  // everyone finished reading the previous tile
#pragma unroll
    for (int p = 0; p < APASS; ++p) *reinterpret_cast<f32x4 *>(&As[(p * RPP + lr) * LD + 4 * l8]) = araw[p];
#pragma unroll
    for (int p = 0; p < BPASS; ++p) *reinterpret_cast<f32x4 *>(&Bs[(p * RPP + lr) * LD + 4 * l8]) = braw[p];
    __syncthreads();
    q += BK;
    if (q >= g.seglen) {  // next valid tap
      q = 0;
      const uint32_t rest = (seg + 1 < 32) ? (tapmask >> (seg + 1)) : 0u;
      seg = rest ? seg + 1 + __builtin_ctz(rest) : g.nseg;
    }
    if (seg < g.nseg) fetch();  // latency hides under the MFMAs below
    const int lrow = lane & 31, lk = 4 * (lane >> 5);
#pragma unroll
    for (int qd = 0; qd < BK / 8; ++qd) {
      f32x4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        af[i] = *reinterpret_cast<const f32x4 *>(&As[(wm0 + 32 * i + lrow) * LD + 8 * qd + lk]);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        bf[j] = *reinterpret_cast<const f32x4 *>(&Bs[(wn0 + 32 * j + lrow) * LD + 8 * qd + lk]);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
        }
    }
  }

  // epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5); rows are images
  const OutMap &om = a.om;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int nb = wn0 + 32 * j;  // uniform per wave
    long long pixo;               // element offset of the output pixel inside an image
    long long imgo;               // elements per output image
    int col = nb + (lane & 31);
    bool inside = true;
    if (om.enabled) {
      const int grp_n = nb / om.chan;
      const int py = grp_n / om.osx, px = grp_n - py * om.osx;
      const int yy = oy * om.osy + py, xx = ox * om.osx + px;
      inside = yy < om.OUT_H && xx < om.OUT_W;
      pixo = static_cast<long long>(yy * om.OUT_W + xx) * a.ldc;
      imgo = static_cast<long long>(om.OUT_H) * om.OUT_W * a.ldc;
      col = nb - grp_n * om.chan + (lane & 31);
    } else {
      pixo = static_cast<long long>(pix) * a.ldc;
      imgo = static_cast<long long>(g.OHW) * a.ldc;
    }
    if (!inside) continue;  // uniform
    const bool full = (grp + 1) * BM <= nimg;  // uniform: all but the last image group
    if (full) {
      // no per-element exec-mask branches, and all ReLU-mask loads are in flight before the first
      // store (stores may alias mask_src for the compiler: a load after a store waits vmcnt(0))
      float mk[TM][16];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int img = grp * BM + wm0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          mk[i][r] = EPI == EPI_MASK ? a.mask_src[img * imgo + pixo + col] : 1.f;
        }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int img = grp * BM + wm0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          a.out[img * imgo + pixo + col] = mk[i][r] > 0.f ? acc[i][j][r] : 0.f;
        }
      continue;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int img = grp * BM + wm0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (img >= nimg) continue;
        const long long o = img * imgo + pixo + col;
        float v = acc[i][j][r];
        if (EPI == EPI_MASK) v = a.mask_src[o] > 0.f ? v : 0.f;
        a.out[o] = v;
      }
  }
}

template <int TAG, int BM, int BN, int WM, int WN, int EPI>
int launch_pix_as(const NTArgs &a, int nimg, hipStream_t stream) {
  const int ngroups = cdiv(nimg, BM);
  const int blocks = cdiv(ngroups, 8) * 8 * a.g.OHW;
  hipLaunchKernelGGL((igemm_nt_pix_kernel<TAG, BM, BN, WM, WN, EPI>), dim3(blocks), dim3(256), 0, stream, a,
                     nimg, ngroups);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

bool pix_enabled() {  // DX_DGRAD_PIX=0: dgrads on the generic kernel (zero taps multiplied)
  const int v = DX_ENV("DX_DGRAD_PIX", 1);
  return v != 0;
}

}  // namespace

// nimg images of g.OHW gathered pixels each (a.M == nimg * g.OHW).  DX_ENOSUP = not covered.
int launch_nt_pix(const NTArgs &a, int nimg, int epi, int stage, hipStream_t stream) {
  const Gather &g = a.g;
  if (!pix_enabled()) return DX_ENOSUP;
  if (epi != EPI_MASK || a.ksplit != 1 || g.idx || g.seglen % 32 || g.nseg > kMaxSeg || nimg < 64) return DX_ENOSUP;
  if (static_cast<long long>(nimg) * g.OHW != a.M) return DX_ENOSUP;
  if (a.om.enabled && (a.om.OHW != g.OHW || a.om.OW != g.OW || a.om.chan % 32)) return DX_ENOSUP;
  switch (stage) {
    case ST_CONV2_DGRAD:
      if (a.N == 64) return launch_pix_as<ST_CONV2_DGRAD, 64, 64, 32, 32, EPI_MASK>(a, nimg, stream);
      break;
    case ST_CONV1_DGRAD:
      if (a.N == 128) return launch_pix_as<ST_CONV1_DGRAD, 64, 128, 32, 64, EPI_MASK>(a, nimg, stream);
      break;
    default: break;
  }
  return DX_ENOSUP;
}

}  // namespace dx
