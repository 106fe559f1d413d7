// fp32-MFMA implicit GEMM kernels (v_mfma_f32_32x32x2_f32: exact f32, k-ordered fma chain).
//
// igemm_nt : C[m][n]  = epi(sum_k A(m,k) Wp[n][k])       conv forward, dgrad, linear layers
// igemm_tn : dW[n][k] = sum_m G[m][n] A(m,k)              wgrad, reduction split over M
//
// A 32x32x2 MFMA takes ONE float of A and ONE of B per lane (lane l: row/col l&31, k = l>>5),
// and issues every 64 cycles per SIMD, so operand traffic is small: the kernels stage tiles
// through LDS with a register prefetch of the next tile (global loads overlap the MFMAs).
//   NT: LDS rows are K-contiguous, stride 36 floats; a lane reads 4 consecutive k of its
//       row with one ds_read_b128 (conflict-free at stride 36) and feeds 4 MFMAs.
//   TN: LDS rows are the reduction index m; a lane reads one float per MFMA, consecutive
//       lanes -> consecutive banks.
#include "igemm_dev.hpp"
#include "tail_greduce_dev.hpp"
#include <cstdlib>

namespace dx {
namespace {

// ------------------------------------------------------------------------------------
// NT kernel
// ------------------------------------------------------------------------------------
// TAG only names the instantiation: every network stage gets its own kernel symbol, so a
// rocprofv3 --stats row is one stage (one shape), not a mix of layers.
template <int TAG, int BM, int BN, int WM, int WN, bool AU8, int EPI, int BK = 32>
__global__ __launch_bounds__(64 * (BM / WM) * (BN / WN)) void igemm_nt_kernel(const NTArgs a) {
  constexpr int NWN = BN / WN;
  constexpr int NT = 64 * (BM / WM) * NWN;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int LD = BK + 4;    // BK k + 4 pad floats: conflict-free ds_read_b128
  constexpr int TPR = BK / 4;   // lanes per tile row (4 k each)
  constexpr int RPP = NT / TPR; // tile rows filled per pass
  static_assert(BM % RPP == 0, "BM must be a multiple of the rows per pass");
  constexpr int APASS = BM / RPP;
  constexpr int BPASS = (BN + RPP - 1) / RPP;
  __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * LD];
  float *As = smem;
  float *Bs = smem + BM * LD;

  const Gather &g = a.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave / NWN) * WM, wn0 = (wave % NWN) * WN;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int kper = a.K / a.ksplit;
  const int kbeg = blockIdx.z * kper, kend = kbeg + kper;
  const int l8 = tid % TPR, lr = tid / TPR;

  RowPos rows[APASS];
#pragma unroll
  for (int p = 0; p < APASS; ++p) {
    const int m = m0 + p * RPP + lr;
    rows[p] = decode_row(g, m, m < a.M);
  }
  const float *wrow[BPASS];
  bool wvalid[BPASS];
#pragma unroll
  for (int p = 0; p < BPASS; ++p) {
    const int nr = p * RPP + lr;
    wvalid[p] = nr < BN && (n0 + nr) < a.N;
    wrow[p] = a.Wp + static_cast<long long>(wvalid[p] ? n0 + nr : 0) * a.K + 4 * l8;
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  int seg = kbeg / g.seglen, q = kbeg - seg * g.seglen;
  typename Raw<AU8>::type araw[APASS];
  float4 braw[BPASS];
  uint32_t aok = 0;  // bit p: pass p of the tile in flight holds real data
  auto fetch = [&](int kt) {
    const int sseg = __builtin_amdgcn_readfirstlane(seg);
    const long long so = static_cast<long long>(g.seg_off[sseg]) + q + 4 * l8;
    aok = 0;
#pragma unroll
    for (int p = 0; p < APASS; ++p) {
      const bool ok = (rows[p].okmask >> sseg) & 1u;
      aok |= (ok ? 1u : 0u) << p;
      araw[p] = load_raw<AU8>(g.src, ok ? rows[p].base + so : 0);
    }
#pragma unroll
    for (int p = 0; p < BPASS; ++p) braw[p] = *reinterpret_cast<const float4 *>(wrow[p] + kt);
  };
  fetch(kbeg);

  for (int kt = kbeg; kt < kend; kt += BK) {
    __syncthreads();  // everyone finished reading the previous tile
#pragma unroll
    for (int p = 0; p < APASS; ++p)
      *reinterpret_cast<float4 *>(&As[(p * RPP + lr) * LD + 4 * l8]) =
          masked(to_float4(araw[p]), (aok >> p) & 1u);
#pragma unroll
    for (int p = 0; p < BPASS; ++p)
      if (p * RPP + lr < BN)
        *reinterpret_cast<float4 *>(&Bs[(p * RPP + lr) * LD + 4 * l8]) = masked(braw[p], wvalid[p]);
    __syncthreads();
    if (kt + BK < kend) {  // prefetch the next tile; its latency hides under the MFMAs
      q += BK;
      if (q >= g.seglen) { q = 0; ++seg; }
      fetch(kt + BK);
    }
    const int lrow = lane & 31, lk = 4 * (lane >> 5);
#pragma unroll
    for (int qd = 0; qd < BK / 8; ++qd) {
      float4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        af[i] = *reinterpret_cast<const float4 *>(&As[(wm0 + 32 * i + lrow) * LD + 8 * qd + lk]);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        bf[j] = *reinterpret_cast<const float4 *>(&Bs[(wn0 + 32 * j + lrow) * LD + 8 * qd + lk]);
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

  // epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  float *out = a.out + (a.ksplit > 1 ? blockIdx.z * a.slab_stride : 0);
  const OutMap &om = a.om;
  auto finish = [&](float v, long long o, int n) {
    if (a.ksplit == 1) {
      if (EPI == EPI_BIAS || EPI == EPI_BIAS_RELU || EPI == EPI_BIAS_TANH) v += a.bias[n];
      if (EPI == EPI_BIAS_RELU) v = v > 0.f ? v : 0.f;
      if (EPI == EPI_BIAS_TANH) v = tanhf(v);
      if (EPI == EPI_MASK) v = a.mask_src[o] > 0.f ? v : 0.f;
      if (EPI == EPI_DTANH) { const float y = a.mask_src[o]; v *= 1.f - y * y; }
    } else if ((EPI == EPI_BIAS || EPI == EPI_BIAS_RELU) && blockIdx.z == 0) {
      v += a.bias[n];  // split-K partials: the bias rides on slab 0 (no activation allowed)
    }
    out[o] = v;
  };
  if (!om.enabled && m0 + BM <= a.M && n0 + BN <= a.N) {
    // Interior tile (uniform): no per-element guards (each is an exec-mask branch), and every
    // load of the epilogue is issued before the first store -- the compiler must assume the
    // stores alias bias / mask_src, so a load placed after a store waits for vmcnt(0) each time
    // (32 serialised round trips per workgroup otherwise).
    constexpr bool kBias = EPI == EPI_BIAS || EPI == EPI_BIAS_RELU || EPI == EPI_BIAS_TANH;
    constexpr bool kAux = EPI == EPI_MASK || EPI == EPI_DTANH;
    float bias_n[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn0 + 32 * j + (lane & 31);
      bias_n[j] = (kBias && (a.ksplit == 1 || blockIdx.z == 0)) ? a.bias[n] : 0.f;
    }
    const long long row0 = static_cast<long long>(m0 + wm0 + 4 * (lane >> 5)) * a.ldc + n0 + wn0 + (lane & 31);
    float aux[kAux ? TM : 1][kAux ? 16 : 1][kAux ? TN : 1];
    if (kAux && a.ksplit == 1) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            aux[kAux ? i : 0][kAux ? r : 0][kAux ? j : 0] =
                a.mask_src[row0 + static_cast<long long>(32 * i + (r & 3) + 8 * (r >> 2)) * a.ldc + 32 * j];
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          float v = acc[i][j][r] + bias_n[j];
          if (a.ksplit == 1) {
            if (EPI == EPI_BIAS_RELU) v = v > 0.f ? v : 0.f;
            if (EPI == EPI_BIAS_TANH) v = tanhf(v);
            if (EPI == EPI_MASK) v = aux[kAux ? i : 0][kAux ? r : 0][kAux ? j : 0] > 0.f ? v : 0.f;
            if (EPI == EPI_DTANH) { const float y = aux[kAux ? i : 0][kAux ? r : 0][kAux ? j : 0]; v *= 1.f - y * y; }
          }
          out[row0 + static_cast<long long>(32 * i + (r & 3) + 8 * (r >> 2)) * a.ldc + 32 * j] = v;
        }
    return;
  }
  // per column sub-tile: pixel-group offsets of the output map (uniform per j: scalar division
  // once per sub-tile instead of runtime divisions per element)
  int gy[TN], gx[TN], cc[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int nb = __builtin_amdgcn_readfirstlane(n0 + wn0 + 32 * j);
    gy[j] = gx[j] = 0;
    cc[j] = nb + (lane & 31);
    if (om.enabled) {
      const int g = nb / om.chan;
      gy[j] = g / om.osx;
      gx[j] = g - gy[j] * om.osx;
      cc[j] = nb - g * om.chan + (lane & 31);
    }
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wm0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (m >= a.M) continue;
      uint32_t img = 0, oy = 0, ox = 0;
      if (om.enabled) {
        img = fdiv(static_cast<uint32_t>(m), om.div_img);
        const uint32_t rem = static_cast<uint32_t>(m) - img * om.OHW;
        oy = fdiv(rem, om.div_row);
        ox = rem - oy * om.OW;
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn0 + 32 * j + (lane & 31);
        if (n >= a.N) continue;
        long long o;
        if (om.enabled) {
          const int yy = static_cast<int>(oy) * om.osy + gy[j];
          const int xx = static_cast<int>(ox) * om.osx + gx[j];
          if (yy >= om.OUT_H || xx >= om.OUT_W) continue;
          o = ((static_cast<long long>(img) * om.OUT_H + yy) * om.OUT_W + xx) * a.ldc + cc[j];
        } else {
          o = static_cast<long long>(m) * a.ldc + n;
        }
        finish(acc[i][j][r], o, n);
      }
    }
}

// ------------------------------------------------------------------------------------
// NT kernel for SMALL problems (rollout batches): 64x64 tile, 64-deep K steps and a register
// ring that keeps PD K tiles in flight.  With few rows per launch the chip is mostly empty and
// a K step is bounded by one memory round trip (~2 us measured), not by its 16 MFMAs per wave;
// PD tiles in flight divide the number of exposed round trips by PD.
// ------------------------------------------------------------------------------------
template <int TAG, int EPI, int PD>
__global__ __launch_bounds__(256) void igemm_nt_small_kernel(const NTArgs a) {
  constexpr int BM = 64, BN = 64, BK = 64, LD = BK + 4, TPR = BK / 4, RPP = 256 / TPR;
  constexpr int APASS = BM / RPP, BPASS = BN / RPP;  // 4 and 4
  __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * LD];
  float *As = smem;
  float *Bs = smem + BM * LD;
  const Gather &g = a.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int kper = a.K / a.ksplit;
  const int kbeg = blockIdx.z * kper, kend = kbeg + kper;
  const int l8 = tid % TPR, lr = tid / TPR;

  RowPos rows[APASS];
#pragma unroll
  for (int p = 0; p < APASS; ++p) {
    const int m = m0 + p * RPP + lr;
    rows[p] = decode_row(g, m, m < a.M);
  }
  const float *wrow[BPASS];
  bool wvalid[BPASS];
#pragma unroll
  for (int p = 0; p < BPASS; ++p) {
    const int nr = p * RPP + lr;
    wvalid[p] = (n0 + nr) < a.N;
    wrow[p] = a.Wp + static_cast<long long>(wvalid[p] ? n0 + nr : 0) * a.K + 4 * l8;
  }
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  float4 araw[PD][APASS], braw[PD][BPASS];
  uint32_t aok[PD];
  int seg = kbeg / g.seglen, q = kbeg - seg * g.seglen;  // position of the NEXT tile to fetch
  auto fetch = [&](int kt, float4 (&ar)[APASS], float4 (&br)[BPASS], uint32_t &ok_bits) {
    const int sseg = __builtin_amdgcn_readfirstlane(seg);
    const long long so = static_cast<long long>(g.seg_off[sseg]) + q + 4 * l8;
    ok_bits = 0;
#pragma unroll
    for (int p = 0; p < APASS; ++p) {
      const bool ok = (rows[p].okmask >> sseg) & 1u;
      ok_bits |= (ok ? 1u : 0u) << p;
      ar[p] = load_raw<false>(g.src, ok ? rows[p].base + so : 0);
    }
#pragma unroll
    for (int p = 0; p < BPASS; ++p) br[p] = *reinterpret_cast<const float4 *>(wrow[p] + kt);
    q += BK;
    if (q >= g.seglen) { q = 0; ++seg; }
  };
#pragma unroll
  for (int d = 0; d < PD; ++d)
    if (kbeg + d * BK < kend) fetch(kbeg + d * BK, araw[d], braw[d], aok[d]);

  for (int k0 = kbeg; k0 < kend; k0 += PD * BK) {
#pragma unroll
    for (int d = 0; d < PD; ++d) {
      const int kt = k0 + d * BK;
      if (kt < kend) {  // uniform
        __syncthreads();
#pragma unroll
        for (int p = 0; p < APASS; ++p)
          *reinterpret_cast<float4 *>(&As[(p * RPP + lr) * LD + 4 * l8]) = masked(araw[d][p], (aok[d] >> p) & 1u);
#pragma unroll
        for (int p = 0; p < BPASS; ++p)
          *reinterpret_cast<float4 *>(&Bs[(p * RPP + lr) * LD + 4 * l8]) = masked(braw[d][p], wvalid[p]);
        __syncthreads();
        if (kt + PD * BK < kend) fetch(kt + PD * BK, araw[d], braw[d], aok[d]);
        const int lrow = lane & 31, lk = 4 * (lane >> 5);
#pragma unroll
        for (int qd = 0; qd < BK / 8; ++qd) {
          const float4 af = *reinterpret_cast<const float4 *>(&As[(wm0 + lrow) * LD + 8 * qd + lk]);
          const float4 bf = *reinterpret_cast<const float4 *>(&Bs[(wn0 + lrow) * LD + 8 * qd + lk]);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, bf.x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, bf.y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, bf.z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, bf.w, acc, 0, 0, 0);
        }
      }
    }
  }
  float *out = a.out + (a.ksplit > 1 ? blockIdx.z * a.slab_stride : 0);
  const int n = n0 + wn0 + (lane & 31);
  auto finish = [&](float v, long long o) {
    if (a.ksplit == 1) {
      if (EPI == EPI_BIAS || EPI == EPI_BIAS_RELU || EPI == EPI_BIAS_TANH) v += a.bias[n];
      if (EPI == EPI_BIAS_RELU) v = v > 0.f ? v : 0.f;
      if (EPI == EPI_BIAS_TANH) v = tanhf(v);
      if (EPI == EPI_MASK) v = a.mask_src[o] > 0.f ? v : 0.f;
      if (EPI == EPI_DTANH) { const float y = a.mask_src[o]; v *= 1.f - y * y; }
    } else if ((EPI == EPI_BIAS || EPI == EPI_BIAS_RELU) && blockIdx.z == 0) {
      v += a.bias[n];
    }
    out[o] = v;
  };
  if (m0 + BM <= a.M && n0 + BN <= a.N) {
    // interior tile (uniform): no per-element exec branches; bias / mask loads before the stores
    // (see igemm_nt_kernel)
    constexpr bool kBias = EPI == EPI_BIAS || EPI == EPI_BIAS_RELU || EPI == EPI_BIAS_TANH;
    constexpr bool kAux = EPI == EPI_MASK || EPI == EPI_DTANH;
    const float bias_v = (kBias && (a.ksplit == 1 || blockIdx.z == 0)) ? a.bias[n] : 0.f;
    const long long row0 = static_cast<long long>(m0 + wm0 + 4 * (lane >> 5)) * a.ldc + n;
    float aux[kAux ? 16 : 1];
    if (kAux && a.ksplit == 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        aux[kAux ? r : 0] = a.mask_src[row0 + static_cast<long long>((r & 3) + 8 * (r >> 2)) * a.ldc];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = acc[r] + bias_v;
      if (a.ksplit == 1) {
        if (EPI == EPI_BIAS_RELU) v = v > 0.f ? v : 0.f;
        if (EPI == EPI_BIAS_TANH) v = tanhf(v);
        if (EPI == EPI_MASK) v = aux[kAux ? r : 0] > 0.f ? v : 0.f;
        if (EPI == EPI_DTANH) { const float y = aux[kAux ? r : 0]; v *= 1.f - y * y; }
      }
      out[row0 + static_cast<long long>((r & 3) + 8 * (r >> 2)) * a.ldc] = v;
    }
    return;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + wm0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (m < a.M && n < a.N) finish(acc[r], static_cast<long long>(m) * a.ldc + n);
  }
}

// ------------------------------------------------------------------------------------
// TN kernel (wgrad): reduction over M in steps of 32 rows
// ------------------------------------------------------------------------------------
template <int TAG, int BN, int BKO, int WN, int WK, bool AU8>
__global__ __launch_bounds__(64 * (BN / WN) * (BKO / WK)) void igemm_tn_kernel(const TNArgs a) {
  constexpr int NWK = BKO / WK;
  constexpr int NT = 64 * (BN / WN) * NWK;
  constexpr int TN = WN / 32, TK = WK / 32;
  constexpr int BMS = 32;
  constexpr int ATPR = BKO / 4;     // threads per A row
  constexpr int ARPP = NT / ATPR;   // A rows per pass
  static_assert(NT % ATPR == 0 && BMS % ARPP == 0, "A tile does not divide");
  constexpr int APASS = BMS / ARPP;
  constexpr int GTPR = BN / 4;
  constexpr int GRPP = NT / GTPR;
  static_assert(NT % GTPR == 0, "G tile does not divide");
  constexpr int GPASS = (BMS + GRPP - 1) / GRPP;
  __shared__ __attribute__((aligned(16))) float smem[BMS * (BN + BKO)];
  float *Gs = smem;
  float *As = smem + BMS * BN;

  const Gather &g = a.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn0 = (wave / NWK) * WN, wk0 = (wave % NWK) * WK;
  // Block -> (k block, n block, M slice).  With a.swizzle the grid is 1-D and numbered so that the
  // blocks an XCD receives (round-robin over the 8 XCDs) are ALL (k, n) blocks of one M slice after
  // another: they re-read the same G rows and overlapping activation rows from that XCD's L2
  // instead of HBM (conv1 wgrad: 1.43 GB -> see DESIGN.md).
  int bk = blockIdx.x, bn = blockIdx.y, bz = blockIdx.z;
  if (a.swizzle) {
    const int per_slice = a.gk * a.gn;
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int zl = seq / per_slice, inner = seq - zl * per_slice;
    bk = inner % a.gk;
    bn = inner / a.gk;
    bz = zl * 8 + xcd;
    if (bz >= a.msplit) return;  // uniform: the grid is padded to a multiple of 8 slices
  }
  const int k0 = bk * BKO, n0 = bn * BN;
  const int mbeg = bz * a.mper;
  const int mend = min(a.M, mbeg + a.mper);

  // this thread's fixed k position (4 consecutive k inside one run)
  const int akq = tid % ATPR, arow = tid / ATPR;
  const int k = k0 + 4 * akq;
  const bool kvalid = k < a.K;
  const int seg = kvalid ? k / g.seglen : 0;
  const long long segoff = static_cast<long long>(g.seg_off[seg]) + (kvalid ? k - seg * g.seglen : 0);
  const int gnq = tid % GTPR, grow = tid / GTPR;
  const bool nvalid = (n0 + 4 * gnq) < a.N;

  f32x16 acc[TN][TK];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TK; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float bias_acc = 0.f;

  typename Raw<AU8>::type araw[APASS];
  float4 graw[GPASS];
  uint32_t aok = 0, gok = 0;
  auto fetch = [&](int ms) {
    aok = gok = 0;
#pragma unroll
    for (int p = 0; p < APASS; ++p) {
      const int m = ms + p * ARPP + arow;
      const bool ok = m < mend && kvalid;
      const RowPos r = decode_row(g, m, ok);
      aok |= (ok ? 1u : 0u) << p;
      araw[p] = load_raw<AU8>(g.src, ok ? r.base + segoff : 0);
    }
#pragma unroll
    for (int p = 0; p < GPASS; ++p) {
      const int mr = p * GRPP + grow;
      const int m = ms + mr;
      const bool ok = mr < BMS && m < mend && nvalid;
      gok |= (ok ? 1u : 0u) << p;
      graw[p] = *reinterpret_cast<const float4 *>(a.G + (ok ? static_cast<long long>(m) * a.ldg + n0 + 4 * gnq : 0));
    }
  };
  if (mbeg < mend) fetch(mbeg);

  for (int ms = mbeg; ms < mend; ms += BMS) {
    __syncthreads();
#pragma unroll
    for (int p = 0; p < APASS; ++p)
      *reinterpret_cast<float4 *>(&As[(p * ARPP + arow) * BKO + 4 * akq]) =
          masked(to_float4(araw[p]), (aok >> p) & 1u);
#pragma unroll
    for (int p = 0; p < GPASS; ++p)
      if (p * GRPP + grow < BMS)
        *reinterpret_cast<float4 *>(&Gs[(p * GRPP + grow) * BN + 4 * gnq]) = masked(graw[p], (gok >> p) & 1u);
    __syncthreads();
    if (ms + BMS < mend) fetch(ms + BMS);
    if (a.bias_slab && bk == 0 && tid < BN) {
#pragma unroll 8
      for (int r = 0; r < BMS; ++r) bias_acc += Gs[r * BN + tid];
    }
    const int lcol = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int e = 0; e < BMS / 2; ++e) {
      float gf[TN], af[TK];
#pragma unroll
      for (int i = 0; i < TN; ++i) gf[i] = Gs[(2 * e + lh) * BN + wn0 + 32 * i + lcol];
#pragma unroll
      for (int j = 0; j < TK; ++j) af[j] = As[(2 * e + lh) * BKO + wk0 + 32 * j + lcol];
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TK; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(gf[i], af[j], acc[i][j], 0, 0, 0);
    }
  }

  float *slab = a.slab + static_cast<long long>(bz) * a.N * a.K;
  if (n0 + BN <= a.N && k0 + BKO <= a.K) {  // interior tile (uniform): unguarded stores
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wn0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
#pragma unroll
        for (int j = 0; j < TK; ++j)
          slab[static_cast<long long>(n) * a.K + k0 + wk0 + 32 * j + (lane & 31)] = acc[i][j][r];
      }
  } else {
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wn0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (n >= a.N) continue;
#pragma unroll
        for (int j = 0; j < TK; ++j) {
          const int kk = k0 + wk0 + 32 * j + (lane & 31);
          if (kk < a.K) slab[static_cast<long long>(n) * a.K + kk] = acc[i][j][r];
        }
      }
  }
  if (a.bias_slab && bk == 0 && tid < BN && n0 + tid < a.N)
    a.bias_slab[static_cast<long long>(bz) * a.N + n0 + tid] = bias_acc;
}

// ------------------------------------------------------------------------------------
// strided permute + slab reduction (weight packing, gradient finalisation)
// ------------------------------------------------------------------------------------
struct JobTable {
  PermuteJob jobs[kMaxJobs];
  int chunk_begin[kMaxJobs + 1];  // prefix sums of the 64-element chunks per job
  int njobs;
};

// One workgroup per 64 consecutive elements of the contiguous side of ONE job (the grid is the
// concatenation of all jobs' chunks: no empty workgroups, no loop -- a second chunk's loads
// would sit behind the first chunk's store, and vmcnt counts stores).  Its 4 waves split the
// slabs (wave g sums slabs z = g, g+4, ...) with the 64 lanes on consecutive elements, so every
// slab read is one coalesced 256-B row when the source is contiguous along the fastest output
// dimension (all finalize jobs); the waves combine through LDS.  Loads are issued 8 deep.
__device__ __forceinline__ void permute_reduce_block(const JobTable &t, int b, float (*red)[64]) {
  int job = 0;
  while (job + 1 < t.njobs && b >= t.chunk_begin[job + 1]) ++job;  // uniform, <= kMaxJobs steps
  const PermuteJob &j = t.jobs[job];
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  const long long i = static_cast<long long>(b - t.chunk_begin[job]) * 64 + lane;
  const bool active = i < j.total;
  long long rest = active ? i : 0;
  const long long d3 = rest % j.D3; rest /= j.D3;
  const long long d2 = rest % j.D2; rest /= j.D2;
  const long long d1 = rest % j.D1; rest /= j.D1;
  const long long strided = rest * j.s0 + d1 * j.s1 + d2 * j.s2 + d3 * j.s3;
  const float *src = j.src + j.off + (j.scatter ? (active ? i : 0) : strided);
  float v = 0.f;
  int z = g;
  for (; z + 28 < j.nslab; z += 32) {  // 8 loads in flight
    float x[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) x[u] = src[(z + 4 * u) * j.slab_stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) v += x[u];
  }
  for (; z < j.nslab; z += 4) v += src[z * j.slab_stride];
  red[g][lane] = v;
  __syncthreads();
  if (g == 0 && active)
    j.dst[j.scatter ? strided : i] = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
}

__global__ __launch_bounds__(256) void permute_reduce_kernel(const JobTable t) {
  __shared__ float red[4][64];
  permute_reduce_block(t, blockIdx.x, red);
}

// the same grid followed by the (49, Jp) blocks of the factored tail's G / s reduction (tail_greduce_dev.hpp)
__global__ __launch_bounds__(256) void permute_reduce_greduce_kernel(const JobTable t, const TailGreduceArgs g, int chunks) {
  __shared__ float red[4][64];
  __shared__ float sred[256];
  const int b = blockIdx.x;
  if (b < chunks) {
    permute_reduce_block(t, b, red);
    return;
  }
  const int q = b - chunks;
  tail_greduce_block(g, q % kGreduceP, q / kGreduceP, red, sred);
}

template <int TAG, int BM, int BN, int WM, int WN, bool AU8, int EPI, int BK = 32>
int launch_nt_as(const NTArgs &a, hipStream_t stream) {
  DX_REQUIRE(a.g.seglen % BK == 0 && (a.K / a.ksplit) % BK == 0,
             "igemm_nt: run length %d / K per split %d not a multiple of the K step %d", a.g.seglen,
             a.K / a.ksplit, BK);
  dim3 grid(cdiv(a.M, BM), cdiv(a.N, BN), a.ksplit);
  dim3 block(64 * (BM / WM) * (BN / WN));
  hipLaunchKernelGGL((igemm_nt_kernel<TAG, BM, BN, WM, WN, AU8, EPI, BK>), grid, block, 0, stream, a);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

template <int TAG, int EPI>
int launch_nt_small(const NTArgs &a, hipStream_t stream) {
  dim3 grid(cdiv(a.M, 64), cdiv(a.N, 64), a.ksplit);
  hipLaunchKernelGGL((igemm_nt_small_kernel<TAG, EPI, 3>), grid, dim3(256), 0, stream, a);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

int tn_swizzle() {  // DX_TN_SWIZZLE=0: plain 3-D grid for the wgrad kernels
  const int v = DX_ENV("DX_TN_SWIZZLE", 1);
  return v;
}

template <int TAG, int BN, int BKO, int WN, int WK, bool AU8>
int launch_tn_as(const TNArgs &a_in, hipStream_t stream) {
  TNArgs a = a_in;
  a.gk = cdiv(a.K, BKO);
  a.gn = cdiv(a.N, BN);
  a.swizzle = (tn_swizzle() && a.msplit >= 64) ? 1 : 0;  // few slices: keep every XCD busy instead
  dim3 grid(a.gk, a.gn, a.msplit);
  if (a.swizzle) grid = dim3(cdiv(a.msplit, 8) * 8 * a.gk * a.gn, 1, 1);
  hipLaunchKernelGGL((igemm_tn_kernel<TAG, BN, BKO, WN, WK, AU8>), grid,
                     dim3(64 * (BN / WN) * (BKO / WK)), 0, stream, a);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

}  // namespace

// Tile shapes.  Big problems (training minibatches) use 64x64 wave tiles -- 64 MFMAs between
// barrier pairs -- small ones (rollout batches) smaller tiles so that >= 256 workgroups exist.
//   N <= 32 : 512x32 (4 waves of 128x32)  |  256x32 (4 waves of 64x32)
//   N >= 64 : 256x64 (4 waves of 64x64)   |  128x64 (4 waves of 64x32)  |  64x64 (4 waves of 32x32)
#ifdef DX_EXPERIMENT_B3
static int split_bf16() {  // DX_SPLIT_BF16=1: big NT stages on the bf16 matrix cores (3-way split)
  const int v = DX_ENV("DX_SPLIT_BF16", 0);
  return v;
}
static int split_min_m() {
  const int v = DX_ENV("DX_SPLIT_MIN_M", 65536);
  return v;
}
#endif
static int nt_big_min_m() {  // DX_NT_BIG_MIN_M: rows from which the N=64 stages use 128x64 tiles
  const int v = DX_ENV("DX_NT_BIG_MIN_M", 65536);
  return v;
}
// Small problems (rollout batches) are latency-bound: 64-deep K steps halve the number of
// barrier / load round trips per tile.
#define DX_NT_N64(ST, EPI)                                                                      \
  if (a.M >= nt_big_min_m()) return launch_nt_as<ST, 128, 64, 64, 32, false, EPI>(a, stream);   \
  if (a.g.seglen % 64 == 0 && (a.K / a.ksplit) % 64 == 0 && !a.om.enabled)                      \
    return launch_nt_small<ST, EPI>(a, stream);                                                 \
  return launch_nt_as<ST, 64, 64, 32, 32, false, EPI>(a, stream)

int launch_nt(const NTArgs &a_in, bool a_u8, int epi, int stage, hipStream_t stream) {
  NTArgs a = a_in;
  DX_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "igemm_nt: empty problem M=%d N=%d K=%d", a.M, a.N, a.K);
  DX_REQUIRE(a.g.seglen % 32 == 0 && a.g.nseg * a.g.seglen == a.K && a.g.nseg <= kMaxSeg,
             "igemm_nt: K=%d must be nseg(%d) x seglen(%d), seglen %% 32 == 0", a.K, a.g.nseg,
             a.g.seglen);
  DX_REQUIRE(a.ksplit >= 1 && a.K % (32 * a.ksplit) == 0, "igemm_nt: K=%d not divisible by 32*ksplit(%d)",
             a.K, a.ksplit);
  DX_REQUIRE(a.ksplit == 1 || epi == EPI_NONE || epi == EPI_BIAS,
             "igemm_nt: split-K partial sums cannot carry epilogue %d", epi);
  DX_REQUIRE(a.g.src && a.Wp && a.out, "igemm_nt: null pointer");
  DX_REQUIRE(a.M < (1 << 30), "igemm_nt: M too large");
  DX_REQUIRE(!a.om.enabled || (a.om.chan > 0 && a.om.chan % 32 == 0),
             "igemm_nt: output map needs a channel count that is a multiple of 32");
  {  // small batches: one 32x32 tile per workgroup, K split over its waves (igemm_lat.hip)
    const int rc = launch_nt_lat(a, a_u8, epi, stage, stream);
    if (rc != DX_ENOSUP) return rc;
  }
#ifdef DX_EXPERIMENT_B3  // build flag DERL_AMD_EXPERIMENTS=1: csrc/experiments/igemm_b3.hip (negative result)
  if (split_bf16() && !a_u8 && a.M >= split_min_m()) {
    const int rc = launch_nt_b3(a, epi, stage, stream);
    if (rc != DX_ENOSUP) return rc;
  }
#endif
  switch (stage) {
    case ST_CONV0_FWD:
      DX_REQUIRE(epi == EPI_BIAS_RELU && a.N <= 32, "igemm_nt: conv0 forward needs bias+relu, N <= 32");
      return a_u8 ? launch_nt_as<ST_CONV0_FWD, 256, 32, 64, 32, true, EPI_BIAS_RELU>(a, stream)
                  : launch_nt_as<ST_CONV0_FWD, 256, 32, 64, 32, false, EPI_BIAS_RELU>(a, stream);
    case ST_CONV1_FWD: DX_NT_N64(ST_CONV1_FWD, EPI_BIAS_RELU);
    case ST_CONV2_FWD: DX_NT_N64(ST_CONV2_FWD, EPI_BIAS_RELU);
    case ST_FC_FWD: DX_NT_N64(ST_FC_FWD, EPI_BIAS);
    case ST_HEADS_FWD:
      if (a.M <= 16384 && a.K % 64 == 0) return launch_nt_small<ST_HEADS_FWD, EPI_BIAS>(a, stream);
      return launch_nt_as<ST_HEADS_FWD, 256, 32, 64, 32, false, EPI_BIAS>(a, stream);
    case ST_HEADS_DGRAD: return launch_nt_as<ST_HEADS_DGRAD, 128, 64, 64, 32, false, EPI_NONE>(a, stream);
    case ST_FC_DGRAD: DX_NT_N64(ST_FC_DGRAD, EPI_MASK);
    // zero-fill gathers: 64x64 tiles win at every size (minibatch 8192: 472 vs 504 us, 600 vs 711 us)
    case ST_CONV2_DGRAD: return launch_nt_as<ST_CONV2_DGRAD, 64, 64, 32, 32, false, EPI_MASK>(a, stream);
    case ST_CONV1_DGRAD: return launch_nt_as<ST_CONV1_DGRAD, 64, 64, 32, 32, false, EPI_MASK>(a, stream);
    case ST_MLP_HIDDEN: DX_NT_N64(ST_MLP_HIDDEN, EPI_BIAS_TANH);
    case ST_MLP_OUT: return launch_nt_as<ST_MLP_OUT, 64, 32, 32, 32, false, EPI_BIAS>(a, stream);
    case ST_MLP_DGRAD: DX_NT_N64(ST_MLP_DGRAD, EPI_DTANH);
    default: return fail(DX_EINVAL, "igemm_nt: unknown stage %d", stage);
  }
}
#undef DX_NT_N64

int launch_tn(const TNArgs &a, bool a_u8, int stage, hipStream_t stream) {
  DX_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "igemm_tn: empty problem");
  DX_REQUIRE(a.g.seglen % 4 == 0 && a.g.nseg * a.g.seglen == a.K && a.g.nseg <= kMaxSeg,
             "igemm_tn: bad K decomposition");
  DX_REQUIRE(a.mper % 32 == 0 && a.msplit >= 1 && static_cast<long long>(a.mper) * a.msplit >= a.M,
             "igemm_tn: bad M split (mper=%d msplit=%d M=%d)", a.mper, a.msplit, a.M);
  DX_REQUIRE(a.g.src && a.G && a.slab, "igemm_tn: null pointer");
  DX_REQUIRE(a.N % 4 == 0 && a.ldg % 4 == 0, "igemm_tn: N and ldg must be multiples of 4");
  DX_REQUIRE(!a.g.check, "igemm_tn: bounds-checked gathers are not supported in wgrad");
  switch (stage) {
    case ST_HEADS_WGRAD: return launch_tn_as<ST_HEADS_WGRAD, 32, 256, 32, 64, false>(a, stream);
    case ST_FC_WGRAD: return launch_tn_as<ST_FC_WGRAD, 64, 128, 64, 32, false>(a, stream);
    case ST_CONV2_WGRAD: return launch_tn_as<ST_CONV2_WGRAD, 64, 128, 64, 32, false>(a, stream);
    case ST_CONV1_WGRAD: return launch_tn_as<ST_CONV1_WGRAD, 64, 128, 64, 32, false>(a, stream);
    case ST_CONV0_WGRAD:
      return a_u8 ? launch_tn_as<ST_CONV0_WGRAD, 32, 256, 32, 64, true>(a, stream)
                  : launch_tn_as<ST_CONV0_WGRAD, 32, 256, 32, 64, false>(a, stream);
    case ST_MLP_WGRAD_HID: return launch_tn_as<ST_MLP_WGRAD_HID, 64, 128, 64, 32, false>(a, stream);
    case ST_MLP_WGRAD_OUT: return launch_tn_as<ST_MLP_WGRAD_OUT, 32, 256, 32, 64, false>(a, stream);
    default: return fail(DX_EINVAL, "igemm_tn: unknown stage %d", stage);
  }
}

namespace {

// Packing of the 3136-wide linear layer (95 % of the parameters) with coalesced traffic on both
// sides.  fcf[n][p*C + c] = W[n][c*P + p] (the reference flattens NCHW, activations here are
// NHWC): one workgroup per row n stages the row in LDS and writes it permuted.
__device__ __forceinline__ void fc_row_permute_block(const float *__restrict__ W, float *__restrict__ out, int P,
                                                     int C, int n, float *row) {
  const int K = P * C;
  const float *src = W + static_cast<long long>(n) * K;
  float *dst = out + static_cast<long long>(n) * K;
  for (int i = threadIdx.x; i < K; i += 256) row[i] = src[i];
  __syncthreads();
  for (int k = threadIdx.x; k < K; k += 256) {
    const int p = k / C, c = k - p * C;
    dst[k] = row[c * P + p];
  }
}

__global__ __launch_bounds__(256) void fc_row_permute_kernel(const float *__restrict__ W, float *__restrict__ out,
                                                            int P, int C) {
  extern __shared__ float row[];
  fc_row_permute_block(W, out, P, C, blockIdx.x, row);
}

// The inverse for the weight gradient: canonical dW[n][c*P + p] = sum_z slab[z][n][p*C + c].  A
// workgroup takes one row n and one half of the channels: it sums the slabs with 16-byte loads
// (two slabs in flight per thread), permutes through LDS (row stride C/2 + 1: the transposed read
// walks p, which with a power-of-two stride would hit one bank 49 times) and writes its
// contiguous [c][p] range.  Was one scalar-load workgroup per row: 27-35 us for 38 MB.
__device__ __forceinline__ void fc_row_unpermute_block(const float *__restrict__ slab, int nslab,
                                                       long long slab_stride, float *__restrict__ out, int P,
                                                       int C, int bx, int by, float *row) {  // row: [P][C/2 + 1]
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const int CH = C / 2, C4 = CH / 4, LD = CH + 1;
  const long long K = static_cast<long long>(P) * C;
  const float *src = slab + bx * K + by * CH;
  for (int q = threadIdx.x; q < P * C4; q += 256) {
    const int p = q / C4, c4 = q - p * C4;
    const float *s = src + p * C + 4 * c4;
    f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
    int z = 0;
    for (; z + 1 < nslab; z += 2) {
      v0 += *reinterpret_cast<const f32x4 *>(s + z * slab_stride);
      v1 += *reinterpret_cast<const f32x4 *>(s + (z + 1) * slab_stride);
    }
    if (z < nslab) v0 += *reinterpret_cast<const f32x4 *>(s + z * slab_stride);
    v0 += v1;
    float *d = row + p * LD + 4 * c4;
    d[0] = v0[0]; d[1] = v0[1]; d[2] = v0[2]; d[3] = v0[3];
  }
  __syncthreads();
  float *dst = out + bx * K + static_cast<long long>(by) * CH * P;
  for (int i = threadIdx.x; i < CH * P; i += 256) {
    const int c = i / P, p = i - c * P;
    dst[i] = row[p * LD + c];
  }
}

__global__ __launch_bounds__(256) void fc_row_unpermute_reduce_kernel(const float *__restrict__ slab, int nslab,
                                                                     long long slab_stride,
                                                                     float *__restrict__ out, int P, int C) {
  extern __shared__ float row[];
  fc_row_unpermute_block(slab, nslab, slab_stride, out, P, C, blockIdx.x, blockIdx.y, row);
}

// Gradient finalisation in ONE launch: the strided slab reductions of permute_reduce_kernel in
// workgroups [0, chunks), the linear layer's [p][c] -> [c][p] reduction in the rest -- the two are
// independent, and as separate launches on one stream the second waited for the first (28 + 14 us).
struct FcFinalize {
  const float *slab;
  int nslab;
  long long slab_stride;
  float *out;
  int N, P, C;
};
__global__ __launch_bounds__(256) void finalize_fused_kernel(const JobTable t, const FcFinalize f, int chunks) {
  __shared__ float red[4][64];
  extern __shared__ float row[];
  const int b = blockIdx.x;
  if (b < chunks) {
    permute_reduce_block(t, b, red);
  } else {
    const int r = b - chunks;
    fc_row_unpermute_block(f.slab, f.nslab, f.slab_stride, f.out, f.P, f.C, r >> 1, r & 1, row);
  }
}

// out[k][n] = in[n][k] (in: R x Cn row-major), 64 x 64 tiles through LDS
__global__ __launch_bounds__(256) void transpose_kernel(const float *__restrict__ in, float *__restrict__ out,
                                                       int R, int Cn) {
  __shared__ float tile[64][65];
  const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
  const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = r0 + ly + 4 * i, c = c0 + lx;
    tile[ly + 4 * i][lx] = (r < R && c < Cn) ? in[static_cast<long long>(r) * Cn + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = c0 + ly + 4 * i, r = r0 + lx;
    if (r < R && c < Cn) out[static_cast<long long>(c) * R + r] = tile[lx][ly + 4 * i];
  }
}

}  // namespace

int launch_fc_grad_finalize(const float *slab, int nslab, long long slab_stride, float *grad, int N, int P,
                            int C, hipStream_t stream) {
  DX_REQUIRE(slab && grad && nslab >= 1 && N > 0 && P * C * 4 <= 64 * 1024 && C % 8 == 0 && slab_stride % 4 == 0 &&
                 aligned(slab, 16),
             "fc_grad_finalize: bad arguments");
  hipLaunchKernelGGL(fc_row_unpermute_reduce_kernel, dim3(N, 2), dim3(256), sizeof(float) * P * (C / 2 + 1), stream,
                     slab, nslab, slab_stride, grad, P, C);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

// fcf [N][P*C] (k = p*C + c) and its transpose fcd [P*C][N] from the canonical W [N][C*P]
int launch_fc_pack(const float *W, float *fcf, float *fcd, int N, int P, int C, hipStream_t stream) {
  DX_REQUIRE(W && fcf && fcd && N > 0 && P > 0 && C > 0 && P * C * 4 <= 64 * 1024, "fc_pack: bad arguments");
  hipLaunchKernelGGL(fc_row_permute_kernel, dim3(N), dim3(256), sizeof(float) * P * C, stream, W, fcf, P, C);
  DX_LAUNCH_CHECK();
  hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(P * C, 64), cdiv(N, 64)), dim3(256), 0, stream, fcf, fcd, N,
                     P * C);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

// Weight packing: the strided mirrors (permute jobs) in workgroups [0, chunks), the linear layer's
// row permutation in the rest (independent; one launch less per optimizer step)
__global__ __launch_bounds__(256) void pack_fused_kernel(const JobTable t, const float *__restrict__ W,
                                                        float *__restrict__ fcf, int P, int C, int chunks) {
  __shared__ float red[4][64];
  extern __shared__ float row[];
  const int b = blockIdx.x;
  if (b < chunks) permute_reduce_block(t, b, red);
  else fc_row_permute_block(W, fcf, P, C, b - chunks, row);
}

// permute jobs + the linear layer's gradient (see finalize_fused_kernel)
int launch_finalize_fused(const PermuteJob *jobs, int njobs, const float *slab, int nslab, long long slab_stride,
                          float *grad, int N, int P, int C, hipStream_t stream) {
  DX_REQUIRE(njobs >= 1 && njobs <= kMaxJobs, "finalize_fused: %d jobs (max %d)", njobs, kMaxJobs);
  DX_REQUIRE(slab && grad && nslab >= 1 && N > 0 && P * C * 4 <= 64 * 1024 && C % 8 == 0 && slab_stride % 4 == 0 &&
                 aligned(slab, 16),
             "finalize_fused: bad arguments");
  JobTable t;
  long long chunks = 0;
  for (int i = 0; i < njobs; ++i) {
    DX_REQUIRE(jobs[i].src && jobs[i].dst && jobs[i].nslab >= 1 && jobs[i].D1 > 0 && jobs[i].D2 > 0 &&
                   jobs[i].D3 > 0 && jobs[i].total > 0,
               "finalize_fused: bad job %d", i);
    t.jobs[i] = jobs[i];
    t.chunk_begin[i] = static_cast<int>(chunks);
    chunks += (jobs[i].total + 63) / 64;
    DX_REQUIRE(chunks < (1LL << 30), "finalize_fused: too many elements");
  }
  t.chunk_begin[njobs] = static_cast<int>(chunks);
  t.njobs = njobs;
  const FcFinalize f{slab, nslab, slab_stride, grad, N, P, C};
  hipLaunchKernelGGL(finalize_fused_kernel, dim3(static_cast<unsigned>(chunks) + 2 * N), dim3(256),
                     sizeof(float) * P * (C / 2 + 1), stream, t, f, static_cast<int>(chunks));
  DX_LAUNCH_CHECK();
  return DX_OK;
}

// permute jobs + fcf [N][P*C] (k = p*C + c) from the canonical W [N][C*P] in one launch, then its
// transpose fcd [P*C][N]
int launch_pack_fused(const PermuteJob *jobs, int njobs, const float *W, float *fcf, float *fcd, int N, int P, int C,
                      hipStream_t stream) {
  DX_REQUIRE(njobs >= 1 && njobs <= kMaxJobs && W && fcf && fcd && N > 0 && P > 0 && C > 0 && P * C * 4 <= 64 * 1024,
             "pack_fused: bad arguments");
  JobTable t;
  long long chunks = 0;
  for (int i = 0; i < njobs; ++i) {
    DX_REQUIRE(jobs[i].src && jobs[i].dst && jobs[i].nslab >= 1 && jobs[i].D1 > 0 && jobs[i].D2 > 0 &&
                   jobs[i].D3 > 0 && jobs[i].total > 0,
               "pack_fused: bad job %d", i);
    t.jobs[i] = jobs[i];
    t.chunk_begin[i] = static_cast<int>(chunks);
    chunks += (jobs[i].total + 63) / 64;
  }
  t.chunk_begin[njobs] = static_cast<int>(chunks);
  t.njobs = njobs;
  hipLaunchKernelGGL(pack_fused_kernel, dim3(static_cast<unsigned>(chunks) + N), dim3(256), sizeof(float) * P * C,
                     stream, t, W, fcf, P, C, static_cast<int>(chunks));
  DX_LAUNCH_CHECK();
  hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(P * C, 64), cdiv(N, 64)), dim3(256), 0, stream, fcf, fcd, N,
                     P * C);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

static int fill_job_table(const PermuteJob *jobs, int njobs, JobTable *t, long long *chunks_out) {
  long long chunks = 0;
  for (int i = 0; i < njobs; ++i) {
    DX_REQUIRE(jobs[i].src && jobs[i].dst && jobs[i].nslab >= 1 && jobs[i].D1 > 0 && jobs[i].D2 > 0 &&
                   jobs[i].D3 > 0 && jobs[i].total > 0,
               "permute_reduce: bad job %d", i);
    t->jobs[i] = jobs[i];
    t->chunk_begin[i] = static_cast<int>(chunks);
    chunks += (jobs[i].total + 63) / 64;
    DX_REQUIRE(chunks < (1LL << 30), "permute_reduce: too many elements");
  }
  t->chunk_begin[njobs] = static_cast<int>(chunks);
  t->njobs = njobs;
  *chunks_out = chunks;
  return DX_OK;
}

int launch_permute_reduce_greduce(const PermuteJob *jobs, int njobs, const TailGreduceArgs &g, hipStream_t stream) {
  DX_REQUIRE(njobs >= 1 && njobs <= kMaxJobs, "permute_reduce_greduce: %d jobs (max %d)", njobs, kMaxJobs);
  DX_REQUIRE(g.gslab && g.sslab && g.Gc && g.s && g.nslab >= 1 && g.Jp % 8 == 0 && g.nj >= 1 && g.nj <= g.Jp, "permute_reduce_greduce: bad tail arguments");
  JobTable t;
  long long chunks = 0;
  if (int rc = fill_job_table(jobs, njobs, &t, &chunks)) return rc;
  hipLaunchKernelGGL(permute_reduce_greduce_kernel, dim3(static_cast<unsigned>(chunks + kGreduceP * g.Jp)), dim3(256), 0, stream, t, g,
                     static_cast<int>(chunks));
  DX_LAUNCH_CHECK();
  return DX_OK;
}

int launch_permute_reduce(const PermuteJob *jobs, int njobs, hipStream_t stream) {
  DX_REQUIRE(njobs >= 0 && njobs <= kMaxJobs, "permute_reduce: %d jobs (max %d)", njobs, kMaxJobs);
  if (njobs == 0) return DX_OK;
  JobTable t;
  long long chunks = 0;
  for (int i = 0; i < njobs; ++i) {
    DX_REQUIRE(jobs[i].src && jobs[i].dst && jobs[i].nslab >= 1 && jobs[i].D1 > 0 && jobs[i].D2 > 0 &&
                   jobs[i].D3 > 0 && jobs[i].total > 0,
               "permute_reduce: bad job %d", i);
    t.jobs[i] = jobs[i];
    t.chunk_begin[i] = static_cast<int>(chunks);
    chunks += (jobs[i].total + 63) / 64;
    DX_REQUIRE(chunks < (1LL << 30), "permute_reduce: too many elements");
  }
  t.chunk_begin[njobs] = static_cast<int>(chunks);
  t.njobs = njobs;
  hipLaunchKernelGGL(permute_reduce_kernel, dim3(static_cast<unsigned>(chunks)), dim3(256), 0, stream, t);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

}  // namespace dx
