// fp32-accurate GEMM on the bf16 matrix cores: every fp32 operand x is split EXACTLY into three
// bf16 values x = hi + mid + lo (8 + 8 + 8 significand bits), and a product a*b is formed from
// the six partial products whose weight is >= 2^-16 of the leading one:
//     a_hi*b_hi + (a_hi*b_mid + a_mid*b_hi) + (a_hi*b_lo + a_mid*b_mid + a_lo*b_hi)
// Each bf16 x bf16 product is exact in fp32 and accumulation is fp32 inside the MFMA, so the
// result differs from an fp32 fma chain only by the dropped terms (2 * 2^-24 relative per
// product, i.e. one fp32 rounding) -- the same tolerance class as a different summation order.
// v_mfma_f32_32x32x16_bf16 issues every 32 cycles for 16 k, six of them replace eight
// v_mfma_f32_32x32x2_f32 (64 cycles each): 192 vs 512 cycles per 32x32x16 block.
//
// Same interface, gathers and epilogues as igemm_nt_kernel; only the tile staging (split while
// writing LDS, three planes of [rows][32 k] bf16 with 80-byte rows: conflict-free ds_read_b128)
// and the MFMA body differ.
#include "../bf16_split.hpp"

namespace dx {
namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
constexpr int kRowB = 80;  // bytes per LDS row: 32 bf16 + 16 pad

template <int TAG, int BM, int BN, int WM, int WN, int EPI>
__global__ __launch_bounds__(64 * (BM / WM) * (BN / WN)) void igemm_nt_b3_kernel(const NTArgs a) {
  constexpr int NWN = BN / WN;
  constexpr int NT = 64 * (BM / WM) * NWN;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int BK = 32;
  constexpr int RPP = NT / 8;
  static_assert(BM % RPP == 0, "BM must be a multiple of the rows per pass");
  constexpr int APASS = BM / RPP;
  // weights: pre-split planes [3][N][K] bf16 (dx_cnn_pack); a tile row is 32 k = 4 x 16 bytes per plane
  constexpr int BCH = BN * 4 * 3;  // 16-byte chunks per K step
  constexpr int BPASS = (BCH + NT - 1) / NT;
  constexpr int A_PLANE = BM * kRowB, B_PLANE = BN * kRowB;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];  // 3 * (A_PLANE + B_PLANE) bytes
  uint8_t *As = smem;                 // planes hi, mid, lo
  uint8_t *Bs = smem + 3 * A_PLANE;

  const Gather &g = a.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave / NWN) * WM, wn0 = (wave % NWN) * WN;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int kper = a.K / a.ksplit;
  const int kbeg = blockIdx.z * kper, kend = kbeg + kper;
  const int l8 = tid & 7, lr = tid >> 3;

  RowPos rows[APASS];
#pragma unroll
  for (int p = 0; p < APASS; ++p) {
    const int m = m0 + p * RPP + lr;
    rows[p] = decode_row(g, m, m < a.M);
  }
  // chunk c of the B tile: plane = c / (BN*4), row = (c / 4) % BN, 16-byte piece = c % 4
  const uint8_t *wsrc[BPASS];
  int wdst[BPASS];
#pragma unroll
  for (int p = 0; p < BPASS; ++p) {
    const int c = min(p * NT + tid, BCH - 1);
    const int plane = c / (BN * 4), row = (c >> 2) % BN, piece = c & 3;
    const int n = min(n0 + row, a.N - 1);  // columns past N are computed but never stored
    wsrc[p] = reinterpret_cast<const uint8_t *>(a.Wb) +
              2 * (plane * a.wb_plane + static_cast<long long>(n) * a.K) + 16 * piece;
    wdst[p] = plane * B_PLANE + row * kRowB + 16 * piece;
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Software pipeline (per K step of 32): the activation tile is fetched TWO steps ahead as raw
  // fp32, split into its three bf16 planes ONE step ahead -- in the shadow of the current step's
  // MFMAs, not between the barriers -- and written to LDS at the start of its own step.  The
  // weight planes need no arithmetic and are fetched one step ahead.
  int seg = kbeg / g.seglen, q = kbeg - seg * g.seglen;  // position of the next A fetch
  f32x4 araw[APASS];
  u32x4 braw[BPASS];
  uint2 shi[APASS], smid[APASS], slo[APASS];
  uint32_t aok = 0;
  auto fetch_a = [&]() {
    const int sseg = __builtin_amdgcn_readfirstlane(seg);
    const long long so = static_cast<long long>(g.seg_off[sseg]) + q + 4 * l8;
    aok = 0;
#pragma unroll
    for (int p = 0; p < APASS; ++p) {
      const bool ok = (rows[p].okmask >> sseg) & 1u;
      aok |= (ok ? 1u : 0u) << p;
      araw[p] = *reinterpret_cast<const f32x4 *>(static_cast<const float *>(g.src) + (ok ? rows[p].base + so : 0));
    }
    q += BK;
    if (q >= g.seglen) { q = 0; ++seg; }
  };
  auto fetch_b = [&](int kt) {
#pragma unroll
    for (int p = 0; p < BPASS; ++p) braw[p] = *reinterpret_cast<const u32x4 *>(wsrc[p] + 2 * kt);
  };
  auto split_a = [&]() {
#pragma unroll
    for (int p = 0; p < APASS; ++p) {
      const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
      const Split4 s = split4(((aok >> p) & 1u) ? araw[p] : zero);
      shi[p] = s.hi; smid[p] = s.mid; slo[p] = s.lo;
    }
  };
  fetch_a();
  fetch_b(kbeg);
  split_a();
  if (kbeg + BK < kend) fetch_a();

  for (int kt = kbeg; kt < kend; kt += BK) {
    __syncthreads();
#pragma unroll
    for (int p = 0; p < APASS; ++p) {
      const int o = (p * RPP + lr) * kRowB + 8 * l8;
      *reinterpret_cast<uint2 *>(As + o) = shi[p];
      *reinterpret_cast<uint2 *>(As + A_PLANE + o) = smid[p];
      *reinterpret_cast<uint2 *>(As + 2 * A_PLANE + o) = slo[p];
    }
#pragma unroll
    for (int p = 0; p < BPASS; ++p)
      if (p * NT + tid < BCH) *reinterpret_cast<u32x4 *>(Bs + wdst[p]) = braw[p];
    __syncthreads();
    if (kt + BK < kend) fetch_b(kt + BK);
    const int lrow = lane & 31, lk = 16 * (lane >> 5);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {  // two 16-deep k steps per 32-k tile
      bf16x8 ah[TM], am[TM], al[TM], bh[TN], bm[TN], bl[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int o = (wm0 + 32 * i + lrow) * kRowB + 32 * ks + lk;
        ah[i] = *reinterpret_cast<const bf16x8 *>(As + o);
        am[i] = *reinterpret_cast<const bf16x8 *>(As + A_PLANE + o);
        al[i] = *reinterpret_cast<const bf16x8 *>(As + 2 * A_PLANE + o);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int o = (wn0 + 32 * j + lrow) * kRowB + 32 * ks + lk;
        bh[j] = *reinterpret_cast<const bf16x8 *>(Bs + o);
        bm[j] = *reinterpret_cast<const bf16x8 *>(Bs + B_PLANE + o);
        bl[j] = *reinterpret_cast<const bf16x8 *>(Bs + 2 * B_PLANE + o);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          // smallest terms first
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bm[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bm[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
      if (ks == 0 && kt + BK < kend) {  // next tile: split now (MFMAs above are in flight), refill the raw regs
        split_a();
        if (kt + 2 * BK < kend) fetch_a();
      }
    }
  }

  // epilogue (identical to igemm_nt_kernel)
  float *out = a.out + (a.ksplit > 1 ? blockIdx.z * a.slab_stride : 0);
  const OutMap &om = a.om;
  int gy[TN], gx[TN], cc[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int nb = __builtin_amdgcn_readfirstlane(n0 + wn0 + 32 * j);
    gy[j] = gx[j] = 0;
    cc[j] = nb + (lane & 31);
    if (om.enabled) {
      const int gg = nb / om.chan;
      gy[j] = gg / om.osx;
      gx[j] = gg - gy[j] * om.osx;
      cc[j] = nb - gg * om.chan + (lane & 31);
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
        float v = acc[i][j][r];
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
      }
    }
}

template <int TAG, int BM, int BN, int WM, int WN, int EPI>
int launch_b3_as(const NTArgs &a, hipStream_t stream) {
  constexpr int lds = 3 * (BM + BN) * kRowB;
  static bool configured = false;  // per instantiation
  if (!configured) {
    DX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(igemm_nt_b3_kernel<TAG, BM, BN, WM, WN, EPI>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    configured = true;
  }
  dim3 grid(cdiv(a.M, BM), cdiv(a.N, BN), a.ksplit);
  hipLaunchKernelGGL((igemm_nt_b3_kernel<TAG, BM, BN, WM, WN, EPI>), grid, dim3(64 * (BM / WM) * (BN / WN)),
                     lds, stream, a);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

int b3_tile() {  // DX_B3_TILE: 0 = 256x64 (4 waves of 64x64), 1 = 128x64 (4 waves of 64x32)
  static int v = -1;
  if (v < 0) { const char *e = getenv("DX_B3_TILE"); v = e ? atoi(e) : 0; }
  return v;
}

template <int TAG, int EPI>
int launch_b3(const NTArgs &a, hipStream_t stream) {
  if (b3_tile() == 1) return launch_b3_as<TAG, 128, 64, 64, 32, EPI>(a, stream);
  return launch_b3_as<TAG, 256, 64, 64, 64, EPI>(a, stream);
}

}  // namespace

// bf16x3-split variants of the big NT stages (float activations, pre-split weight planes, N >= 64).
// Returns DX_ENOSUP for a stage it does not cover so that the caller falls back to the
// fp32-MFMA kernel.
int launch_nt_b3(const NTArgs &a, int epi, int stage, hipStream_t stream) {
  if (!a.Wb || a.g.seglen % 32 || (a.K / a.ksplit) % 32 || a.N < 64 || a.K % 8) return DX_ENOSUP;
  switch (stage) {
    case ST_CONV1_FWD: return epi == EPI_BIAS_RELU ? launch_b3<ST_CONV1_FWD, EPI_BIAS_RELU>(a, stream) : DX_ENOSUP;
    case ST_CONV2_FWD: return epi == EPI_BIAS_RELU ? launch_b3<ST_CONV2_FWD, EPI_BIAS_RELU>(a, stream) : DX_ENOSUP;
    case ST_FC_FWD: return epi == EPI_BIAS ? launch_b3<ST_FC_FWD, EPI_BIAS>(a, stream) : DX_ENOSUP;
    case ST_FC_DGRAD: return epi == EPI_MASK ? launch_b3<ST_FC_DGRAD, EPI_MASK>(a, stream) : DX_ENOSUP;
    case ST_CONV2_DGRAD: return epi == EPI_MASK ? launch_b3<ST_CONV2_DGRAD, EPI_MASK>(a, stream) : DX_ENOSUP;
    case ST_CONV1_DGRAD: return epi == EPI_MASK ? launch_b3<ST_CONV1_DGRAD, EPI_MASK>(a, stream) : DX_ENOSUP;
    default: return DX_ENOSUP;
  }
}

}  // namespace dx
