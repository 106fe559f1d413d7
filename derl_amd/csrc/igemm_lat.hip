// Latency-shaped NT GEMM for small batches (rollout act, multi-GPU minibatch shards).
//
// The LDS-tiled kernels in igemm.hip walk K serially per 64x64 tile: at rollout sizes the chip
// holds a few dozen workgroups, each a chain of 8..49 load -> barrier -> LDS -> MFMA round
// trips (13-20 us per layer whatever the batch).  Here a workgroup owns ONE 32x32 output tile
// and its NW waves split K NW ways:
//   * every global load of a wave's K range (coalesced: CK/4 adjacent lanes per row) is issued
//     before its first MFMA, so a tile costs about one memory round trip plus KW/2 MFMAs;
//   * operands reach the MFMA layout (lane l: row/col l&31, k-half l>>5) through a WAVE-PRIVATE
//     LDS chunk of CK k-columns: written and read by the same wave, ordered by the LDS queue,
//     so there is no workgroup barrier inside the K loop.  (Loading straight into the MFMA
//     layout puts every lane on its own cache line: measured 16 B/cycle/CU, 3-4x slower.)
//   * the NW partial tiles meet in LDS once, in a fixed order (deterministic sums).
// Rows / columns past M / N compute garbage that is never stored (GEMM rows are independent),
// so there is no masking anywhere in the loop.
#include "igemm_dev.hpp"
#include <cstdlib>

namespace dx {
namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

template <bool U8> struct LatRaw { using type = f32x4; };
template <> struct LatRaw<true> { using type = uint32_t; };

__device__ __forceinline__ f32x4 lat_expand(f32x4 v) { return v; }
__device__ __forceinline__ f32x4 lat_expand(uint32_t w) {
  f32x4 v;
  v.x = dequant_u8(w & 0xff); v.y = dequant_u8((w >> 8) & 0xff);
  v.z = dequant_u8((w >> 16) & 0xff); v.w = dequant_u8(w >> 24);
  return v;
}

// SEGLEN: elements per contiguous K run (0 = K is one run); NW waves of KW k-elements each,
// staged CK at a time.
template <int TAG, int EPI, bool AU8, int SEGLEN, int NW, int KW, int CK>
__global__ __launch_bounds__(64 * NW) void igemm_nt_lat_kernel(const NTArgs a) {
  static_assert(KW % CK == 0 && CK % 8 == 0 && (SEGLEN == 0 || SEGLEN % CK == 0), "K chunks stay inside a run");
  constexpr int NC = KW / CK;    // chunks per wave
  constexpr int LPR = CK / 4;    // lanes per row of a chunk (4 k each)
  constexpr int RPI = 64 / LPR;  // rows per load instruction
  constexpr int NI = 32 / RPI;   // load instructions per chunk and operand
  constexpr int LD = CK + 4;     // conflict-free ds_read_b128 (36- or 20-float rows)
  constexpr int STAGE = NW * 2 * 32 * LD, RED = NW * 16 * 64;
  __shared__ __attribute__((aligned(16))) float smem[STAGE > RED ? STAGE : RED];
  const Gather &g = a.g;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float *As = smem + wave * (2 * 32 * LD);
  float *Bs = As + 32 * LD;
  const int lr = lane / LPR, lq = lane % LPR;
  const int kwbeg = blockIdx.z * (NW * KW) + wave * KW;
  const long long pitch = g.seg_off[1];  // runs are equally spaced (checked on the host)

  long long abase[NI];
  const float *wrow[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int m = blockIdx.x * 32 + i * RPI + lr, n = blockIdx.y * 32 + i * RPI + lr;
    abase[i] = decode_row(g, m, m < a.M).base + 4 * lq;
    wrow[i] = a.Wp + static_cast<long long>(n < a.N ? n : 0) * a.K + kwbeg + 4 * lq;
  }
  typename LatRaw<AU8>::type araw[NC][NI];
  f32x4 braw[NC][NI];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int k = kwbeg + c * CK;  // uniform
    long long koff = k;
    if constexpr (SEGLEN != 0) {
      const int seg = k / SEGLEN;
      koff = seg * pitch + (k - seg * SEGLEN);
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      if constexpr (AU8)
        araw[c][i] = *reinterpret_cast<const uint32_t *>(static_cast<const uint8_t *>(g.src) + abase[i] + koff);
      else
        araw[c][i] = *reinterpret_cast<const f32x4 *>(static_cast<const float *>(g.src) + abase[i] + koff);
      braw[c][i] = *reinterpret_cast<const f32x4 *>(wrow[i] + c * CK);
    }
  }
  // keep every load ahead of the first MFMA: left alone, the scheduler sinks them next to
  // their uses (2 in flight = one exposed round trip per K step)
  __builtin_amdgcn_sched_barrier(0);
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      *reinterpret_cast<f32x4 *>(&As[(i * RPI + lr) * LD + 4 * lq]) = lat_expand(araw[c][i]);
      *reinterpret_cast<f32x4 *>(&Bs[(i * RPI + lr) * LD + 4 * lq]) = braw[c][i];
    }
#pragma unroll
    for (int j = 0; j < CK / 8; ++j) {
      const f32x4 af = *reinterpret_cast<const f32x4 *>(&As[r * LD + 8 * j + 4 * h]);
      const f32x4 bf = *reinterpret_cast<const f32x4 *>(&Bs[r * LD + 8 * j + 4 * h]);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, bf.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, bf.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, bf.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, bf.w, acc, 0, 0, 0);
    }
  }
  __syncthreads();  // the partial tiles reuse the staging memory
  float *red = smem;
#pragma unroll
  for (int i = 0; i < 16; ++i) red[(wave * 16 + i) * 64 + lane] = acc[i];
  __syncthreads();
  // waves 0..3 finish accumulator registers 4w..4w+3 (C/D layout: col = lane&31,
  // row = (i&3) + 8*(i>>2) + 4*(lane>>5)): 128-byte row stores
  if (wave >= 4) return;
  float *out = a.out + (a.ksplit > 1 ? blockIdx.z * a.slab_stride : 0);
  const int nn = blockIdx.y * 32 + r;
  // one bias load before the stores (a load after a store waits for it: possible alias);
  // split-K partials: the bias rides on slab 0
  const bool has_bias = (EPI == EPI_BIAS || EPI == EPI_BIAS_RELU) && (a.ksplit == 1 || blockIdx.z == 0);
  const float bias_v = has_bias && nn < a.N ? a.bias[nn] : 0.f;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int i = 4 * wave + t;
    float v = red[i * 64 + lane];
#pragma unroll
    for (int w = 1; w < NW; ++w) v += red[(w * 16 + i) * 64 + lane];
    const int mm = blockIdx.x * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
    if (mm >= a.M || nn >= a.N) continue;
    v += bias_v;
    if (a.ksplit == 1 && EPI == EPI_BIAS_RELU) v = v > 0.f ? v : 0.f;
    out[static_cast<long long>(mm) * a.ldc + nn] = v;
  }
}

template <int TAG, int EPI, bool AU8, int SEGLEN, int NW, int KW, int CK>
int launch_lat_as(const NTArgs &a, hipStream_t stream) {
  dim3 grid(cdiv(a.M, 32), cdiv(a.N, 32), a.ksplit);
  hipLaunchKernelGGL((igemm_nt_lat_kernel<TAG, EPI, AU8, SEGLEN, NW, KW, CK>), grid, dim3(64 * NW), 0, stream, a);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

// Largest tile count each stage hands to the latency kernels (measured crossovers against the
// LDS-tiled kernels on MI355X: conv0 wins up to 3200+ tiles, conv1 at 1296, conv2 / linear lose
// from ~800).  DX_LAT_MAX_TILES overrides all of them (0 = off).
int lat_max_tiles(int stage) {
  const int env = DX_ENV("DX_LAT_MAX_TILES", -1);
  if (env >= 0) return env;
  switch (stage) {
    case ST_CONV0_FWD: return 4096;
    case ST_CONV1_FWD: return 2048;
    case ST_CONV2_FWD: return 600;
    case ST_FC_FWD: return 600;
    default: return 1 << 20;
  }
}

}  // namespace

// DX_ENOSUP = shape not covered (the caller falls back to the LDS-tiled kernels).
int launch_nt_lat(const NTArgs &a, bool a_u8, int epi, int stage, hipStream_t stream) {
  const Gather &g = a.g;
  const long long tiles = static_cast<long long>(cdiv(a.M, 32)) * cdiv(a.N, 32) * a.ksplit;
  if (tiles > lat_max_tiles(stage) || g.check || a.om.enabled) return DX_ENOSUP;
  for (int s = 1; s < g.nseg; ++s)
    if (g.seg_off[s] != s * g.seg_off[1]) return DX_ENOSUP;
  const int kper = a.K / a.ksplit;
  if (a.K % a.ksplit) return DX_ENOSUP;
  switch (stage) {
    case ST_CONV0_FWD:
      if (a_u8 && epi == EPI_BIAS_RELU && g.seglen == 32 && kper == 256)
        return launch_lat_as<ST_CONV0_FWD, EPI_BIAS_RELU, true, 32, 4, 64, 32>(a, stream);
      break;
    case ST_CONV1_FWD:
      if (!a_u8 && epi == EPI_BIAS_RELU && g.seglen == 128 && kper == 512)
        return launch_lat_as<ST_CONV1_FWD, EPI_BIAS_RELU, false, 128, 4, 128, 32>(a, stream);
      break;
    case ST_CONV2_FWD:
      if (!a_u8 && epi == EPI_BIAS_RELU && g.seglen == 192 && kper == 576)
        return launch_lat_as<ST_CONV2_FWD, EPI_BIAS_RELU, false, 192, 6, 96, 32>(a, stream);
      break;
    case ST_FC_FWD:
      if (a_u8 || epi != EPI_BIAS || g.nseg != 1) break;
      if (kper == 448) return launch_lat_as<ST_FC_FWD, EPI_BIAS, false, 0, 7, 64, 32>(a, stream);
      break;
    case ST_HEADS_FWD:
      if (!a_u8 && epi == EPI_BIAS && g.nseg == 1 && kper == 512)
        return launch_lat_as<ST_HEADS_FWD, EPI_BIAS, false, 0, 4, 128, 32>(a, stream);
      break;
    default: break;
  }
  return DX_ENOSUP;
}

}  // namespace dx
