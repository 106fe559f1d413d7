// The rollout's 3136 -> 512 linear layer (derl/models.py:112-115 inside the batched Policy.act forward
// of derl/policies.py:61-80) for up to a few hundred rows: a WEIGHT-STATIONARY split-K kernel.
//
// The latency kernel it replaces (igemm_lat.hip) gives every 32 x 32 output tile its own workgroup, so
// each of the 4-8 row tiles streams the whole 6.4 MB weight matrix again (51 MB through L2 for 0.2 GMAC
// at 128 rows: 0.21 of the fp32-MFMA peak).  Here the grid is cut over the WEIGHTS only -- 16 column
// tiles of 32 x 14 K parts of 224 = 224 workgroups, one per CU -- and a workgroup keeps its 32 x 224
// slice of W (28 KB) in LDS for the whole launch while it walks the rows in chunks of 128: every weight
// is read from memory exactly once per launch, the activations (1.6 MB at 128 rows: L2-resident after
// the first touch of each XCD) once per column tile.  All loads of a chunk are issued before the first
// is waited for (one memory round trip), rows are 228-float pitched in LDS (conflict-free ds_read_b128
// for the 32x32x2 fp32 MFMA: a lane reads 4 consecutive k of its row), the 8 waves are 4 row tiles x 2
// K halves and meet in LDS once.  Output: 14 partial slabs [part][rows][512] (the bias rides on slab 0)
// that the heads launch sums -- the split-K contract of the kernel it replaces, with 14 parts.
#include "igemm.hpp"
#include "igemm_dev.hpp"

namespace dx {
namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kK = 3136, kN = 512, kParts = 14, kKP = kK / kParts;  // 224 k per part
constexpr int kPitch = kKP + 4;                                     // floats per LDS row
constexpr int kRows = 128;                                          // rows per chunk
constexpr int oW = 0, oA = oW + 32 * kPitch * 4, kLdsBytes = oA + kRows * kPitch * 4;
static_assert(kKP % 16 == 0 && kLdsBytes <= 160 * 1024, "K part in whole b128 steps of both halves; LDS budget");
static_assert(4 * 16 * 64 * 4 <= kRows * kPitch * 4, "reduction scratch fits the row chunk's bytes");

struct FcRolloutArgs {
  const float *A;     // [M][3136]
  const float *W;     // [512][3136] packed (k in activation order)
  const float *bias;  // [512]
  float *slabs;       // [14][M][512]
  int M;
};

__global__ __launch_bounds__(512) void fc_rollout_kernel(const FcRolloutArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  float *Ws = reinterpret_cast<float *>(smem + oW);
  float *As = reinterpret_cast<float *>(smem + oA);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int part = blockIdx.x % kParts, ntile = blockIdx.x / kParts;
  const int mt = wave & 3, kh = wave >> 2;
  const int r = lane & 31, h = lane >> 5;
  const int chunks = (a.M + kRows - 1) / kRows;
  const int n = 32 * ntile + r;
  const float bias = part == 0 ? a.bias[n] : 0.f;  // (before any store: a load behind a store waits for it)

  // this workgroup's slice of W: 32 rows x 56 pieces of 16 bytes, 3.5 per thread
  f32x4 wv[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int i = min(tid + 512 * u, 32 * 56 - 1);
    wv[u] = *reinterpret_cast<const f32x4 *>(a.W + static_cast<long long>(32 * ntile + i / 56) * kK + part * kKP + 4 * (i % 56));
  }
  for (int c = 0; c < chunks; ++c) {
    const int row0 = c * kRows;
    // the chunk's rows: 128 x 56 pieces, 14 per thread, rows past M read row M - 1 (their results are not stored)
    f32x4 av[14];
#pragma unroll
    for (int u = 0; u < 14; ++u) {
      const int i = tid + 512 * u;
      const int row = min(row0 + i / 56, a.M - 1);
      av[u] = *reinterpret_cast<const f32x4 *>(a.A + static_cast<long long>(row) * kK + part * kKP + 4 * (i % 56));
    }
    __builtin_amdgcn_sched_barrier(0);
    if (c == 0) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = tid + 512 * u;
        if (i < 32 * 56) *reinterpret_cast<f32x4 *>(Ws + (i / 56) * kPitch + 4 * (i % 56)) = wv[u];
      }
    } else {
      __syncthreads();  // the previous chunk's reduction has left the row region
    }
#pragma unroll
    for (int u = 0; u < 14; ++u) {
      const int i = tid + 512 * u;
      *reinterpret_cast<f32x4 *>(As + (i / 56) * kPitch + 4 * (i % 56)) = av[u];
    }
    __syncthreads();
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const float *ap = As + (32 * mt + r) * kPitch + kh * (kKP / 2) + 4 * h;
    const float *wp = Ws + r * kPitch + kh * (kKP / 2) + 4 * h;
#pragma unroll
    for (int s = 0; s < kKP / 16; ++s) {  // 14 steps of 8 k
      const f32x4 af = *reinterpret_cast<const f32x4 *>(ap + 8 * s);
      const f32x4 bf = *reinterpret_cast<const f32x4 *>(wp + 8 * s);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, bf.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, bf.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, bf.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, bf.w, acc, 0, 0, 0);
    }
    __syncthreads();  // every wave has read its rows: the K halves meet in the row region
    float *red = As;
    if (kh == 1) {
#pragma unroll
      for (int i = 0; i < 16; ++i) red[(mt * 16 + i) * 64 + lane] = acc[i];
    }
    __syncthreads();
    if (kh == 0) {  // C/D layout: column = lane & 31, row = (i & 3) + 8 (i >> 2) + 4 (lane >> 5)
      float *out = a.slabs + (static_cast<long long>(part) * a.M + row0 + 32 * mt) * kN + n;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
        const float v = (acc[i] + red[(mt * 16 + i) * 64 + lane]) + bias;
        if (row0 + 32 * mt + row < a.M) out[static_cast<long long>(row) * kN] = v;
      }
    }
  }
}

}  // namespace

int fc_rollout_parts() { return kParts; }

bool fc_rollout_supported(int M, int N, int K) { return M >= 1 && N == kN && K == kK; }

// slabs [14][M][512] <- partial sums of A [M][3136] x W^T [512][3136] over 14 K parts, + bias on slab 0
int launch_fc_rollout(const float *A, const float *W, const float *bias, float *slabs, int M, hipStream_t stream) {
  DX_REQUIRE(A && W && bias && slabs && M >= 1, "fc_rollout: bad arguments");
  DX_REQUIRE(aligned(A, 16) && aligned(W, 16), "fc_rollout: operands must be 16-byte aligned");
  DX_LDS_OPT_IN(fc_rollout_kernel, kLdsBytes);
  const FcRolloutArgs a{A, W, bias, slabs, M};
  hipLaunchKernelGGL(fc_rollout_kernel, dim3(kParts * (kN / 32)), dim3(512), kLdsBytes, stream, a);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

}  // namespace dx
