// Plain NT GEMM C[m][n] = epi(sum_k A[m][k] * W[n][k]) for the linear layer's forward and dgrad
// (derl/models.py:112-115 and its autograd backward), built like the weight-gradient kernels
// of round 2: operands arrive by LDS-DMA in a ring of three K stages with ONE barrier per stage
// behind a counted vmcnt, and the fragment reads are pinned behind the first MFMA of each group.
//
//  * Workgroup tile 128 x 128, 8 waves of 64 x 32 (two 32x32 MFMA tiles each), K stage = 32: a
//    stage is 128 + 128 rows of 128 bytes = 32 pieces of 1 KiB (8 rows each), 4 per wave.
//  * Both operands are k-contiguous, so a lane reads 4 consecutive k of ITS row with one
//    ds_read_b128 (4 k-pairs = 4 MFMAs per fragment pair); 32 lanes = 32 rows at a 128-byte
//    stride would hit 16-way bank conflicts in a linear image, and the DMA writes LDS linearly --
//    so the swizzle is applied to the SOURCE address: slot s of row R holds the row's 16-byte
//    chunk s ^ ((R >> 1) & 7), and a reader of chunk c looks in slot c ^ ((R >> 1) & 7)
//    (conflict-free for every 16-lane group of a ds_read_b128).
//  * Epilogues: bias (forward) or ReLU mask from the kept activation (dgrad), as in igemm.hip.
#include <algorithm>
#include <vector>

#include "igemm_dev.hpp"

namespace dx {
namespace {

using f4 = __attribute__((ext_vector_type(4))) float;

constexpr int kBM = 128, kBN = 128, kBK = 32;
constexpr int kStageFloats = (kBM + kBN) * kBK;  // 8192 floats = 32 KiB

template <int EPI, int STAGES>
__global__ __launch_bounds__(512, 2) void nt_dma_kernel(const NtDmaArgs a, unsigned long long *stamps) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const unsigned long long t_entry = (kDiag && stamps) ? __builtin_amdgcn_s_memrealtime() : 0;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  const int gn = (a.N + kBN - 1) / kBN;
  const int bn = blockIdx.x % gn, bm = blockIdx.x / gn;
  const int m0 = bm * kBM, n0 = bn * kBN;
  const int ksteps = a.K / kBK;

  // this wave's 4 pieces of a stage: pieces 4w .. 4w+3 of the 32 (0-15: A rows, 16-31: W rows)
  const float *src[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int piece = 4 * wave + p;
    const int R = (piece & 15) * 8 + (lane >> 3);         // row inside the A or W tile
    const int chunk = (lane & 7) ^ ((R >> 1) & 7);        // swizzle on the source
    if (piece < 16) {
      src[p] = a.A + static_cast<long long>(m0 + R) * a.lda + 4 * chunk;
    } else {
      const int n = min(n0 + R, a.N - 1);                 // rows past N: a valid row, never stored
      src[p] = a.W + static_cast<long long>(n) * a.K + 4 * chunk;
    }
  }
#define DX_NT_ISSUE(S, SLOT)                                                                     \
  {                                                                                              \
    float *dst_ = smem + (SLOT) * kStageFloats + 4 * wave * 256;                                 \
    _Pragma("unroll") for (int p = 0; p < 4; ++p)                                                \
        __builtin_amdgcn_global_load_lds(src[p] + (S) * kBK, dst_ + p * 256, 16, 0, 0);          \
  }

  const int wm = wave >> 2, wn = wave & 3;  // 2 x 4 waves of 64 x 32
  f32x16 acc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // lane offsets (bytes) of the four 8-k groups of a stage: chunk 2q + hi of row l31 (+32 t)
  const char *lds = reinterpret_cast<const char *>(smem);
  const unsigned x = (l31 >> 1) & 7;  // (row >> 1) & 7 is the same for rows l31, l31 + 32, + 64 ...
  unsigned aoff[4], boff[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const unsigned slot = ((2 * q + hi) ^ x) * 16;
    aoff[q] = (wm * 64 + l31) * 128 + slot;
    boff[q] = (kBM + wn * 32 + l31) * 128 + slot;
  }

  // STAGES - 1 stages in flight: the DMA of stage s + STAGES - 1 is issued behind the barrier of
  // stage s (every wave is then done reading stage s - 1, whose slot it overwrites)
  DX_NT_ISSUE(0, 0)
  if (STAGES == 3 && ksteps > 1) DX_NT_ISSUE(1, 1)
  int slot = 0;
  unsigned long long t_first = 0, c_first = 0;
  for (int s = 0; s < ksteps; ++s) {
    if (kDiag && stamps && s == 1) { t_first = __builtin_amdgcn_s_memrealtime(); c_first = __builtin_amdgcn_s_memtime(); }
    if (STAGES == 3 && s + 1 < ksteps) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // not __syncthreads(): its fence would wait for the stage in flight
    asm volatile("" ::: "memory");
    if (s + STAGES - 1 < ksteps) {
      const int nslot = slot >= 1 ? slot - 1 : slot + STAGES - 1;
      DX_NT_ISSUE(s + STAGES - 1, nslot)
    }
    const char *base = lds + slot * kStageFloats * 4;
    f4 af[2][2], bf[2];
    af[0][0] = *reinterpret_cast<const f4 *>(base + aoff[0]);
    af[0][1] = *reinterpret_cast<const f4 *>(base + aoff[0] + 32 * 128);
    bf[0] = *reinterpret_cast<const f4 *>(base + boff[0]);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = q & 1, nx = c ^ 1;
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c][0][0], bf[c][0], acc[0], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (q + 1 < 4) {  // the next group's fragments, requested in the first MFMA's shadow
        af[nx][0] = *reinterpret_cast<const f4 *>(base + aoff[q + 1]);
        af[nx][1] = *reinterpret_cast<const f4 *>(base + aoff[q + 1] + 32 * 128);
        bf[nx] = *reinterpret_cast<const f4 *>(base + boff[q + 1]);
      }
      __builtin_amdgcn_sched_barrier(0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c][1][0], bf[c][0], acc[1], 0, 0, 0);
#pragma unroll
      for (int e = 1; e < 4; ++e) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c][0][e], bf[c][e], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c][1][e], bf[c][e], acc[1], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    slot = slot == STAGES - 1 ? 0 : slot + 1;
  }
#undef DX_NT_ISSUE
  const unsigned long long t_loop = (kDiag && stamps) ? __builtin_amdgcn_s_memrealtime() : 0;
  const unsigned long long c_loop = (kDiag && stamps) ? __builtin_amdgcn_s_memtime() : 0;

  // epilogue: C/D layout of the 32x32 MFMA: column = l31, row = (r & 3) + 8 (r >> 2) + 4 hi
  const int n = n0 + wn * 32 + l31;
  if (n < a.N) {
  const float bias = (EPI == EPI_BIAS) ? a.bias[n] : 0.f;
  const long long row0 = static_cast<long long>(m0 + wm * 64 + 4 * hi) * a.ldc + n;
  float aux[2][16];
  if (EPI == EPI_MASK) {  // every load before the first store (vmcnt counts stores too)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        aux[t][r] = a.mask_src[row0 + static_cast<long long>(32 * t + (r & 3) + 8 * (r >> 2)) * a.ldc];
  }
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = acc[t][r] + bias;
      if (EPI == EPI_MASK) v = aux[t][r] > 0.f ? v : 0.f;
      a.out[row0 + static_cast<long long>(32 * t + (r & 3) + 8 * (r >> 2)) * a.ldc] = v;
    }
  }
  if (kDiag && stamps && lane == 0) {  // DX_NT_DIAG: 100 MHz ticks (entry, first stage done, loop end, exit) + loop cycles
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long *o = stamps + (static_cast<long long>(blockIdx.x) * 8 + wave) * 5;
    o[0] = t_entry; o[1] = t_first; o[2] = t_loop; o[3] = __builtin_amdgcn_s_memrealtime(); o[4] = c_loop - c_first;
  }
}

// STAGES = 3 (96 KiB, one workgroup per CU) when the grid has at most one workgroup per CU anyway,
// 2 (64 KiB, two per CU: one's epilogue runs under the other's K loop) for the wide dgrad
template <int EPI, int STAGES>
int launch_as(const NtDmaArgs &a, hipStream_t stream) {
  constexpr int lds = STAGES * kStageFloats * 4;
  DX_LDS_OPT_IN((nt_dma_kernel<EPI, STAGES>), lds);
  const int grid = (a.M / kBM) * cdiv(a.N, kBN);
#if DX_DIAG
  const int diag = DX_ENV("DX_NT_DIAG", 0);
#else
  constexpr int diag = 0;
#endif
  if (!diag) {
    hipLaunchKernelGGL((nt_dma_kernel<EPI, STAGES>), dim3(grid), dim3(512), lds, stream, a, nullptr);
    DX_LAUNCH_CHECK();
    return DX_OK;
  }
#if DX_DIAG
  // diagnostic: in-kernel stamps, summarised on stderr (synchronous; never on the product path)
  unsigned long long *dev = nullptr;
  const size_t count = static_cast<size_t>(grid) * 8 * 5;
  DX_HIP(hipMalloc(&dev, count * 8));
  hipLaunchKernelGGL((nt_dma_kernel<EPI, STAGES>), dim3(grid), dim3(512), lds, stream, a, dev);
  DX_LAUNCH_CHECK();
  DX_HIP(hipStreamSynchronize(stream));
  std::vector<unsigned long long> h(count);
  DX_HIP(hipMemcpy(h.data(), dev, count * 8, hipMemcpyDeviceToHost));
  DX_HIP(hipFree(dev));
  unsigned long long first = ~0ull, last = 0;
  std::vector<double> pro, loop, epi, cyc;
  for (size_t i = 0; i < count; i += 5) {
    first = std::min(first, h[i]); last = std::max(last, h[i + 3]);
    pro.push_back((h[i + 1] - h[i]) * 0.01); loop.push_back((h[i + 2] - h[i + 1]) * 0.01);
    epi.push_back((h[i + 3] - h[i + 2]) * 0.01); cyc.push_back(static_cast<double>(h[i + 4]));
  }
  auto med = [](std::vector<double> &v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  const int ks = a.K / kBK - 1;
  const double lm = med(loop), cm = med(cyc);
  fprintf(stderr, "[nt_dma M=%d N=%d K=%d grid=%d] span %.1f us | per wave: to first stage %.2f us, %d stages %.2f us "
          "(%.0f cycles/stage, ideal %d, clock %.2f GHz), epilogue %.2f us\n", a.M, a.N, a.K, grid, (last - first) * 0.01,
          med(pro), ks, lm, cm / ks, 32 * 64 * (grid > 256 && STAGES == 2 ? 4 : 2), cm / lm * 1e-3, med(epi));
#endif
  return DX_OK;
}

}  // namespace

bool nt_dma_supported(int M, int N, int K) { return M >= 1024 && M % kBM == 0 && K % kBK == 0 && K >= 64 && N >= 128 && N % 4 == 0; }

int launch_nt_dma(const NtDmaArgs &a, int epi, hipStream_t stream) {
  DX_REQUIRE(a.A && a.W && a.out && nt_dma_supported(a.M, a.N, a.K) && a.lda >= a.K && a.ldc >= a.N,
             "nt_dma: unsupported problem M=%d N=%d K=%d", a.M, a.N, a.K);
  DX_REQUIRE(aligned(a.A, 16) && aligned(a.W, 16) && a.lda % 4 == 0, "nt_dma: operands must be 16-byte aligned");
  if (epi == EPI_BIAS) {
    DX_REQUIRE(a.bias != nullptr, "nt_dma: bias epilogue without a bias");
    return launch_as<EPI_BIAS, 3>(a, stream);
  }
  if (epi == EPI_MASK) {
    DX_REQUIRE(a.mask_src != nullptr, "nt_dma: mask epilogue without a mask source");
    return launch_as<EPI_MASK, 2>(a, stream);
  }
  return fail(DX_ENOSUP, "nt_dma: epilogue %d not instantiated", epi);
}

}  // namespace dx
