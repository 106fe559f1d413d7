// Diagnostics: what the fp32 matrix pipe of this part sustains with no memory traffic at all
// (the ceiling the GEMM stages are judged against in DESIGN.md next to the data-sheet peak).
#include "common.hpp"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// every wave: `iters` rounds of NACC independent v_mfma_f32_32x32x2_f32 accumulations
template <int NACC>
__global__ __launch_bounds__(256) void mfma_f32_loop_kernel(int iters, float *out) {
  f32x16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  float a = 1.f + threadIdx.x * 1e-3f, b = 1.f - threadIdx.x * 1e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  if (s == 12345.678f) out[0] = s;  // keeps the loop alive, never true in practice
}

}  // namespace

// Launches blocks x 256 threads, each wave issuing iters x 4 MFMAs of 32x32x2 (4096 flop each).
// The caller times it with events; flops = blocks * 4 waves * iters * 4 * 4096.
extern "C" int dx_diag_mfma_f32(int blocks, int iters, float *out, void *stream) {
  DX_REQUIRE(blocks >= 1 && iters >= 1 && out, "dx_diag_mfma_f32: bad argument");
  hipLaunchKernelGGL(mfma_f32_loop_kernel<4>, dim3(blocks), dim3(256), 0, dx::as_stream(stream), iters, out);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

// The same with ONE accumulator per wave (every MFMA depends on the previous one): iters x 4 MFMAs.
extern "C" int dx_diag_mfma_f32_chain(int blocks, int iters, float *out, void *stream) {
  DX_REQUIRE(blocks >= 1 && iters >= 1 && out, "dx_diag_mfma_f32_chain: bad argument");
  hipLaunchKernelGGL(mfma_f32_loop_kernel<1>, dim3(blocks), dim3(256), 0, dx::as_stream(stream), 4 * iters, out);
  DX_LAUNCH_CHECK();
  return DX_OK;
}
