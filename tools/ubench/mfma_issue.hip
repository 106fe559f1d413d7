// In-kernel cycles per v_mfma_f32_16x16x32_bf16 / v_mfma_f32_32x32x16_bf16 (s_memtime around the loop, and the shader clock
// from s_memrealtime) for one or two waves per SIMD and 1 / 2 / 4 independent accumulators: what does ONE wave's
// dependent stream sustain?
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/mfma_issue.hip -o tools/ubench/mfma_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

__device__ unsigned mix(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ bf16x8 rnd8(unsigned seed) {
  u32x4 v;
  for (int i = 0; i < 4; ++i) v[i] = (mix(seed * 4 + i) & 0x807f807fu) | 0x3f803f80u;
  return __builtin_bit_cast(bf16x8, v);
}

template <int SHAPE, int NACC>
__global__ void loop_kernel(int iters, unsigned long long *stamps, float *out) {
  const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = rnd8(tid * 8 + i); b[i] = rnd8(tid * 8 + 4 + i); }
  float s = 0.f;
  __syncthreads();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_amdgcn_s_memtime();
  if constexpr (SHAPE == 0) {
    f32x4 acc[NACC] = {};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[(i >> 2) & 3], acc[i % NACC], 0, 0, 0);
    for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][3];
  } else {
    f32x16 acc[NACC] = {};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 3], b[(i >> 2) & 3], acc[i % NACC], 0, 0, 0);
    for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][9];
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if ((threadIdx.x & 63) == 0) {
    stamps[(blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) * 2] = t1 - t0;
    stamps[(blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) * 2 + 1] = r1 - r0;
  }
  if (s == 123.456f) out[0] = s;
}

template <int SHAPE, int NACC>
void run(int threads, const char *name) {
  const int blocks = 256, iters = 4000, waves = blocks * threads / 64;
  unsigned long long *d;
  float *o;
  hipMalloc(&d, waves * 16);
  hipMalloc(&o, 4);
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((loop_kernel<SHAPE, NACC>), dim3(blocks), dim3(threads), 0, 0, iters, d, o);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(waves * 2);
  hipMemcpy(h.data(), d, waves * 16, hipMemcpyDeviceToHost);
  double cyc = 0, ticks = 0;
  for (int w = 0; w < waves; ++w) { cyc += h[2 * w]; ticks += h[2 * w + 1]; }
  const double n = 16.0 * iters;
  printf("%-28s %d waves/SIMD, %d accumulators: %.2f cycles per MFMA per wave (%.2f per SIMD), clock %.0f MHz, %.0f TFLOP/s chip\n", name,
         threads / 256, NACC, cyc / waves / n, cyc / waves / n / (threads / 256), cyc / ticks * 100.0,
         (SHAPE == 0 ? 16384.0 : 32768.0) * n * waves / (ticks / waves / 100e6) / 1e12);
  hipFree(d);
  hipFree(o);
}

int main() {
  run<0, 1>(256, "16x16x32"); run<0, 2>(256, "16x16x32"); run<0, 4>(256, "16x16x32");
  run<0, 1>(512, "16x16x32"); run<0, 2>(512, "16x16x32"); run<0, 4>(512, "16x16x32");
  run<1, 1>(256, "32x32x16"); run<1, 2>(256, "32x32x16"); run<1, 4>(256, "32x32x16");
  run<1, 2>(512, "32x32x16");
  return 0;
}
