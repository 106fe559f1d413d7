// Package power and sustained rate of bare matrix-instruction loops (operands in registers, pseudo-random data):
// what does a multiply-accumulate COST on this part, per instruction shape?  rocm-smi is read from the host while
// the queued launches run.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/mfma_power.hip -o tools/ubench/mfma_power
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

__device__ unsigned mix(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ bf16x8 rnd8(unsigned seed) {  // eight bf16 in [1, 2) with random mantissas (random sign)
  u32x4 v;
  for (int i = 0; i < 4; ++i) {
    const unsigned r = mix(seed * 4 + i);
    v[i] = (r & 0x807f807fu) | 0x3f803f80u;
  }
  return __builtin_bit_cast(bf16x8, v);
}

// mode 0: v_mfma_f32_16x16x32_bf16, 1: v_mfma_f32_32x32x16_bf16, 2: v_mfma_f32_32x32x2_f32, 3: v_mfma_f32_16x16x4_f32
template <int MODE>
__global__ __launch_bounds__(512) void loop_kernel(int iters, float *out) {
  const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = rnd8(tid * 8 + i); b[i] = rnd8(tid * 8 + 4 + i); }
  float s = 0.f;
  if constexpr (MODE == 0) {
    f32x4 acc[4] = {};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[(i + j) & 3], acc[j], 0, 0, 0);
    for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][3];
  } else if constexpr (MODE == 1) {
    f32x16 acc[4] = {};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[(i + j) & 3], acc[j], 0, 0, 0);
    for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][9];
  } else if constexpr (MODE == 2) {
    f32x16 acc[4] = {};
    float fa[4], fb[4];
    for (int i = 0; i < 4; ++i) { fa[i] = __builtin_bit_cast(float, (mix(tid * 8 + i) & 0x807fffffu) | 0x3f800000u); fb[i] = __builtin_bit_cast(float, (mix(tid * 8 + 4 + i) & 0x807fffffu) | 0x3f800000u); }
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[(i + j) & 3], acc[j], 0, 0, 0);
    for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][9];
  } else {
    f32x4 acc[4] = {};
    float fa[4], fb[4];
    for (int i = 0; i < 4; ++i) { fa[i] = __builtin_bit_cast(float, (mix(tid * 8 + i) & 0x807fffffu) | 0x3f800000u); fb[i] = __builtin_bit_cast(float, (mix(tid * 8 + 4 + i) & 0x807fffffu) | 0x3f800000u); }
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i], fb[(i + j) & 3], acc[j], 0, 0, 0);
    for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][3];
  }
  if (s == 123.456f) out[0] = s;  // (keeps the loop)
}

static double read_power() {
  FILE *p = popen("rocm-smi --showpower --csv 2>/dev/null", "r");
  if (!p) return -1;
  char line[512];
  double w = -1;
  while (fgets(line, sizeof line, p)) {
    if (strncmp(line, "card", 4) == 0) {
      const char *c = strchr(line, ',');
      if (c) w = atof(c + 1);
    }
  }
  pclose(p);
  return w;
}

template <int MODE>
void run(const char *name, int threads, double macs_per_inst, double cycles_per_inst, float *out) {
  const int iters = 20000, insts = iters * 16;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(loop_kernel<MODE>, dim3(256), dim3(threads), 0, 0, iters, out);
  (void)hipDeviceSynchronize();
  const int launches = 60;
  hipEventRecord(e0);
  for (int l = 0; l < launches; ++l) hipLaunchKernelGGL(loop_kernel<MODE>, dim3(256), dim3(threads), 0, 0, iters, out);
  hipEventRecord(e1);
  double pw[3];
  for (int i = 0; i < 3; ++i) pw[i] = read_power();  // (the queue is still running: each read takes ~0.2 s)
  (void)hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double waves = 256.0 * threads / 64, total_insts = waves * insts * launches;
  const double tmac = total_insts * macs_per_inst / (ms * 1e-3) / 1e12;
  const double ghz = static_cast<double>(insts) * launches * cycles_per_inst * (threads / 256.0) / (ms * 1e-3) / 1e9;  // if the pipe is saturated
  printf("%-28s %d waves/SIMD: %7.1f ms, %7.1f TMAC/s, power %.0f %.0f %.0f W -> %.2f pJ/MAC at the last reading (clock if pipe-bound %.2f GHz)\n", name,
         threads / 256, ms, tmac, pw[0], pw[1], pw[2], pw[2] / tmac, ghz);
}

int main() {
  float *out;
  hipMalloc(&out, 64);
  for (int threads : {256, 512}) {
    run<0>("bf16 16x16x32", threads, 8192, 16, out);
    run<1>("bf16 32x32x16", threads, 16384, 32, out);
    run<2>("f32 32x32x2", threads, 2048, 64, out);
    run<3>("f32 16x16x4", threads, 1024, 32, out);
  }
  return 0;
}
