// Microbenchmark of convstack.hip's conv1 K-half loop (conv_half<1, KH, 6, 8>): the same code, alone in a
// kernel, with cycle stamps around it.  Modes pick which waves run it and what the LDS reads look like.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I derl_amd/csrc tools/ubench/conv1_loop.hip -o /tmp/conv1_loop
#define launch_convstack ubench_unused_launch_convstack
#include "../../derl_amd/csrc/convstack.hip"
#undef launch_convstack
#include <cstdarg>

namespace dx {
char *error_buffer() { static thread_local char b[512]; return b; }
void count_launch() {}

namespace {
// mode bit 0: waves 4-7 idle; bit 1: waves 0-3 idle; bit 2: every lane reads the SAME address (broadcast);
// bit 3: registers only (no LDS reads: the MFMA chain by itself)
__global__ __launch_bounds__(512) void ubench_kernel(const uint8_t *wsrc, float *out, unsigned long long *stamps, int mode) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nt = wave & 3, kh2 = wave >> 2;
  for (int i = tid; i < kLdsBytes / 4; i += 512) reinterpret_cast<uint32_t *>(smem)[i] = 0x3f803f80u;
  u32x4 w1[8][3];
#pragma unroll
  for (int s = 0; s < 8; ++s)
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) w1[s][pl] = *reinterpret_cast<const u32x4 *>(wsrc + ((s * 3 + pl) * 64 + lane) * 16);
  __syncthreads();
  const int n16 = lane & 15, kq = lane >> 4;
  int pb[6];
#pragma unroll
  for (int mt = 0; mt < 6; ++mt) {
    const int p = min(16 * mt + n16, kP1 - 1), oy = p / 9, ox = p - 9 * oy;
    pb[mt] = (mode & 4) ? oY0 : oY0 + 2 * oy * kY0R + 2 * ox * kY0P + 16 * kq;
  }
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc[6] = {zero4, zero4, zero4, zero4, zero4, zero4};
  __builtin_amdgcn_sched_barrier(0);
  const unsigned long long t0 = __builtin_readcyclecounter();
  __builtin_amdgcn_sched_barrier(0);
  const bool idle = ((mode & 1) && kh2 == 1) || ((mode & 2) && kh2 == 0);
  if (!idle) {
    if (mode & 16) {
      for (int i = 0; i < 100; ++i) __builtin_amdgcn_s_sleep(100);  // calibration: 100 x 64 x 100 shader cycles
    } else if (mode & 8) {
      u32x4 x[2][3];
      load_pair<1, 0, 0, 0, 6>(smem, pb, x);
#pragma unroll
      for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int pr = 0; pr < 3; ++pr) {
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) asm volatile("" : "+v"(x[h][pl]));  // (distinct values as far as the compiler knows)
          acc[2 * pr] = mac_rest(mac_first(acc[2 * pr], w1[s], x[0]), w1[s], x[0]);
          acc[2 * pr + 1] = mac_rest(mac_first(acc[2 * pr + 1], w1[s], x[1]), w1[s], x[1]);
        }
    } else {
      u32x4 unused[9][3];
      const NextWeights<9, 0> none{nullptr, 0u, unused};
      if (kh2 == 0) conv_half<1, 0, 6, 8>(smem, pb, w1, acc, none);
      else conv_half<1, 1, 6, 8>(smem, pb, w1, acc, none);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int m = 0; m < 6; ++m) s += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
  out[blockIdx.x * 512 + tid] = s;
  if (lane == 0) stamps[blockIdx.x * 8 + wave] = t1 - t0;
}
}  // namespace
}  // namespace dx

int main() {
  using namespace dx;
  const int B = 256;
  uint8_t *wsrc; float *out; unsigned long long *stamps;
  hipMalloc(&wsrc, 8 * 3 * 64 * 16); hipMemset(wsrc, 0x3f, 8 * 3 * 64 * 16);
  hipMalloc(&out, B * 512 * 4); hipMalloc(&stamps, B * 8 * 8);
  hipFuncSetAttribute(reinterpret_cast<const void *>(ubench_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
  for (int mode : {16, 0, 1, 2, 4, 5, 8, 9}) {
    std::vector<unsigned long long> h(B * 8);
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(ubench_kernel, dim3(B), dim3(512), kLdsBytes, 0, wsrc, out, stamps, mode);
      (void)hipDeviceSynchronize();
    }
    hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
    double w[8] = {};
    for (int b = 0; b < B; ++b) for (int i = 0; i < 8; ++i) w[i] += double(h[b * 8 + i]) / B;
    printf("mode %d: cycles per wave", mode);
    for (int i = 0; i < 8; ++i) printf(" %6.0f", w[i]);
    printf("   (288 MFMAs per wave: %.1f / %.1f cycles per MFMA)\n", w[0] / 288, w[4] / 288);
  }
  return 0;
}
