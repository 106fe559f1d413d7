#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
  const int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false);
  return v + __builtin_bit_cast(float, moved);
}
__device__ __forceinline__ float swap_sum32(float a, float b) {
  const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
  return __builtin_bit_cast(float, r[0]) + __builtin_bit_cast(float, r[1]);
}
__device__ __forceinline__ float swap_sum16(float a, float b) {
  const auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
  return __builtin_bit_cast(float, r[0]) + __builtin_bit_cast(float, r[1]);
}
__device__ __forceinline__ float row_sum_all(float v) {
  v = dpp_add<0xB1, 0xf>(v);
  v = dpp_add<0x4E, 0xf>(v);
  v = dpp_add<0x141, 0xf>(v);
  return dpp_add<0x140, 0xf>(v);
}
__global__ void k(float *o) {
  const int l = threadIdx.x;
  float p0 = 1.f, p1 = 10.f, p2 = 100.f, p3 = 1000.f + l;
  const float s32a = swap_sum32(p0, p2), s32b = swap_sum32(p1, p3);
  const float s16 = swap_sum16(s32a, s32b);
  o[l] = s32a; o[64 + l] = s32b; o[128 + l] = s16; o[192 + l] = row_sum_all(s16);
}
int main() {
  float *d, h[256];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char *names[4] = {"s32a", "s32b", "s16", "rowsum"};
  for (int i = 0; i < 4; ++i) { printf("%s:", names[i]); for (int l = 0; l < 64; l += 8) printf(" [%d]=%g", l, h[64 * i + l]); printf("\n"); }
  printf("expected rowsum: 64, 640, 6400, 64000+2016=66016\n");
  return 0;
}
