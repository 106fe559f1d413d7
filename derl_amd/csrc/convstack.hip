// The rollout's conv stack as ONE launch with ONE workgroup per image: conv(4->32, k8, s4) + ReLU ->
// conv(32->64, k4, s2) + ReLU -> conv(64->64, k3, s1) + ReLU of derl/models.py:104-111 (the batched
// Policy.act forward of derl/policies.py:61-80), for 84 x 84 x 4 uint8 frames.
//
// Why: at rollout sizes (128-256 images) the layer-by-layer kernels (igemm_lat.hip: one 32x32 tile per
// workgroup) stream every operand of every tile through L2 -- 128 KB for 0.5 MMAC -- and ran at
// 0.21-0.29 of the fp32-MFMA peak, three dependent launches per step.  The whole conv stack of an image
// is image-local, and 256 images are 256 CUs: here a workgroup keeps ITS image's frame (28 KB), y0
// (20 x 20 x 32) and y1 (9 x 9 x 64) in LDS, never writes them to memory, and only y2 (7 x 7 x 64)
// leaves the CU.  The activations are read by the matrix instructions straight from the LDS images
// (a lane's address is its pixel's base + a compile-time tap offset: one ds_read_b128, no address
// arithmetic in the loop, pixel pitches of 36 / 72 floats keep the 16-lane read groups on distinct
// banks); the weights of conv1 / conv2 never touch LDS at all: wave (n tile, K half) is the ONLY reader
// of its 16 x 256 / 16 x 288 slice of the packed matrix and loads it straight into the B-fragment
// layout of v_mfma_f32_16x16x4_f32 (lane (n, kq) = 4 consecutive k of row n: 16 / 18 loads per lane,
// all issued before the first layer starts, L2-resident because every workgroup reads the same 272 KB).
//   conv0: bf16 matrix cores with EXACT operands (uint8 pixels; fp32 weights pre-split into three
//          bf16 planes by dx_cnn_pack), v_mfma_f32_32x32x16_bf16, fp32 accumulation -- conv0_b16.hip's
//          arithmetic; the planes sit in LDS (528-byte rows), 13 pixel tiles of 32 over 8 waves.
//   conv1: M = 81 pixels = 5 tiles of 16 + ONE pixel, N = 4 tiles of 16, K = 512 in two halves over
//          the 8 waves.  The 81st pixel rides on the vector ALUs beside the matrix pipe (4 fma per K
//          step and lane from one broadcast LDS read) instead of a sixth, 94 % empty, tile.
//   conv2: M = 49 = 3 tiles + one pixel, N = 4 tiles, K = 576 in two halves, the same way.
// Results equal the layer-by-layer path to fp32 rounding (other summation order); exact fp32 products
// everywhere (v_mfma_f32_16x16x4_f32 is an fma chain, the bf16 products are exact).
#include "igemm.hpp"
#include "igemm_dev.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace dx {
namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

constexpr int kIn = 84, kFrameB = kIn * kIn * 4, kRowB = kIn * 4;  // uint8 NHWC frame, 4 stacked channels
constexpr int kP0 = 400, kP1 = 81, kP2 = 49;                       // output pixels of the three layers
constexpr int kY0P = 36, kY1P = 72;                                // floats per y0 / y1 pixel in LDS (32 / 64 + pad)
constexpr int kWRowB = 528, kWPlaneB = 32 * kWRowB;                // conv0 weight planes in LDS: 256 bf16 + 16 B pad
constexpr int oFrame = 0, oY0 = oFrame + kFrameB, oY1 = oY0 + kP0 * kY0P * 4, oW0 = oY1 + kP1 * kY1P * 4,
              kLdsBytes = oW0 + 3 * kWPlaneB;
constexpr int kRedFloats = 4 * 21 * 64;  // the K halves meet here (aliases the frame, dead after conv0)
static_assert(oY0 % 16 == 0 && oY1 % 16 == 0 && oW0 % 16 == 0 && kLdsBytes <= 160 * 1024, "LDS layout");
static_assert(kRedFloats * 4 <= kFrameB, "reduction scratch fits the frame's bytes");

struct ConvStackArgs {
  const uint8_t *obs;    // (B, 84, 84, 4) uint8
  const uint16_t *Wb0;   // conv0 weights, three bf16 planes [3][32][256] (k = (kh, kw, c))
  const float *bias0;
  const float *W1;       // conv1 packed [64][512] (k = (kh, kw, ic))
  const float *bias1;
  const float *W2;       // conv2 packed [64][576]
  const float *bias2;
  float *y2;             // (B, 7, 7, 64) NHWC
  int B;
  unsigned long long *stamps;  // DX_DIAG only (DX_CS_DIAG=1): [B][8] shader-clock stamps of wave 0, else NULL
};

// two bytes -> two bf16 (exact: the fp32 of an integer < 256 has a zero low half)
__device__ __forceinline__ uint32_t cs_bytes_to_bf16x2(float f0, float f1) {
  return __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, f1), __builtin_bit_cast(uint32_t, f0), 0x07060302u);
}
__device__ __forceinline__ bf16x8 cs_expand8(uint2 w) {
  u32x4 r;
  r.x = cs_bytes_to_bf16x2(static_cast<float>(w.x & 0xff), static_cast<float>((w.x >> 8) & 0xff));
  r.y = cs_bytes_to_bf16x2(static_cast<float>((w.x >> 16) & 0xff), static_cast<float>(w.x >> 24));
  r.z = cs_bytes_to_bf16x2(static_cast<float>(w.y & 0xff), static_cast<float>((w.y >> 8) & 0xff));
  r.w = cs_bytes_to_bf16x2(static_cast<float>((w.y >> 16) & 0xff), static_cast<float>(w.y >> 24));
  return __builtin_bit_cast(bf16x8, r);
}
// x / 255 to within the last bit (conv0_b16.hip: div255)
__device__ __forceinline__ float cs_div255(float x) {
  const float r = 1.0f / 255.0f;
  const float q = x * r;
  return __builtin_fmaf(__builtin_fmaf(-q, 255.0f, x), r, q);
}

// conv0 for NT_ 32-pixel tiles (tile0, tile0 + 8) of this wave: one pass over the 16 K chunks, every
// weight fragment read from LDS once per chunk for both tiles
template <int NT_>
__device__ __forceinline__ void conv0_tiles(const uint8_t *smem, int tile0, int lane, float bias) {
  const int r = lane & 31, kg = lane >> 5;
  int pb[NT_];
  f32x16 acc[NT_];
#pragma unroll
  for (int t = 0; t < NT_; ++t) {
    const int p = min(32 * (tile0 + 8 * t) + r, kP0 - 1);  // rows past the image compute a copy that is not stored
    const int oy = p / 20, ox = p - 20 * oy;
    pb[t] = (4 * oy * kIn + 4 * ox) * 4 + 8 * kg;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  }
  const int wb = oW0 + r * kWRowB + 16 * kg;
#pragma unroll
  for (int c = 0; c < 16; ++c) {  // chunk c: kernel row c / 2, bytes 16 (c % 2) .. + 15 of its 32
    const int aoff = oFrame + (c >> 1) * kRowB + 16 * (c & 1);
    bf16x8 af[NT_];
#pragma unroll
    for (int t = 0; t < NT_; ++t) af[t] = cs_expand8(*reinterpret_cast<const uint2 *>(smem + pb[t] + aoff));
#pragma unroll
    for (int pl = 2; pl >= 0; --pl) {
      const bf16x8 bf = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(smem + wb + pl * kWPlaneB + 32 * c));
#pragma unroll
      for (int t = 0; t < NT_; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[t], bf, acc[t], 0, 0, 0);
    }
  }
  float *y0 = reinterpret_cast<float *>(const_cast<uint8_t *>(smem) + oY0);
#pragma unroll
  for (int t = 0; t < NT_; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) {  // C/D layout: column = lane & 31, row = (i & 3) + 8 (i >> 2) + 4 (lane >> 5)
      const int p = 32 * (tile0 + 8 * t) + (i & 3) + 8 * (i >> 2) + 4 * kg;
      const float v = cs_div255(acc[t][i]) + bias;
      if (p < kP0) y0[p * kY0P + r] = v > 0.f ? v : 0.f;
    }
}

// K half KH of conv1 for this wave's 16 output channels: 5 pixel tiles on the matrix pipe, pixel 80
// on the vector ALUs.  pb = byte address of the lane's pixel (input pixel (2 oy, 2 ox)) + its k group.
template <int KH>
__device__ __forceinline__ void conv1_half(const uint8_t *smem, const int (&pb)[5], int pbx, const f32x4 (&b)[16],
                                           f32x4 (&acc)[5], float &accx) {
#pragma unroll
  for (int s = 0; s < 16; ++s) {  // K step s: tap (kh, kw) = (2 KH + s / 8, (s / 2) % 4), input channels 16 (s % 2) ..
    const int off = oY0 + (((2 * KH + (s >> 3)) * 20 + ((s >> 1) & 3)) * kY0P + (s & 1) * 16) * 4;
    f32x4 av[5];
#pragma unroll
    for (int mt = 0; mt < 5; ++mt) av[mt] = *reinterpret_cast<const f32x4 *>(smem + pb[mt] + off);
    const f32x4 ax = *reinterpret_cast<const f32x4 *>(smem + pbx + off);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int mt = 0; mt < 5; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt][j], b[s][j], acc[mt], 0, 0, 0);
    accx = __builtin_fmaf(ax[0], b[s][0], accx);
    accx = __builtin_fmaf(ax[1], b[s][1], accx);
    accx = __builtin_fmaf(ax[2], b[s][2], accx);
    accx = __builtin_fmaf(ax[3], b[s][3], accx);
  }
}

// K half KH of conv2 (K steps 18 KH .. 18 KH + 17 of 36: tap = step / 4, input channels 16 (step % 4) ..)
template <int KH>
__device__ __forceinline__ void conv2_half(const uint8_t *smem, const int (&pb)[3], int pbx, const f32x4 (&b)[18],
                                           f32x4 (&acc)[3], float &accx) {
#pragma unroll
  for (int s = 0; s < 18; ++s) {
    const int g = 18 * KH + s, tap = g >> 2;
    const int off = oY1 + (((tap / 3) * 9 + tap % 3) * kY1P + (g & 3) * 16) * 4;
    f32x4 av[3];
#pragma unroll
    for (int mt = 0; mt < 3; ++mt) av[mt] = *reinterpret_cast<const f32x4 *>(smem + pb[mt] + off);
    const f32x4 ax = *reinterpret_cast<const f32x4 *>(smem + pbx + off);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int mt = 0; mt < 3; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt][j], b[s][j], acc[mt], 0, 0, 0);
    accx = __builtin_fmaf(ax[0], b[s][0], accx);
    accx = __builtin_fmaf(ax[1], b[s][1], accx);
    accx = __builtin_fmaf(ax[2], b[s][2], accx);
    accx = __builtin_fmaf(ax[3], b[s][3], accx);
  }
}

__global__ __launch_bounds__(512) void convstack_image_kernel(const ConvStackArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nt = wave & 3, kh2 = wave >> 2;  // conv1 / conv2: output-channel tile and K half of this wave
  const int n16 = lane & 15, kq = lane >> 4;
  const int oc = 16 * nt + n16;
  const uint8_t *src = a.obs + static_cast<long long>(blockIdx.x) * kFrameB;
  unsigned long long tk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define DX_CS_MARK(i) if (kDiag && a.stamps) tk[i] = __builtin_amdgcn_s_memtime();
  DX_CS_MARK(0)

  // ---- everything this workgroup reads from memory, issued before anything is waited for ----
  u32x4 fr[4];  // the frame: 1,764 pieces of 16 bytes
#pragma unroll
  for (int u = 0; u < 4; ++u) fr[u] = *reinterpret_cast<const u32x4 *>(src + 16 * min(tid + 512 * u, kFrameB / 16 - 1));
  u32x4 wv[6];  // conv0's planes: 3 x 32 rows x 32 pieces
#pragma unroll
  for (int u = 0; u < 6; ++u) {
    const int i = u * 512 + tid;
    wv[u] = *reinterpret_cast<const u32x4 *>(a.Wb0 + (i >> 10) * 8192 + ((i >> 5) & 31) * 256 + (i & 31) * 8);
  }
  const float bias0 = a.bias0[lane & 31], bias1 = a.bias1[oc], bias2 = a.bias2[oc];
  f32x4 b1[16];  // this wave's B fragments of conv1: W1[oc][256 kh2 + 16 s + 4 kq ..]
  {
    const float *w1 = a.W1 + oc * 512 + kh2 * 256 + 4 * kq;
#pragma unroll
    for (int s = 0; s < 16; ++s) b1[s] = *reinterpret_cast<const f32x4 *>(w1 + 16 * s);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int u = 0; u < 4; ++u)
    if (tid + 512 * u < kFrameB / 16) *reinterpret_cast<u32x4 *>(smem + oFrame + 16 * (tid + 512 * u)) = fr[u];
#pragma unroll
  for (int u = 0; u < 6; ++u) {
    const int i = u * 512 + tid;
    *reinterpret_cast<u32x4 *>(smem + oW0 + (i >> 10) * kWPlaneB + ((i >> 5) & 31) * kWRowB + (i & 31) * 16) = wv[u];
  }
  __syncthreads();
  DX_CS_MARK(1)

  // ---- conv0: 13 tiles of 32 pixels, waves 0-4 take two ----
  if (wave < 5) conv0_tiles<2>(smem, wave, lane, bias0);
  else conv0_tiles<1>(smem, wave, lane, bias0);
  // conv2's B fragments travel while conv1 runs
  f32x4 b2[18];
  {
    const float *w2 = a.W2 + oc * 576 + kh2 * 288 + 4 * kq;
#pragma unroll
    for (int s = 0; s < 18; ++s) b2[s] = *reinterpret_cast<const f32x4 *>(w2 + 16 * s);
  }
  __syncthreads();  // y0 complete; the frame's bytes are free
  DX_CS_MARK(2)

  float *red = reinterpret_cast<float *>(smem + oFrame);
  float *y1 = reinterpret_cast<float *>(smem + oY1);
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  {  // ---- conv1 ----
    int pb[5];
#pragma unroll
    for (int mt = 0; mt < 5; ++mt) {
      const int p = 16 * mt + n16, oy = p / 9, ox = p - 9 * oy;
      pb[mt] = ((2 * oy * 20 + 2 * ox) * kY0P + 4 * kq) * 4;
    }
    const int pbx = ((16 * 20 + 16) * kY0P + 4 * kq) * 4;  // output pixel (8, 8)
    f32x4 acc[5] = {zero4, zero4, zero4, zero4, zero4};
    float accx = 0.f;
    if (kh2 == 0) conv1_half<0>(smem, pb, pbx, b1, acc, accx);
    else conv1_half<1>(smem, pb, pbx, b1, acc, accx);
    DX_CS_MARK(3)
    if (kh2 == 1) {
#pragma unroll
      for (int mt = 0; mt < 5; ++mt)
#pragma unroll
        for (int j = 0; j < 4; ++j) red[(nt * 21 + 4 * mt + j) * 64 + lane] = acc[mt][j];
      red[(nt * 21 + 20) * 64 + lane] = accx;
    }
    __syncthreads();
    if (kh2 == 0) {  // C/D layout: column (output channel) = lane & 15, row (pixel) = 4 (lane >> 4) + j
#pragma unroll
      for (int mt = 0; mt < 5; ++mt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float v = (acc[mt][j] + red[(nt * 21 + 4 * mt + j) * 64 + lane]) + bias1;
          y1[(16 * mt + 4 * kq + j) * kY1P + oc] = v > 0.f ? v : 0.f;
        }
      float x = accx + red[(nt * 21 + 20) * 64 + lane];  // this lane's k groups of pixel 80; the four groups meet below
      x += __shfl_xor(x, 16);
      x += __shfl_xor(x, 32);
      x += bias1;
      if (kq == 0) y1[80 * kY1P + oc] = x > 0.f ? x : 0.f;
    }
    __syncthreads();
    DX_CS_MARK(4)
  }
  {  // ---- conv2 ----
    int pb[3];
#pragma unroll
    for (int mt = 0; mt < 3; ++mt) {
      const int p = 16 * mt + n16, oy = p / 7, ox = p - 7 * oy;
      pb[mt] = ((oy * 9 + ox) * kY1P + 4 * kq) * 4;
    }
    const int pbx = ((6 * 9 + 6) * kY1P + 4 * kq) * 4;  // output pixel (6, 6)
    f32x4 acc[3] = {zero4, zero4, zero4};
    float accx = 0.f;
    if (kh2 == 0) conv2_half<0>(smem, pb, pbx, b2, acc, accx);
    else conv2_half<1>(smem, pb, pbx, b2, acc, accx);
    DX_CS_MARK(5)
    if (kh2 == 1) {
#pragma unroll
      for (int mt = 0; mt < 3; ++mt)
#pragma unroll
        for (int j = 0; j < 4; ++j) red[(nt * 21 + 4 * mt + j) * 64 + lane] = acc[mt][j];
      red[(nt * 21 + 20) * 64 + lane] = accx;
    }
    __syncthreads();
    if (kh2 == 0) {
      float *out = a.y2 + static_cast<long long>(blockIdx.x) * (kP2 * 64);
#pragma unroll
      for (int mt = 0; mt < 3; ++mt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float v = (acc[mt][j] + red[(nt * 21 + 4 * mt + j) * 64 + lane]) + bias2;
          out[(16 * mt + 4 * kq + j) * 64 + oc] = v > 0.f ? v : 0.f;
        }
      float x = accx + red[(nt * 21 + 20) * 64 + lane];
      x += __shfl_xor(x, 16);
      x += __shfl_xor(x, 32);
      x += bias2;
      if (kq == 0) out[48 * 64 + oc] = x > 0.f ? x : 0.f;
    }
  }
  DX_CS_MARK(6)
#undef DX_CS_MARK
  if (kDiag && a.stamps && tid == 0) {
    tk[7] = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < 8; ++i) a.stamps[blockIdx.x * 8 + i] = tk[i];
  }
}

}  // namespace

bool convstack_supported(int in_h, int in_w, int in_c) { return in_h == kIn && in_w == kIn && in_c == 4; }

// obs (B, 84, 84, 4) uint8 -> y2 (B, 7, 7, 64) in ONE launch (rollout only: y0 / y1 are not kept)
int launch_convstack_image(const uint8_t *obs, const uint16_t *Wb0, const float *bias0, const float *W1,
                           const float *bias1, const float *W2, const float *bias2, float *y2, int B,
                           hipStream_t stream) {
  DX_REQUIRE(obs && Wb0 && bias0 && W1 && bias1 && W2 && bias2 && y2 && B >= 1, "convstack: bad arguments");
  DX_REQUIRE(aligned(obs, 16) && aligned(Wb0, 16) && aligned(W1, 16) && aligned(W2, 16),
             "convstack: frames and packed weights must be 16-byte aligned");
  static int configured_device = -1;
  int dev = 0;
  DX_HIP(hipGetDevice(&dev));
  if (configured_device != dev) {
    DX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(convstack_image_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
    configured_device = dev;
  }
  ConvStackArgs a{obs, Wb0, bias0, W1, bias1, W2, bias2, y2, B, nullptr};
#if DX_DIAG
  if (getenv("DX_CS_DIAG")) {  // in-kernel phase cycles (wave 0 of every workgroup), summarised on stderr (synchronous)
    unsigned long long *dev = nullptr;
    DX_HIP(hipMalloc(&dev, static_cast<size_t>(B) * 64));
    a.stamps = dev;
    hipLaunchKernelGGL(convstack_image_kernel, dim3(B), dim3(512), kLdsBytes, stream, a);
    DX_LAUNCH_CHECK();
    DX_HIP(hipStreamSynchronize(stream));
    std::vector<unsigned long long> h(static_cast<size_t>(B) * 8);
    DX_HIP(hipMemcpy(h.data(), dev, h.size() * 8, hipMemcpyDeviceToHost));
    DX_HIP(hipFree(dev));
    double ph[6] = {0, 0, 0, 0, 0, 0};
    for (int b = 0; b < B; ++b)
      for (int i = 0; i < 6; ++i) ph[i] += static_cast<double>(h[b * 8 + i + 1] - h[b * 8 + i]) / B;
    fprintf(stderr, "[convstack B=%d] cycles per workgroup (wave 0): loads + LDS fill %.0f, conv0 %.0f, conv1 loop %.0f, "
            "conv1 reduce + y1 %.0f, conv2 loop %.0f, conv2 reduce + store %.0f, total %.0f\n", B, ph[0], ph[1], ph[2], ph[3],
            ph[4], ph[5], ph[0] + ph[1] + ph[2] + ph[3] + ph[4] + ph[5]);
    return DX_OK;
  }
#endif
  hipLaunchKernelGGL(convstack_image_kernel, dim3(B), dim3(512), kLdsBytes, stream, a);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

}  // namespace dx
