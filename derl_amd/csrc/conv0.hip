// Direct kernels for the first convolution (8x8 stride 4 over uint8 NHWC frames, 4 channels,
// 32 output channels; derl/models.py:103,117-124): forward and weight gradient.
//
// The generic implicit GEMM re-reads every input byte 4x (im2col overlap), converts it once per
// block K-step and round-trips the float tile through LDS with two barriers per 32 k; with
// N = 32 there is little MFMA work to hide that behind (78 / 58 TFLOP/s measured).  Here a
// workgroup stages the RAW uint8 input rows of a 256-pixel output tile (<= 23 KB, two
// contiguous byte ranges) and, for the forward, the whole packed weight matrix (32 x 256 fp32)
// into LDS once, and every lane builds its MFMA operand straight from the bytes:
//   forward : lane (pixel, h) reads the 4 channel bytes of input pixel (4oy+kh, 4ox+2t+h) with one
//             ds_read_b32, dequantises x/255 exactly and feeds 4 MFMAs; B = 4 consecutive k of W.
//   wgrad   : A = dY0^T from an LDS tile [256 pixels][32 oc], B = one input byte per lane
//             (k = kh*32 + kw*4 + c is 32 consecutive bytes of an input row), reduction over the
//             256 pixels of the tile; a workgroup loops over tiles and writes one slab.
// No barrier inside the 256-MFMA body of a tile; workgroups loop over tiles (persistent).
#include "conv0_tile.hpp"

namespace dx {
namespace {

__device__ __forceinline__ float dq(uint32_t x) {  // exact x / 255 (see igemm.hip dequant_u8)
#ifdef DX_CONV0_FASTDQ
  return static_cast<float>(x) * (1.0f / 255.0f);  // timing experiment: 2 VALU instead of 4
#else
  const float r = 1.0f / 255.0f;
  const float xf = static_cast<float>(x);
  const float q = xf * r;
  return __builtin_fmaf(__builtin_fmaf(-q, 255.0f, xf), r, q);
#endif
}

constexpr int kWLd = 260;     // LDS row stride of the packed weights (256 + 4 pad floats)

__global__ __launch_bounds__(256) void conv0_fwd_kernel(const Conv0Args a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  float *Ws = reinterpret_cast<float *>(smem);
  uint8_t *patch = smem + 32 * kWLd * sizeof(float);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rowB = a.in_w * 4;
  for (int i = tid; i < 32 * 64; i += 256) {
    const int row = i >> 6, c4 = i & 63;
    *reinterpret_cast<float4 *>(&Ws[row * kWLd + 4 * c4]) =
        *reinterpret_cast<const float4 *>(a.Wp + row * 256 + 4 * c4);
  }
  const int lrow = lane & 31, h = lane >> 5;
  const float bias = a.bias[lrow];
  u32x4 pre[kPatchRegs];
  if (blockIdx.x < a.ntiles) patch_load(a, tile_segments(a, blockIdx.x * kTile), pre);
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int m0 = tile * kTile;
    const Seg s = tile_segments(a, m0);
    __syncthreads();  // every wave finished reading the previous patch (and Ws is written)
    patch_store(s, pre, patch);
    __syncthreads();
    if (tile + gridDim.x < a.ntiles) patch_load(a, tile_segments(a, (tile + gridDim.x) * kTile), pre);
    int rb[2];
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) rb[t2] = pixel_base(a, s, wave * 64 + t2 * 32 + lrow) + 4 * h;
    f32x16 acc[2];
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t2][r] = 0.f;
    const float *wl = Ws + lrow * kWLd + 4 * h;
    for (int kh = 0; kh < 8; ++kh) {
      const uint8_t *prow = patch + kh * rowB;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        // k = (kh, kw = 2t + h, c = 0..3): 4 consecutive k of the packed weights, one input pixel
        const float4 bf = *reinterpret_cast<const float4 *>(wl + (kh * 8 + 2 * t) * 4);
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
          const uint32_t w = *reinterpret_cast<const uint32_t *>(prow + rb[t2] + 8 * t);
          acc[t2] = __builtin_amdgcn_mfma_f32_32x32x2f32(dq(w & 0xff), bf.x, acc[t2], 0, 0, 0);
          acc[t2] = __builtin_amdgcn_mfma_f32_32x32x2f32(dq((w >> 8) & 0xff), bf.y, acc[t2], 0, 0, 0);
          acc[t2] = __builtin_amdgcn_mfma_f32_32x32x2f32(dq((w >> 16) & 0xff), bf.z, acc[t2], 0, 0, 0);
          acc[t2] = __builtin_amdgcn_mfma_f32_32x32x2f32(dq(w >> 24), bf.w, acc[t2], 0, 0, 0);
        }
      }
    }
    // bias + ReLU, NHWC store: col = lane&31 (oc), row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wave * 64 + t2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m < a.M) {
          const float v = acc[t2][r] + bias;
          a.out[static_cast<long long>(m) * 32 + lrow] = v > 0.f ? v : 0.f;
        }
      }
  }
}

__global__ __launch_bounds__(256) void conv0_wgrad_kernel(const Conv0Args a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  float *Gs = reinterpret_cast<float *>(smem);                 // [256][32]
  int *rbtab = reinterpret_cast<int *>(smem + kTile * 32 * 4);  // [256]
  uint8_t *patch = smem + kTile * 32 * 4 + kTile * 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rowB = a.in_w * 4;
  const int lcol = lane & 31, h = lane >> 5;
  // this wave's two k tiles are kernel rows kh = 2*wave and 2*wave + 1; lane j -> byte j of the row
  const int koff0 = (2 * wave) * rowB + lcol, koff1 = koff0 + rowB;
  f32x16 acc[2];
#pragma unroll
  for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t2][r] = 0.f;
  float bias_acc = 0.f;
  u32x4 pre[kPatchRegs];
  f32x4 gpre[8];
  // dY0 tile: 256 rows x 32 floats, contiguous; rows beyond M are zeroed when stored
  auto g_load = [&](int m0) {
    const f32x4 *g4 = reinterpret_cast<const f32x4 *>(a.G + static_cast<long long>(m0) * 32);
    const int nvalid = min(kTile, a.M - m0) * 8;
#pragma unroll
    for (int u = 0; u < 8; ++u) gpre[u] = g4[min(tid + u * 256, nvalid - 1)];
  };
  if (blockIdx.x < a.ntiles) {
    patch_load(a, tile_segments(a, blockIdx.x * kTile), pre);
    g_load(blockIdx.x * kTile);
  }
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int m0 = tile * kTile;
    const Seg s = tile_segments(a, m0);
    __syncthreads();
    patch_store(s, pre, patch);
    {
      f32x4 *d4 = reinterpret_cast<f32x4 *>(Gs);
      const int nvalid = min(kTile, a.M - m0) * 8;
#pragma unroll
      for (int u = 0; u < 8; ++u)
        d4[tid + u * 256] = (tid + u * 256) < nvalid ? gpre[u] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    rbtab[tid] = pixel_base(a, s, tid);
    __syncthreads();
    if (tile + gridDim.x < a.ntiles) {
      patch_load(a, tile_segments(a, (tile + gridDim.x) * kTile), pre);
      g_load((tile + gridDim.x) * kTile);
    }
    if (lane < 32) {  // bias gradient: each wave sums its quarter of the rows
#pragma unroll 8
      for (int r = 0; r < 64; ++r) bias_acc += Gs[(wave * 64 + r) * 32 + lane];
    }
#pragma unroll 4
    for (int sidx = 0; sidx < kTile / 2; ++sidx) {
      const int m = 2 * sidx + h;
      const float g = Gs[m * 32 + lcol];
      const int rb = rbtab[m];
      const float b0 = dq(patch[rb + koff0]);
      const float b1 = dq(patch[rb + koff1]);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(g, b0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(g, b1, acc[1], 0, 0, 0);
    }
  }
  // slab[block][oc][k]: rows = oc, cols = k
  float *slab = a.slab + static_cast<long long>(blockIdx.x) * 32 * 256;
#pragma unroll
  for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int oc = (r & 3) + 8 * (r >> 2) + 4 * h;
      slab[oc * 256 + (2 * wave + t2) * 32 + lcol] = acc[t2][r];
    }
  if (a.bias_slab) {
    __syncthreads();
    float *red = Gs;  // [4][32]
    if (lane < 32) red[wave * 32 + lane] = bias_acc;
    __syncthreads();
    if (tid < 32)
      a.bias_slab[static_cast<long long>(blockIdx.x) * 32 + tid] = red[tid] + red[32 + tid] + red[64 + tid] + red[96 + tid];
  }
}

}  // namespace

bool conv0_direct_supported(int in_h, int in_w, int in_c, int h0, int w0) {
  const int patch = (4 * (kTile / (w0 > 0 ? w0 : 1) + 3) + 8) * in_w * 4;
  return in_c == 4 && in_w % 4 == 0 && h0 * w0 >= kTile && w0 >= 4 && in_h >= 8 &&
         patch <= kPatchRegs * 256 * 16;
}

int launch_conv0_fwd(const Conv0Args &a, hipStream_t stream) {
  DX_REQUIRE(a.obs && a.Wp && a.bias && a.out && a.M > 0, "conv0_fwd: bad arguments");
  const int lds = 32 * kWLd * 4 + patch_bytes(a);
  DX_REQUIRE(lds <= 160 * 1024 && patch_bytes(a) <= kPatchRegs * 256 * 16,
             "conv0_fwd: tile does not fit (%d LDS bytes, patch %d)", lds, patch_bytes(a));
  DX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(conv0_fwd_kernel),
                             hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  int grid = a.ntiles < 512 ? a.ntiles : 512;
  hipLaunchKernelGGL(conv0_fwd_kernel, dim3(grid), dim3(256), lds, stream, a);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

int launch_conv0_wgrad(const Conv0Args &a, int nblocks, hipStream_t stream) {
  DX_REQUIRE(a.obs && a.G && a.slab && a.M > 0 && nblocks >= 1 && nblocks <= a.ntiles,
             "conv0_wgrad: bad arguments");
  const int lds = kTile * 32 * 4 + kTile * 4 + patch_bytes(a);
  DX_REQUIRE(lds <= 160 * 1024 && patch_bytes(a) <= kPatchRegs * 256 * 16,
             "conv0_wgrad: tile does not fit (%d LDS bytes, patch %d)", lds, patch_bytes(a));
  DX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(conv0_wgrad_kernel),
                             hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipLaunchKernelGGL(conv0_wgrad_kernel, dim3(nblocks), dim3(256), lds, stream, a);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

}  // namespace dx
