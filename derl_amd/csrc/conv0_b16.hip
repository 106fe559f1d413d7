// First convolution (8x8 stride 4, uint8 NHWC frames -> 32 channels; derl/models.py:103,117-124)
// on the bf16 matrix cores WITHOUT giving up fp32 accuracy:
//   * a uint8 pixel is an integer < 256: exact in bf16 (8 significand bits);
//   * an fp32 operand w splits exactly into three bf16 terms w = hi + mid + lo (igemm_b3.hip);
//   * every bf16 x bf16 product is exact in fp32 and v_mfma_f32_32x32x16_bf16 accumulates in fp32,
// so  sum_k x_k * w_k  costs THREE bf16 MFMAs of 32 cycles per 16 k instead of eight fp32 MFMAs of
// 64 cycles (5.3x less matrix time), with only the fp32 accumulation roundings any fp32 chain has.
// The 1/255 of the reference's `observations / 255` (models.py:121-123) is one division (div255) of the
// finished sum (forward) or of the slab element (weight gradient) instead of one rounding per term.
//
//   forward : A = 8 channel-bytes of two adjacent input pixels per lane (one ds_read_b64 from the
//             staged uint8 patch, converted in registers), B = three bf16 planes of the packed
//             weights in LDS (528-byte rows: conflict-free ds_read_b128).
//   wgrad   : dW[oc][k] = sum_m dY[m][oc] * x[m][k]; A = dY^T as three bf16 planes [oc][m] built once
//             per 256-pixel tile, B = the bytes of 8 pixels at this lane's k (8 ds_read_u8).
// Tiling, patch staging and the persistent tile loop are those of conv0.hip.
#include <type_traits>

#include "conv0_tile.hpp"

namespace dx {
namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;

__device__ __forceinline__ uint32_t pack2(float a, float b) {
  bf16x2 v = {static_cast<__bf16>(a), static_cast<__bf16>(b)};
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float lo_f32(uint32_t p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float hi_f32(uint32_t p) { return __builtin_bit_cast(float, p & 0xffff0000u); }

// exact split of two floats into packed (hi, mid, lo) bf16 pairs
__device__ __forceinline__ void split2(float x0, float x1, uint32_t &h, uint32_t &m, uint32_t &l) {
  h = pack2(x0, x1);
  const float r0 = x0 - lo_f32(h), r1 = x1 - hi_f32(h);
  m = pack2(r0, r1);
  l = pack2(r0 - lo_f32(m), r1 - hi_f32(m));
}

// two bytes -> two bf16 (exact: the fp32 of an integer < 256 has a zero low half)
__device__ __forceinline__ uint32_t bytes_to_bf16x2(float f0, float f1) {
  return __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, f1), __builtin_bit_cast(uint32_t, f0), 0x07060302u);
}

__device__ __forceinline__ bf16x8 expand8(uint2 w) {
  u32x4 r;
  r.x = bytes_to_bf16x2(static_cast<float>(w.x & 0xff), static_cast<float>((w.x >> 8) & 0xff));
  r.y = bytes_to_bf16x2(static_cast<float>((w.x >> 16) & 0xff), static_cast<float>(w.x >> 24));
  r.z = bytes_to_bf16x2(static_cast<float>(w.y & 0xff), static_cast<float>((w.y >> 8) & 0xff));
  r.w = bytes_to_bf16x2(static_cast<float>((w.y >> 16) & 0xff), static_cast<float>(w.y >> 24));
  return __builtin_bit_cast(bf16x8, r);
}

// x / 255 to within the last bit: multiply by the rounded reciprocal, one Newton residual step
// (3 instructions instead of the ~10 of an IEEE division; 32 of them per lane and tile)
__device__ __forceinline__ float div255(float x) {
  const float r = 1.0f / 255.0f;
  const float q = x * r;
  return __builtin_fmaf(__builtin_fmaf(-q, 255.0f, x), r, q);
}

constexpr int kWRowB = 528;                 // bytes per LDS weight row: 256 bf16 + 16 pad
constexpr int kWPlaneB = 32 * kWRowB;       // one plane of the packed weights

__global__ __launch_bounds__(256) void conv0_fwd_b16_kernel(const Conv0Args a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint8_t *Wpl = smem;                      // planes hi, mid, lo
  uint8_t *patch = smem + 3 * kWPlaneB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rowB = a.in_w * 4;
  for (int i = tid; i < 32 * 64; i += 256) {  // packed weights [32][256] fp32 -> three bf16 planes
    const int row = i >> 6, c4 = i & 63;
    const float4 w = *reinterpret_cast<const float4 *>(a.Wp + row * 256 + 4 * c4);
    uint2 h, m, l;
    split2(w.x, w.y, h.x, m.x, l.x);
    split2(w.z, w.w, h.y, m.y, l.y);
    const int o = row * kWRowB + c4 * 8;
    *reinterpret_cast<uint2 *>(Wpl + o) = h;
    *reinterpret_cast<uint2 *>(Wpl + kWPlaneB + o) = m;
    *reinterpret_cast<uint2 *>(Wpl + 2 * kWPlaneB + o) = l;
  }
  const int lrow = lane & 31, h = lane >> 5;
  const float bias = a.bias[lrow];
  u32x4 pre[kPatchRegs];
  Seg nxt = tile_segments(a, min(static_cast<int>(blockIdx.x), a.ntiles - 1) * kTile);  // the tile whose patch is in `pre`
  if (blockIdx.x < a.ntiles) patch_load(a, nxt, pre);
  // Whole tiles in the loop, the (at most one) ragged last tile after it: with the guarded stores of
  // a ragged tile inside the loop the number of stores per pass is not static, hipcc then waits
  // vmcnt(0) for the prefetched patch at the top of the next pass -- i.e. for every store of this
  // one.  (Worth 2-3 % here; neither this, a two-tile-deep prefetch, software-pipelined fragment reads
  // nor dropping the byte -> bf16 conversions moves the kernel off 175-180 us for 0.68 GB of traffic
  // and 70 us of MFMA work.)
  auto one_tile = [&](int tile, auto ragged) {
    constexpr bool RAGGED = decltype(ragged)::value;
    const int m0 = tile * kTile;
    const Seg s = nxt;
    __syncthreads();  // every wave finished reading the previous patch (and the planes are written)
    patch_store(s, pre, patch);
    __syncthreads();
    if (tile + gridDim.x < a.ntiles) {
      nxt = tile_segments(a, (tile + gridDim.x) * kTile);
      patch_load(a, nxt, pre);
    }
    int rb[2];
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) rb[t2] = pixel_base(a, s, wave * 64 + t2 * 32 + lrow) + 8 * h;
    f32x16 acc[2];
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t2][r] = 0.f;
    const uint8_t *wl = Wpl + lrow * kWRowB + 16 * h;
#pragma unroll 2
    for (int kh = 0; kh < 8; ++kh) {
      const uint8_t *prow = patch + kh * rowB;
#pragma unroll
      for (int c16 = 0; c16 < 2; ++c16) {
        // 16 k = (kh, kw = 4*c16 .. 4*c16+3, c): lane half h takes kw = 4*c16 + 2h, +1 (8 bytes)
        const int wo = (kh * 2 + c16) * 32;
        const bf16x8 bh = *reinterpret_cast<const bf16x8 *>(wl + wo);
        const bf16x8 bm = *reinterpret_cast<const bf16x8 *>(wl + kWPlaneB + wo);
        const bf16x8 bl = *reinterpret_cast<const bf16x8 *>(wl + 2 * kWPlaneB + wo);
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
          const bf16x8 af = expand8(*reinterpret_cast<const uint2 *>(prow + rb[t2] + 16 * c16));
          acc[t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bl, acc[t2], 0, 0, 0);
          acc[t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bm, acc[t2], 0, 0, 0);
          acc[t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bh, acc[t2], 0, 0, 0);
        }
      }
    }
    // /255, bias, ReLU, NHWC store: col = lane&31 (oc), row = (r&3) + 8*(r>>2) + 4*(lane>>5).
    // Full tiles (all but the last) store through one per-lane base pointer with immediate
    // offsets; a guarded store per element compiles to 32 exec-mask branches.
    float *obase = a.out + (static_cast<long long>(m0) + wave * 64 + 4 * h) * 32 + lrow;
    if (!RAGGED) {
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = div255(acc[t2][r]) + bias;
          obase[(t2 * 32 + (r & 3) + 8 * (r >> 2)) * 32] = v > 0.f ? v : 0.f;
        }
    } else {
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wave * 64 + t2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (m < a.M) {
            const float v = div255(acc[t2][r]) + bias;
            obase[(t2 * 32 + (r & 3) + 8 * (r >> 2)) * 32] = v > 0.f ? v : 0.f;
          }
        }
    }
  };
  const int nfull = a.M / kTile;
  int tile = blockIdx.x;
  for (; tile < nfull; tile += gridDim.x) one_tile(tile, std::false_type{});
  if (tile < a.ntiles) one_tile(tile, std::true_type{});
}

constexpr int kGRowB = 528;            // bytes per row of a dY^T plane: 256 bf16 (m) + 16 pad
constexpr int kGPlaneB = 32 * kGRowB;

__global__ __launch_bounds__(256) void conv0_wgrad_b16_kernel(const Conv0Args a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint8_t *Gpl = smem;                                             // dY^T planes hi, mid, lo: [oc][m]
  int *rbtab = reinterpret_cast<int *>(smem + 3 * kGPlaneB);       // [256]
  float *red = reinterpret_cast<float *>(smem + 3 * kGPlaneB + kTile * 4);  // [8][32] bias partials
  uint8_t *patch = smem + 3 * kGPlaneB + kTile * 4 + 8 * 32 * 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rowB = a.in_w * 4;
  const int lcol = lane & 31, h = lane >> 5;
  // Wave roles: kgrp = wave & 1 owns kernel rows 4*kgrp .. 4*kgrp+3 (128 k), mgrp = wave >> 1 owns
  // pixels 128*mgrp .. +127 of every tile.  A lane reads one DWORD of an input row per pixel --
  // lane j: row 4*kgrp + j/8, bytes 4*(j%8) .. +3 -- and byte t of it is this lane's operand of k
  // tile t (k = 32*row + 4*(j%8) + t): one LDS read feeds four k tiles x three planes = 12 MFMAs
  // (a byte per read fed 3), which is what this kernel was bound by.
  const int kgrp = wave & 1, mgrp = wave >> 1;
  const int koff = (4 * kgrp + (lcol >> 3)) * rowB + 4 * (lcol & 7);
  // staging role: thread -> (oc = tid & 31, 32 pixels mg*32 .. mg*32+31)
  const int soc = tid & 31, mg = tid >> 5;
  f32x16 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bias_acc = 0.f;
  u32x4 pre[kPatchRegs];
  float gpre[32];
  auto g_load = [&](int m0) {  // dY0[m][oc]: 32 lanes = 128 contiguous bytes per pixel
    const int nvalid = min(kTile, a.M - m0);
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      const int m = mg * 32 + u;
      gpre[u] = a.G[static_cast<long long>(m0 + min(m, nvalid - 1)) * 32 + soc];
    }
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
      const int nvalid = min(kTile, a.M - m0);
#pragma unroll
      for (int q = 0; q < 4; ++q) {  // 8 pixels -> one 16-byte store per plane
        u32x4 ph, pm, pl;
        uint32_t hh[4], mm[4], ll[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int m = mg * 32 + q * 8 + 2 * e;
          const float g0 = m < nvalid ? gpre[q * 8 + 2 * e] : 0.f;
          const float g1 = m + 1 < nvalid ? gpre[q * 8 + 2 * e + 1] : 0.f;
          bias_acc += g0 + g1;
          split2(g0, g1, hh[e], mm[e], ll[e]);
        }
        ph.x = hh[0]; ph.y = hh[1]; ph.z = hh[2]; ph.w = hh[3];
        pm.x = mm[0]; pm.y = mm[1]; pm.z = mm[2]; pm.w = mm[3];
        pl.x = ll[0]; pl.y = ll[1]; pl.z = ll[2]; pl.w = ll[3];
        const int o = soc * kGRowB + (mg * 32 + q * 8) * 2;
        *reinterpret_cast<u32x4 *>(Gpl + o) = ph;
        *reinterpret_cast<u32x4 *>(Gpl + kGPlaneB + o) = pm;
        *reinterpret_cast<u32x4 *>(Gpl + 2 * kGPlaneB + o) = pl;
      }
    }
    rbtab[tid] = pixel_base(a, s, tid);
    __syncthreads();
    if (tile + gridDim.x < a.ntiles) {
      patch_load(a, tile_segments(a, (tile + gridDim.x) * kTile), pre);
      g_load((tile + gridDim.x) * kTile);
    }
    const uint8_t *gl = Gpl + lcol * kGRowB + 16 * h;
#pragma unroll 2
    for (int cc = 0; cc < kTile / 32; ++cc) {  // 16 pixels per MFMA: lane half h takes pixels 16c + 8h .. +7
      const int c = mgrp * (kTile / 32) + cc;
      const bf16x8 ah = *reinterpret_cast<const bf16x8 *>(gl + c * 32);
      const bf16x8 am = *reinterpret_cast<const bf16x8 *>(gl + kGPlaneB + c * 32);
      const bf16x8 al = *reinterpret_cast<const bf16x8 *>(gl + 2 * kGPlaneB + c * 32);
      uint32_t w[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) w[e] = *reinterpret_cast<const uint32_t *>(patch + rbtab[16 * c + 8 * h + e] + koff);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = static_cast<float>((w[e] >> (8 * t)) & 0xff);
        u32x4 b;
        b.x = bytes_to_bf16x2(x[0], x[1]); b.y = bytes_to_bf16x2(x[2], x[3]);
        b.z = bytes_to_bf16x2(x[4], x[5]); b.w = bytes_to_bf16x2(x[6], x[7]);
        const bf16x8 bf = __builtin_bit_cast(bf16x8, b);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bf, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bf, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bf, acc[t], 0, 0, 0);
      }
    }
  }
  // The two pixel halves meet in LDS (the dY planes are dead now), then
  // slab[block][oc][k]: rows = oc, cols = k; the input scale 1/255 is applied here
  __syncthreads();
  float *part = reinterpret_cast<float *>(smem);  // [kgrp][t][r][lane]
  if (mgrp == 1) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) part[((kgrp * 4 + t) * 16 + r) * 64 + lane] = acc[t][r];
  }
  __syncthreads();
  if (mgrp == 0) {
    float *slab = a.slab + static_cast<long long>(blockIdx.x) * 32 * 256;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int oc = (r & 3) + 8 * (r >> 2) + 4 * h;
        const int k = (4 * kgrp + (lcol >> 3)) * 32 + 4 * (lcol & 7) + t;
        slab[oc * 256 + k] = div255(acc[t][r] + part[((kgrp * 4 + t) * 16 + r) * 64 + lane]);
      }
  }
  if (a.bias_slab) {
    __syncthreads();
    red[mg * 32 + soc] = bias_acc;
    __syncthreads();
    if (tid < 32) {
      float v = 0.f;
#pragma unroll
      for (int g = 0; g < 8; ++g) v += red[g * 32 + tid];
      a.bias_slab[static_cast<long long>(blockIdx.x) * 32 + tid] = v;
    }
  }
}

// Rollout batches (a few thousand output pixels): one 32-pixel x 32-channel tile per workgroup,
// the four waves split the 16 K chunks (kernel rows 2w, 2w+1), every operand goes straight from
// global memory into the MFMA layout -- 8 input bytes and 3 x 16 bytes of the pre-split weight
// planes (dx_cnn_pack) per lane and chunk, all 16 loads in flight before the first of 12 MFMAs --
// and the four partial tiles meet in LDS once (igemm_lat.hip has the fp32 layers of this path).
__global__ __launch_bounds__(256) void conv0_lat_b16_kernel(const Conv0Args a, const uint16_t *Wb) {
  __shared__ float red[4][16][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int rowB = a.in_w * 4;
  const int P = a.h0 * a.w0;
  const int m = min(static_cast<int>(blockIdx.x) * 32 + r, a.M - 1);  // clamped rows are not stored
  int img = m / P;
  const int pix = m - img * P;
  const int oy = pix / a.w0, ox = pix - oy * a.w0;
  if (a.idx) img = a.idx[img];
  const uint8_t *src = a.obs + static_cast<long long>(img) * a.in_h * rowB + (4 * oy + 2 * wave) * rowB + ox * 16 + 8 * h;
  const uint16_t *wsrc = Wb + r * 256 + (2 * wave) * 32 + 8 * h;
  uint2 araw[4];
  u32x4 braw[3][4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {  // chunk c: kernel row 2*wave + (c >> 1), bytes 16*(c & 1) .. +15
    araw[c] = *reinterpret_cast<const uint2 *>(src + (c >> 1) * rowB + 16 * (c & 1));
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
      braw[pl][c] = *reinterpret_cast<const u32x4 *>(wsrc + pl * 8192 + (c >> 1) * 32 + 16 * (c & 1));
  }
  __builtin_amdgcn_sched_barrier(0);  // all loads before the first MFMA (see igemm_lat.hip)
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const bf16x8 af = expand8(araw[c]);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8, braw[2][c]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8, braw[1][c]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8, braw[0][c]), acc, 0, 0, 0);
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) red[wave][i][lane] = acc[i];
  __syncthreads();
  const float bias = a.bias[r];
#pragma unroll
  for (int t = 0; t < 4; ++t) {  // wave w finishes accumulator registers 4w .. 4w+3
    const int i = 4 * wave + t;
    const float sum = ((red[0][i][lane] + red[1][i][lane]) + red[2][i][lane]) + red[3][i][lane];
    const int mm = blockIdx.x * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
    if (mm < a.M) {
      const float v = div255(sum) + bias;
      a.out[static_cast<long long>(mm) * 32 + r] = v > 0.f ? v : 0.f;
    }
  }
}

}  // namespace

int launch_conv0_fwd_b16(const Conv0Args &a, hipStream_t stream) {
  DX_REQUIRE(a.obs && a.Wp && a.bias && a.out && a.M > 0, "conv0_fwd_b16: bad arguments");
  const int lds = 3 * kWPlaneB + patch_bytes(a);
  DX_REQUIRE(lds <= 160 * 1024 && patch_bytes(a) <= kPatchRegs * 256 * 16,
             "conv0_fwd_b16: tile does not fit (%d LDS bytes, patch %d)", lds, patch_bytes(a));
  DX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(conv0_fwd_b16_kernel),
                             hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  int grid = a.ntiles < 512 ? a.ntiles : 512;
  hipLaunchKernelGGL(conv0_fwd_b16_kernel, dim3(grid), dim3(256), lds, stream, a);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

// DX_ENOSUP when the shape is not covered (odd row pitch) or the batch is too big for this path
int launch_conv0_lat_b16(const Conv0Args &a, const uint16_t *Wb, hipStream_t stream) {
  DX_REQUIRE(a.obs && Wb && a.bias && a.out && a.M > 0, "conv0_lat_b16: bad arguments");
  if (a.in_w % 2 || (static_cast<long long>(a.in_h) * a.in_w) % 2 || a.w0 < 1) return DX_ENOSUP;
  hipLaunchKernelGGL(conv0_lat_b16_kernel, dim3(cdiv(a.M, 32)), dim3(256), 0, stream, a, Wb);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

int launch_conv0_wgrad_b16(const Conv0Args &a, int nblocks, hipStream_t stream) {
  DX_REQUIRE(a.obs && a.G && a.slab && a.M > 0 && nblocks >= 1 && nblocks <= a.ntiles,
             "conv0_wgrad_b16: bad arguments");
  const int lds = 3 * kGPlaneB + kTile * 4 + 8 * 32 * 4 + patch_bytes(a);
  DX_REQUIRE(lds <= 160 * 1024 && patch_bytes(a) <= kPatchRegs * 256 * 16,
             "conv0_wgrad_b16: tile does not fit (%d LDS bytes, patch %d)", lds, patch_bytes(a));
  DX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(conv0_wgrad_b16_kernel),
                             hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipLaunchKernelGGL(conv0_wgrad_b16_kernel, dim3(nblocks), dim3(256), lds, stream, a);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

}  // namespace dx
