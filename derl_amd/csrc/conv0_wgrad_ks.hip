// Weight gradient of the first convolution (8x8 stride 4 on uint8 84 x 84 x 4 frames; the autograd backward of
// derl/models.py:103,117-124 triggered by derl/alg/common.py:70), round 6: the contraction over output pixels SPLIT
// ACROSS THE WAVES, a wave holding all 32 channels x half the taps of the result.
//
//   dW[oc][kh][kw][ci] = 1/255 * sum over (img, oy, ox) of dY0[img][oy][ox][oc] * frame[img][4 oy + kh][4 ox + kw][ci]
//
// A GEMM with M = 32 output channels, N = 256 taps and K = 400 pixels per image.  With M that small, any split of N (or
// M) over the waves of a workgroup makes every wave read the whole gradient operand again: conv0_b16.hip's tile kernel
// (two k halves x two pixel halves, 32x32x16, bytes gathered with ds_read_u8-style reads and converted PER USE: 48
// conversions per 12 MFMAs) runs with the matrix pipe 0.42 busy and 24 % of its LDS cycles in bank conflicts; round 4's
// transposed-read kernel with four accumulator tiles per wave read a K step's operands 3x over and was slower still.
// Here a wave owns 2 x 8 accumulator tiles (all 32 channels x HALF the taps: 64 registers; the whole 32 x 256 result in
// 128 registers was built first and spilled) and takes whole K STEPS of 32 pixels: per step 12 transposed reads of the
// gradient fragments (2 channel tiles x 3 planes) and 16 of the frame's (8 tap tiles) feed 48 v_mfma_f32_16x16x32_bf16
// -- the frame leaves LDS ONCE per image, the gradient planes twice, 0.58 reads per MFMA.
//   * the frame is converted to bf16 ONCE per image, when it is copied into LDS (a byte is exact in bf16): LDS image
//     [84 rows][84 pixels][4 channels] bf16 in the frame's own order, so the 16 taps (kw, ci) of half a kernel row are 32
//     contiguous bytes at pixel (4 oy + kh, 4 ox) and ds_read_b64_tr_b16 -- every lane supplies the address of ITS pixel
//     -- hands each lane 4 consecutive K slots of its tap: no im2col, no gather table.  The natural 672-byte row pitch is
//     conflict-free for the 8 consecutive pixels a 32-lane half reads (32-byte pixel stride; an 8-pixel group that
//     crosses an output row lands 4 x 672 = 128 (mod 256) bytes on).
//   * dY0 (fp32) is split EXACTLY into three bf16 planes [pixel][32 channels] (64-byte rows, the two 32-byte channel
//     halves swapped on rows with bit 2 set: conflict-free transposed reads without padding); byte x bf16 products are
//     exact in fp32, the planes are accumulated smallest first, fp32 accumulation as in every fp32 chain.
//   * 400 pixels = 12.5 K steps.  Wave w = (tap half w >> 2, K group w & 3) takes the steps 3 (w & 3) .. + 2 and a QUARTER
//     of the half-valid step 12 (two of its eight tap tiles, the step's upper K half known to be zero): every wave
//     multiplies 3.25 half-steps per image, 13 steps per image where 12.5 are useful (executed / useful 1.04).
//   * accumulators stay in registers over all images of the workgroup (one workgroup per CU, images blockIdx.x, + grid,
//     ...); the next image's rows travel in registers while this one multiplies; at the end the four K groups' partial
//     results meet in LDS in a fixed order (deterministic), are scaled by 1/255 and leave as ONE slab per workgroup
//     (<= 256 slabs where the tile kernel wrote 512).
//   * measured and NOT kept (git history: "conv0_wgrad_ks v3"): half-image units in two LDS buffers with the two waves of a
//     SIMD in opposite phases (one copies the next unit while the other multiplies): 193-199 us against this version's
//     173-177 on the same boxes, the same with every wave in the same order -- the launch is not bound by the order of its
//     phases but by what it spends: with nothing but the multiplication running it takes 121 us (1.35 PFLOP/s executed,
//     what the other bf16 stages reach under the package power limit), with nothing but its loads 120 us (5.6 TB/s).
// LDS: 56,448 (frame) + 3 x 25,664 (gradient planes incl. one zero row) = 133,440 bytes + the workgroup's gather entries.
#include "bf16_split.hpp"
#include "igemm_dev.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace dx {
namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using s16x4 = __attribute__((ext_vector_type(4))) short;
using lds_s16x4 = __attribute__((address_space(3))) s16x4;

constexpr int kFrameB = 84 * 84 * 4;           // bytes of a uint8 frame
constexpr int kXRow = 84 * 4 * 2;              // bytes of a bf16 LDS row
constexpr int kXB = 84 * kXRow;
constexpr int kPix = 400, kOW = 20, kGRow = 64;
constexpr int kGPlane = (kPix + 1) * kGRow;    // row 400: zeros (K slots past the image)
constexpr int oX = 0, oG = kXB, kEnd = oG + 3 * kGPlane;
constexpr int kMaxImages = 4096;               // frames per workgroup: their gather entries sit in LDS behind the image
constexpr int kNG4 = kPix * 32 / 4, kGR = (kNG4 + 511) / 512;      // float4 pieces of an image's gradient rows; per lane
constexpr int kNX16 = kFrameB / 16, kXR = (kNX16 + 511) / 512;     // 16-byte pieces of a frame; per lane
static_assert(kFrameB % 16 == 0 && oG % 64 == 0 && kGPlane % 64 == 0 && kEnd + 4 * kMaxImages <= 160 * 1024, "LDS layout");
static_assert(4 * 32 * 256 * 4 <= kEnd, "the final reduction (four partial results) reuses the image's LDS");

__device__ __forceinline__ s16x4 ks_tr(const uint8_t *smem, int off) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(smem + off));  // (flat -> LDS address space)
}
__device__ __forceinline__ bf16x8 ks_frag(s16x4 lo, s16x4 hi) {
  const __attribute__((ext_vector_type(8))) short v = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  return __builtin_bit_cast(bf16x8, v);
}
// two bytes -> two bf16 (exact: the fp32 of an integer < 256 has a zero low half)
__device__ __forceinline__ uint32_t ks_bytes2(uint32_t w, int lo) {
  const float f0 = static_cast<float>((w >> (8 * lo)) & 0xff), f1 = static_cast<float>((w >> (8 * lo + 8)) & 0xff);
  return __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, f1), __builtin_bit_cast(uint32_t, f0), 0x07060302u);
}
__device__ __forceinline__ float ks_div255(float x) {  // x / 255 to within the last bit (conv0_b16.hip)
  const float r = 1.0f / 255.0f;
  const float q = x * r;
  return __builtin_fmaf(__builtin_fmaf(-q, 255.0f, x), r, q);
}

// this lane's row addresses: the gradient row of K slot q of the wave's FIRST step (slot 4 + q is 16 pixels = 1,024 bytes
// on, the next step 32 pixels = 2,048 bytes -- the same swizzle bit) and the frame rows of K slots q / 4 + q of its four steps
struct KsRows { int g, x0[4], x1[4]; };

// MFMA, read, MFMA, read, ... for READS LDS reads among a tile's six MFMAs (a burst of reads holds the wave's in-order
// stream while the LDS takes them: convstack_train.hip)
template <int READS>
__device__ __forceinline__ void ks_pin() {
#pragma unroll
  for (int r = 0; r < READS; ++r) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
  }
  if (READS < 6) __builtin_amdgcn_sched_group_barrier(0x008, 6 - READS, 0);
}

// One image's multiplication by this wave: three full K steps x its eight tap tiles, then the half-valid step 12 on tap
// tiles J0, J0 + 1 (its K slots 16 .. 31 lie past the image: zeros in registers, nothing read).  Behind the current tile's
// MFMAs: the frame fragment of the next tile.  (The next step's gradient fragments read one step ahead into a second
// register set were built and spilled: 256 registers + 19; they are read at the top of their step.)
template <int J0, int STEP12>  // STEP12: step 12 is this many steps behind the wave's first (12 - 3 kg)
__device__ __forceinline__ void ks_multiply(const uint8_t *smem, const KsRows &rows, f32x4 (&acc)[2][8]) {
  constexpr int NT = 26;  // tiles: (step k / 8, tap tile k % 8); 24, 25: the half step's two
  const s16x4 zero = {0, 0, 0, 0};
  auto gfrag = [&](int s, int i, int pl) {
    const int o = pl * kGPlane + (s < 3 ? s : STEP12) * 32 * kGRow;
    return ks_frag(ks_tr(smem, (rows.g ^ (32 * i)) + o), s < 3 ? ks_tr(smem, (rows.g ^ (32 * i)) + o + 16 * kGRow) : zero);
  };
  auto xfrag = [&](int k) {
    const int s = k < 24 ? k / 8 : 3, j = k < 24 ? k % 8 : J0 + (k - 24);
    const int o = (j >> 1) * kXRow + (j & 1) * 32;
    return ks_frag(ks_tr(smem, rows.x0[s] + o), k < 24 ? ks_tr(smem, rows.x1[s] + o) : zero);
  };
  bf16x8 gf[2][3];
  bf16x8 xf[2];  // this tile's and the next tile's frame fragment, alternately
  xf[0] = xfrag(0);
#pragma unroll
  for (int k = 0; k < NT; ++k) {
    const int s = k < 24 ? k / 8 : 3, jj = k < 24 ? k % 8 : k - 24, j = k < 24 ? jj : J0 + jj;  // (compile-time after unrolling)
    if (jj == 0) {  // a step's six gradient fragments (the registers held the previous step's until its last MFMA)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) gf[i][pl] = gfrag(s, i, pl);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (k + 1 < NT) xf[(k + 1) & 1] = xfrag(k + 1);
#pragma unroll
    for (int i = 0; i < 2; ++i) {  // planes smallest first
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[i][2], xf[k & 1], acc[i][j], 0, 0, 0);
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[i][1], xf[k & 1], acc[i][j], 0, 0, 0);
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[i][0], xf[k & 1], acc[i][j], 0, 0, 0);
    }
    const int xreads = k + 1 < NT ? (k + 1 >= 24 ? 1 : 2) : 0;
    switch (xreads) {  // (compile-time)
      case 0: ks_pin<0>(); break;
      case 1: ks_pin<1>(); break;
      default: ks_pin<2>(); break;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// `variant` (DX_DIAG only, DX_KS_VARIANT; WRONG results, timing experiments): 1 = no multiplication, 2 = no loads after
// the first image's, 4 = no copy into LDS (bits may be combined)
__global__ __launch_bounds__(512) void conv0_wgrad_ks_kernel(const Conv0Args a, int B, unsigned long long *stamps, int variant, int descending) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grid = static_cast<int>(gridDim.x), first = static_cast<int>(blockIdx.x);
  const int nimg = (B - first + grid - 1) / grid;
  // The gather table entries of this workgroup's frames, read ONCE into LDS: a scalar load per image, even issued an image
  // ahead, is waited for by the next barrier's lgkmcnt(0) -- a memory round trip per image (1,260 cycles per image in the
  // stamps of this kernel's second version).  `descending`: images are walked from the LAST one down -- the first layer's
  // gradient rows were written by the launch before this one in ascending order, so the tail of them is what the 256 MB
  // last-level cache still holds (180-181 us against 185-188 ascending, same box).
  int *raws = reinterpret_cast<int *>(smem + kEnd);
  auto img_of = [&](int t) { return descending ? B - 1 - (first + t * grid) : first + t * grid; };
  for (int i = tid; i < nimg; i += 512) raws[i] = a.idx ? a.idx[img_of(i)] : img_of(i);
  __syncthreads();
  auto raw_of = [&](int t) { return __builtin_amdgcn_readfirstlane(raws[t]); };
  // DX_DIAG only: shader cycles per phase, summed over this workgroup's images (wave 0)
  unsigned long long ph[4] = {0, 0, 0, 0}, tprev = 0;
#define DX_KS_MARK(i) if (kDiag && stamps) { const unsigned long long now = __builtin_amdgcn_s_memtime(); ph[i] += now - tprev; tprev = now; }

  // ---- staging: piece tid + 512 u of the gradient rows (float4: pixel (tid >> 3) + 64 u, channels 4 (tid & 7) ..) and of
  // the frame (16 bytes = 4 pixels -> two 16-byte stores of bf16.  8-byte pieces -- consecutive lanes then write
  // consecutive 16 bytes, no 2-way conflict on the stores -- were measured 6 % SLOWER: 7 loads per lane instead of 4) ----
  const int gdst = oG + (tid >> 3) * kGRow + 32 * (((tid >> 2) & 1) ^ ((tid >> 5) & 1)) + 8 * (tid & 3);
  const int xdst = oX + 32 * tid;
  // an image's rows travel in registers while the previous image multiplies (the gradient rows TWO images ahead were
  // measured: no faster, 28 registers more)
  f32x4 ga[kGR];
  u32x4 xr[kXR];
  auto fetch_g = [&](f32x4 (&gr)[kGR], int t) {
    if (kDiag && (variant & 2) && t != 0) return;
    const int img = img_of(t);
    const f32x4 *gs = reinterpret_cast<const f32x4 *>(a.G) + static_cast<long long>(img) * kNG4;
#pragma unroll
    for (int u = 0; u < kGR; ++u) gr[u] = gs[min(tid + 512 * u, kNG4 - 1)];
  };
  auto fetch_x = [&](int t) {
    if (kDiag && (variant & 2) && t != 0) return;
    const int raw = raw_of(t);
    const u32x4 *xs = reinterpret_cast<const u32x4 *>(a.obs + static_cast<long long>(raw) * kFrameB);
#pragma unroll
    for (int u = 0; u < kXR; ++u) xr[u] = xs[min(tid + 512 * u, kNX16 - 1)];
  };

  // ---- this wave's K steps and this lane's rows in them: K slot 8 g + 4 r + q holds pixel 32 s + 16 r + 4 g + q ----
  const int g4 = lane >> 4, q = (lane >> 2) & 3, p4 = lane & 3;
  const int kg = wave & 3, nh = wave >> 2;  // K group, tap half (kernel rows 4 nh .. 4 nh + 3)
  KsRows rows;
  rows.g = oG + (32 * 3 * kg + 4 * g4 + q) * kGRow + 32 * (g4 & 1) + 8 * p4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int s = i < 3 ? 3 * kg + i : 12;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int pc = min(32 * s + 16 * h + 4 * g4 + q, kPix - 1), oy = pc / kOW, ox = pc - oy * kOW;  // (past the image: never read)
      (h ? rows.x1[i] : rows.x0[i]) = oX + (4 * oy + 4 * nh) * kXRow + 32 * ox + 8 * p4;
    }
  }

  f32x4 acc[2][8];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};  // column sums of the gradient rows this lane copies (channels 4 (tid & 7) ..)

  // one image: its rows -> LDS, the registers refilled (gradient rows of image t + 2, frame of image t + 1), multiply
  auto image = [&](int t, f32x4 (&gr)[kGR]) {
    if (kDiag && stamps) tprev = __builtin_amdgcn_s_memtime();
    if (t > 0) __syncthreads();  // every wave is done with the previous image
    DX_KS_MARK(0)
    const bool copies = !(kDiag && (variant & 4));
    if (!copies) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the loads are still waited for)
#pragma unroll
    for (int u = 0; u < kXR; ++u)
      if (copies && tid + 512 * u < kNX16) {
        const u32x4 w = xr[u];
        *reinterpret_cast<u32x4 *>(smem + xdst + 32 * 512 * u) =
            u32x4{ks_bytes2(w.x, 0), ks_bytes2(w.x, 2), ks_bytes2(w.y, 0), ks_bytes2(w.y, 2)};
        *reinterpret_cast<u32x4 *>(smem + xdst + 32 * 512 * u + 16) =
            u32x4{ks_bytes2(w.z, 0), ks_bytes2(w.z, 2), ks_bytes2(w.w, 0), ks_bytes2(w.w, 2)};
      }
    if (t + 1 < nimg) fetch_x(t + 1);
#pragma unroll
    for (int u = 0; u < kGR; ++u)
      if (copies && tid + 512 * u < kNG4) {
        const Split4 s = split4(gr[u]);
        *reinterpret_cast<uint2 *>(smem + gdst + 64 * kGRow * u) = s.hi;
        *reinterpret_cast<uint2 *>(smem + gdst + 64 * kGRow * u + kGPlane) = s.mid;
        *reinterpret_cast<uint2 *>(smem + gdst + 64 * kGRow * u + 2 * kGPlane) = s.lo;
        bsum += gr[u];
      }
    if (t + 1 < nimg) fetch_g(gr, t + 1);
    DX_KS_MARK(1)
    __syncthreads();
    DX_KS_MARK(2)
    if (!(kDiag && (variant & 1))) {
      switch (kg) {  // (wave-uniform; the accumulator tiles need compile-time indices)
        case 0: ks_multiply<0, 12>(smem, rows, acc); break;
        case 1: ks_multiply<2, 9>(smem, rows, acc); break;
        case 2: ks_multiply<4, 6>(smem, rows, acc); break;
        default: ks_multiply<6, 3>(smem, rows, acc); break;
      }
    }
    DX_KS_MARK(3)
  };
  fetch_x(0);
  fetch_g(ga, 0);
  for (int t = 0; t < nimg; ++t) image(t, ga);
  if (kDiag && stamps && tid == 0) {
    unsigned long long *o = stamps + static_cast<long long>(blockIdx.x) * 4;
    for (int i = 0; i < 4; ++i) o[i] = ph[i];
  }
#undef DX_KS_MARK

  // ---- the four K groups' partial results meet in LDS, added in order ----
  float *part = reinterpret_cast<float *>(smem);  // [kg][nh][i][j][r][lane]
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) part[((((kg * 2 + nh) * 2 + i) * 8 + j) * 4 + r) * 64 + lane] = acc[i][j][r];
  __syncthreads();
  // slab[oc][k], k = 32 kh + 4 kw + ci: element (oc, k) sits in tap half k / 128, tile (i = oc / 16, j = (k % 128) / 16),
  // register r = oc % 4 of lane 16 ((oc % 16) / 4) + k % 16 (the MFMA's output map: D[4 (lane >> 4) + r][lane & 15])
  float *slab = a.slab + static_cast<long long>(first) * 32 * 256;
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int e = tid + 512 * u, oc = e >> 8, k = e & 255;
    const int at = (((((k >> 7) * 2 + (oc >> 4)) * 8 + ((k >> 4) & 7)) * 4 + (oc & 3)) * 64) + 16 * ((oc & 15) >> 2) + (k & 15);
    slab[e] = ks_div255(((part[at] + part[at + 8192]) + part[at + 2 * 8192]) + part[at + 3 * 8192]);
  }
  // ---- bias gradient: the 64 lanes with the same tid & 7 hold partial sums of the same four channels; eight groups of
  // eight in a fixed order, then the eight group sums ----
  if (a.bias_slab) {
    __syncthreads();
    reinterpret_cast<f32x4 *>(smem)[tid] = bsum;
    __syncthreads();
    float s = 0.f;
    if (tid < 256) {
      const int oc = tid & 31, g = tid >> 5;
#pragma unroll
      for (int m = 8 * g; m < 8 * g + 8; ++m) s += reinterpret_cast<const float *>(smem)[(8 * m + (oc >> 2)) * 4 + (oc & 3)];
    }
    __syncthreads();
    if (tid < 256) reinterpret_cast<float *>(smem)[tid] = s;
    __syncthreads();
    if (tid < 32) {
      float v = 0.f;
#pragma unroll
      for (int g = 0; g < 8; ++g) v += reinterpret_cast<const float *>(smem)[32 * g + tid];
      a.bias_slab[static_cast<long long>(first) * 32 + tid] = v;
    }
  }
}

}  // namespace

bool bwd_descending(int stage) {  // (igemm.hpp: which way a backward stage walks the minibatch)
  const int mask = DX_ENV("DX_BWD_ORDER", 21);
  switch (stage) {
    case ST_CONV2_WGRAD: return (mask & 1) != 0;
    case ST_CONV2_DGRAD: return (mask & 2) != 0;
    case ST_CONV1_WGRAD: return (mask & 4) != 0;
    case ST_CONV1_DGRAD: return (mask & 8) != 0;
    case ST_CONV0_WGRAD: return (mask & 16) != 0;
    default: return false;
  }
}

// DX_CONV0_KS=0: conv0_b16.hip's tile kernel (two k halves x two pixel halves per 256-pixel tile)
bool conv0_wgrad_ks_on() { return DX_ENV("DX_CONV0_KS", 1) != 0; }
bool conv0_wgrad_ks_supported(int in_h, int in_w, int in_c, int h0, int w0) {
  return in_h == 84 && in_w == 84 && in_c == 4 && h0 == 20 && w0 == 20;
}
// slabs (= workgroups) a minibatch of B frames is reduced over: one workgroup per CU, at most one per frame
int conv0_wgrad_ks_workgroups(long long B) { return static_cast<int>(B < 256 ? B : 256); }

int launch_conv0_wgrad_ks(const Conv0Args &a, int nblocks, hipStream_t stream) {
  DX_REQUIRE(a.obs && a.G && a.slab && a.M > 0 && a.M % kPix == 0, "conv0_wgrad_ks: bad arguments");
  DX_REQUIRE(conv0_wgrad_ks_supported(a.in_h, a.in_w, 4, a.h0, a.w0), "conv0_wgrad_ks: 84 x 84 x 4 frames only (%d x %d -> %d x %d)",
             a.in_h, a.in_w, a.h0, a.w0);
  const int B = a.M / kPix;
  DX_REQUIRE(nblocks >= 1 && nblocks <= B && nblocks <= 256, "conv0_wgrad_ks: %d workgroups for %d frames", nblocks, B);
  DX_REQUIRE(aligned(a.obs, 16) && aligned(a.G, 16), "conv0_wgrad_ks: frames and gradient rows must be 16-byte aligned");
  const int per_wg = cdiv(B, nblocks), lds = kEnd + 4 * per_wg;
  DX_REQUIRE(per_wg <= kMaxImages, "conv0_wgrad_ks: %d frames per workgroup (max %d)", per_wg, kMaxImages);
  DX_LDS_OPT_IN(conv0_wgrad_ks_kernel, kEnd + 4 * kMaxImages);
  const int descending = bwd_descending(ST_CONV0_WGRAD) ? 1 : 0;
  int variant = 0;
#if DX_DIAG
  variant = DX_ENV("DX_KS_VARIANT", 0);
  if (getenv("DX_C0_DIAG")) {  // in-kernel phase cycles, summarised on stderr (synchronous)
    unsigned long long *dev = nullptr;
    DX_HIP(hipMalloc(&dev, static_cast<size_t>(nblocks) * 64));
    hipLaunchKernelGGL(conv0_wgrad_ks_kernel, dim3(nblocks), dim3(512), lds, stream, a, B, dev, variant, descending);
    DX_LAUNCH_CHECK();
    DX_HIP(hipStreamSynchronize(stream));
    std::vector<unsigned long long> h(static_cast<size_t>(nblocks) * 4);
    DX_HIP(hipMemcpy(h.data(), dev, h.size() * 8, hipMemcpyDeviceToHost));
    DX_HIP(hipFree(dev));
    double sum[4] = {0, 0, 0, 0};
    for (int b = 0; b < nblocks; ++b)
      for (int i = 0; i < 4; ++i) sum[i] += static_cast<double>(h[static_cast<size_t>(b) * 4 + i]);
    fprintf(stderr, "[conv0_wgrad_ks B=%d grid=%d] cycles per image (wave 0): barrier-in %.0f, copy (waits for its loads) "
            "%.0f, barrier %.0f, multiply %.0f\n", B, nblocks, sum[0] / B, sum[1] / B, sum[2] / B, sum[3] / B);
    return DX_OK;
  }
#endif
  hipLaunchKernelGGL(conv0_wgrad_ks_kernel, dim3(nblocks), dim3(512), lds, stream, a, B, static_cast<unsigned long long *>(nullptr), variant, descending);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

}  // namespace dx
