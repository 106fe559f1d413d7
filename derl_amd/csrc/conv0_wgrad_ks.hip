// Weight gradient of the first convolution (8x8 stride 4 on uint8 84 x 84 x 4 frames; the autograd backward of
// derl/models.py:103,117-124 triggered by derl/alg/common.py:70), round 6: the contraction over output pixels SPLIT
// ACROSS THE WAVES, and an image's copy into LDS running UNDER the previous half image's multiplication.
//
//   dW[oc][kh][kw][ci] = 1/255 * sum over (img, oy, ox) of dY0[img][oy][ox][oc] * frame[img][4 oy + kh][4 ox + kw][ci]
//
// A GEMM with M = 32 output channels, N = 256 taps and K = 400 pixels per image that moves 79 KB per image for 9.8 M
// multiply-adds (x3 bf16 terms): it is bound by how continuously those bytes stream, not by the matrix cores.
// conv0_b16.hip's tile kernel (two k halves x two pixel halves per 256-pixel tile, 32x32x16, bytes gathered and converted
// PER USE: 48 conversions per 12 MFMAs) runs with the matrix pipe 0.42 busy and 24 % of its LDS cycles in bank conflicts
// (178-188 us at minibatch 8192, 3.6 TB/s); round 4's transposed-read kernel read a K step's operands 3x over and was
// slower.  This kernel's first two versions (whole-image units, one LDS image: copy -> barrier -> multiply -> barrier)
// measured 173-177 us: per image 3,400 cycles of copying (vector ALU: the exact split and the byte conversion), 6,100 of
// multiplying and 1,300 of barriers, one after the other (in-kernel stamps, DX_C0_DIAG) -- the matrix pipe idle while
// the vector ALU worked and vice versa.  Now:
//   * a UNIT is half an image: unit A = pixels 0 .. 191 (6 K steps of 32 pixels; frame rows 0 .. 43), unit B = pixels
//     192 .. 399 (6 steps + a half-valid one; frame rows 36 .. 83).  LDS holds both: while unit u multiplies from its
//     buffer, unit u + 1 is copied into the other.  The two waves of a SIMD (w, w + 4) run the two jobs in OPPOSITE
//     order -- waves 0 .. 3 copy first and multiply second, waves 4 .. 7 multiply first -- so a SIMD's vector ALU and its
//     matrix pipe work at the same time on different waves; one barrier per unit.
//   * wave w = (K group w >> 2, tap quarter w & 3: kernel rows 2 (w & 3), + 1) holds 2 x 4 accumulator tiles (32
//     registers) and takes 3 K steps per unit + half of unit B's half step: per step 12 transposed reads of the gradient
//     fragments (2 channel tiles x 3 planes), 8 of the frame's and 24 v_mfma_f32_16x16x32_bf16; the next step's
//     fragments are read behind this step's MFMAs.  13 steps per image where 12.5 are useful (executed / useful 1.04).
//   * the frame is converted to bf16 ONCE, when it is copied into LDS (a byte is exact in bf16): LDS image [row][84
//     pixels][4 channels] bf16 in the frame's own order, so the 16 taps (kw, ci) of half a kernel row are 32 contiguous
//     bytes at pixel (4 oy + kh, 4 ox) and ds_read_b64_tr_b16 -- every lane supplies the address of ITS pixel -- hands
//     each lane 4 consecutive K slots of its tap: no im2col, no gather table.  The natural 672-byte row pitch is
//     conflict-free for the 8 consecutive pixels a 32-lane half reads (32-byte pixel stride; an 8-pixel group that
//     crosses an output row lands 4 x 672 = 128 (mod 256) bytes on).
//   * dY0 (fp32) is split EXACTLY into three bf16 planes [pixel][32 channels] (64-byte rows, the two 32-byte channel
//     halves swapped on rows with bit 2 set: conflict-free transposed reads without padding); byte x bf16 products are
//     exact in fp32, the planes are accumulated smallest first, fp32 accumulation as in every fp32 chain.
//   * a unit's rows travel in registers from the moment the previous unit of its kind was copied (two units = one image
//     ahead, refilled at once: 79 KB per CU always in flight); accumulators stay in registers over all images of the
//     workgroup (one per CU; images blockIdx.x, + grid, ...); at the end the two K groups' partial results meet in LDS
//     in a fixed order (deterministic), are scaled by 1/255 and leave as ONE slab per workgroup (<= 256 slabs where the
//     tile kernel wrote 512).
// LDS: 2 x (48 frame rows x 672 + 3 planes x 209 gradient rows x 64) = 144,768 bytes.
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
constexpr int kFrameRowB = 84 * 4;
constexpr int kXRow = 84 * 4 * 2;              // bytes of a bf16 LDS row
constexpr int kPix = 400, kOW = 20, kGRow = 64;
// unit A / B: pixels, first frame row of unit B, frame rows
constexpr int kPixA = 192, kPixB = kPix - kPixA, kRowB0 = 36, kRowsA = 44, kRowsB = 48;
static_assert(4 * ((kPixA - 1) / kOW) + 8 <= kRowsA && 4 * (kPixA / kOW) >= kRowB0 && kRowB0 + kRowsB == 84, "frame rows of the units");
constexpr int kXBuf = kRowsB * kXRow;                  // 32,256
constexpr int kGPlane = (kPixB + 1) * kGRow;           // 208 rows + one of slack (the half step's K slots 16 .. 31 are never read: zeros in registers)
constexpr int kBuf = kXBuf + 3 * kGPlane;              // 72,384
constexpr int kEnd = 2 * kBuf;
constexpr int kMaxImages = 4096;                       // frames per workgroup: their gather entries sit in LDS behind the buffers
constexpr int kNGA = kPixA * 8, kNGB = kPixB * 8, kGR = (kNGB + 511) / 512;                        // float4 pieces of a unit's gradient rows; per lane
constexpr int kNXA = kRowsA * kFrameRowB / 8, kNXB = kRowsB * kFrameRowB / 8, kXR = (kNXB + 511) / 512;  // 8-byte pieces of its frame rows
static_assert(kXBuf % 64 == 0 && kGPlane % 64 == 0 && kBuf % 64 == 0 && kEnd + 4 * kMaxImages <= 160 * 1024, "LDS layout");
static_assert(2 * 32 * 256 * 4 <= kEnd, "the final reduction (two partial results) reuses the images' LDS");
static_assert(kNGA % 512 == 0 && kGR == 4 && kXR == 4, "pieces per lane");

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

// this lane's row addresses in one unit: the gradient row of K slot q of the wave's FIRST step (slot 4 + q is 16 pixels =
// 1,024 bytes on, the next step 32 pixels = 2,048 bytes -- the same swizzle bit; past unit B's last pixel nothing is read)
// and the frame rows of K slots q / 4 + q of its (up to) four steps
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

// One unit's multiplication: three full K steps x this wave's four tap tiles, then (LEFT: unit B) the half-valid step on
// tap tiles J0, J0 + 1.  Read behind the current tile's MFMAs: the frame fragment of the tile AFTER the next (one tile
// ahead -- 4 MFMAs = 64 cycles between issue and use -- left the waves waiting for LDS at every tile: 25 cycles per MFMA
// per SIMD with nothing but this loop running) and, two fragments per tile, the next step's six gradient fragments.
template <bool LEFT, int J0>
__device__ __forceinline__ void ks_multiply(const uint8_t *smem, const KsRows &rows, f32x4 (&acc)[2][4]) {
  constexpr int NT = LEFT ? 14 : 12;  // tiles: (step k / 4, tap tile k % 4); 12, 13: the half step's two
  const s16x4 zero = {0, 0, 0, 0};
  auto gfrag = [&](int s, int i, int pl, bool half) {  // (the half step is unit B's step 6: three or six steps behind the wave's first)
    const int o = pl * kGPlane + (s < 3 ? s : (J0 == 0 ? 6 : 3)) * 32 * kGRow;
    return ks_frag(ks_tr(smem, (rows.g ^ (32 * i)) + o), half ? zero : ks_tr(smem, (rows.g ^ (32 * i)) + o + 16 * kGRow));
  };
  auto xfrag = [&](int k) {
    const int s = k < 12 ? k / 4 : 3, j = k < 12 ? k % 4 : J0 + (k - 12);
    const int o = (j >> 1) * kXRow + (j & 1) * 32;
    return ks_frag(ks_tr(smem, rows.x0[s] + o), k >= 12 ? zero : ks_tr(smem, rows.x1[s] + o));
  };
  bf16x8 gf[2][2][3], xf[3];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) gf[0][i][pl] = gfrag(0, i, pl, false);
  xf[0] = xfrag(0);
  xf[1] = xfrag(1);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int k = 0; k < NT; ++k) {
    const int s = k < 12 ? k / 4 : 3, jj = k < 12 ? k % 4 : k - 12, j = k < 12 ? jj : J0 + jj;  // (compile-time after unrolling)
    const bool next_half = LEFT && s == 2, last_step = s + 1 == (LEFT ? 4 : 3);
    if (k + 2 < NT) xf[(k + 2) % 3] = xfrag(k + 2);
    const bool gnext = !last_step && jj < 3;  // fragments 2 jj, 2 jj + 1 of the next step's six
    if (gnext) {
#pragma unroll
      for (int f = 2 * jj; f < 2 * jj + 2; ++f) gf[(s + 1) & 1][f / 3][f % 3] = gfrag(s + 1, f / 3, f % 3, next_half);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {  // planes smallest first
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[s & 1][i][2], xf[k % 3], acc[i][j], 0, 0, 0);
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[s & 1][i][1], xf[k % 3], acc[i][j], 0, 0, 0);
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[s & 1][i][0], xf[k % 3], acc[i][j], 0, 0, 0);
    }
    const int xreads = k + 2 < NT ? (k + 2 >= 12 ? 1 : 2) : 0;
    const int greads = gnext ? 2 * (next_half ? 1 : 2) : 0;
    switch (xreads + greads) {  // (compile-time)
      case 0: ks_pin<0>(); break;
      case 1: ks_pin<1>(); break;
      case 2: ks_pin<2>(); break;
      case 3: ks_pin<3>(); break;
      case 4: ks_pin<4>(); break;
      case 5: ks_pin<5>(); break;
      default: ks_pin<6>(); break;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// `variant` (DX_DIAG only, DX_KS_VARIANT; WRONG results, timing experiments): 1 = no multiplication, 2 = no loads after
// the prologue's, 4 = no copy into LDS (bits may be combined)
__global__ __launch_bounds__(512) void conv0_wgrad_ks_kernel(const Conv0Args a, int B, unsigned long long *stamps, int variant) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grid = static_cast<int>(gridDim.x), first = static_cast<int>(blockIdx.x);
  const int nimg = (B - first + grid - 1) / grid;
  // The gather table entries of this workgroup's frames, read ONCE into LDS: a scalar load per image, even issued an image
  // ahead, is waited for by the next barrier's lgkmcnt(0) -- a memory round trip per image (the "nothing but barriers"
  // variant of this kernel took 43 us at 32 images per workgroup)
  int *raws = reinterpret_cast<int *>(smem + kEnd);
  for (int i = tid; i < nimg; i += 512) raws[i] = a.idx ? a.idx[first + i * grid] : first + i * grid;
  __syncthreads();
  auto raw_of = [&](int t) { return __builtin_amdgcn_readfirstlane(raws[t]); };
  // DX_DIAG only: shader cycles per phase, summed over this workgroup's units (waves 0 and 4)
  unsigned long long ph[3] = {0, 0, 0}, tprev = 0;
#define DX_KS_MARK(i) if (kDiag && stamps) { const unsigned long long now = __builtin_amdgcn_s_memtime(); ph[i] += now - tprev; tprev = now; }

  // LDS: buffer 0 (unit A) / buffer 1 (unit B): frame rows, then the three gradient planes
  constexpr int oXA = 0, oGA = kXBuf, oXB = kBuf, oGB = kBuf + kXBuf;

  // ---- copying: piece tid + 512 u of a unit's gradient rows (float4: local pixel (tid >> 3) + 64 u, channels 4 (tid & 7) ..)
  // and of its frame rows (8 bytes = 2 pixels -> 16 bytes of bf16: consecutive lanes write consecutive 16 bytes) ----
  const int gdst = (tid >> 3) * kGRow + 32 * (((tid >> 2) & 1) ^ ((tid >> 5) & 1)) + 8 * (tid & 3);
  const int xdst = 16 * tid;
  f32x4 ga[kGR], gb[kGR];  // unit A's / unit B's rows in flight
  uint2 xa[kXR], xb[kXR];
  auto fetch = [&](f32x4 (&gr)[kGR], uint2 (&xr)[kXR], int img, int raw, bool unit_b) {
      if (kDiag && (variant & 2) && img != first) return;
    const f32x4 *gs = reinterpret_cast<const f32x4 *>(a.G) + static_cast<long long>(img) * (kPix * 8) + (unit_b ? kNGA : 0);
    const uint2 *xs = reinterpret_cast<const uint2 *>(a.obs + static_cast<long long>(raw) * kFrameB + (unit_b ? kRowB0 * kFrameRowB : 0));
    const int ng = unit_b ? kNGB : kNGA, nx = unit_b ? kNXB : kNXA;
#pragma unroll
    for (int u = 0; u < kXR; ++u) xr[u] = xs[min(tid + 512 * u, nx - 1)];
#pragma unroll
    for (int u = 0; u < kGR; ++u)
      if (unit_b || u < kNGA / 512) gr[u] = gs[min(tid + 512 * u, ng - 1)];
  };
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};  // column sums of the gradient rows this lane copies (channels 4 (tid & 7) ..)
  auto copy = [&](const f32x4 (&gr)[kGR], const uint2 (&xr)[kXR], bool unit_b) {
    if (kDiag && (variant & 4)) {  // (the loads are still waited for)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      return;
    }
    const int ox = unit_b ? oXB : oXA, og = unit_b ? oGB : oGA;
    const int ng = unit_b ? kNGB : kNGA, nx = unit_b ? kNXB : kNXA;
#pragma unroll
    for (int u = 0; u < kXR; ++u)
      if (tid + 512 * u < nx) {
        const uint2 w = xr[u];
        *reinterpret_cast<u32x4 *>(smem + ox + xdst + 16 * 512 * u) =
            u32x4{ks_bytes2(w.x, 0), ks_bytes2(w.x, 2), ks_bytes2(w.y, 0), ks_bytes2(w.y, 2)};
      }
#pragma unroll
    for (int u = 0; u < kGR; ++u)
      if ((unit_b || u < kNGA / 512) && tid + 512 * u < ng) {
        const Split4 s = split4(gr[u]);
        *reinterpret_cast<uint2 *>(smem + og + gdst + 64 * kGRow * u) = s.hi;
        *reinterpret_cast<uint2 *>(smem + og + gdst + 64 * kGRow * u + kGPlane) = s.mid;
        *reinterpret_cast<uint2 *>(smem + og + gdst + 64 * kGRow * u + 2 * kGPlane) = s.lo;
        bsum += gr[u];
      }
  };

  // ---- this wave's K steps and this lane's rows in them: K slot 8 g + 4 r + q holds the unit's pixel 32 s + 16 r + 4 g + q ----
  const int g4 = lane >> 4, q = (lane >> 2) & 3, p4 = lane & 3;
  const int kg = wave >> 2, nq = wave & 3;  // K group (steps 3 kg ..), tap quarter (kernel rows 2 nq, 2 nq + 1)
  bool copies_first = wave < 4;             // (its partner on the SIMD, wave + 4, multiplies first)
  if (kDiag && (variant & 8)) copies_first = true;    // (timing experiments: every wave in the same order)
  if (kDiag && (variant & 16)) copies_first = false;
  auto rows_of = [&](bool unit_b) {
    KsRows r;
    r.g = (unit_b ? oGB : oGA) + (32 * 3 * kg + 4 * g4 + q) * kGRow + 32 * (g4 & 1) + 8 * p4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int s = i < 3 ? 3 * kg + i : 6;  // (unit B's half step)
      int x[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int pl = 32 * s + 16 * h + 4 * g4 + q, npix = unit_b ? kPixB : kPixA;  // the unit's local pixel
        const int pc = min(pl, npix - 1) + (unit_b ? kPixA : 0), oy = pc / kOW, ox = pc - oy * kOW;  // (past the unit: never read)
        x[h] = (unit_b ? oXB : oXA) + (4 * oy + 2 * nq - (unit_b ? kRowB0 : 0)) * kXRow + 32 * ox + 8 * p4;
      }
      r.x0[i] = x[0]; r.x1[i] = x[1];
    }
    return r;
  };
  const KsRows rows_a = rows_of(false), rows_b = rows_of(true);

  f32x4 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto multiply_a = [&]() {
    if (kDiag && (variant & 1)) return;
    ks_multiply<false, 0>(smem, rows_a, acc);
  };
  auto multiply_b = [&]() {
    if (kDiag && (variant & 1)) return;
    if (kg == 0) ks_multiply<true, 0>(smem, rows_b, acc);  // (wave-uniform; the accumulator tiles need compile-time indices)
    else ks_multiply<true, 2>(smem, rows_b, acc);
  };

  // ---- prologue: image 0's two units in flight, unit A copied, its registers refilled with image 1's ----
  {
    const int raw0 = raw_of(0);
    fetch(ga, xa, first, raw0, false);
    fetch(gb, xb, first, raw0, true);
  }
  copy(ga, xa, false);
  if (nimg > 1) fetch(ga, xa, first + grid, raw_of(1), false);
  __syncthreads();
  if (kDiag && stamps) tprev = __builtin_amdgcn_s_memtime();
  for (int t = 0; t < nimg; ++t) {
    // period 1: unit A of image t multiplies, unit B of image t is copied (its registers then take image t + 1's)
    if (copies_first) {
      copy(gb, xb, true);
      if (t + 1 < nimg) fetch(gb, xb, first + (t + 1) * grid, raw_of(t + 1), true);
      DX_KS_MARK(0)
      multiply_a();
      DX_KS_MARK(1)
    } else {
      multiply_a();
      DX_KS_MARK(1)
      copy(gb, xb, true);
      if (t + 1 < nimg) fetch(gb, xb, first + (t + 1) * grid, raw_of(t + 1), true);
      DX_KS_MARK(0)
    }
    __syncthreads();
    DX_KS_MARK(2)
    // period 2: unit B of image t multiplies, unit A of image t + 1 is copied (its registers then take image t + 2's)
    const bool more = t + 1 < nimg;
    if (copies_first) {
      if (more) {
        copy(ga, xa, false);
        if (t + 2 < nimg) fetch(ga, xa, first + (t + 2) * grid, raw_of(t + 2), false);
      }
      DX_KS_MARK(0)
      multiply_b();
      DX_KS_MARK(1)
    } else {
      multiply_b();
      DX_KS_MARK(1)
      if (more) {
        copy(ga, xa, false);
        if (t + 2 < nimg) fetch(ga, xa, first + (t + 2) * grid, raw_of(t + 2), false);
      }
      DX_KS_MARK(0)
    }
    __syncthreads();
    DX_KS_MARK(2)
  }
  if (kDiag && stamps && (tid == 0 || tid == 256)) {
    unsigned long long *o = stamps + (static_cast<long long>(blockIdx.x) * 2 + (tid >> 8)) * 4;
    for (int i = 0; i < 3; ++i) o[i] = ph[i];
  }
#undef DX_KS_MARK

  // ---- the two K groups' partial results meet in LDS (every wave is past the loop's last barrier) ----
  float *part = reinterpret_cast<float *>(smem);  // [kg][nq][i][j][r][lane]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) part[((((kg * 4 + nq) * 2 + i) * 4 + j) * 4 + r) * 64 + lane] = acc[i][j][r];
  __syncthreads();
  // slab[oc][k], k = 32 kh + 4 kw + ci: element (oc, k) sits in tap quarter k / 64, tile (i = oc / 16, j = (k % 64) / 16),
  // register r = oc % 4 of lane 16 ((oc % 16) / 4) + k % 16 (the MFMA's output map: D[4 (lane >> 4) + r][lane & 15])
  float *slab = a.slab + static_cast<long long>(first) * 32 * 256;
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int e = tid + 512 * u, oc = e >> 8, k = e & 255;
    const int at = (((((k >> 6) * 2 + (oc >> 4)) * 4 + ((k >> 4) & 3)) * 4 + (oc & 3)) * 64) + 16 * ((oc & 15) >> 2) + (k & 15);
    slab[e] = ks_div255(part[at] + part[at + 8192]);
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
  int variant = 0;
#if DX_DIAG
  variant = DX_ENV("DX_KS_VARIANT", 0);
  if (getenv("DX_C0_DIAG")) {  // in-kernel phase cycles, summarised on stderr (synchronous)
    unsigned long long *dev = nullptr;
    DX_HIP(hipMalloc(&dev, static_cast<size_t>(nblocks) * 64));
    hipLaunchKernelGGL(conv0_wgrad_ks_kernel, dim3(nblocks), dim3(512), lds, stream, a, B, dev, variant);
    DX_LAUNCH_CHECK();
    DX_HIP(hipStreamSynchronize(stream));
    std::vector<unsigned long long> h(static_cast<size_t>(nblocks) * 8);
    DX_HIP(hipMemcpy(h.data(), dev, h.size() * 8, hipMemcpyDeviceToHost));
    DX_HIP(hipFree(dev));
    double sum[2][3] = {{0, 0, 0}, {0, 0, 0}};
    for (int b = 0; b < nblocks; ++b)
      for (int w = 0; w < 2; ++w)
        for (int i = 0; i < 3; ++i) sum[w][i] += static_cast<double>(h[(static_cast<size_t>(b) * 2 + w) * 4 + i]);
    for (int w = 0; w < 2; ++w)
      fprintf(stderr, "[conv0_wgrad_ks B=%d grid=%d] cycles per image, wave %d (%s): copy (waits for its loads) %.0f, multiply %.0f, "
              "barriers %.0f\n", B, nblocks, 4 * w, w ? "multiplies first" : "copies first", sum[w][0] / B, sum[w][1] / B, sum[w][2] / B);
    return DX_OK;
  }
#endif
  hipLaunchKernelGGL(conv0_wgrad_ks_kernel, dim3(nblocks), dim3(512), lds, stream, a, B, static_cast<unsigned long long *>(nullptr), variant);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

}  // namespace dx
