// Diagnostics: what the fp32 matrix pipe of this part sustains with no memory traffic at all
// (the ceiling the GEMM stages are judged against in DESIGN.md next to the data-sheet peak).
#include "../common.hpp"
#include "../../../include/derl_amd_diag.h"

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

typedef float f32x4 __attribute__((ext_vector_type(4)));

// The inner loop of igemm_nt_kernel<.., 128, 64, 64, 32, ..> without its global loads, LDS writes
// and barriers: per 8-k sub-step a wave reads two A fragments and one B fragment (ds_read_b128,
// rows 36 floats apart) and issues 8 MFMAs on its 64x32 register tile.  MODE 1 reads the next
// sub-step's fragments before the current MFMAs (explicit double buffer).
template <int MODE>
__global__ __launch_bounds__(256) void lds_mfma_loop_kernel(int iters, float *out) {
  constexpr int LD = 36;
  __shared__ __attribute__((aligned(16))) float smem[192 * LD];
  // DX_DIAG_RANDOM (iters < 0): full-range pseudo-random operands instead of 8 distinct values --
  // the matrix pipe's power, and with it the clock the chip holds, depends on the data
  const bool random_data = iters < 0;
  if (random_data) iters = -iters;
  for (int i = threadIdx.x; i < 192 * LD; i += 256) {
    unsigned h = (i + 1) * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    smem[i] = random_data ? (static_cast<float>(h & 0xffffff) / 8388608.f - 1.f) * 1e-3f : 1.f + (i & 7) * 0.125f;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float *pa = smem + ((wave >> 1) * 64 + (lane & 31)) * LD + (lane >> 5) * 4;
  const float *pb = smem + (128 + (wave & 1) * 32 + (lane & 31)) * LD + (lane >> 5) * 4;
  f32x16 acc0, acc1;
#pragma unroll
  for (int j = 0; j < 16; ++j) { acc0[j] = 0.f; acc1[j] = 0.f; }
  if (MODE == 0) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int ss = 0; ss < 4; ++ss) {
        const f32x4 a0 = *reinterpret_cast<const f32x4 *>(pa + ss * 8);
        const f32x4 a1 = *reinterpret_cast<const f32x4 *>(pa + 32 * LD + ss * 8);
        const f32x4 b = *reinterpret_cast<const f32x4 *>(pb + ss * 8);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[q], b[q], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[q], b[q], acc1, 0, 0, 0);
        }
      }
      asm volatile("" ::: "memory");
    }
  } else {
    f32x4 a0 = *reinterpret_cast<const f32x4 *>(pa), a1 = *reinterpret_cast<const f32x4 *>(pa + 32 * LD),
          b = *reinterpret_cast<const f32x4 *>(pb);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int ss = 0; ss < 4; ++ss) {
        const int nx = ((ss + 1) & 3) * 8;
        const f32x4 na0 = *reinterpret_cast<const f32x4 *>(pa + nx);
        const f32x4 na1 = *reinterpret_cast<const f32x4 *>(pa + 32 * LD + nx);
        const f32x4 nb = *reinterpret_cast<const f32x4 *>(pb + nx);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[q], b[q], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[q], b[q], acc1, 0, 0, 0);
        }
        a0 = na0; a1 = na1; b = nb;
      }
      asm volatile("" ::: "memory");
    }
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) s += acc0[j] + acc1[j];
  if (s == 12345.678f) out[0] = s;
}

// The same LDS-fed loop with EIGHT accumulator tiles per wave (the register shape of the wgrad
// kernels).  MODE 2: every MFMA goes to another tile (a tile is revisited after 8 MFMAs);
// MODE 3: the 4 MFMAs of a tile in a row (accumulate chains), tile after tile.
template <int MODE>
__global__ __launch_bounds__(256) void lds_mfma8_loop_kernel(int iters, float *out) {
  constexpr int LD = 36;
  __shared__ __attribute__((aligned(16))) float smem[192 * LD];
  for (int i = threadIdx.x; i < 192 * LD; i += 256) {
    unsigned h = (i + 1) * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    smem[i] = (static_cast<float>(h & 0xffffff) / 8388608.f - 1.f) * 1e-3f;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float *pa = smem + ((wave >> 1) * 64 + (lane & 31)) * LD + (lane >> 5) * 4;
  const float *pb = smem + (128 + (wave & 1) * 32 + (lane & 31)) * LD + (lane >> 5) * 4;
  f32x16 acc[8];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;
  f32x4 a0 = *reinterpret_cast<const f32x4 *>(pa), a1 = *reinterpret_cast<const f32x4 *>(pa + 32 * LD),
        b = *reinterpret_cast<const f32x4 *>(pb);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ss = 0; ss < 4; ++ss) {
      const int nx = ((ss + 1) & 3) * 8;
      const f32x4 na0 = *reinterpret_cast<const f32x4 *>(pa + nx);
      const f32x4 na1 = *reinterpret_cast<const f32x4 *>(pa + 32 * LD + nx);
      const f32x4 nb = *reinterpret_cast<const f32x4 *>(pb + nx);
      if (MODE == 2) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          acc[2 * q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[q], b[q], acc[2 * q], 0, 0, 0);
          acc[2 * q + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[q], b[q], acc[2 * q + 1], 0, 0, 0);
        }
      } else {
        const int t0 = (ss & 1) * 4;  // two tiles per sub-step, four chained MFMAs each
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[t0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[q], b[q], acc[t0], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[t0 + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[q], b[q], acc[t0 + 1], 0, 0, 0);
      }
      a0 = na0; a1 = na1; b = nb;
    }
    asm volatile("" ::: "memory");
  }
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int j = 0; j < 16; ++j) s += acc[t][j];
  if (s == 12345.678f) out[0] = s;
}

// Step-by-step rebuild of the NT GEMM K loop: one 128x64 tile per workgroup, K = 32 * ktiles,
// A [tiles*128][K] and B [64][K] row-major, no epilogue (the sums only keep the loop alive).
//   WRITES: the register -> LDS stage with its two barriers per K tile
//   LOADS : the global loads of the next K tile issued before the MFMAs (else registers are reused)
template <bool WRITES, bool LOADS, bool SHARED = false>
__global__ __launch_bounds__(256) void gemm_loop_kernel(const float *A, const float *B, int ktiles, float *out) {
  constexpr int LD = 36;
  __shared__ __attribute__((aligned(16))) float smem[192 * LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l8 = tid & 7, lr = tid >> 3;
  const long long K = 32LL * ktiles;
  const float *ga = A + (static_cast<long long>(SHARED ? blockIdx.x & 7 : blockIdx.x) * 128 + lr) * K + l8 * 4;
  const float *gb = B + lr * K + l8 * 4;
  float *wa = smem + lr * LD + l8 * 4;
  for (int i = tid; i < 192 * LD; i += 256) smem[i] = 1.f;
  f32x4 ra[4], rb[2];
#pragma unroll
  for (int p = 0; p < 4; ++p) ra[p] = *reinterpret_cast<const f32x4 *>(ga + p * 32 * K);
#pragma unroll
  for (int p = 0; p < 2; ++p) rb[p] = *reinterpret_cast<const f32x4 *>(gb + p * 32 * K);
  const float *pa = smem + ((wave >> 1) * 64 + (lane & 31)) * LD + (lane >> 5) * 4;
  const float *pb = smem + (128 + (wave & 1) * 32 + (lane & 31)) * LD + (lane >> 5) * 4;
  f32x16 acc0, acc1;
#pragma unroll
  for (int j = 0; j < 16; ++j) { acc0[j] = 0.f; acc1[j] = 0.f; }
  __syncthreads();
  for (int kt = 0; kt < ktiles; ++kt) {
    if (WRITES) {
      __syncthreads();
#pragma unroll
      for (int p = 0; p < 4; ++p) *reinterpret_cast<f32x4 *>(wa + p * 32 * LD) = ra[p];
#pragma unroll
      for (int p = 0; p < 2; ++p) *reinterpret_cast<f32x4 *>(wa + (128 + p * 32) * LD) = rb[p];
      __syncthreads();
    }
    if (LOADS && kt + 1 < ktiles) {
#pragma unroll
      for (int p = 0; p < 4; ++p) ra[p] = *reinterpret_cast<const f32x4 *>(ga + p * 32 * K + (kt + 1) * 32);
#pragma unroll
      for (int p = 0; p < 2; ++p) rb[p] = *reinterpret_cast<const f32x4 *>(gb + p * 32 * K + (kt + 1) * 32);
    }
#pragma unroll
    for (int ss = 0; ss < 4; ++ss) {
      const f32x4 a0 = *reinterpret_cast<const f32x4 *>(pa + ss * 8);
      const f32x4 a1 = *reinterpret_cast<const f32x4 *>(pa + 32 * LD + ss * 8);
      const f32x4 b = *reinterpret_cast<const f32x4 *>(pb + ss * 8);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[q], b[q], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[q], b[q], acc1, 0, 0, 0);
      }
    }
    asm volatile("" ::: "memory");
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) s += acc0[j] + acc1[j];
  if (s == 12345.678f) out[0] = s;
}

// The full loop with the global loads issued TWO K tiles ahead (two register sets, loop unrolled by 2).
__global__ __launch_bounds__(256) void gemm_loop_ahead2_kernel(const float *A, const float *B, int ktiles, float *out) {
  constexpr int LD = 36;
  __shared__ __attribute__((aligned(16))) float smem[192 * LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l8 = tid & 7, lr = tid >> 3;
  const long long K = 32LL * ktiles;
  const float *ga = A + (static_cast<long long>(blockIdx.x) * 128 + lr) * K + l8 * 4;
  const float *gb = B + lr * K + l8 * 4;
  float *wa = smem + lr * LD + l8 * 4;
  f32x4 xa[4], xb[2], ya[4], yb[2];
  auto load = [&](f32x4 (&ra)[4], f32x4 (&rb)[2], int kt) {
    const int k = (kt < ktiles ? kt : ktiles - 1) * 32;  // clamped: the tail reloads the last tile
#pragma unroll
    for (int p = 0; p < 4; ++p) ra[p] = *reinterpret_cast<const f32x4 *>(ga + p * 32 * K + k);
#pragma unroll
    for (int p = 0; p < 2; ++p) rb[p] = *reinterpret_cast<const f32x4 *>(gb + p * 32 * K + k);
  };
  load(xa, xb, 0);
  load(ya, yb, 1);
  const float *pa = smem + ((wave >> 1) * 64 + (lane & 31)) * LD + (lane >> 5) * 4;
  const float *pb = smem + (128 + (wave & 1) * 32 + (lane & 31)) * LD + (lane >> 5) * 4;
  f32x16 acc0, acc1;
#pragma unroll
  for (int j = 0; j < 16; ++j) { acc0[j] = 0.f; acc1[j] = 0.f; }
  auto step = [&](f32x4 (&ra)[4], f32x4 (&rb)[2], int kt) {
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 4; ++p) *reinterpret_cast<f32x4 *>(wa + p * 32 * LD) = ra[p];
#pragma unroll
    for (int p = 0; p < 2; ++p) *reinterpret_cast<f32x4 *>(wa + (128 + p * 32) * LD) = rb[p];
    __syncthreads();
    load(ra, rb, kt + 2);
#pragma unroll
    for (int ss = 0; ss < 4; ++ss) {
      const f32x4 a0 = *reinterpret_cast<const f32x4 *>(pa + ss * 8);
      const f32x4 a1 = *reinterpret_cast<const f32x4 *>(pa + 32 * LD + ss * 8);
      const f32x4 b = *reinterpret_cast<const f32x4 *>(pb + ss * 8);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[q], b[q], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[q], b[q], acc1, 0, 0, 0);
      }
    }
  };
  for (int kt = 0; kt < ktiles; kt += 2) {
    step(xa, xb, kt);
    if (kt + 1 < ktiles) step(ya, yb, kt + 1);
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) s += acc0[j] + acc1[j];
  if (s == 12345.678f) out[0] = s;
}

// Double-buffered LDS, ONE barrier per K tile: the registers holding tile kt+1 are written to the
// other buffer before the MFMAs of tile kt, then the loads of tile kt+2 are issued.
template <int BK>
__global__ __launch_bounds__(256) void gemm_loop_db_kernel(const float *A, const float *B, int ksteps, float *out) {
  constexpr int LD = BK + 4, TPR = BK / 4, RPP = 256 / TPR, AP = 128 / RPP, BP = 64 / RPP, ST = 192 * LD;
  __shared__ __attribute__((aligned(16))) float smem[2 * ST];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lk = tid % TPR, lr = tid / TPR;
  const long long K = static_cast<long long>(BK) * ksteps;
  const float *ga = A + (static_cast<long long>(blockIdx.x) * 128 + lr) * K + lk * 4;
  const float *gb = B + lr * K + lk * 4;
  float *wa = smem + lr * LD + lk * 4;
  f32x4 ra[AP], rb[BP];
  auto load = [&](int kt) {
    const int k = (kt < ksteps ? kt : ksteps - 1) * BK;
#pragma unroll
    for (int p = 0; p < AP; ++p) ra[p] = *reinterpret_cast<const f32x4 *>(ga + p * RPP * K + k);
#pragma unroll
    for (int p = 0; p < BP; ++p) rb[p] = *reinterpret_cast<const f32x4 *>(gb + p * RPP * K + k);
  };
  auto store = [&](int stage) {
#pragma unroll
    for (int p = 0; p < AP; ++p) *reinterpret_cast<f32x4 *>(wa + stage * ST + p * RPP * LD) = ra[p];
#pragma unroll
    for (int p = 0; p < BP; ++p) *reinterpret_cast<f32x4 *>(wa + stage * ST + (128 + p * RPP) * LD) = rb[p];
  };
  load(0);
  store(0);
  load(1);
  const float *pa = smem + ((wave >> 1) * 64 + (lane & 31)) * LD + (lane >> 5) * 4;
  const float *pb = smem + (128 + (wave & 1) * 32 + (lane & 31)) * LD + (lane >> 5) * 4;
  f32x16 acc0, acc1;
#pragma unroll
  for (int j = 0; j < 16; ++j) { acc0[j] = 0.f; acc1[j] = 0.f; }
  __syncthreads();
  for (int kt = 0; kt < ksteps; ++kt) {
    const int stage = kt & 1;
    store(stage ^ 1);  // tile kt+1 (its readers finished before the barrier that ended step kt-1)
    load(kt + 2);
#pragma unroll
    for (int ss = 0; ss < BK / 8; ++ss) {
      const f32x4 a0 = *reinterpret_cast<const f32x4 *>(pa + stage * ST + ss * 8);
      const f32x4 a1 = *reinterpret_cast<const f32x4 *>(pa + stage * ST + 32 * LD + ss * 8);
      const f32x4 b = *reinterpret_cast<const f32x4 *>(pb + stage * ST + ss * 8);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[q], b[q], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[q], b[q], acc1, 0, 0, 0);
      }
    }
    __syncthreads();
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) s += acc0[j] + acc1[j];
  if (s == 12345.678f) out[0] = s;
}

// LDS fed by global_load_lds_dwordx4 (no staging registers, no ds_write): three stages of
// (BM + 64) rows x 128 B, unpadded with the 16-byte chunks XOR-swizzled by (row & 7) on the SOURCE
// address; two K tiles in flight, ONE raw s_barrier per tile behind a counted s_waitcnt vmcnt.
template <int BM>
__device__ __forceinline__ void gemm_loop_glds_body(const float *A, const float *B, int ktiles, float *out) {
  constexpr int NW = BM / 32, ROWS = BM + 64, ST = ROWS * 32, PER_WAVE = ROWS / 8 / NW;
  static_assert(ROWS / 8 % NW == 0, "rows must split evenly over the waves");
  __shared__ __attribute__((aligned(16))) float smem[3 * ST];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long K = 32LL * ktiles;
  // this lane's source rows: instruction j of this wave fills stage rows (j*NW + wave)*8 .. +8
  const float *src[PER_WAVE];
#pragma unroll
  for (int j = 0; j < PER_WAVE; ++j) {
    const int r = (j * NW + wave) * 8 + (lane >> 3);
    const int logical = (lane & 7) ^ (r & 7);
    src[j] = (r < BM ? A + (static_cast<long long>(blockIdx.x) * BM + r) * K : B + static_cast<long long>(r - BM) * K) +
             4 * logical;
  }
#define DX_ISSUE(kt_)                                                                                   \
  do {                                                                                                  \
    float *dst_ = smem + ((kt_) % 3) * ST;                                                              \
    _Pragma("unroll") for (int j = 0; j < PER_WAVE; ++j)                                                \
        __builtin_amdgcn_global_load_lds(src[j] + 32 * (kt_), dst_ + (j * NW + wave) * 8 * 32, 16, 0, 0); \
  } while (0)
  DX_ISSUE(0);
  if (ktiles > 1) DX_ISSUE(1);
  const int wm0 = (wave >> 1) * 64, wn0 = BM + (wave & 1) * 32, row = lane & 31, kh = lane >> 5;
  f32x16 acc0, acc1;
#pragma unroll
  for (int j = 0; j < 16; ++j) { acc0[j] = 0.f; acc1[j] = 0.f; }
  for (int kt = 0; kt < ktiles; ++kt) {
    if (kt + 1 < ktiles) {
      if (PER_WAVE == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();  // tile kt has landed for every wave; stage (kt+2)%3 is no longer read
    if (kt + 2 < ktiles) DX_ISSUE(kt + 2);
    const float *st = smem + (kt % 3) * ST;
#pragma unroll
    for (int ss = 0; ss < 4; ++ss) {
      const int ch = ((2 * ss + kh) ^ (row & 7)) * 4;
      const f32x4 a0 = *reinterpret_cast<const f32x4 *>(st + (wm0 + row) * 32 + ch);
      const f32x4 a1 = *reinterpret_cast<const f32x4 *>(st + (wm0 + 32 + row) * 32 + ch);
      const f32x4 b = *reinterpret_cast<const f32x4 *>(st + (wn0 + row) * 32 + ch);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[q], b[q], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[q], b[q], acc1, 0, 0, 0);
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) s += acc0[j] + acc1[j];
  if (s == 12345.678f) out[0] = s;
#undef DX_ISSUE
}
__global__ __launch_bounds__(256) void gemm_loop_glds128_kernel(const float *A, const float *B, int ktiles, float *out) {
  gemm_loop_glds_body<128>(A, B, ktiles, out);
}
__global__ __launch_bounds__(512) void gemm_loop_glds256_kernel(const float *A, const float *B, int ktiles, float *out) {
  gemm_loop_glds_body<256>(A, B, ktiles, out);
}

}  // namespace

// tiles x (128 x 64 x 32*ktiles) GEMM tiles; what: 0 = LDS reads + MFMA only, 1 = + LDS writes and
// barriers, 2 = + global loads (the full K loop), 3 = the same with the loads two K tiles ahead.  A must hold tiles*128*32*ktiles floats, B 64*32*ktiles.
extern "C" int dx_diag_gemm_loop_f32(const float *A, const float *B, int tiles, int ktiles, int what, float *out,
                                     void *stream) {
  DX_REQUIRE(A && B && out && tiles >= 1 && ktiles >= 1 && what >= 0 && what <= 8, "dx_diag_gemm_loop_f32: bad argument");
  hipStream_t s = dx::as_stream(stream);
  if (what == 0) hipLaunchKernelGGL((gemm_loop_kernel<false, false>), dim3(tiles), dim3(256), 0, s, A, B, ktiles, out);
  if (what == 1) hipLaunchKernelGGL((gemm_loop_kernel<true, false>), dim3(tiles), dim3(256), 0, s, A, B, ktiles, out);
  if (what == 2) hipLaunchKernelGGL((gemm_loop_kernel<true, true>), dim3(tiles), dim3(256), 0, s, A, B, ktiles, out);
  if (what == 4)  // the full loop, every workgroup re-reading one of 8 A tiles: no HBM traffic
    hipLaunchKernelGGL((gemm_loop_kernel<true, true, true>), dim3(tiles), dim3(256), 0, s, A, B, ktiles, out);
  if (what == 5) hipLaunchKernelGGL(gemm_loop_db_kernel<32>, dim3(tiles), dim3(256), 0, s, A, B, ktiles, out);
  if (what == 6) hipLaunchKernelGGL(gemm_loop_db_kernel<16>, dim3(tiles), dim3(256), 0, s, A, B, 2 * ktiles, out);
  if (what == 7) hipLaunchKernelGGL(gemm_loop_glds128_kernel, dim3(tiles), dim3(256), 0, s, A, B, ktiles, out);
  if (what == 8) hipLaunchKernelGGL(gemm_loop_glds256_kernel, dim3(tiles / 2), dim3(512), 0, s, A, B, ktiles, out);
  if (what == 3) hipLaunchKernelGGL(gemm_loop_ahead2_kernel, dim3(tiles), dim3(256), 0, s, A, B, ktiles, out);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

// LDS-fed variant: blocks x 4 waves x iters K tiles of 32 MFMAs (4096 flop each).  mode 0 = reads
// right before use, 1 = one sub-step ahead.
extern "C" int dx_diag_lds_mfma_f32(int blocks, int iters, int mode, float *out, void *stream) {
  DX_REQUIRE(blocks >= 1 && iters != 0 && out && mode >= 0 && mode <= 3, "dx_diag_lds_mfma_f32: bad argument");
  if (mode == 2)
    hipLaunchKernelGGL(lds_mfma8_loop_kernel<2>, dim3(blocks), dim3(256), 0, dx::as_stream(stream), iters, out);
  else if (mode == 3)
    hipLaunchKernelGGL(lds_mfma8_loop_kernel<3>, dim3(blocks), dim3(256), 0, dx::as_stream(stream), iters, out);
  else if (mode == 0)
    hipLaunchKernelGGL(lds_mfma_loop_kernel<0>, dim3(blocks), dim3(256), 0, dx::as_stream(stream), iters, out);
  else
    hipLaunchKernelGGL(lds_mfma_loop_kernel<1>, dim3(blocks), dim3(256), 0, dx::as_stream(stream), iters, out);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

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
