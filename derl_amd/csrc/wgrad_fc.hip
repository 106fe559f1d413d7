// Weight gradient of the 3136 -> 512 linear layer on fp32 MFMA:
//
//   dW[n][k] = sum_m G[m][n] * A[m][k]     G = dL/dhid (M x 512), A = y2 flattened NHWC (M x K)
//
// (the autograd backward of derl/models.py:112-115's linear layer, derl/alg/common.py:70).  The
// generic implicit-GEMM wgrad (igemm_tn_kernel, 64 n x 128 k tiles) re-reads every A row from 8
// workgroups that sit on 8 different XCDs and every G row from 25: 0.98 GB of HBM-side fetches
// per launch for 0.12 GB of operands (profiles/r01_pmc_traffic.json), and it feeds every MFMA
// with 1.5 scalar LDS reads.  This kernel is built around two measurements: HBM bytes cost
// package power (the clock drops under 3 TB/s of traffic next to full-rate MFMA), and every
// non-MFMA instruction of a wave costs matrix-pipe time.
//
//  * A workgroup (8 waves) owns ALL 512 columns n of one 128-wide k block for one slice of the
//    rows: A is read from HBM exactly once; G (16.8 MB at M = 8192) is re-read per k block but
//    lives in L2 / the Infinity Cache.  Operand bytes per MAC drop from 0.094 to 0.039.
//  * Rows arrive by LDS-DMA (global_load_lds_dwordx4): a stage is 16 rows = 32 KiB of G + 8 KiB of
//    A, whole 1 KiB pieces of contiguous global rows, no staging registers, no ds_write; a ring
//    of three stages, ONE barrier per stage behind a counted vmcnt (two stages stay in flight).
//  * Operand reads are as wide as the natural row-major layout allows, with a PERMUTED column
//    map instead of a transposed tile: lane l of a row reads G[m][128 wn + 4 l .. + 3] with one
//    ds_read_b128 -- the A-operands of FOUR n tiles (tile j = columns 4 l + j) -- and
//    A[m][64 wk + 2 l .. + 1] with one ds_read_b64 -- the B-operands of TWO k tiles: 2 LDS reads
//    feed 8 MFMAs (v_mfma_f32_32x32x2_f32; rows m, m + 1 are the two k-lanes of the
//    instruction).  All row offsets inside a stage are immediates.
//  * Accumulators (8 tiles, 128 registers) go to the slab [slice][n][k] of the workgroup's row
//    slice with 8-byte stores; slices are summed by the deterministic finalize launch as before.
//    The bias gradient (column sums of G) is a separate small launch (colsum).
#include "igemm_dev.hpp"
#include <cstdlib>

namespace dx {
namespace {

using f4 = __attribute__((ext_vector_type(4))) float;
using f2 = __attribute__((ext_vector_type(2))) float;

constexpr int kN = 512, kBK = 128, kRows = 16, kStages = 3;
constexpr int kGFloats = kRows * kN, kAFloats = kRows * kBK, kStageFloats = kGFloats + kAFloats;  // 40 KiB
constexpr int kPiecesPerWave = (kStageFloats * 4 / 1024) / 8;  // 40 pieces over 8 waves

__global__ __launch_bounds__(512, 2) void fc_wgrad_kernel(const FcWgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  const int bk = blockIdx.x % a.gk, z = blockIdx.x / a.gk;
  const int k0 = bk * kBK;
  const int total = a.M / kRows;  // stages over all rows (M % 16 == 0 checked by the launcher)
  const int sbeg = static_cast<int>(static_cast<long long>(z) * total / a.msplit);
  const int send = static_cast<int>(static_cast<long long>(z + 1) * total / a.msplit);

  // this wave's pieces of a stage: G pieces 4w .. 4w+3 (piece p = row p / 2, columns 256 (p % 2)
  // + 4 lane), A piece w (rows 2w + hi, columns 4 l31; lanes past K stay inactive)
  // uniform base + constant 32-bit lane offset: the SGPR-base form, no address VALU (igemm_dev.hpp)
  const uint32_t glane = lane * 16u;
  const int acol = k0 + 4 * l31;
  const bool avalid = acol < a.K;
  const uint32_t alane = (static_cast<uint32_t>(hi) * a.K + (avalid ? 4 * l31 : 0)) * 4u;
#define DX_FC_ISSUE(S, SLOT)                                                                     \
  {                                                                                              \
    float *dst_ = smem + (SLOT) * kStageFloats;                                                  \
    const long long row_ = static_cast<long long>(S) * kRows + 2 * wave;                         \
    _Pragma("unroll") for (int p = 0; p < 4; ++p)                                                \
        dma_piece(a.G + (row_ + (p >> 1)) * kN + (p & 1) * 256, glane, dst_ + (4 * wave + p) * 256); \
    if (avalid) dma_piece(a.A + row_ * a.K + k0, alane, dst_ + kGFloats + wave * 256);           \
  }

  const int wn = wave & 3, wk = wave >> 2;
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // lane part of the operand addresses (bytes); the pair's rows are immediates
  const char *lds = reinterpret_cast<const char *>(smem);
  const unsigned gl = (hi * kN + 128 * wn + 4 * l31) * 4;
  const unsigned al = (kGFloats + hi * kBK + 64 * wk + 2 * l31) * 4;

  // diag bit 3: in-kernel stamps around the stage loop (shader cycles and 100 MHz ticks), written
  // INSTEAD of the result: [workgroup][wave] -> {loop cycles, loop ticks, stages, kernel-entry tick}
  const unsigned long long t_entry = (kDiag && (a.diag & 8)) ? __builtin_amdgcn_s_memrealtime() : 0;
  if (sbeg < send) DX_FC_ISSUE(sbeg, 0)
  if (sbeg + 1 < send) DX_FC_ISSUE(sbeg + 1, 1)
  const unsigned long long c0 = (kDiag && (a.diag & 8)) ? __builtin_amdgcn_s_memtime() : 0;
  const unsigned long long r0 = (kDiag && (a.diag & 8)) ? __builtin_amdgcn_s_memrealtime() : 0;
  int slot = 0;
  for (int s = sbeg; s < send; ++s) {
    // stage s has landed for this wave (the pieces of stage s + 1 may still be in flight); after
    // the barrier it has for everyone, and nobody reads slot (s + 2) % 3 = (s - 1) % 3 any more
    if (s + 1 < send) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kPiecesPerWave) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // not __syncthreads(): its fence makes the compiler wait vmcnt(0), i.e. for the stage in flight
    if (!(kDiag && (a.diag & 2))) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (s + 2 < send && !(kDiag && (a.diag & 1))) {
      const int nslot = slot >= 1 ? slot - 1 : slot + 2;
      DX_FC_ISSUE(s + 2, nslot)
    }
    const char *gp = lds + (gl + slot * kStageFloats * 4);
    const char *ap = lds + (al + slot * kStageFloats * 4);
    // two register sets: the operands of pair e + 1 are requested in the shadow of the FIRST MFMA
    // of pair e (sched_barrier pins that order).  Left alone hipcc
    // sinks the reads below the MFMAs; issued in front of them it waits with lgkmcnt(0), i.e. for
    // the reads it has just issued.
    f4 gq[2];
    f2 aq[2];
    gq[0] = *reinterpret_cast<const f4 *>(gp);
    aq[0] = *reinterpret_cast<const f2 *>(ap);
#pragma unroll
    for (int e = 0; e < kRows / 2; ++e) {
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(gq[e & 1][0], aq[e & 1][0], acc[0][0], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (e + 1 < kRows / 2) {
        gq[(e + 1) & 1] = *reinterpret_cast<const f4 *>(gp + (e + 1) * 2 * kN * 4);
        aq[(e + 1) & 1] = *reinterpret_cast<const f2 *>(ap + (e + 1) * 2 * kBK * 4);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 1; t < 8; ++t)
        acc[t >> 1][t & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(gq[e & 1][t >> 1], aq[e & 1][t & 1], acc[t >> 1][t & 1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    slot = slot == 2 ? 0 : slot + 1;
  }
#undef DX_FC_ISSUE

  // tile (i, j): MFMA row r' = (r & 3) + 8 (r >> 2) + 4 hi is column n = 128 wn + 4 r' + i,
  // MFMA column l31 is k = k0 + 64 wk + 2 l31 + j
  if (kDiag && (a.diag & 8)) {
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) {
      unsigned long long *o = reinterpret_cast<unsigned long long *>(a.slab) + (static_cast<long long>(blockIdx.x) * 8 + wave) * 4;
      o[0] = c1 - c0; o[1] = r1 - r0; o[2] = send - sbeg; o[3] = t_entry;
    }
    return;
  }
  const int kk = k0 + 64 * wk + 2 * l31;
  if (k0 + 64 * wk >= a.K) return;  // wave-uniform: the empty half of the last k block
  if (kDiag && (a.diag & 4) && acc[0][0][0] != 12345.678f) return;
  float *slab = a.slab + static_cast<long long>(z) * kN * a.K + kk;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = 128 * wn + 4 * ((r & 3) + 8 * (r >> 2) + 4 * hi) + i;
      *reinterpret_cast<f2 *>(slab + static_cast<long long>(n) * a.K) = f2{acc[i][0][r], acc[i][1][r]};
    }
}

// out[chunk][n] = sum of G[m][n] over the rows of the chunk (bias gradient partials; the finalize
// launch sums the chunks).  One workgroup per (chunk, 256 columns): thread = column quad x row
// phase, 16-byte loads, phases combined through LDS.
__global__ __launch_bounds__(256) void colsum_kernel(const float *__restrict__ G, float *__restrict__ out, int M,
                                                    int N, int rows_per_chunk) {
  __shared__ f4 part[256];
  const int chunk = blockIdx.x, cb = blockIdx.y;  // cb: block of 256 columns
  const int q = threadIdx.x & 63, phase = threadIdx.x >> 6;
  const int col = cb * 256 + 4 * q;
  const int mbeg = chunk * rows_per_chunk, mend = min(M, mbeg + rows_per_chunk);
  f4 sum = {0.f, 0.f, 0.f, 0.f};
  if (col < N)
    for (int m = mbeg + phase; m < mend; m += 4) sum += *reinterpret_cast<const f4 *>(G + static_cast<long long>(m) * N + col);
  part[threadIdx.x] = sum;
  __syncthreads();
  if (phase == 0 && col < N) {
    const f4 t = (part[q] + part[q + 64]) + (part[q + 128] + part[q + 192]);
    *reinterpret_cast<f4 *>(out + static_cast<long long>(chunk) * N + col) = t;
  }
}

}  // namespace

bool fc_wgrad_supported(int M, int N, int K) { return N == kN && K % 64 == 0 && K >= 128 && M % kRows == 0 && M >= 16 * kRows; }

// msplit row slices x ceil(K / 128) k blocks, about one workgroup per CU
int fc_wgrad_slices(int M, int K) {
  const int gk = cdiv(K, kBK);
  int ms = 256 / gk;
  const int cap = M / (4 * kRows);  // >= 4 stages per slice
  if (ms > cap) ms = cap;
  return ms < 1 ? 1 : ms;
}

int launch_fc_wgrad(const FcWgradArgs &a_in, hipStream_t stream) {
  FcWgradArgs a = a_in;
  DX_REQUIRE(a.G && a.A && a.slab && fc_wgrad_supported(a.M, kN, a.K) && a.msplit >= 1 &&
                 a.msplit <= a.M / kRows,
             "fc_wgrad: unsupported problem M=%d K=%d slices=%d", a.M, a.K, a.msplit);
  DX_REQUIRE(aligned(a.G, 16) && aligned(a.A, 16) && aligned(a.slab, 8), "fc_wgrad: misaligned pointer");
  a.gk = cdiv(a.K, kBK);
#if DX_DIAG
  a.diag = DX_ENV("DX_FC_DIAG", 0);
#else
  a.diag = 0;
#endif
  constexpr int lds = kStages * kStageFloats * 4;
  DX_LDS_OPT_IN(fc_wgrad_kernel, lds);
  hipLaunchKernelGGL(fc_wgrad_kernel, dim3(a.gk * a.msplit), dim3(512), lds, stream, a);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

int launch_colsum(const float *G, float *out, int M, int N, int chunks, hipStream_t stream) {
  DX_REQUIRE(G && out && M > 0 && N > 0 && N % 4 == 0 && chunks >= 1, "colsum: bad arguments");
  DX_REQUIRE(aligned(G, 16) && aligned(out, 16), "colsum: misaligned pointer");
  const int rows = cdiv(M, chunks);
  hipLaunchKernelGGL(colsum_kernel, dim3(chunks, cdiv(N, 256)), dim3(256), 0, stream, G, out, M, N, rows);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

}  // namespace dx
