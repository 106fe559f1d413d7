// The FACTORED tail of the Nature-CNN actor-critic: linear layer + both heads as ONE affine map.
//
// derl's NatureCNNBase ends in `linear(3136 -> 512)` with NO activation behind it (derl/models.py:112-115:
// the ReLUs follow the three convolutions only), and NatureCNNModel applies its output layers straight
// to that (derl/models.py:198-203).  For output_units = [A, 1] the network's last three matrices are
// therefore one affine map of the flattened conv output y2:
//     out = Wh (Wfc y2 + bfc) + bh = Wc y2 + beff,   Wc = Wh Wfc  ((A + 1) x 3136),  beff = Wh bfc + bh,
// and the reverse-mode gradients derl's autograd forms (derl/alg/common.py:70) re-associate the same way:
//     dL/dWfc = dhid^T y2 = Wh^T (dout^T y2) = Wh^T G        G = dout^T y2  ((A + 1) x 3136)
//     dL/dWh  = dout^T hid = G Wfc^T + s bfc^T               s = sum_b dout_b
//     dL/dbfc = Wh^T s,   dL/dbh = s,   dL/dy2 = dhid Wfc = dout Wc.
// These are the SAME function and the SAME gradients (every parameter of the reference's state_dict gets
// its full gradient; nothing is approximated), in the association that costs (A + 1) x 3136 multiplies
// per sample where the layer-by-layer association costs 512 x 3136: the three largest-K GEMM stages of an
// update (linear forward, weight gradient, data gradient: 680 us of a 3.0 ms update at minibatch 8192)
// become two passes over y2 that are bound by HBM.  Rounding differs from the layer-by-layer order at
// the fp32 level only (both orders against float64 on random data of the layer's shapes: 3-6e-7 of the largest element either way).
//
//   tail_pack_kernel Wc (NHWC column order of y2) and beff from the canonical parameters, every update
//   tail_loss_kernel (heads.hip) out, loss, dout
//   tail_bwd_kernel  dy2 = relu'(y2) * (dout Wc), G partials, s partials: one pass over y2
//   tail_greduce / tail_grads  G -> dWfc, dWh, dbfc, dbh straight into the flat gradient buffer
#include "pack_direct_dev.hpp"
#include "tail_greduce_dev.hpp"

namespace dx {
namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kK = 3136, kNH = 512, kP = 49;  // flat width (49 pixels x 64 channels), hidden width, pixels
// The A + 1 outputs are padded to Jp = 8, 16 or 24 rows (groups of eight: up to 18 actions + the value -- the full Atari
// action set, derl/env/make_env.py:94-106); every per-output array below has Jp rows, rows beyond A + 1 are zero.
constexpr int kJ = 8, kJMax = 24, kMaxOutputs = 19;
static_assert(kMaxOutputs <= kJMax, "the padded row counts are 8, 16, 24");
constexpr int kChunks = 8;                                       // n chunks of the Wc product
constexpr int kBwdThreads = 448, kBwdCols = kK / kBwdThreads;    // 7 columns per thread

struct TailWeights {
  const float *Wfc, *bfc;  // canonical [512][c * 49 + p], [512]
  const float *Wp, *bp;    // policy head [A][512], [A]
  const float *Wv, *bv;    // value head [1][512], [1]
  int A;
};

__device__ __forceinline__ float head_weight(const TailWeights &w, int j, int n) {
  return j < w.A ? w.Wp[j * kNH + n] : (j == w.A ? w.Wv[n] : 0.f);
}

// Wc = Wh Wfc (A + 1 rows of 3136) and beff = Wh bfc + bh in ONE launch (round 5; two until then: eight chunk partials
// through global memory and a second launch to add them).  Workgroup = 32 columns of Wfc (its canonical order,
// kc = c * 49 + p) x 8 output rows (blockIdx.y); thread (col, ng, chunk) forms the product over 16 hidden units of chunk
// `chunk`, the four ng of a chunk meet pairwise in a fixed order, then the eight chunks in order: the same sums in the
// same order as the two launches formed (bit-identical Wc).  The result goes out in y2's column order (k = p * 64 + c)
// and, optionally, in the fragment order of the rollout's conv-stack kernel (convstack.hip:
// [wave = 4 (p / 32) + c / 16][tile (p % 32) / 16][row j of Jp][lane = 16 ((c % 16) / 4) + p % 16][c % 4]; the rows of
// pixels 49 .. 63 are never written: `packed` is zero-filled once by its owner).
// Workgroups beyond the product's: one for beff, then the direct pack's pieces (pack_direct_dev.hpp) when `direct`
// has planes to write -- between two updates of an epoch every mirror the next minibatch reads comes from this launch.
constexpr int kPackCols = 32, kPackBlocks = kK / kPackCols, kPackThreads = 1024;
static_assert(kK % kPackCols == 0 && kPackThreads == kPackCols * 4 * kChunks && kNH == kChunks * 64, "thread = (column, 16 hidden units of a chunk)");

__global__ __launch_bounds__(kPackThreads) void tail_pack_kernel(const TailWeights w, float *Wc, float *beff, float *Wcf, int Jp,
                                                                const PackDirectArgs direct) {
  const int t = threadIdx.x;
  if (blockIdx.x > kPackBlocks) {  // the conv layers' bf16 planes
    if (blockIdx.y == 0) pack_direct_piece(direct, (blockIdx.x - kPackBlocks - 1) * kPackThreads + t);
    return;
  }
  const int j0 = 8 * blockIdx.y;
  if (blockIdx.x == kPackBlocks) {  // beff[j] = Wh[j] . bfc + bh[j]: 256 threads, the two-launch version's order
    __shared__ float bred[4][kJ];
    const bool worker = t < 256;  // (all 1,024 threads stay for the barrier; the first 256 do the work)
    float part[kJ];
#pragma unroll
    for (int j = 0; j < kJ; ++j) part[j] = 0.f;
    for (int n = t; worker && n < kNH; n += 256) {
      const float b = w.bfc[n];
#pragma unroll
      for (int j = 0; j < kJ; ++j) part[j] = fmaf(head_weight(w, j0 + j, n), b, part[j]);
    }
#pragma unroll
    for (int j = 0; j < kJ; ++j) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) part[j] += __shfl_xor(part[j], o);
      if (worker && (t & 63) == 0) bred[t >> 6][j] = part[j];
    }
    __syncthreads();
    if (t < kJ) {
      const int j = j0 + t;
      const float bh = j < w.A ? w.bp[j] : (j == w.A ? w.bv[0] : 0.f);
      beff[j] = (((bred[0][t] + bred[1][t]) + bred[2][t]) + bred[3][t]) + bh;
    }
    return;
  }
  __shared__ float sWh[kJ][kNH];                  // 16 KB
  __shared__ float red[kChunks][4][kJ][kPackCols];  // 32 KB
  const int col = t & (kPackCols - 1), ng = (t >> 5) & 3, chunk = t >> 7;
  const int kc = blockIdx.x * kPackCols + col;
  float v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = w.Wfc[static_cast<long long>(chunk * 64 + ng * 16 + i) * kK + kc];
  for (int i = t; i < kJ * kNH; i += kPackThreads) sWh[i >> 9][i & (kNH - 1)] = head_weight(w, j0 + (i >> 9), i & (kNH - 1));
  __syncthreads();
  float acc[kJ];
#pragma unroll
  for (int j = 0; j < kJ; ++j) acc[j] = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int j = 0; j < kJ; ++j) acc[j] = fmaf(sWh[j][chunk * 64 + ng * 16 + i], v[i], acc[j]);
#pragma unroll
  for (int j = 0; j < kJ; ++j) red[chunk][ng][j][col] = acc[j];
  __syncthreads();
  if (t < kJ * kPackCols) {
    const int j = t >> 5, c32 = t & (kPackCols - 1);
    float tot = 0.f;
#pragma unroll
    for (int ch = 0; ch < kChunks; ++ch)
      tot += ((red[ch][0][j][c32] + red[ch][1][j][c32]) + red[ch][2][j][c32]) + red[ch][3][j][c32];
    const int kcc = blockIdx.x * kPackCols + c32, c = kcc / kP, p = kcc - c * kP;
    Wc[(j0 + j) * kK + p * 64 + c] = tot;
    if (Wcf) {
      const int wave = 4 * (p >> 5) + (c >> 4), m = (p >> 4) & 1, lane = 16 * ((c >> 2) & 3) + (p & 15);
      Wcf[((wave * 2 + m) * Jp + j0 + j) * 256 + lane * 4 + (c & 3)] = tot;
    }
  }
}

struct TailBwdArgs {
  const float *y2;     // [B][3136]
  const float *dhead;  // [B][32]: dL/dout in columns 0..A
  const float *Wc;     // [Jp][3136]
  float *dy2;          // [B][3136]
  float *gslab;        // [gridDim.x][Jp][3136] partial G
  float *sslab;        // [gridDim.x][Jp] partial s
  int B, rows_per_wg, Jp;
};

// One pass over y2: dy2 = relu'(y2) * (dout Wc) and this workgroup's partial G = dout^T y2, s = sum dout.
// A thread owns 7 columns (t, t + 448, ...: coalesced dword accesses) with their Wc and G entries in
// registers; the A + 1 gradient values of a row are the same for every thread.
template <int NJ>
__global__ __launch_bounds__(kBwdThreads) void tail_bwd_kernel(const TailBwdArgs a) {
  const int t = threadIdx.x;
  float wc[NJ][kBwdCols], g[NJ][kBwdCols], sacc[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    sacc[j] = 0.f;
#pragma unroll
    for (int i = 0; i < kBwdCols; ++i) {
      wc[j][i] = a.Wc[j * kK + t + kBwdThreads * i];
      g[j][i] = 0.f;
    }
  }
  const int r0 = blockIdx.x * a.rows_per_wg, r1 = min(a.B, r0 + a.rows_per_wg);
  constexpr int kRows = 4;  // rows in flight
  for (int r = r0; r < r1; r += kRows) {
    float y[kRows][kBwdCols], d[kRows][NJ];
#pragma unroll
    for (int u = 0; u < kRows; ++u) {
      const long long row = min(r + u, a.B - 1);
#pragma unroll
      for (int i = 0; i < kBwdCols; ++i) y[u][i] = a.y2[row * kK + t + kBwdThreads * i];
#pragma unroll
      for (int j = 0; j < NJ; ++j) d[u][j] = a.dhead[row * 32 + j];  // the same address in every lane
    }
#pragma unroll
    for (int u = 0; u < kRows; ++u) {
      if (r + u >= r1) break;  // uniform
      float *out = a.dy2 + static_cast<long long>(r + u) * kK + t;
#pragma unroll
      for (int i = 0; i < kBwdCols; ++i) {
        float dsum = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          dsum = fmaf(d[u][j], wc[j][i], dsum);
          g[j][i] = fmaf(d[u][j], y[u][i], g[j][i]);
        }
        out[kBwdThreads * i] = y[u][i] > 0.f ? dsum : 0.f;
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j) sacc[j] += d[u][j];
    }
  }
  float *slab = a.gslab + static_cast<long long>(blockIdx.x) * a.Jp * kK;
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int i = 0; i < kBwdCols; ++i) slab[j * kK + t + kBwdThreads * i] = g[j][i];
  if (t == 0) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) a.sslab[blockIdx.x * a.Jp + j] = sacc[j];
  }
}

// The same pass for 9 .. 19 outputs (up to 18 actions + the value): 2 NJ registers per column no longer fit seven
// columns per thread, so a workgroup takes HALF of a row (blockIdx.y; 392 threads x one float4 = 1,568 columns) and two
// rows in flight; the outputs' gradients of a row are uniform (scalar loads), y2 and dy2 move as 16-byte accesses.
constexpr int kWideThreads = 392;  // x 4 columns x 2 halves = 3136
template <int NJ>
__global__ __launch_bounds__(448) void tail_bwd_wide_kernel(const TailBwdArgs a) {
  const int t = threadIdx.x;
  if (t >= kWideThreads) return;
  const int q = kWideThreads * blockIdx.y + t;  // float4 index within a row
  f32x4 wc[NJ], g[NJ];
  float sacc[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    sacc[j] = 0.f;
    wc[j] = reinterpret_cast<const f32x4 *>(a.Wc + j * kK)[q];
    g[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int r0 = blockIdx.x * a.rows_per_wg, r1 = min(a.B, r0 + a.rows_per_wg);
  constexpr int kRows = 2;  // rows in flight
  for (int r = r0; r < r1; r += kRows) {
    f32x4 y[kRows];
    float d[kRows][NJ];
#pragma unroll
    for (int u = 0; u < kRows; ++u) {
      const long long row = min(r + u, a.B - 1);
      y[u] = reinterpret_cast<const f32x4 *>(a.y2 + row * kK)[q];
#pragma unroll
      for (int j = 0; j < NJ; ++j) d[u][j] = a.dhead[row * 32 + j];  // the same address in every lane
    }
#pragma unroll
    for (int u = 0; u < kRows; ++u) {
      if (r + u >= r1) break;  // uniform
      f32x4 dsum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          dsum[i] = fmaf(d[u][j], wc[j][i], dsum[i]);
          g[j][i] = fmaf(d[u][j], y[u][i], g[j][i]);
        }
#pragma unroll
      for (int i = 0; i < 4; ++i) dsum[i] = y[u][i] > 0.f ? dsum[i] : 0.f;
      reinterpret_cast<f32x4 *>(a.dy2 + static_cast<long long>(r + u) * kK)[q] = dsum;
#pragma unroll
      for (int j = 0; j < NJ; ++j) sacc[j] += d[u][j];
    }
  }
  float *slab = a.gslab + static_cast<long long>(blockIdx.x) * a.Jp * kK;
#pragma unroll
  for (int j = 0; j < NJ; ++j) reinterpret_cast<f32x4 *>(slab + j * kK)[q] = g[j];
  if (t == 0 && blockIdx.y == 0) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) a.sslab[blockIdx.x * a.Jp + j] = sacc[j];
  }
}

__global__ __launch_bounds__(256) void tail_greduce_kernel(const TailGreduceArgs a) {
  __shared__ float red[4][64];
  __shared__ float sred[256];
  tail_greduce_block(a, blockIdx.x, blockIdx.y, red, sred);
}

struct TailGradArgs {
  TailWeights w;
  const float *Gc;  // [Jp][3136] canonical column order
  const float *s;   // [Jp]
  float *dWfc, *dbfc, *dWp, *dbp, *dWv, *dbv;  // views of the flat gradient buffer (canonical layout)
  int nj;           // A + 1
};

constexpr int kGradBlocksW = kNH * (kK / 4) / 256;  // 1568: dWfc, one float4 per thread

// dWfc = Wh^T G | dWh = G Wfc^T + s bfc^T | dbfc = Wh^T s, dbh = s -- three roles in one grid
template <int Jp>  // 8, 16 or 24 padded output rows: compile-time so that the common case (up to 7 actions) keeps eight registers per array
__global__ __launch_bounds__(256) void tail_grads_kernel(const TailGradArgs a) {
  const int t = threadIdx.x, blk = blockIdx.x;
  if (blk < kGradBlocksW) {
    const int idx = blk * 256 + t, n = idx / (kK / 4), q = idx - n * (kK / 4);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    // every row's loads in flight at once (a loop over the A + 1 real rows waited for each row's pair of loads in
    // turn); rows beyond A + 1 are zero in Gc (tail_greduce_kernel) and in Wh: they add +0
    float wh[Jp];
    float4 g4[Jp];
#pragma unroll
    for (int j = 0; j < Jp; ++j) {
      wh[j] = head_weight(a.w, j, n);
      g4[j] = reinterpret_cast<const float4 *>(a.Gc + j * kK)[q];
    }
#pragma unroll
    for (int j = 0; j < Jp; ++j) {
      acc.x = fmaf(wh[j], g4[j].x, acc.x); acc.y = fmaf(wh[j], g4[j].y, acc.y);
      acc.z = fmaf(wh[j], g4[j].z, acc.z); acc.w = fmaf(wh[j], g4[j].w, acc.w);
    }
    reinterpret_cast<float4 *>(a.dWfc + static_cast<long long>(n) * kK)[q] = acc;
    return;
  }
  if (blk < kGradBlocksW + kNH) {
    __shared__ float red[4][Jp];
    const int n = blk - kGradBlocksW;
    float part[Jp];
#pragma unroll
    for (int j = 0; j < Jp; ++j) part[j] = 0.f;
    // the row's 13 column steps with all their loads in flight (one step's loads waited for before the next step's
    // were issued: 13 dependent round trips, 10 us of a 14.8 us launch at any batch)
    constexpr int kSteps = (kK + 255) / 256;
    float wfc[kSteps];
#pragma unroll
    for (int i = 0; i < kSteps; ++i) wfc[i] = t + 256 * i < kK ? a.w.Wfc[static_cast<long long>(n) * kK + t + 256 * i] : 0.f;
#pragma unroll
    for (int i0 = 0; i0 < kSteps; i0 += 4) {  // four steps' Gc rows at a time: 4 Jp loads in flight
      float gc[4][Jp];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < Jp; ++j) {
          const int kc = t + 256 * (i0 + u);
          gc[u][j] = (i0 + u < kSteps && kc < kK) ? a.Gc[j * kK + kc] : 0.f;  // rows >= A + 1 of Gc are zero
        }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (i0 + u < kSteps) {
#pragma unroll
          for (int j = 0; j < Jp; ++j) part[j] = fmaf(gc[u][j], wfc[i0 + u], part[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < Jp; ++j) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) part[j] += __shfl_xor(part[j], o);
      if ((t & 63) == 0) red[t >> 6][j] = part[j];
    }
    __syncthreads();
    if (t < a.nj) {
      const float v = (((red[0][t] + red[1][t]) + red[2][t]) + red[3][t]) + a.s[t] * a.w.bfc[n];
      if (t < a.w.A) a.dWp[t * kNH + n] = v;
      else a.dWv[n] = v;
    }
    return;
  }
  for (int n = t; n < kNH; n += 256) {
    float v = 0.f;
    for (int j = 0; j < a.nj; ++j) v = fmaf(head_weight(a.w, j, n), a.s[j], v);
    a.dbfc[n] = v;
  }
  if (t < a.w.A) a.dbp[t] = a.s[t];
  if (t == a.w.A) a.dbv[0] = a.s[t];
}

template <int NJ>
int launch_tail_bwd_as(const TailBwdArgs &a, int nwg, hipStream_t stream) {
  if constexpr (NJ <= 8) hipLaunchKernelGGL(tail_bwd_kernel<NJ>, dim3(nwg), dim3(kBwdThreads), 0, stream, a);
  else hipLaunchKernelGGL(tail_bwd_wide_kernel<NJ>, dim3(nwg, 2), dim3(448), 0, stream, a);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

TailWeights tail_weights(const float *params, const long long *off_w, const long long *off_b, int A) {
  return TailWeights{params + off_w[3], params + off_b[3], params + off_w[4], params + off_b[4],
                     params + off_w[5], params + off_b[5], A};
}

}  // namespace

bool tail_supported(int flat, int num_actions) { return flat == kK && num_actions >= 1 && num_actions + 1 <= kMaxOutputs; }
int tail_rows(int num_actions) { return 8 * cdiv(num_actions + 1, 8); }  // Jp: the outputs padded to groups of eight

// floats of scratch: Wc product partials (inside `packed`), and G slabs + Gc + s (inside `slabs`)
long long tail_pack_scratch_floats(int num_actions) { return static_cast<long long>(kChunks) * tail_rows(num_actions) * kK; }
int tail_bwd_workgroups(int B) {
  const int nwg = cdiv(B, 8) < 256 ? cdiv(B, 8) : 256;  // >= 8 rows per workgroup, one workgroup per CU at most
  return cdiv(B, cdiv(B, nwg));
}
long long tail_slab_floats(int B, int num_actions) {
  return (static_cast<long long>(tail_bwd_workgroups(B)) + 2) * tail_rows(num_actions) * kK + 64;
}
// >= tail_slab_floats(B) of EVERY B <= max_batch: tail_bwd_workgroups is not monotone (2100 rows -> 234 workgroups of
// 9, 2048 rows -> 256 of 8), its bound min(ceil(max_batch / 8), 256) is
long long tail_slab_capacity_floats(int max_batch, int num_actions) {
  const int nwg = cdiv(max_batch, 8) < 256 ? cdiv(max_batch, 8) : 256;
  return (static_cast<long long>(nwg) + 2) * tail_rows(num_actions) * kK + 64;
}

// Wc [Jp][3136] (y2's column order), beff [Jp] from the canonical parameters; scratch: tail_pack_scratch_floats(A)
int launch_tail_pack(const float *params, const long long *off_w, const long long *off_b, int A, float *Wc, float *beff,
                     float *scratch, float *Wcf, hipStream_t stream, const TailDirectPlanes *direct) {
  DX_REQUIRE(params && Wc && beff && scratch && A >= 1 && A + 1 <= kMaxOutputs, "tail_pack: bad arguments");
  const TailWeights w = tail_weights(params, off_w, off_b, A);
  const int Jp = tail_rows(A);
  PackDirectArgs d{};
  int extra = 0;
  if (direct) {  // the conv layers' bf16 planes ride in extra workgroups of the same launch
    DX_REQUIRE(direct->p0 && direct->f1 && direct->f2 && direct->d1 && direct->d2, "tail_pack: direct planes missing");
    DX_REQUIRE(aligned(direct->p0, 16) && aligned(direct->f1, 16) && aligned(direct->f2, 16) && aligned(direct->d1, 16) &&
                   aligned(direct->d2, 16), "tail_pack: a direct plane is not 16-byte aligned (written with 16-byte stores)");
    d = PackDirectArgs{params + off_w[0], params + off_w[1], params + off_w[2], direct->p0, direct->f1, direct->f2, direct->d1, direct->d2};
    extra = cdiv(kPackDirectPieces, kPackThreads);
  }
  (void)scratch;  // (the two-launch version's chunk partials; kept in the signature for its callers)
  hipLaunchKernelGGL(tail_pack_kernel, dim3(kPackBlocks + 1 + extra, Jp / 8), dim3(kPackThreads), 0, stream, w, Wc, beff, Wcf, Jp, d);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

// where launch_tail_bwd's partial G / s live inside `scratch` and how its rows are split (heads.hip's fused
// loss + backward launch writes the same slabs)
TailBwdPlan tail_bwd_plan(float *scratch, int B, int A) {
  const int nwg = tail_bwd_workgroups(B), Jp = tail_rows(A);
  return TailBwdPlan{scratch, scratch + static_cast<long long>(nwg) * Jp * kK, Jp, nwg, cdiv(B, nwg)};
}

// dy2 and the partial G / s of the minibatch (scratch: tail_slab_floats(B) floats); then
// launch_tail_grads turns them into the linear layer's and the heads' gradients
int launch_tail_bwd(const float *y2, const float *dhead, const float *Wc, float *dy2, float *scratch, int B, int A,
                    hipStream_t stream) {
  DX_REQUIRE(y2 && dhead && Wc && dy2 && scratch && B >= 1 && A >= 1 && A + 1 <= kMaxOutputs, "tail_bwd: bad arguments");
  const int nwg = tail_bwd_workgroups(B), Jp = tail_rows(A);
  const TailBwdArgs a{y2, dhead, Wc, dy2, scratch, scratch + static_cast<long long>(nwg) * Jp * kK, B, cdiv(B, nwg), Jp};
  switch (A + 1) {
    case 2: return launch_tail_bwd_as<2>(a, nwg, stream);
    case 3: return launch_tail_bwd_as<3>(a, nwg, stream);
    case 4: return launch_tail_bwd_as<4>(a, nwg, stream);
    case 5: return launch_tail_bwd_as<5>(a, nwg, stream);
    case 6: return launch_tail_bwd_as<6>(a, nwg, stream);
    case 7: return launch_tail_bwd_as<7>(a, nwg, stream);
    case 8: return launch_tail_bwd_as<8>(a, nwg, stream);
    case 9: return launch_tail_bwd_as<9>(a, nwg, stream);
    case 10: return launch_tail_bwd_as<10>(a, nwg, stream);
    case 11: return launch_tail_bwd_as<11>(a, nwg, stream);
    case 12: return launch_tail_bwd_as<12>(a, nwg, stream);
    case 13: return launch_tail_bwd_as<13>(a, nwg, stream);
    case 14: return launch_tail_bwd_as<14>(a, nwg, stream);
    case 15: return launch_tail_bwd_as<15>(a, nwg, stream);
    case 16: return launch_tail_bwd_as<16>(a, nwg, stream);
    case 17: return launch_tail_bwd_as<17>(a, nwg, stream);
    case 18: return launch_tail_bwd_as<18>(a, nwg, stream);
    default: return launch_tail_bwd_as<19>(a, nwg, stream);
  }
}

// where the reduction reads and writes inside `scratch` (launch_tail_bwd's layout)
TailGreduceArgs tail_greduce_args(float *scratch, int B, int A) {
  const int nwg = tail_bwd_workgroups(B), Jp = tail_rows(A);
  float *Gc = scratch + (static_cast<long long>(nwg) + 1) * Jp * kK;
  return TailGreduceArgs{scratch, scratch + static_cast<long long>(nwg) * Jp * kK, nwg, A + 1, Jp, Gc, Gc + Jp * kK};
}

// `reduced`: the G / s reduction has run already (igemm.hip: launch_permute_reduce_greduce, beside the conv layers'
// slab reduction); otherwise it is this call's first launch
int launch_tail_grads(const float *params, float *grads, const long long *off_w, const long long *off_b, int A,
                      float *scratch, int B, hipStream_t stream, bool reduced) {
  DX_REQUIRE(params && grads && scratch && B >= 1 && A >= 1 && A + 1 <= kMaxOutputs, "tail_grads: bad arguments");
  const TailGreduceArgs r = tail_greduce_args(scratch, B, A);
  const int Jp = r.Jp;
  if (!reduced) {
    hipLaunchKernelGGL(tail_greduce_kernel, dim3(kP, Jp), dim3(256), 0, stream, r);
    DX_LAUNCH_CHECK();
  }
  const TailGradArgs a{tail_weights(params, off_w, off_b, A), r.Gc, r.s, grads + off_w[3], grads + off_b[3], grads + off_w[4],
                       grads + off_b[4], grads + off_w[5], grads + off_b[5], A + 1};
  const dim3 ggrid(kGradBlocksW + kNH + 1);
  if (Jp == 8) hipLaunchKernelGGL(tail_grads_kernel<8>, ggrid, dim3(256), 0, stream, a);
  else if (Jp == 16) hipLaunchKernelGGL(tail_grads_kernel<16>, ggrid, dim3(256), 0, stream, a);
  else hipLaunchKernelGGL(tail_grads_kernel<24>, ggrid, dim3(256), 0, stream, a);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

}  // namespace dx
