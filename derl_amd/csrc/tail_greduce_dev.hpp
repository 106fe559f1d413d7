// The fixed-order reduction of the factored tail's partial G / s slabs (tail.hip: tail_bwd_kernel / heads.hip:
// tail_loss_bwd_kernel write them) as a device function, so that it can ride in the same launch as the conv layers'
// slab reduction (igemm.hip: permute_reduce_greduce_kernel) -- the two are independent and both bound by reading slabs.
#pragma once
#include "common.hpp"

namespace dx {

struct TailGreduceArgs {
  const float *gslab, *sslab;  // [nslab][Jp][3136], [nslab][Jp]
  int nslab, nj, Jp;           // slabs, A + 1, padded rows
  float *Gc, *s;               // [Jp][3136] in Wfc's canonical column order (c * 49 + p), [Jp]
};
constexpr int kGreduceK = 3136, kGreduceP = 49;

// block (p, j) of a (49, Jp) grid, 256 threads: G[j][k] = sum over the workgroups' partials (fixed order), in y2's column
// order and in the canonical order of Wfc's columns (k = p * 64 + c  ->  c * 49 + p); block (0, j) also sums s[j]
__device__ __forceinline__ void tail_greduce_block(const TailGreduceArgs &a, int p, int j, float (*red)[64], float *sred) {
  const int t = threadIdx.x, c = t & 63, sg = t >> 6;
  if (j >= a.nj) {  // rows beyond the A + 1 outputs read as zero in the heads' dot products (uniform branch)
    if (t < 64) a.Gc[j * kGreduceK + c * kGreduceP + p] = 0.f;
    if (p == 0 && t == 0) a.s[j] = 0.f;
    return;
  }
  float v = 0.f;
  for (int z0 = sg; z0 < a.nslab; z0 += 64) {  // sixteen loads in flight, added in slab order
    float x[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int z = z0 + 4 * u;
      x[u] = z < a.nslab ? a.gslab[(static_cast<long long>(z) * a.Jp + j) * kGreduceK + p * 64 + c] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) v += x[u];
  }
  red[sg][c] = v;
  if (p == 0) {
    float sv = 0.f;
    for (int z = t; z < a.nslab; z += 256) sv += a.sslab[z * a.Jp + j];
    sred[t] = sv;
  }
  __syncthreads();
  if (t < 64) a.Gc[j * kGreduceK + c * kGreduceP + p] = ((red[0][c] + red[1][c]) + red[2][c]) + red[3][c];
  if (p == 0 && t == 0) {
    float tot = 0.f;
    for (int i = 0; i < 256; ++i) tot += sred[i];
    a.s[j] = tot;
  }
}

}  // namespace dx
