// Vectorised observation / return normalisation of the MuJoCo env stack on the device
// (derl/env/mujoco_wrappers.py:8-124, class Normalize + RunningMeanVar).
//
// Per env step, on a batch of N envs with D-dimensional observations:
//   obs stats  : batch mean / population variance per dimension, merged into the running
//                (mean, var, count) with the parallel-variance rule (:48-61);
//   obs out    : clip((obs - mean) / sqrt(var + eps), +-clipobs) with the UPDATED stats (:99-110);
//   returns    : ret = ret * gamma + reward per env (:114); scalar running stats of `ret` (:117);
//   reward out : clip(reward / sqrt(ret_var + eps), +-cliprew) (:118-119); ret[resets] = 0 (:120).
// The reference does this in float64 NumPy on the host; the state here is float64 too (inputs
// are the float32 tensors of the device-resident env), outputs are float32.  HBM-bound and tiny:
// 16 B per observation element (read for the mean, the variance and the output; one write) +
// state; three launches: per-(row block, dimension) partial moments, normalise (every block merges
// the partials it needs, read-only), then one block commits the statistics and does the returns.
#include "common.hpp"

namespace {

constexpr int kRowGroups = 32;  // 1024 threads = 32 dims x 32 row groups
constexpr int kMaxRowBlocks = 128;

// sums `v` over the 32 row groups of a block for each of the 32 dims (fixed order)
__device__ __forceinline__ double group_sum(double v, double (*red)[32], int dim, int grp) {
  red[grp][dim] = v;
  __syncthreads();
  double s = 0.0;
#pragma unroll 8
  for (int g = 0; g < kRowGroups; ++g) s += red[g][dim];
  __syncthreads();
  return s;
}

struct Moments {
  double mean, var;
};

// batch statistics of N rows from the per-row-block partials {mean, M2} (Chan et al. merge in
// row-block order), then the running-statistics update of mujoco_wrappers.py:48-61
__device__ __forceinline__ Moments merged_stats(const double *partial, int d, int D, int N, int rows_per_block,
                                                int nblocks, double mean, double var, double count) {
  double bmean = 0.0, m2 = 0.0, n = 0.0;
  for (int b = 0; b < nblocks; ++b) {
    const int rows = min(rows_per_block, N - b * rows_per_block);
    const double pm = partial[(static_cast<long long>(b) * D + d) * 2];
    const double pm2 = partial[(static_cast<long long>(b) * D + d) * 2 + 1];
    const double delta = pm - bmean, tot = n + rows;
    bmean += delta * rows / tot;
    m2 += pm2 + delta * delta * (n * rows / tot);
    n = tot;
  }
  const double bvar = m2 / N;
  const double delta = bmean - mean, tot = count + N;
  Moments r;
  r.mean = mean + delta * N / tot;
  r.var = var * (count / tot) + bvar * (N / tot) + delta * delta * (count * N / (tot * tot));
  return r;
}

// pass 1: {mean, M2} of every (row block, dimension); grid (row blocks, ceil(D / 32))
__global__ __launch_bounds__(1024) void normalize_partial_kernel(const float *__restrict__ obs, int N, int D,
                                                                int rows_per_block,
                                                                double *__restrict__ partial) {
  __shared__ double red[kRowGroups][32];
  const int dim_l = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int d = blockIdx.y * 32 + dim_l;
  const bool dv = d < D;
  const int r0 = blockIdx.x * rows_per_block, r1 = min(N, r0 + rows_per_block);
  double s = 0.0;
  for (int i = r0 + grp; i < r1; i += kRowGroups) s += dv ? static_cast<double>(obs[static_cast<long long>(i) * D + d]) : 0.0;
  const double mean = group_sum(s, red, dim_l, grp) / (r1 - r0);
  double q = 0.0;
  for (int i = r0 + grp; i < r1; i += kRowGroups) {
    const double x = dv ? static_cast<double>(obs[static_cast<long long>(i) * D + d]) - mean : 0.0;
    q += x * x;
  }
  const double m2 = group_sum(q, red, dim_l, grp);
  if (dv && grp == 0) {
    partial[(static_cast<long long>(blockIdx.x) * D + d) * 2] = mean;
    partial[(static_cast<long long>(blockIdx.x) * D + d) * 2 + 1] = m2;
  }
}

// pass 2: every block derives the updated statistics of its 32 dimensions from the partials
// (read-only: the state itself is written by normalize_ret_kernel afterwards) and normalises
// its rows; grid (row blocks, ceil(D / 32))
__global__ __launch_bounds__(1024) void normalize_apply_kernel(const float *__restrict__ obs, int N, int D,
                                                              int rows_per_block, int nblocks,
                                                              const double *__restrict__ stats,
                                                              const double *__restrict__ partial, float clip,
                                                              double eps, int update,
                                                              float *__restrict__ out) {
  __shared__ double sh_mean[32], sh_inv[32];
  const int dim_l = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int d = blockIdx.y * 32 + dim_l;
  const bool dv = d < D;
  if (grp == 0) {
    Moments m{dv ? stats[d] : 0.0, dv ? stats[D + d] : 1.0};
    if (update && dv) m = merged_stats(partial, d, D, N, rows_per_block, nblocks, m.mean, m.var, stats[2 * D]);
    sh_mean[dim_l] = m.mean;
    sh_inv[dim_l] = 1.0 / sqrt(m.var + eps);
  }
  __syncthreads();
  const double mean = sh_mean[dim_l], inv = sh_inv[dim_l];
  const int r0 = blockIdx.x * rows_per_block, r1 = min(N, r0 + rows_per_block);
  for (int i = r0 + grp; i < r1; i += kRowGroups) {
    if (!dv) continue;
    const long long o = static_cast<long long>(i) * D + d;
    double y = (static_cast<double>(obs[o]) - mean) * inv;
    y = y > clip ? clip : (y < -clip ? -clip : y);
    out[o] = static_cast<float>(y);
  }
}

__device__ __forceinline__ double block_sum_1024(double v, double *red) {
  red[threadIdx.x] = v;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    if (static_cast<int>(threadIdx.x) < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  const double r = red[0];
  __syncthreads();
  return r;
}

// one block: discounted returns, their running stats, reward scaling; also advances the count
// of the observation statistics (the per-dimension blocks only read it)
__global__ __launch_bounds__(1024) void normalize_ret_kernel(const float *__restrict__ rewards,
                                                            const uint8_t *__restrict__ resets, int N,
                                                            double *__restrict__ ret,
                                                            double *__restrict__ ret_stats,
                                                            double *__restrict__ obs_stats, int D,
                                                            const double *__restrict__ partial,
                                                            int rows_per_block, int nblocks, int obs_update,
                                                            float clip, double gamma, double eps,
                                                            float *__restrict__ rew_out) {
  __shared__ double red[1024];
  if (obs_stats && obs_update) {  // commit the observation statistics (same merge as pass 2)
    const double count = obs_stats[2 * D];
    for (int d = threadIdx.x; d < D; d += 1024) {
      const Moments m = merged_stats(partial, d, D, N, rows_per_block, nblocks, obs_stats[d], obs_stats[D + d], count);
      obs_stats[d] = m.mean;
      obs_stats[D + d] = m.var;
    }
    __syncthreads();  // every thread has read the old count
    if (threadIdx.x == 0) obs_stats[2 * D] = count + N;
  }
  if (!rewards) return;  // reset(): observations only
  double s = 0.0;
  for (int i = threadIdx.x; i < N; i += 1024) {
    const double r = ret[i] * gamma + static_cast<double>(rewards[i]);
    ret[i] = r;
    s += r;
  }
  double scale = 1.0;
  if (ret_stats) {
    const double bmean = block_sum_1024(s, red) / N;
    double q = 0.0;
    for (int i = threadIdx.x; i < N; i += 1024) {
      const double x = ret[i] - bmean;
      q += x * x;
    }
    const double bvar = block_sum_1024(q, red) / N;
    const double mean = ret_stats[0], var = ret_stats[1], count = ret_stats[2];
    const double delta = bmean - mean, tot = count + N;
    const double new_var = var * (count / tot) + bvar * (N / tot) + delta * delta * (count * N / (tot * tot));
    __syncthreads();  // every thread has read the old stats
    if (threadIdx.x == 0) {
      ret_stats[0] = mean + delta * N / tot;
      ret_stats[1] = new_var;
      ret_stats[2] = tot;
    }
    scale = 1.0 / sqrt(new_var + eps);
  }
  for (int i = threadIdx.x; i < N; i += 1024) {
    double y = static_cast<double>(rewards[i]) * scale;
    if (ret_stats) y = y > clip ? clip : (y < -clip ? -clip : y);
    rew_out[i] = static_cast<float>(y);
    if (resets && resets[i]) ret[i] = 0.0;
  }
}

}  // namespace

extern "C" int dx_normalize_step_f32(const float *obs, int N, int D, const float *rewards,
                                     const uint8_t *resets, double *obs_stats, double *ret_stats,
                                     double *ret, double *workspace, long long workspace_count,
                                     float clipobs, float cliprew, double gamma, double eps,
                                     int update_stats, float *obs_out, float *rew_out, void *stream) {
  DX_TRACE("dx_normalize_step_f32");
  DX_REQUIRE(N >= 1 && D >= 1, "dx_normalize_step_f32: bad shape N=%d D=%d", N, D);
  DX_REQUIRE(obs && obs_out, "dx_normalize_step_f32: null observations");
  DX_REQUIRE(!rewards || (ret && rew_out), "dx_normalize_step_f32: rewards need ret and rew_out");
  hipStream_t s = dx::as_stream(stream);
  // rows per block: >= 64, at most kMaxRowBlocks blocks
  int rows_per_block = 64;
  while (dx::cdiv(N, rows_per_block) > kMaxRowBlocks) rows_per_block *= 2;
  const int nblocks = dx::cdiv(N, rows_per_block);
  if (obs_stats) {
    DX_REQUIRE(workspace && workspace_count >= 2LL * nblocks * D,
               "dx_normalize_step_f32: workspace of %lld doubles needed (got %lld)", 2LL * nblocks * D,
               workspace_count);
    const dim3 grid(nblocks, dx::cdiv(D, 32));
    if (update_stats) {
      hipLaunchKernelGGL(normalize_partial_kernel, grid, dim3(1024), 0, s, obs, N, D, rows_per_block, workspace);
      DX_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(normalize_apply_kernel, grid, dim3(1024), 0, s, obs, N, D, rows_per_block, nblocks,
                       obs_stats, workspace, clipobs, eps, update_stats, obs_out);
    DX_LAUNCH_CHECK();
  } else if (obs_out != obs) {
    DX_HIP(hipMemcpyAsync(obs_out, obs, sizeof(float) * static_cast<size_t>(N) * D, hipMemcpyDeviceToDevice, s));
  }
  if (rewards || (obs_stats && update_stats)) {
    hipLaunchKernelGGL(normalize_ret_kernel, dim3(1), dim3(1024), 0, s, rewards, resets, N, ret, ret_stats,
                       obs_stats, D, workspace, rows_per_block, nblocks, update_stats, cliprew, gamma, eps,
                       rew_out);
    DX_LAUNCH_CHECK();
  }
  return DX_OK;
}
