// Episode-reward statistics of a batched env on the device -- derl/env/summarize.py:8-63
// (RewardSummarizer.step applied to every step of a device-resident rollout in ONE launch;
// SURVEY.md 8f-3).  State per env: running episode reward, episode length (frozen once the env
// has finished an episode since the last summary), "has ended" flag, a ring of the last Q episode
// rewards.  After each step, if recording and EVERY env has ended an episode since the last
// summary, one row {total_reward, episode_length, min_reward, max_reward, reward_mean_Q,
// step_count} is emitted and the lengths / flags are cleared (summarize.py:21-52).
// One workgroup walks the T steps (the "all envs ended" test is a block-wide AND per step);
// 5 B per (step, env) + state: microseconds per rollout.
#include "common.hpp"

namespace {

struct SumArgs {
  const float *rewards;    // (T, N)
  const uint8_t *resets;   // (T, N)
  int T, N, Q, record, max_rows;
  double *acc, *ep_len;    // (N)
  uint8_t *ended;          // (N)
  double *queue;           // (N, Q)
  int *qlen, *qpos;        // (N)
  long long *step_count;   // (1)
  double *rows;            // (max_rows, 6)
  int *nrows;              // (1), incremented
};

__device__ __forceinline__ double block_reduce(double v, double *red, int op) {  // 0 sum, 1 min, 2 max
  red[threadIdx.x] = v;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    if (static_cast<int>(threadIdx.x) < s) {
      const double a = red[threadIdx.x], b = red[threadIdx.x + s];
      red[threadIdx.x] = op == 0 ? a + b : (op == 1 ? (a < b ? a : b) : (a > b ? a : b));
    }
    __syncthreads();
  }
  const double r = red[0];
  __syncthreads();
  return r;
}

__global__ __launch_bounds__(1024) void reward_summary_kernel(const SumArgs a) {
  __shared__ double red[1024];
  long long steps = *a.step_count;
  int nrows = *a.nrows;
  __syncthreads();  // every thread has read the counters before thread 0 rewrites them
  for (int t = 0; t < a.T; ++t) {
    int all = 1;
    for (int i = threadIdx.x; i < a.N; i += 1024) {
      double r = a.acc[i] + static_cast<double>(a.rewards[static_cast<long long>(t) * a.N + i]);
      uint8_t ended = a.ended[i];
      if (!ended) a.ep_len[i] += 1.0;
      if (a.resets[static_cast<long long>(t) * a.N + i]) {
        const int pos = a.qpos[i];
        a.queue[static_cast<long long>(i) * a.Q + pos] = r;
        a.qpos[i] = pos + 1 == a.Q ? 0 : pos + 1;
        if (a.qlen[i] < a.Q) a.qlen[i] += 1;
        r = 0.0;
        ended = 1;
        a.ended[i] = 1;
      }
      a.acc[i] = r;
      all &= ended;
    }
    steps += a.N;
    if (!__syncthreads_and(all) || !a.record) continue;  // uniform
    // add_summaries (summarize.py:25-38)
    double s_last = 0.0, s_len = 0.0, mn = INFINITY, mx = -INFINITY, s_mean = 0.0;
    for (int i = threadIdx.x; i < a.N; i += 1024) {
      const int n = a.qlen[i], pos = a.qpos[i];
      const double *q = a.queue + static_cast<long long>(i) * a.Q;
      const double last = q[pos == 0 ? a.Q - 1 : pos - 1];
      double m = 0.0;
      for (int k = 0; k < n; ++k) m += q[k];
      s_last += last;
      s_len += a.ep_len[i];
      mn = last < mn ? last : mn;
      mx = last > mx ? last : mx;
      s_mean += m / n;
      a.ep_len[i] = 0.0;
      a.ended[i] = 0;
    }
    s_last = block_reduce(s_last, red, 0);
    s_len = block_reduce(s_len, red, 0);
    mn = block_reduce(mn, red, 1);
    mx = block_reduce(mx, red, 2);
    s_mean = block_reduce(s_mean, red, 0);
    if (threadIdx.x == 0 && nrows < a.max_rows) {
      double *row = a.rows + static_cast<long long>(nrows) * 6;
      row[0] = s_last / a.N; row[1] = s_len / a.N; row[2] = mn; row[3] = mx; row[4] = s_mean / a.N;
      row[5] = static_cast<double>(steps);
    }
    ++nrows;
  }
  if (threadIdx.x == 0) {
    *a.step_count = steps;
    *a.nrows = nrows < a.max_rows ? nrows : a.max_rows;
  }
}

}  // namespace

extern "C" int dx_reward_summary_f32(const float *rewards, const uint8_t *resets, int T, int N, int Q,
                                     int record, double *acc, double *ep_len, uint8_t *ended,
                                     double *queue, int *qlen, int *qpos, long long *step_count,
                                     double *rows, int max_rows, int *nrows, void *stream) {
  DX_TRACE("dx_reward_summary_f32");
  DX_REQUIRE(T >= 1 && N >= 1 && Q >= 1 && max_rows >= 0, "dx_reward_summary_f32: bad shape T=%d N=%d Q=%d", T, N, Q);
  DX_REQUIRE(rewards && resets && acc && ep_len && ended && queue && qlen && qpos && step_count && rows && nrows,
             "dx_reward_summary_f32: null pointer");
  SumArgs a{rewards, resets, T, N, Q, record, max_rows, acc, ep_len, ended, queue, qlen, qpos, step_count, rows, nrows};
  hipLaunchKernelGGL(reward_summary_kernel, dim3(1), dim3(1024), 0, dx::as_stream(stream), a);
  DX_LAUNCH_CHECK();
  return DX_OK;
}
