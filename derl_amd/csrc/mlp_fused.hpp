// Fused two-net MLP kernels (mlp_fused.hip), used by mlp.hip when the observation fits 64 columns.
#pragma once
#include "common.hpp"

namespace dx {

struct MlpFusedArgs {
  const float *params;
  long long off_w[6], off_b[6];
  int D, Dp, P, B;
  const float *obs;          // forward input (B, D)
  float *xpad;               // (B, Dp) zero-padded copy, written by the forward
  float *h1[2], *h2[2];      // (B, 64) each
  float *head;               // (B, 32)
  const float *dhead;        // backward input (B, 32)
  float *slabs;              // backward output, laid out like dx_mlp_backward's slabs
  long long slab_per_net;
  int ms_cap, nslab;         // slab capacity per layer / workgroups (= slabs written) per net
};

// `T` steps of the Gaussian policy against the MuJoCo-shaped synthetic env in ONE launch (mlp_fused.hip): every env's
// chain observation -> both nets -> sample -> next observation is local to its workgroup
struct MlpRolloutArgs {
  MlpFusedArgs f;        // params / offsets / D, Dp, P; B = number of envs
  long long off_logstd;
  float *obs;            // (T + 1, N, D), obs[0] given, the rest written
  float *actions;        // (T, N, P)
  float *log_prob, *values, *rewards;  // (T, N)
  uint8_t *resets;       // (T, N)
  int T;
  uint64_t policy_seed, policy_counter, env_seed, env_counter;
  float p_reset;
};
int launch_mlp_rollout_synth(const MlpRolloutArgs &a, hipStream_t stream);

bool mlp_fused_supported(int obs_pad);
int mlp_fused_tile_rows(int B, int obs_pad);
int launch_mlp_forward_fused(const MlpFusedArgs &a, hipStream_t stream);
int launch_mlp_backward_fused(const MlpFusedArgs &a, hipStream_t stream);

}  // namespace dx
