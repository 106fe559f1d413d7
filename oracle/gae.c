/* Plain-C restatement of the GAE backward recursion -- TEST INFRASTRUCTURE ONLY.
 *
 * Follows derl/runners/trajectory_transforms.py:45-65 (see oracle/gae.py for the
 * notes on float64 intermediates).  Built by __graft_entry__.build() into
 * oracle/liboracle_gae.so and used by tests and by bench.py's cpu_baseline sweep;
 * the product (derl_amd/) never links or loads it.
 *
 * Layout: time-major (T, N) row-major, exactly the np.asarray stacking of
 * derl/runners/onpolicy.py:20-27.
 */
#include <stdint.h>
#include <stddef.h>

int oracle_gae_f32(const float *rewards, const uint8_t *resets, const float *values,
                   const float *last_values, int T, int N, double gamma, double lambda,
                   float *advantages, float *value_targets) {
  if (T <= 0 || N <= 0) return 0;
  for (int n = 0; n < N; ++n) {
    size_t i = (size_t)(T - 1) * N + n;
    double nr = 1.0 - (double)(resets[i] != 0);
    float first = (float)((double)rewards[i] - (double)values[i]); /* :46 */
    advantages[i] = (float)((double)first + nr * gamma * (double)last_values[n]); /* :53 */
  }
  for (int t = T - 1; t > 0; --t) { /* :56-62 */
    const size_t cur = (size_t)t * N, prev = (size_t)(t - 1) * N;
    for (int n = 0; n < N; ++n) {
      double nr = 1.0 - (double)(resets[prev + n] != 0);
      double delta = (double)rewards[prev + n] + nr * gamma * (double)values[cur + n]
                     - (double)values[prev + n];
      advantages[prev + n] = (float)(delta + nr * gamma * lambda * (double)advantages[cur + n]);
    }
  }
  for (size_t i = 0; i < (size_t)T * N; ++i)
    value_targets[i] = advantages[i] + values[i]; /* :63 float32 add */
  return 0;
}
