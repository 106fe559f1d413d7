// Host side of derl/runners/onpolicy.py:44-49 (IterateWithMinibatches: one np.random.permutation
// per epoch, the shuffles compose): the composed permutations of ALL epochs of a rollout, drawn
// from NumPy's legacy global generator WITHOUT holding the Python GIL.  The caller hands over the
// generator's MT19937 state (np.random.get_state()) and takes the advanced state back
// (np.random.set_state()), so the stream is exactly the one the reference consumes.
//
// numpy (pinned by the reference as >= 1.16.4; 2.2.6 here) -- published algorithm restated:
//   RandomState.permutation(n) = arange(n) shuffled by RandomState.shuffle; for a 1-D array that is
//   `for i in reversed(range(1, n)): j = random_interval(i); swap(x[i], x[j])` (_shuffle_raw);
//   random_interval(max) masks next_uint32() with the smallest 2^k - 1 >= max and rejects values
//   > max (distributions.c); next_uint32 is the MT19937 tempering of the state words, regenerated
//   624 at a time.
// At BASELINE config 3 (131,072 samples x 10 epochs) the NumPy calls held the interpreter for 9 ms
// of a 16 ms iteration; from ctypes this runs beside it (ctypes drops the GIL around a call).
#include "common.hpp"
#include <cstring>
#include <vector>

namespace {

struct Mt { uint32_t *key; int pos; };

void mt_regenerate(Mt &s) {
  constexpr int N = 624, M = 397;
  constexpr uint32_t kMatrixA = 0x9908b0dfu, kUpper = 0x80000000u, kLower = 0x7fffffffu;
  uint32_t *mt = s.key;
  int kk = 0;
  for (; kk < N - M; ++kk) {
    const uint32_t y = (mt[kk] & kUpper) | (mt[kk + 1] & kLower);
    mt[kk] = mt[kk + M] ^ (y >> 1) ^ (-(y & 1u) & kMatrixA);
  }
  for (; kk < N - 1; ++kk) {
    const uint32_t y = (mt[kk] & kUpper) | (mt[kk + 1] & kLower);
    mt[kk] = mt[kk + (M - N)] ^ (y >> 1) ^ (-(y & 1u) & kMatrixA);
  }
  const uint32_t y = (mt[N - 1] & kUpper) | (mt[0] & kLower);
  mt[N - 1] = mt[M - 1] ^ (y >> 1) ^ (-(y & 1u) & kMatrixA);
  s.pos = 0;
}

inline uint32_t mt_next32(Mt &s) {
  if (s.pos == 624) mt_regenerate(s);
  uint32_t y = s.key[s.pos++];
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}

inline uint32_t random_interval32(Mt &s, uint32_t max) {
  if (max == 0) return 0;
  uint32_t mask = max;
  mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
  uint32_t value;
  while ((value = (mt_next32(s) & mask)) > max) {}
  return value;
}

}  // namespace

extern "C" int dx_host_compose_permutations(uint32_t *mt_key_host, int *mt_pos_host, long long n, int epochs,
                                            int shuffle, int32_t *orders_out_host) {
  DX_REQUIRE(mt_key_host && mt_pos_host && orders_out_host, "dx_host_compose_permutations: null pointer");
  DX_REQUIRE(n >= 1 && n <= 0x7fffffffLL && epochs >= 1, "dx_host_compose_permutations: need 1 <= n < 2^31 and epochs >= 1");
  DX_REQUIRE(*mt_pos_host >= 0 && *mt_pos_host <= 624, "dx_host_compose_permutations: MT19937 position %d outside [0, 624]", *mt_pos_host);
  Mt s{mt_key_host, *mt_pos_host};
  std::vector<int32_t> perm(static_cast<size_t>(n));
  std::vector<int32_t> order(static_cast<size_t>(n)), next(static_cast<size_t>(n));
  for (long long i = 0; i < n; ++i) order[i] = static_cast<int32_t>(i);
  for (int e = 0; e < epochs; ++e) {
    if (shuffle) {
      for (long long i = 0; i < n; ++i) perm[i] = static_cast<int32_t>(i);
      for (long long i = n - 1; i >= 1; --i) {
        const uint32_t j = random_interval32(s, static_cast<uint32_t>(i));
        const int32_t tmp = perm[i]; perm[i] = perm[j]; perm[j] = tmp;
      }
      for (long long i = 0; i < n; ++i) next[i] = order[perm[i]];  // order = order[permutation]
      order.swap(next);
    }
    std::memcpy(orders_out_host + static_cast<long long>(e) * n, order.data(), sizeof(int32_t) * n);
  }
  *mt_pos_host = s.pos;
  return DX_OK;
}
