// fp32 weight mirrors -> three bf16 planes (hi, mid, lo with hi + mid + lo == the fp32 value
// exactly): what the first conv layer's bf16-MFMA kernels read (conv0_b16.hip).  Part of
// dx_cnn_pack (derl/models.py:94-124's first conv; DESIGN.md section 3, exact-split arithmetic).
#include "bf16_split.hpp"

namespace dx {

// fp32 [n] -> three bf16 planes dst[0..n), dst[n..2n), dst[2n..3n) with src == hi + mid + lo exactly
namespace {
struct SplitTable {
  const float *src[kMaxSplitJobs];
  uint16_t *dst[kMaxSplitJobs];
  long long count[kMaxSplitJobs];
};
__global__ __launch_bounds__(256) void split_planes_kernel(const SplitTable t) {
  const float *src = t.src[blockIdx.y];
  uint16_t *dst = t.dst[blockIdx.y];
  const long long n = t.count[blockIdx.y];  // multiple of 4
  for (long long i = (static_cast<long long>(blockIdx.x) * 256 + threadIdx.x) * 4; i < n;
       i += static_cast<long long>(gridDim.x) * 1024) {
    const Split4 s = split4(*reinterpret_cast<const f32x4 *>(src + i));
    *reinterpret_cast<uint2 *>(dst + i) = s.hi;
    *reinterpret_cast<uint2 *>(dst + n + i) = s.mid;
    *reinterpret_cast<uint2 *>(dst + 2 * n + i) = s.lo;
  }
}
}  // namespace

int launch_split_planes(const float *const *src, uint16_t *const *dst, const long long *count, int njobs,
                        hipStream_t stream) {
  DX_REQUIRE(njobs >= 0 && njobs <= kMaxSplitJobs, "split_planes: %d jobs (max %d)", njobs, kMaxSplitJobs);
  if (njobs == 0) return DX_OK;
  SplitTable t;
  long long biggest = 0;
  for (int i = 0; i < njobs; ++i) {
    DX_REQUIRE(src[i] && dst[i] && count[i] > 0 && count[i] % 4 == 0, "split_planes: bad job %d", i);
    t.src[i] = src[i]; t.dst[i] = dst[i]; t.count[i] = count[i];
    if (count[i] > biggest) biggest = count[i];
  }
  int bx = cdiv(biggest, 1024);
  if (bx > 1024) bx = 1024;
  hipLaunchKernelGGL(split_planes_kernel, dim3(bx, njobs), dim3(256), 0, stream, t);
  DX_LAUNCH_CHECK();
  return DX_OK;
}

}  // namespace dx
