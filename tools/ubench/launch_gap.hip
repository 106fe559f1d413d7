// Cost of a DEPENDENT kernel boundary on one stream: N tiny kernels back to back as plain launches, and the same
// N kernels as one captured hipGraph.  Each kernel adds 1 to a word its predecessor wrote (a true dependency).
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/launch_gap.hip -o tools/ubench/launch_gap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void bump(unsigned *w, int blocks_work) {
  if (blockIdx.x == 0 && threadIdx.x == 0) w[0] += 1;
  // a little parallel work so that the grid is not trivially one wave
  if (blocks_work) w[1 + blockIdx.x * blockDim.x + threadIdx.x] += 1;
}

int main(int argc, char **argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 200;
  unsigned *w;
  CK(hipMalloc(&w, 4 * (1 + 1024 * 256)));
  CK(hipMemset(w, 0, 4 * (1 + 1024 * 256)));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int grid : {1, 256, 1024}) {
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0, s));
      for (int i = 0; i < N; ++i) hipLaunchKernelGGL(bump, dim3(grid), dim3(256), 0, s, w, grid > 1);
      CK(hipEventRecord(e1, s));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep == 2) printf("grid %4d: %d plain launches: %.2f us each\n", grid, N, ms * 1e3 / N);
    }
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(bump, dim3(grid), dim3(256), 0, s, w, grid > 1);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0, s));
      CK(hipGraphLaunch(ge, s));
      CK(hipEventRecord(e1, s));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep == 2) printf("grid %4d: one graph of %d kernel nodes: %.2f us each\n", grid, N, ms * 1e3 / N);
    }
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  unsigned h; CK(hipMemcpy(&h, w, 4, hipMemcpyDeviceToHost));
  printf("counter %u\n", h);
  return 0;
}
