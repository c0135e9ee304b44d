// probe: how fast can ONE workgroup per CU stream an L2-resident buffer (every workgroup reads the SAME 1 MB, as the
// workgroups of a fused-FFN kernel would read the weights)? Prints GB/s per CU for several waves-per-workgroup and
// loads-in-flight settings.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int UNROLL>
__global__ __launch_bounds__(1024) void k(const u32x4* __restrict__ buf, int n16, int reps, unsigned* sink) {
  u32x4 acc = {0u, 0u, 0u, 0u};
  const int nthr = blockDim.x;
  for (int r = 0; r < reps; ++r) {
    for (int i = threadIdx.x; i + (UNROLL - 1) * nthr < n16; i += UNROLL * nthr) {
      u32x4 v[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) v[u] = buf[i + u * nthr];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) acc ^= v[u];
    }
  }
  if (acc[0] == 0x12345u) sink[0] = acc[1] ^ acc[2] ^ acc[3];
}

int main() {
  const int bytes = 1 << 20, n16 = bytes / 16, reps = 64;
  u32x4* buf; unsigned* sink;
  (void)hipMalloc(&buf, bytes); (void)hipMalloc(&sink, 4); (void)hipMemset(buf, 1, bytes);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int grid : {256, 512}) for (int threads : {256, 512, 1024}) for (int un : {4, 8}) {
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      (void)hipEventRecord(e0, 0);
      if (un == 4) hipLaunchKernelGGL(k<4>, dim3(grid), dim3(threads), 0, 0, buf, n16, reps, sink);
      else hipLaunchKernelGGL(k<8>, dim3(grid), dim3(threads), 0, 0, buf, n16, reps, sink);
      (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    const double per_wg = (double)bytes * reps / (best * 1e-3) / 1e9;
    printf("grid %3d x %4d threads, %d loads in flight/thread: %.1f us, %.1f GB/s per workgroup, %.2f TB/s aggregate\n", grid, threads, un,
           best * 1e3, per_wg, per_wg * grid / 1e3);
  }
  return 0;
}
