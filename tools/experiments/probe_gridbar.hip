// probe: cost of a grid-wide barrier (atomic arrive + spin) inside one launch, with a data hand-off between workgroups
// through global memory at every barrier (producer WG i writes, consumer WG (i+1)%G reads), for several grid sizes and
// for workgroups confined to one XCD (blockIdx % 8 == 0). Every spin loop is bounded: a lost barrier ends the kernel.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned target) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    __threadfence();
    atomicAdd(counter, 1u);
    int spins = 0;
    while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (++spins > 2000000) { ok = false; break; }
      __builtin_amdgcn_s_sleep(1);
    }
    __threadfence();
  }
  __syncthreads();
  return ok;
}

// mode 0: every workgroup participates; mode 1: only blockIdx % 8 == 0 (one XCD by round-robin dispatch)
__global__ __launch_bounds__(256) void k(unsigned* counter, float* buf, int n_bar, int mode, int* fail, int payload) {
  int wg = blockIdx.x, G = gridDim.x;
  if (mode == 1) {
    if (wg % 8 != 0) return;
    wg /= 8; G = (G + 7) / 8;
  }
  float acc = 0.f;
  for (int b = 0; b < n_bar; ++b) {
    // hand-off: write `payload` floats, then after the barrier read the neighbour's
    for (int i = threadIdx.x; i < payload; i += 256) buf[(size_t)(b & 1) * G * payload + (size_t)wg * payload + i] = (float)(b + wg + i);
    if (!grid_barrier(counter, (unsigned)(G * (b + 1)))) { if (threadIdx.x == 0) atomicAdd(fail, 1); return; }
    const int nb = (wg + 1) % G;
    for (int i = threadIdx.x; i < payload; i += 256) {
      const float v = buf[(size_t)(b & 1) * G * payload + (size_t)nb * payload + i];
      if (v != (float)(b + nb + i)) atomicAdd(fail, 1000);
      acc += v;
    }
  }
  if (acc == -1.f) buf[0] = acc;
}

int main() {
  unsigned* counter; float* buf; int* fail;
  hipMalloc(&counter, 4); hipMalloc(&buf, 64 << 20); hipMalloc(&fail, 4);
  hipMemset(fail, 0, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode : {0, 1}) for (int G : {16, 32, 64, 128, 256}) for (int payload : {256, 4096}) {
    const int grid = mode == 1 ? G * 8 : G;
    if (mode == 1 && G > 32) continue;
    float best[2] = {1e9f, 1e9f};
    for (int which = 0; which < 2; ++which) {
      const int n_bar = which == 0 ? 2 : 22;
      for (int rep = 0; rep < 5; ++rep) {
        hipMemsetAsync(counter, 0, 4, 0);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, counter, buf, n_bar, mode, fail, payload);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best[which]) best[which] = ms;
      }
    }
    int f; hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost);
    printf("mode %d G %3d payload %5d floats: %.2f us per barrier+handoff (2 barriers %.1f us, 22 barriers %.1f us) fail=%d\n", mode, G, payload,
           (best[1] - best[0]) * 1e3f / 20.f, best[0] * 1e3f, best[1] * 1e3f, f);
    fflush(stdout);
  }
  return 0;
}
