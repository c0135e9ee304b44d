// probe: how much dynamic LDS can one workgroup get on this device, and does it need hipFuncSetAttribute?
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(1024) void k(float* out, int n) {
  extern __shared__ float sm[];
  for (int i = threadIdx.x; i < n; i += blockDim.x) sm[i] = (float)i;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = sm[n - 1];
}
int main() {
  float* d; hipMalloc(&d, 1024);
  int v = 0;
  hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, 0); printf("MaxSharedMemoryPerBlock %d\n", v);
  for (int kb : {32, 64, 65, 96, 128, 159, 160}) {
    size_t bytes = (size_t)kb * 1024;
    hipError_t e = hipSuccess;
    if (kb > 64) e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    hipLaunchKernelGGL(k, dim3(4), dim3(256), bytes, 0, d, (int)(bytes / 4));
    hipError_t l = hipGetLastError();
    hipError_t s = hipDeviceSynchronize();
    float h = 0; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("%3d KB: attr=%s launch=%s sync=%s out=%.0f\n", kb, hipGetErrorString(e), hipGetErrorString(l), hipGetErrorString(s), h);
  }
  return 0;
}
