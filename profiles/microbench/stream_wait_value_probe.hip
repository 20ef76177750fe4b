// probe: can a stream wait (hipStreamWaitValue32) on a counter that a RUNNING kernel increments, and does the
// kernel launched behind the wait see the data the first kernel wrote before the increment?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void producer(float *data, int n, uint32_t **flag, int milestones, long long spin) {
  for (int m = 0; m < milestones; ++m) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) data[(size_t)(blockIdx.x * milestones + m) * n + i] = (float)(m + 1);
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();  // release at agent scope
      atomicAdd(flag[m], 1u);
    }
    const long long t0 = clock64();
    while (clock64() - t0 < spin) {}
    __syncthreads();
  }
}
__global__ void consumer(const float *data, int n, int blocks, int milestones, int m, int *bad) {
  for (int b = blockIdx.x; b < blocks; b += gridDim.x)
    for (int i = threadIdx.x; i < n; i += blockDim.x)
      if (data[(size_t)(b * milestones + m) * n + i] != (float)(m + 1)) atomicAdd(bad, 1);
}
int main() {
  int can = 0;
  CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
  printf("CanUseStreamWaitValue = %d\n", can);
  if (!can) return 0;
  const int blocks = 64, n = 8192, milestones = 4;
  float *data; uint32_t *flags[8]; uint32_t **dflags; int *bad;
  CK(hipMalloc(&data, (size_t)blocks * milestones * n * 4));
  CK(hipMemset(data, 0, (size_t)blocks * milestones * n * 4));
  for (int m = 0; m < milestones; ++m) { CK(hipExtMallocWithFlags((void **)&flags[m], 8, hipMallocSignalMemory)); CK(hipMemset(flags[m], 0, 8)); }
  CK(hipMalloc(&dflags, sizeof(uint32_t *) * 8)); CK(hipMemcpy(dflags, flags, sizeof(uint32_t *) * 8, hipMemcpyHostToDevice));
  CK(hipMalloc(&bad, 4)); CK(hipMemset(bad, 0, 4));
  hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t ev[8]; for (auto &x : ev) CK(hipEventCreate(&x));
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(ev[0], s1));
  hipLaunchKernelGGL(producer, dim3(blocks), dim3(256), 0, s1, data, n, dflags, milestones, 2000000LL);  // ~1 ms per milestone
  CK(hipEventRecord(ev[1], s1));
  for (int m = 0; m < milestones; ++m) {
    CK(hipStreamWaitValue32(s2, flags[m], (uint32_t)blocks, hipStreamWaitValueGte, 0xffffffffu));
    hipLaunchKernelGGL(consumer, dim3(64), dim3(256), 0, s2, data, n, blocks, milestones, m, bad);
    CK(hipEventRecord(ev[2 + m], s2));
  }
  CK(hipDeviceSynchronize());
  int hbad = -1; CK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
  float tp = 0; CK(hipEventElapsedTime(&tp, ev[0], ev[1]));
  printf("producer %.3f ms; stale reads: %d\n", tp, hbad);
  for (int m = 0; m < milestones; ++m) { float t = 0; CK(hipEventElapsedTime(&t, ev[0], ev[2 + m])); printf("  consumer of milestone %d done at %.3f ms\n", m, t); }
  return 0;
}
