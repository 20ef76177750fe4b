// LDS micro-benchmark 2 for gfx950: the backward arc walk's instruction MIX (two random gathers per
// cell plus one update of a third array) at the kernel's geometry.  Reports cycles per cell
// (per wave-instruction triple) CU-wide, for different ways of doing the update.
//   hipcc --offload-arch=gfx950 -O3 -o lds_mix lds_mix.hip && ./lds_mix
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int NT = 1024;
constexpr int ITERS = 512;
constexpr int UNROLL = 8;

enum Mode { GATHER2, GATHER2_ADDU32, GATHER2_WRITE, GATHER1_ADDU32, ADDU32_ONLY, GATHER2_ADDU32_BATCHED, GATHER2_ADDF32 };

template <int MODE>
__global__ __launch_bounds__(NT) void bench(const uint32_t *__restrict__ idx, float *out, long long *cycles) {
  extern __shared__ __align__(16) float lds[];
  float *A = lds, *B = lds + 8192, *Gm = lds + 8192 + 4096;
  const int tid = threadIdx.x;
  for (int i = tid; i < 8192 + 4096 + 4096; i += NT) lds[i] = 1.0f;
  uint32_t ia[UNROLL], ib[UNROLL];
#pragma unroll
  for (int u = 0; u < UNROLL; ++u) {
    ia[u] = idx[u * NT + tid] % 8192;
    ib[u] = idx[(UNROLL + u) * NT + tid] % 4096;
  }
  float acc = 0.f;
  uint32_t *Gu = reinterpret_cast<uint32_t *>(Gm);
  __syncthreads();
  long long t0 = clock64();
  for (int it = 0; it < ITERS; ++it) {
    float a[UNROLL], b[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      if (MODE != ADDU32_ONLY) a[u] = A[ia[u]];
      if (MODE == GATHER2 || MODE == GATHER2_ADDU32 || MODE == GATHER2_WRITE || MODE == GATHER2_ADDU32_BATCHED ||
          MODE == GATHER2_ADDF32)
        b[u] = B[ib[u]];
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      float v = (MODE == ADDU32_ONLY) ? 1.0f : a[u];
      if (MODE == GATHER2 || MODE == GATHER2_ADDU32 || MODE == GATHER2_WRITE || MODE == GATHER2_ADDF32) v *= b[u];
      acc += v;
      if (MODE == GATHER2_ADDU32 || MODE == GATHER1_ADDU32 || MODE == ADDU32_ONLY) atomicAdd(&Gu[ib[u]], (uint32_t)v);
      if (MODE == GATHER2_WRITE) Gm[ib[u]] = v;
      if (MODE == GATHER2_ADDF32) atomicAdd(&Gm[ib[u]], v);
    }
    if (MODE == GATHER2_ADDU32_BATCHED) {
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) atomicAdd(&Gu[ib[u]], (uint32_t)(a[u] * b[u]));
    }
    asm volatile("" ::: "memory");
  }
  __syncthreads();
  long long t1 = clock64();
  if (tid == 0) cycles[blockIdx.x] = t1 - t0;
  out[blockIdx.x * NT + tid] = acc + lds[tid];
}

template <int MODE>
void run(const char *name, const std::vector<uint32_t> &h_idx) {
  const int blocks = 256;
  uint32_t *d_idx;
  float *d_out;
  long long *d_cyc;
  (void)hipMalloc(&d_idx, h_idx.size() * 4);
  (void)hipMemcpy(d_idx, h_idx.data(), h_idx.size() * 4, hipMemcpyHostToDevice);
  (void)hipMalloc(&d_out, (size_t)blocks * NT * 4);
  (void)hipMalloc(&d_cyc, blocks * 8);
  size_t lds = (8192 + 4096 + 4096) * 4;
  (void)hipFuncSetAttribute((const void *)bench<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(bench<MODE>, dim3(blocks), dim3(NT), lds, 0, d_idx, d_out, d_cyc);
  (void)hipDeviceSynchronize();
  std::vector<long long> cyc(blocks);
  (void)hipMemcpy(cyc.data(), d_cyc, blocks * 8, hipMemcpyDeviceToHost);
  double mean = 0;
  for (auto c : cyc) mean += (double)c;
  mean /= blocks;
  printf("%-58s cycles per cell (CU-wide) %8.2f\n", name, mean / ((double)ITERS * UNROLL * (NT / 64)));
  (void)hipFree(d_idx);
  (void)hipFree(d_out);
  (void)hipFree(d_cyc);
}

int main() {
  std::vector<uint32_t> rnd(2 * UNROLL * NT);
  srand(3);
  for (auto &v : rnd) v = (uint32_t)rand();
  run<GATHER2>("2 random gathers", rnd);
  run<GATHER2_WRITE>("2 random gathers + random ds_write_b32", rnd);
  run<GATHER2_ADDU32>("2 random gathers + random ds_add_u32 (interleaved)", rnd);
  run<GATHER2_ADDU32_BATCHED>("2 random gathers + random ds_add_u32 (8 adds batched)", rnd);
  run<GATHER1_ADDU32>("1 random gather  + random ds_add_u32", rnd);
  run<ADDU32_ONLY>("random ds_add_u32 only", rnd);
  run<GATHER2_ADDF32>("2 random gathers + random ds_add_f32", rnd);
  return 0;
}
