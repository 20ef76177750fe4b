// LDS micro-benchmark for gfx950: cycles per wave-instruction of the LDS operations the chain
// kernels are built from, at the kernels' own geometry (1024-thread workgroups, one per CU).
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o lds_ops lds_ops.hip && ./lds_ops
// Indices are held in registers, so the timed loop contains only the LDS instruction under test
// (plus one VALU op per result).  Prints shader cycles per wave-instruction as seen by the whole CU
// with all 16 waves issuing back to back, i.e. the reciprocal throughput that bounds a kernel made
// of that operation.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int NT = 1024;
constexpr int ITERS = 512;
constexpr int UNROLL = 16;

enum Op { READ32, READ64, READ128, ADDF32, WRITE32, ADDU32, ADDU64, MAXU32, ADDU32_RTN };

template <int OP>
__global__ __launch_bounds__(NT) void bench(const uint32_t *__restrict__ idx, float *out, long long *cycles,
                                            int lds_floats) {
  extern __shared__ __align__(16) float lds[];
  const int tid = threadIdx.x;
  for (int i = tid; i < lds_floats; i += NT) lds[i] = 0.0f;
  uint32_t my[UNROLL];
#pragma unroll
  for (int u = 0; u < UNROLL; ++u) my[u] = idx[u * NT + tid];
  float acc = 0.f;
  uint32_t *ldsu = reinterpret_cast<uint32_t *>(lds);
  __syncthreads();
  long long t0 = clock64();
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      if (OP == READ32) acc += lds[my[u]];
      if (OP == READ64) {
        float2 v = *reinterpret_cast<float2 *>(&lds[my[u] & ~1u]);
        acc += v.x + v.y;
      }
      if (OP == READ128) {
        float4 v = *reinterpret_cast<float4 *>(&lds[my[u] & ~3u]);
        acc += v.x + v.y + v.z + v.w;
      }
      if (OP == ADDF32) atomicAdd(&lds[my[u]], 1.0f);
      if (OP == ADDU32) atomicAdd(&ldsu[my[u]], 3u);
      if (OP == ADDU32_RTN) acc += (float)atomicAdd(&ldsu[my[u]], 3u);
      if (OP == MAXU32) atomicMax(&ldsu[my[u]], (uint32_t)(it + u));
      if (OP == ADDU64)
        atomicAdd(reinterpret_cast<unsigned long long *>(&ldsu[my[u] & ~1u]), (unsigned long long)3);
      if (OP == WRITE32) lds[my[u]] = acc + (float)u;
    }
    // keep the compiler from hoisting/merging iterations
    asm volatile("" ::: "memory");
  }
  __syncthreads();
  long long t1 = clock64();
  if (tid == 0) cycles[blockIdx.x] = t1 - t0;
  out[blockIdx.x * NT + tid] = acc + lds[tid];
}

template <int OP>
double run(const char *name, const std::vector<uint32_t> &h_idx, int lds_floats, int blocks) {
  uint32_t *d_idx;
  float *d_out;
  long long *d_cyc;
  (void)hipMalloc(&d_idx, h_idx.size() * 4);
  (void)hipMemcpy(d_idx, h_idx.data(), h_idx.size() * 4, hipMemcpyHostToDevice);
  (void)hipMalloc(&d_out, (size_t)blocks * NT * 4);
  (void)hipMalloc(&d_cyc, blocks * 8);
  size_t lds = (size_t)lds_floats * 4;
  (void)hipFuncSetAttribute((const void *)bench<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int rep = 0; rep < 2; ++rep)
    hipLaunchKernelGGL(bench<OP>, dim3(blocks), dim3(NT), lds, 0, d_idx, d_out, d_cyc, lds_floats);
  (void)hipDeviceSynchronize();
  std::vector<long long> cyc(blocks);
  (void)hipMemcpy(cyc.data(), d_cyc, blocks * 8, hipMemcpyDeviceToHost);
  double mean = 0;
  for (auto c : cyc) mean += (double)c;
  mean /= blocks;
  double per_inst = mean / ((double)ITERS * UNROLL * (NT / 64));
  printf("%-46s cycles per wave-instruction (CU-wide) %8.3f\n", name, per_inst);
  (void)hipFree(d_idx);
  (void)hipFree(d_out);
  (void)hipFree(d_cyc);
  return per_inst;
}

int main() {
  const int blocks = 256;
  const int n = UNROLL * NT;
  std::vector<uint32_t> lin(n), rnd8k(n), rnd4k(n), same(n), rnd_nobank(n), two_way(n), four_way(n);
  srand(1);
  for (int i = 0; i < n; ++i) {
    const int lane = i % 64;
    lin[i] = (uint32_t)(i % NT);
    rnd8k[i] = (uint32_t)(rand() % 8192);
    rnd4k[i] = (uint32_t)(rand() % 4096);
    same[i] = 5;
    rnd_nobank[i] = (uint32_t)((rand() % 256) * 32 + (lane % 32));          // random row, bank = lane
    two_way[i] = (uint32_t)((rand() % 256) * 32 + (lane % 32) / 2 * 2);    // exact 2-way conflicts
    four_way[i] = (uint32_t)((rand() % 256) * 32 + (lane % 32) / 4 * 4);   // exact 4-way conflicts
  }
  const int L = 80 * 1024 / 4;  // 80 KB of LDS
  run<READ32>("ds_read_b32   lane-linear", lin, L, blocks);
  run<READ32>("ds_read_b32   random row, bank = lane", rnd_nobank, L, blocks);
  run<READ32>("ds_read_b32   random, exact 2-way conflicts", two_way, L, blocks);
  run<READ32>("ds_read_b32   random, exact 4-way conflicts", four_way, L, blocks);
  run<READ32>("ds_read_b32   uniformly random in 8192", rnd8k, L, blocks);
  run<READ32>("ds_read_b32   all lanes same address", same, L, blocks);
  run<READ64>("ds_read_b64   uniformly random in 8192", rnd8k, L, blocks);
  run<READ128>("ds_read_b128  uniformly random in 8192", rnd8k, L, blocks);
  run<WRITE32>("ds_write_b32  lane-linear", lin, L, blocks);
  run<WRITE32>("ds_write_b32  uniformly random in 8192", rnd8k, L, blocks);
  run<ADDU32>("ds_add_u32    lane-linear", lin, L, blocks);
  run<ADDU32>("ds_add_u32    random row, bank = lane", rnd_nobank, L, blocks);
  run<ADDU32>("ds_add_u32    uniformly random in 4096", rnd4k, L, blocks);
  run<ADDU32>("ds_add_u32    all lanes same address", same, L, blocks);
  run<ADDU32_RTN>("ds_add_rtn_u32 uniformly random in 4096", rnd4k, L, blocks);
  run<MAXU32>("ds_max_u32    uniformly random in 4096", rnd4k, L, blocks);
  run<ADDU64>("ds_add_u64    uniformly random in 4096", rnd4k, L, blocks);
  run<ADDF32>("ds_add_f32    lane-linear", lin, L, blocks);
  run<ADDF32>("ds_add_f32    uniformly random in 4096", rnd4k, L, blocks);
  return 0;
}
