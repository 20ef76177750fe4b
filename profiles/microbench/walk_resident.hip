// Microbenchmark: a register-resident arc walk (8 cells per chunk: 8 x {v_and/v_lshr, ds_read_b32, v_fmac}, one
// scalar bit test per cell, a commit every 7th cell) at 1024 threads x NCH chunks vs 512 threads x 2*NCH chunks.
// Answers: can two waves per SIMD keep the LDS gather pipe as busy as four?
//   hipcc -O3 --offload-arch=gfx950 walk_resident.hip -o walk_resident && ./walk_resident
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) float lds_f;
struct Chunk { u4 wa, wb, oc; };

__device__ __forceinline__ uint32_t lo16(uint32_t x) { uint32_t r; asm volatile("v_and_b32 %0, 0xffff, %1" : "=v"(r) : "v"(x)); return r; }
__device__ __forceinline__ uint32_t hi16(uint32_t x) { uint32_t r; asm volatile("v_lshrrev_b32 %0, 16, %1" : "=v"(r) : "v"(x)); return r; }

template <int NCH, int THREADS>
__global__ __launch_bounds__(THREADS) void walk_kernel(const u4 *cells, const uint32_t *masks, float *out, int frames, long long *cyc) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x;
  for (int i = tid; i < 8192 + 8192; i += THREADS) lds[i] = 1.0f + i * 1e-6f;
  Chunk res[NCH];
  for (int c = 0; c < NCH; ++c) {
    res[c].wa = cells[(c * 3 + 0) * THREADS + tid];
    res[c].wb = cells[(c * 3 + 1) * THREADS + tid];
    res[c].oc = cells[(c * 3 + 2) * THREADS + tid];
  }
  typedef __attribute__((address_space(4))) const uint32_t cu32;
  cu32 *mk = (cu32 *)masks;
  __syncthreads();
  float total = 0.f;
  const long long t0 = clock64();
  for (int f = 0; f < frames; ++f) {
    float acc = 0.f;
    int k = 0;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const uint32_t m = mk[c];
      uint32_t o[8] = {lo16(res[c].oc.x), hi16(res[c].oc.x), lo16(res[c].oc.y), hi16(res[c].oc.y),
                       lo16(res[c].oc.z), hi16(res[c].oc.z), lo16(res[c].oc.w), hi16(res[c].oc.w)};
      float a[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = *(lds_f *)(o[i]);
      const uint32_t w[8] = {res[c].wa.x, res[c].wa.y, res[c].wa.z, res[c].wa.w, res[c].wb.x, res[c].wb.y, res[c].wb.z, res[c].wb.w};
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        acc = fmaf(a[i], __uint_as_float(w[i]), acc);
        if ((m >> i) & 1u) {
          *(lds_f *)(32768u + 16u * tid + 4u * (k & 3) + ((k >> 2) * 16u * THREADS)) = acc;
          acc = 0.f;
          ++k;
        }
      }
    }
    total += acc;
    __syncthreads();
  }
  const long long t1 = clock64();
  out[blockIdx.x * THREADS + tid] = total;
  if (tid == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NCH, int THREADS>
void run(const char *name, int conflict_free) {
  const int frames = 200;
  std::vector<u4> cells((size_t)NCH * 3 * THREADS);
  srand(1);
  for (int c = 0; c < NCH; ++c)
    for (int t = 0; t < THREADS; ++t) {
      u4 w;
      w.x = w.y = w.z = w.w = 0x3f000000u;
      cells[(c * 3 + 0) * THREADS + t] = w;
      cells[(c * 3 + 1) * THREADS + t] = w;
      u4 o;
      uint32_t off[8];
      for (int i = 0; i < 8; ++i) {
        int pos = conflict_free ? ((rand() % 256) * 32 + (t % 32)) : (rand() % 8192);
        off[i] = (uint32_t)pos * 4u;
      }
      o.x = off[0] | off[1] << 16; o.y = off[2] | off[3] << 16; o.z = off[4] | off[5] << 16; o.w = off[6] | off[7] << 16;
      cells[(c * 3 + 2) * THREADS + t] = o;
    }
  std::vector<uint32_t> masks(NCH);
  int cell = 0;
  for (int c = 0; c < NCH; ++c) {
    uint32_t m = 0;
    for (int i = 0; i < 8; ++i, ++cell)
      if (cell % 7 == 6) m |= 1u << i;
    masks[c] = m;
  }
  u4 *dc; uint32_t *dm; float *dout; long long *dcyc;
  hipMalloc(&dc, cells.size() * sizeof(u4)); hipMalloc(&dm, masks.size() * 4); hipMalloc(&dout, 256 * THREADS * 4); hipMalloc(&dcyc, 8);
  hipMemcpy(dc, cells.data(), cells.size() * sizeof(u4), hipMemcpyHostToDevice);
  hipMemcpy(dm, masks.data(), masks.size() * 4, hipMemcpyHostToDevice);
  auto k = walk_kernel<NCH, THREADS>;
  hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(256), dim3(THREADS), 150 * 1024, 0, dc, dm, dout, frames, dcyc);
  hipDeviceSynchronize();
  long long cyc;
  hipMemcpy(&cyc, dcyc, 8, hipMemcpyDeviceToHost);
  printf("%-28s %s: %7.0f cycles per walk of %d cells/CU (%.2f cells/cycle)\n", name, conflict_free ? "conflict-free" : "random banks ",
         (double)cyc / frames, NCH * 8 * THREADS, (double)NCH * 8 * THREADS * frames / cyc);
  hipFree(dc); hipFree(dm); hipFree(dout); hipFree(dcyc);
}

int main() {
  for (int cf = 0; cf < 2; ++cf) {
    run<7, 1024>("1024 thr x 7 chunks", cf);
    run<14, 512>("512 thr x 14 chunks", cf);
    run<4, 1024>("1024 thr x 4 chunks", cf);
    run<8, 512>("512 thr x 8 chunks", cf);
  }
  return 0;
}
