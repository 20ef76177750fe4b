// Microbenchmark: inner-loop variants of the register-resident arc walk (1024 threads, 7 chunks of 8 cells per lane,
// gathers from a 32 KB LDS array).  Which form of "sum w * SRC[off] over rows with wave-uniform row ends" runs
// fastest on a CU when LDS gathers, VALU and scalar tests all come from the same 16 waves?
//   V0: one scalar flag test per cell, scalar FMA                (round-2 kernel as committed)
//   V1: one flag test per quad of cells, packed FMA             (rows padded to quads)
//   V2: V0 with the next chunk's gathers issued before this chunk's FMAs (software pipelined)
//   V3: V1 software pipelined
//   V4: no flags at all, packed FMA (upper bound)
//   hipcc -O3 --offload-arch=gfx950 walk_variants.hip -o walk_variants && ./walk_variants
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) float lds_f;
struct Chunk { u4 wa, wb, oc; };
constexpr int THREADS = 1024, NCH = 7;

__device__ __forceinline__ uint32_t lo16(uint32_t x) { uint32_t r; asm volatile("v_and_b32 %0, 0xffff, %1" : "=v"(r) : "v"(x)); return r; }
__device__ __forceinline__ uint32_t hi16(uint32_t x) { uint32_t r; asm volatile("v_lshrrev_b32 %0, 16, %1" : "=v"(r) : "v"(x)); return r; }

__device__ __forceinline__ void gather(const Chunk &q, float (&a)[8]) {
  uint32_t o[8] = {lo16(q.oc.x), hi16(q.oc.x), lo16(q.oc.y), hi16(q.oc.y), lo16(q.oc.z), hi16(q.oc.z), lo16(q.oc.w), hi16(q.oc.w)};
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = *(lds_f *)(o[i]);
}
__device__ __forceinline__ void commit(float v, int &k, int tid) {
  *(lds_f *)(32768u + 16u * tid + 4u * (k & 3) + ((k >> 2) * 16u * THREADS)) = v;
  ++k;
}
template <int MODE>  // 0: per-cell flags, 1: per-quad flags + pk, 4: no flags + pk
__device__ __forceinline__ void consume(const Chunk &q, const float (&a)[8], uint32_t m, v2f &acc, int &k, int tid) {
  const uint32_t w[8] = {q.wa.x, q.wa.y, q.wa.z, q.wa.w, q.wb.x, q.wb.y, q.wb.z, q.wb.w};
  if (MODE == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      acc.x = fmaf(a[i], __uint_as_float(w[i]), acc.x);
      if ((m >> i) & 1u) { commit(acc.x, k, tid); acc.x = 0.f; }
    }
  } else {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int u = 2 * h; u < 2 * h + 2; ++u)
        acc = __builtin_elementwise_fma(v2f{a[2 * u], a[2 * u + 1]}, v2f{__uint_as_float(w[2 * u]), __uint_as_float(w[2 * u + 1])}, acc);
      if (MODE == 1 && ((m >> (4 * h + 3)) & 1u)) { commit(acc.x + acc.y, k, tid); acc = v2f{0.f, 0.f}; }
    }
  }
}

template <int MODE, bool PIPE>
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
    v2f acc = {0.f, 0.f};
    int k = 0;
    if (!PIPE) {
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        float a[8];
        gather(res[c], a);
        consume<MODE>(res[c], a, mk[c], acc, k, tid);
      }
    } else {
      float a0[8], a1[8];
      gather(res[0], a0);
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        if (c & 1) {
          if (c + 1 < NCH) gather(res[c + 1], a0);
          consume<MODE>(res[c], a1, mk[c], acc, k, tid);
        } else {
          if (c + 1 < NCH) gather(res[c + 1], a1);
          consume<MODE>(res[c], a0, mk[c], acc, k, tid);
        }
      }
    }
    total += acc.x + acc.y;
    if (MODE == 4) commit(total, k, tid);
    __syncthreads();
  }
  const long long t1 = clock64();
  out[blockIdx.x * THREADS + tid] = total;
  if (tid == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE, bool PIPE>
void run(const char *name, int pattern) {
  const int frames = 200;
  std::vector<u4> cells((size_t)NCH * 3 * THREADS);
  srand(1);
  for (int c = 0; c < NCH; ++c)
    for (int t = 0; t < THREADS; ++t) {
      u4 w;
      w.x = w.y = w.z = w.w = 0x3f000000u;
      cells[(c * 3 + 0) * THREADS + t] = w;
      cells[(c * 3 + 1) * THREADS + t] = w;
      uint32_t off[8];
      for (int i = 0; i < 8; ++i) {
        int pos;
        if (pattern == 0) pos = (rand() % 256) * 32 + (t % 32);                          // conflict-free
        else if (pattern == 1) pos = (rand() % 256) * 32 + ((t % 32) & ~1) + (rand() & 1);  // about 1.5-way
        else pos = rand() % 8192;                                                       // uniformly random banks
        off[i] = (uint32_t)pos * 4u;
      }
      u4 o;
      o.x = off[0] | off[1] << 16; o.y = off[2] | off[3] << 16; o.z = off[4] | off[5] << 16; o.w = off[6] | off[7] << 16;
      cells[(c * 3 + 2) * THREADS + t] = o;
    }
  std::vector<uint32_t> masks(NCH);
  int cell = 0;
  for (int c = 0; c < NCH; ++c) {
    uint32_t m = 0;
    for (int i = 0; i < 8; ++i, ++cell) {
      if (MODE == 0 && cell % 7 == 6) m |= 1u << i;
      if (MODE == 1 && cell % 8 == 7) m |= 1u << i;  // rows of 8 (padded from 7): flags on the 8th cell only
    }
    masks[c] = m;
  }
  u4 *dc; uint32_t *dm; float *dout; long long *dcyc;
  hipMalloc(&dc, cells.size() * sizeof(u4)); hipMalloc(&dm, masks.size() * 4); hipMalloc(&dout, 256 * THREADS * 4); hipMalloc(&dcyc, 8);
  hipMemcpy(dc, cells.data(), cells.size() * sizeof(u4), hipMemcpyHostToDevice);
  hipMemcpy(dm, masks.data(), masks.size() * 4, hipMemcpyHostToDevice);
  auto k = walk_kernel<MODE, PIPE>;
  hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(256), dim3(THREADS), 150 * 1024, 0, dc, dm, dout, frames, dcyc);
  hipDeviceSynchronize();
  long long cyc;
  hipMemcpy(&cyc, dcyc, 8, hipMemcpyDeviceToHost);
  const char *pn[3] = {"conflict-free", "~1.5-way     ", "random banks "};
  printf("%-34s %s: %6.0f cycles per walk (%5.0f per chunk-round, %.2f cells/cycle)\n", name, pn[pattern], (double)cyc / frames,
         (double)cyc / frames / NCH, (double)NCH * 8 * THREADS * frames / cyc);
  hipFree(dc); hipFree(dm); hipFree(dout); hipFree(dcyc);
}

int main() {
  for (int p = 0; p < 3; ++p) {
    run<0, false>("V0 per-cell flags", p);
    run<1, false>("V1 per-quad flags + pk_fma", p);
    run<0, true>("V2 per-cell flags, pipelined", p);
    run<1, true>("V3 per-quad + pk, pipelined", p);
    run<4, false>("V4 no flags + pk (bound)", p);
    run<4, true>("V5 no flags + pk, pipelined", p);
  }
  return 0;
}
