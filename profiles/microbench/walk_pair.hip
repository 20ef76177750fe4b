// Microbenchmark: the register-resident arc walk for ONE sequence (ds_read_b32 gathers, as den_tied_kernel.hip)
// against the same walk for TWO sequences at once (the gather source interleaved [position][2], one ds_read_b64
// per cell, two accumulators; the cell stream, its unpacking and the row-end tests are shared).
//   P0: one sequence, b32 gathers, commits by ds_write_addtid_b32             (the round-2 walk)
//   P1: two sequences, b64 gathers, commits by two ds_write_addtid_b32
//   P2: two sequences, b64 gathers, row sums kept in registers (uniform dynamic index -> v_movreld)
//   hipcc -O3 --offload-arch=gfx950 walk_pair.hip -o walk_pair && ./walk_pair
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v8f __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) float lds_f;
typedef __attribute__((address_space(3))) v2f lds_f2;
struct Chunk { u4 wa, wb, oc; };
constexpr int THREADS = 1024, NCH = 4;

__device__ __forceinline__ uint32_t lo16(uint32_t x) { uint32_t r; asm volatile("v_and_b32 %0, 0xffff, %1" : "=v"(r) : "v"(x)); return r; }
__device__ __forceinline__ uint32_t hi16(uint32_t x) { uint32_t r; asm volatile("v_lshrrev_b32 %0, 16, %1" : "=v"(r) : "v"(x)); return r; }

__device__ __forceinline__ void addtid(uint32_t row, float v) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tds_write_addtid_b32 %0" : : "v"(v), "s"(row) : "memory", "m0");
}

template <int MODE>
__global__ __launch_bounds__(THREADS) void walk_kernel(const u4 *cells, const uint32_t *masks, float *out, int frames, long long *cyc,
                                                       uint32_t accbase) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 16384 + 8192; i += THREADS) lds[i] = 1.0f + i * 1e-6f;
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
    float acc0 = 0.f, acc1 = 0.f;
    uint32_t row = accbase + 256u * 16u * (uint32_t)wave;
    int k = 0;
    float ra[8] = {0, 0, 0, 0, 0, 0, 0, 0}, rb[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const Chunk &q = res[c];
      const uint32_t m = mk[c];
      uint32_t o[8] = {lo16(q.oc.x), hi16(q.oc.x), lo16(q.oc.y), hi16(q.oc.y), lo16(q.oc.z), hi16(q.oc.z), lo16(q.oc.w), hi16(q.oc.w)};
      const uint32_t w[8] = {q.wa.x, q.wa.y, q.wa.z, q.wa.w, q.wb.x, q.wb.y, q.wb.z, q.wb.w};
      if (MODE == 0) {
        float a[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = *(lds_f *)(o[i]);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          acc0 = fmaf(a[i], __uint_as_float(w[i]), acc0);
          if (__builtin_expect((m >> i) & 1u, 0)) { addtid(row, acc0); row += 256u; acc0 = 0.f; }
        }
      } else {
        v2f a[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = *(lds_f2 *)(o[i]);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc0) : "v"(a[i].x), "v"(w[i]));
          asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc1) : "v"(a[i].y), "v"(w[i]));
          if (__builtin_expect((m >> i) & 1u, 0)) {
            if (MODE == 1) { addtid(row, acc0); addtid(row + 256u, acc1); row += 512u; }
            else {
              switch (k) {
                case 0: ra[0] = acc0; rb[0] = acc1; break;
                case 1: ra[1] = acc0; rb[1] = acc1; break;
                case 2: ra[2] = acc0; rb[2] = acc1; break;
                case 3: ra[3] = acc0; rb[3] = acc1; break;
                case 4: ra[4] = acc0; rb[4] = acc1; break;
                case 5: ra[5] = acc0; rb[5] = acc1; break;
                case 6: ra[6] = acc0; rb[6] = acc1; break;
                default: ra[7] = acc0; rb[7] = acc1; break;
              }
              ++k;
            }
            acc0 = 0.f; acc1 = 0.f;
          }
        }
      }
    }
    total += acc0 + acc1;
    if (MODE == 2) total += ra[0] + ra[1] + ra[2] + ra[3] + ra[4] + ra[5] + ra[6] + ra[7] + rb[0] + rb[1] + rb[2] + rb[3] + rb[4] + rb[5] + rb[6] + rb[7];
    __syncthreads();
  }
  const long long t1 = clock64();
  out[blockIdx.x * THREADS + tid] = total;
  if (tid == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE>
void run(const char *name, int pattern) {
  const int frames = 200;
  const int scale = MODE == 0 ? 4 : 8;  // bytes per position of the gather source
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
        if (pattern == 0) pos = (rand() % 256) * 32 + (t % 32);
        else if (pattern == 1) pos = (rand() % 256) * 32 + ((t % 32) & ~1) + (rand() & 1);
        else pos = rand() % 8192;
        off[i] = (uint32_t)pos * scale;
      }
      u4 o;
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
  auto k = walk_kernel<MODE>;
  hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(256), dim3(THREADS), 160 * 1024, 0, dc, dm, dout, frames, dcyc, 65536u + 32768u);
  hipDeviceSynchronize();
  long long cyc;
  hipMemcpy(&cyc, dcyc, 8, hipMemcpyDeviceToHost);
  const char *pn[3] = {"conflict-free", "~1.5-way     ", "random banks "};
  const int seqs = MODE == 0 ? 1 : 2;
  printf("%-44s %s: %6.0f cycles per walk, %.2f cells/cycle, %.2f sequence-cells/cycle\n", name, pn[pattern], (double)cyc / frames,
         (double)NCH * 8 * THREADS * frames / cyc, (double)seqs * NCH * 8 * THREADS * frames / cyc);
  hipFree(dc); hipFree(dm); hipFree(dout); hipFree(dcyc);
}

int main() {
  for (int p = 0; p < 3; ++p) {
    run<0>("P0 one sequence, b32, addtid commits", p);
    run<1>("P1 two sequences, b64, addtid commits", p);
    run<2>("P2 two sequences, b64, register row sums", p);
  }
  return 0;
}
