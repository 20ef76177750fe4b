// Microbenchmark: row sums accumulated DIRECTLY into register rows through the GPR-index mode, no row-end branch.
//
// walk_variants.hip showed what the row ends cost the arc walk: one s_bitcmp + s_cbranch per cell and a commit
// sequence per row (per-cell flags 3830 cycles per walk conflict-free against 2380 without flags).  Here the FMA of a
// cell writes v[ROW0 + M0[7:0]] (s_set_gpr_idx_on with VSRC2_REL | VDST_REL): a row end is "M0 += 1" on the scalar
// unit (s_bitcmp1 + s_addc), never a branch, and the row sums never pass through LDS.
//   G0: FMA block per chunk, gathers of the chunk issued right before it
//   G1: the next chunk's gathers issued ahead of this chunk's FMA block
//   G2: as G1 with per-cell M0 images (one SALU per cell) instead of s_bitcmp1 + s_addc
// Every variant is checked against the row sums computed on the host.
//   hipcc -O3 --offload-arch=gfx950 walk_gpridx.hip -o walk_gpridx && ./walk_gpridx
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) float lds_f;
struct Chunk { u4 wa, wb, oc; };
constexpr int THREADS = 1024, NCH = 7, ROWS = 8;

__device__ __forceinline__ uint32_t lo16(uint32_t x) { uint32_t r; asm volatile("v_and_b32 %0, 0xffff, %1" : "=v"(r) : "v"(x)); return r; }
__device__ __forceinline__ uint32_t hi16(uint32_t x) { uint32_t r; asm volatile("v_lshrrev_b32 %0, 16, %1" : "=v"(r) : "v"(x)); return r; }

__device__ __forceinline__ void gather(const Chunk &q, float (&a)[8]) {
  uint32_t o[8] = {lo16(q.oc.x), hi16(q.oc.x), lo16(q.oc.y), hi16(q.oc.y), lo16(q.oc.z), hi16(q.oc.z), lo16(q.oc.w), hi16(q.oc.w)};
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = *(lds_f *)(o[i]);
}

// The nine row registers (eight rows + one that takes the padding cells behind the last row end) live in v[118:126].
struct Rows { float r0, r1, r2, r3, r4, r5, r6, r7, r8; };
#define ROW_OPS(R) "+{v118}"(R.r0), "+{v119}"(R.r1), "+{v120}"(R.r2), "+{v121}"(R.r3), "+{v122}"(R.r4), "+{v123}"(R.r5), "+{v124}"(R.r6), "+{v125}"(R.r7), "+{v126}"(R.r8)

// one chunk: 8 x { v_fma into row M0 ; row end ? M0 += 1 }.  m: bit i = a row ends with cell i.
#define CELL(i) "v_fma_f32 v118, %[a" #i "], %[w" #i "], v118\n\ts_bitcmp1_b32 %[m], " #i "+%[sh]\n\ts_addc_u32 m0, m0, 0\n\t"
template <int SH>
__device__ __forceinline__ void fma_block(const Chunk &q, const float (&a)[8], uint32_t m, uint32_t &k, Rows &R) {
  asm volatile("s_set_gpr_idx_on %[k], 0xc\n\t" CELL(0) CELL(1) CELL(2) CELL(3) CELL(4) CELL(5) CELL(6) CELL(7)
               "s_and_b32 %[k], m0, 0xff\n\ts_set_gpr_idx_off"
               : [k] "+s"(k), ROW_OPS(R)
               : [a0] "v"(a[0]), [a1] "v"(a[1]), [a2] "v"(a[2]), [a3] "v"(a[3]), [a4] "v"(a[4]), [a5] "v"(a[5]), [a6] "v"(a[6]), [a7] "v"(a[7]),
                 [w0] "v"(q.wa.x), [w1] "v"(q.wa.y), [w2] "v"(q.wa.z), [w3] "v"(q.wa.w), [w4] "v"(q.wb.x), [w5] "v"(q.wb.y), [w6] "v"(q.wb.z), [w7] "v"(q.wb.w),
                 [m] "s"(m), [sh] "n"(SH)
               : "m0", "scc");
}
// per-cell M0 images: mi[0..3] hold 0xC000 | row for cells (0,1), (2,3), (4,5), (6,7); the image is applied BEFORE the cell
#define CELLI(i, j, op) op "\n\tv_fma_f32 v118, %[a" #i "], %[w" #i "], v118\n\t"
__device__ __forceinline__ void fma_block_img(const Chunk &q, const float (&a)[8], u4 mi, Rows &R) {
  asm volatile("s_set_gpr_idx_on %[i0], 0xc\n\t"
               CELLI(0, 0, "s_mov_b32 m0, %[i0]") CELLI(1, 0, "s_lshr_b32 m0, %[i0], 16") CELLI(2, 1, "s_mov_b32 m0, %[i1]") CELLI(3, 1, "s_lshr_b32 m0, %[i1], 16")
               CELLI(4, 2, "s_mov_b32 m0, %[i2]") CELLI(5, 2, "s_lshr_b32 m0, %[i2], 16") CELLI(6, 3, "s_mov_b32 m0, %[i3]") CELLI(7, 3, "s_lshr_b32 m0, %[i3], 16")
               "s_set_gpr_idx_off"
               : ROW_OPS(R)
               : [a0] "v"(a[0]), [a1] "v"(a[1]), [a2] "v"(a[2]), [a3] "v"(a[3]), [a4] "v"(a[4]), [a5] "v"(a[5]), [a6] "v"(a[6]), [a7] "v"(a[7]),
                 [w0] "v"(q.wa.x), [w1] "v"(q.wa.y), [w2] "v"(q.wa.z), [w3] "v"(q.wa.w), [w4] "v"(q.wb.x), [w5] "v"(q.wb.y), [w6] "v"(q.wb.z), [w7] "v"(q.wb.w),
                 [i0] "s"(mi.x), [i1] "s"(mi.y), [i2] "s"(mi.z), [i3] "s"(mi.w)
               : "m0", "scc");
}

template <int MODE>
__global__ __launch_bounds__(THREADS) void walk_kernel(const u4 *cells, const uint32_t *masks, const u4 *images, float *out, float *rows_out, int frames,
                                                        long long *cyc) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x;
  for (int i = tid; i < 8192; i += THREADS) lds[i] = 1.0f + (i % 97) * 0.03125f;
  Chunk res[NCH];
  for (int c = 0; c < NCH; ++c) {
    res[c].wa = cells[(c * 3 + 0) * THREADS + tid];
    res[c].wb = cells[(c * 3 + 1) * THREADS + tid];
    res[c].oc = cells[(c * 3 + 2) * THREADS + tid];
  }
  typedef __attribute__((address_space(4))) const uint32_t cu32;
  typedef __attribute__((address_space(4))) const u4 cu4;
  cu32 *mk = (cu32 *)masks;
  cu4 *im = (cu4 *)images;
  __syncthreads();
  float total = 0.f;
  Rows R;
  const long long t0 = clock64();
  for (int f = 0; f < frames; ++f) {
    R = Rows{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    uint32_t k = 0;
    if (MODE == 0) {
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        float a[8];
        gather(res[c], a);
        fma_block<0>(res[c], a, mk[c], k, R);
      }
    } else {
      float a0[8], a1[8];
      gather(res[0], a0);
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        if (c & 1) {
          if (c + 1 < NCH) gather(res[c + 1], a0);
          if (MODE == 1) fma_block<0>(res[c], a1, mk[c], k, R); else fma_block_img(res[c], a1, im[c], R);
        } else {
          if (c + 1 < NCH) gather(res[c + 1], a1);
          if (MODE == 1) fma_block<0>(res[c], a0, mk[c], k, R); else fma_block_img(res[c], a0, im[c], R);
        }
      }
    }
    total += R.r0 + R.r1 + R.r2 + R.r3 + R.r4 + R.r5 + R.r6 + R.r7;
    __syncthreads();
  }
  const long long t1 = clock64();
  out[blockIdx.x * THREADS + tid] = total;
  if (blockIdx.x == 0) {
    const float rr[9] = {R.r0, R.r1, R.r2, R.r3, R.r4, R.r5, R.r6, R.r7, R.r8};
    for (int i = 0; i < 9; ++i) rows_out[i * THREADS + tid] = rr[i];
  }
  if (tid == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE>
void run(const char *name, int pattern) {
  const int frames = 200;
  std::vector<u4> cells((size_t)NCH * 3 * THREADS);
  std::vector<uint32_t> offs((size_t)NCH * 8 * THREADS);
  std::vector<float> wts((size_t)NCH * 8 * THREADS);
  srand(1);
  for (int c = 0; c < NCH; ++c)
    for (int t = 0; t < THREADS; ++t) {
      float w[8];
      uint32_t off[8];
      for (int i = 0; i < 8; ++i) {
        int pos;
        if (pattern == 0) pos = (rand() % 256) * 32 + (t % 32);
        else if (pattern == 1) pos = (rand() % 256) * 32 + ((t % 32) & ~1) + (rand() & 1);
        else pos = rand() % 8192;
        off[i] = (uint32_t)pos * 4u;
        w[i] = 0.25f + (rand() % 64) / 128.0f;
        offs[((size_t)c * 8 + i) * THREADS + t] = pos;
        wts[((size_t)c * 8 + i) * THREADS + t] = w[i];
      }
      u4 wa, wb, o;
      memcpy(&wa, w, 16);
      memcpy(&wb, w + 4, 16);
      o.x = off[0] | off[1] << 16; o.y = off[2] | off[3] << 16; o.z = off[4] | off[5] << 16; o.w = off[6] | off[7] << 16;
      cells[(c * 3 + 0) * THREADS + t] = wa;
      cells[(c * 3 + 1) * THREADS + t] = wb;
      cells[(c * 3 + 2) * THREADS + t] = o;
    }
  // rows of 7 cells: the eighth row ends with the very last cell
  std::vector<uint32_t> masks(NCH);
  std::vector<u4> images(NCH);
  std::vector<int> row_of(NCH * 8);
  int cell = 0, row = 0;
  for (int c = 0; c < NCH; ++c) {
    uint32_t m = 0, img[4] = {0, 0, 0, 0};
    for (int i = 0; i < 8; ++i, ++cell) {
      row_of[cell] = row;
      img[i / 2] |= (0xC000u | (uint32_t)row) << (16 * (i & 1));
      if (cell % 7 == 6) { m |= 1u << i; ++row; }
    }
    masks[c] = m;
    images[c] = u4{img[0], img[1], img[2], img[3]};
  }
  u4 *dc, *di; uint32_t *dm; float *dout, *drows; long long *dcyc;
  hipMalloc(&dc, cells.size() * sizeof(u4)); hipMalloc(&dm, masks.size() * 4); hipMalloc(&di, images.size() * 16);
  hipMalloc(&dout, 256 * THREADS * 4); hipMalloc(&drows, 9 * THREADS * 4); hipMalloc(&dcyc, 8);
  hipMemcpy(dc, cells.data(), cells.size() * sizeof(u4), hipMemcpyHostToDevice);
  hipMemcpy(dm, masks.data(), masks.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(di, images.data(), images.size() * 16, hipMemcpyHostToDevice);
  auto k = walk_kernel<MODE>;
  hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(256), dim3(THREADS), 150 * 1024, 0, dc, dm, di, dout, drows, frames, dcyc);
  hipDeviceSynchronize();
  long long cyc;
  hipMemcpy(&cyc, dcyc, 8, hipMemcpyDeviceToHost);
  std::vector<float> rows(9 * THREADS);
  hipMemcpy(rows.data(), drows, rows.size() * 4, hipMemcpyDeviceToHost);
  // host check
  double worst = 0;
  for (int t = 0; t < THREADS; ++t) {
    double ref[9] = {0};
    for (int cc = 0; cc < NCH * 8; ++cc) ref[row_of[cc]] += (double)wts[(size_t)cc * THREADS + t] * (1.0 + (offs[(size_t)cc * THREADS + t] % 97) * 0.03125);
    for (int r = 0; r < 9; ++r) worst = fmax(worst, fabs(ref[r] - rows[r * THREADS + t]) / fmax(1.0, fabs(ref[r])));
  }
  const char *pn[3] = {"conflict-free", "~1.5-way     ", "random banks "};
  printf("%-40s %s: %6.0f cycles per walk (%5.0f per chunk-round, %.2f cells/cycle)  max rel err %.2e %s\n", name, pn[pattern],
         (double)cyc / frames, (double)cyc / frames / NCH, (double)NCH * 8 * THREADS * frames / cyc, worst, worst < 1e-5 ? "ok" : "WRONG");
  hipFree(dc); hipFree(dm); hipFree(di); hipFree(dout); hipFree(drows); hipFree(dcyc);
}

int main() {
  for (int p = 0; p < 3; ++p) {
    run<0>("G0 gpr-idx rows", p);
    run<1>("G1 gpr-idx rows, gathers ahead", p);
    run<2>("G2 gpr-idx rows, M0 images, ahead", p);
  }
  return 0;
}
