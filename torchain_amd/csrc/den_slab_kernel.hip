// Streamed denominator forward-backward: the path for graphs whose per-frame working set does not fit
// one CU's LDS (more than 16384 states or pdfs, or a layout beyond 160 KB) -- which includes many real
// Kaldi den.fst files (tens of thousands of states, hundreds of thousands of arcs).
//
// Same computation as den_kernels.hip ([K] DenominatorComputation::Forward() + Backward(),
// chain-denominator.cc; reference call site src/my_lib_chain.cpp:129-131), different mapping: with the
// state vectors in HBM/L2 anyway, LANES RUN OVER SEQUENCES and a frame is a handful of launches (the frame
// recursion is a grid-wide dependency).
//
// Layout (round 4; rounds 1-3 kept [state][64 sequences] matrices and one state per wave): sequences are cut
// into SLABS of G = 16 or 32 (per graph, by measured size classes: den_graph.cpp build_schedules), and every per-frame
// matrix is [slab][row][G] -- a state's (pdf's) values for the slab's sequences are one 64- or 128-byte segment.
// Why: a frame's arc sums gather A * S * 4 bytes (410 MB at 400 k arcs x 256 sequences) from a matrix that is re-read degree times; an XCD's L2 is 4 MB, and with 64-wide rows the
// slice of the matrix one XCD works on was 10 MB at 40 k states, so every gathered row came from the Infinity
// Cache (profiles/r03_streamed_counters_x2.txt: L2 misses = 1.03 x the gathered bytes, 5.6-6.0 TB/s in all three
// large kernels).  A slab's slice at G = 16 is 2.5 MB: it stays in L2 while the XCD walks the slab, and the block
// index is decoded so that an XCD takes the slabs one after the other (blocks b and b + 8 share an XCD).
//
// A wave works on 64 / G rows at a time, one per group of G lanes ("bundle"): the rows of a list are sorted by
// length, so the rows of a bundle have (nearly) the same number of entries, and the entries of a bundle are
// laid out so that ONE dword load per lane brings a chunk of steps for all its rows (lane G q + 16 r + W i + c
// holds dword c of step i of row q, for every 16-lane row r of the group); the step's values then reach the
// sixteen lanes of a DPP row through row broadcasts (row_newbcast, gfx90a+), folded by the compiler into the
// address add and the multiply.  No scalar loads in the loop, two VALU operations and one 256-byte gather per step.
//
// Sums over states use per-block partials reduced by a second small kernel in a fixed order; gamma of tied
// graphs is accumulated per state by the backward kernel as unsigned fixed point (31 fractional bits, integer
// atomics in L2: order-independent; storing the occupations per state and summing them per pdf in the kernel that
// writes the derivative was measured: 28.4 -> 30.9 ms on the 40000-state graph), gamma of general graphs by one group
// of lanes per pdf from a by-pdf arc list.  Results do not depend on scheduling.
//
// The history holds UN-dashed alpha_t and the per-frame sums; alpha'_t = alpha_t + leaky*pi*asum_t and
// beta_t = beta'_t + bsum_t are formed on the fly by their consumers.
#include <type_traits>

#include "den_device.h"

namespace tc {

namespace {

constexpr int kBT = 256;              // threads per block: 4 waves
// a block = kSlabRowsPerBlock (64) rows of a list = G bundles, G / 4 per wave, interleaved (bundles are sorted by length)
constexpr int kRT = 1024;             // reduction kernels

__device__ __forceinline__ float big_exp(float x) {
  x = x < -30.0f ? -30.0f : x;  // compare-and-clamp keeps NaN
  x = x > 30.0f ? 30.0f : x;
  return __expf(x);
}

// small per-sequence arrays inside p.big_small, all indexed [..][Sp] with s = slab * G + j
struct BigSmall {
  float *asum;       // [T + 1][Sp]
  float *bsum;       // [2][Sp]      leaky * sum_h pi(h) beta'(h), frames t+1 / t alternate
  float *inv_tot;    // [Sp]
  float *part_a;     // [slab][state blocks][G]   also the backward's bsum partials
  float *part_ab;    // [slab][state blocks][G]
  float *part_g;     // [slab][pdf tiles][G]
  float *part_y2;    // [slab][pdf tiles][G]  running sum of y^2 per tile of 64 pdfs
};

__host__ __device__ inline int slab_pdf_tiles(int P) { return (P + 63) / 64; }

__host__ __device__ inline BigSmall big_small(const DenParams &p) {
  const int64_t Sp = p.big_Sp;
  const int64_t hb = p.big.hb, pb = slab_pdf_tiles(p.P);
  BigSmall s;
  float *q = p.big_small;
  s.asum = q;
  q += (int64_t)(p.T + 1) * Sp;
  s.bsum = q;
  q += 2 * Sp;
  s.inv_tot = q;
  q += Sp;
  s.part_a = q;
  q += hb * Sp;
  s.part_ab = q;
  q += hb * Sp;
  s.part_g = q;
  q += pb * Sp;
  s.part_y2 = q;
  q += pb * Sp;
  return s;
}

// ---- block index -> (slab, block within the slab) ------------------------------------------------------
// Blocks b and b + 8 run on one XCD (observed placement, used for speed only): XCD x takes the x-th eighth of
// the slab-major order, i.e. whole slabs one after the other when there are at least eight of them.
struct SlabBlock {
  int slab, blk;
  bool ok;
};
__device__ __forceinline__ SlabBlock slab_block(int per_slab, int slabs) {
  const int total = per_slab * slabs, per_xcd = (total + 7) / 8;
  const int lin = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  SlabBlock b;
  b.ok = (int)(blockIdx.x >> 3) < per_xcd && lin < total;
  b.slab = b.ok ? lin / per_slab : 0;
  b.blk = b.ok ? lin % per_slab : 0;
  return b;
}
inline int slab_grid(int per_slab, int slabs) { return 8 * ((per_slab * slabs + 7) / 8); }

// ---- lists ---------------------------------------------------------------------------------------------
template <int L>
__device__ __forceinline__ uint32_t rowb(uint32_t v) {  // lane L of this lane's DPP row of 16
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x150 + L, 0xf, 0xf, false);
}
template <int L>
__device__ __forceinline__ float rowbf(uint32_t v) { return __uint_as_float(rowb<L>(v)); }

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

__device__ __forceinline__ float ld(const float *base, uint32_t byte_off) {
  return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + byte_off);
}
__device__ __forceinline__ void st(float *base, uint32_t byte_off, float v) {
  *reinterpret_cast<float *>(reinterpret_cast<char *>(base) + byte_off) = v;
}

// a lane's place in its wave for slabs of G sequences: row group q, sequence j, byte offset of j in a row
template <int G>
struct Lane {
  uint32_t lane, q, j, j4;
  __device__ __forceinline__ explicit Lane(uint32_t l) : lane(l), q(l / G), j(l % G), j4((l % G) * 4) {}
};

// Walks bundle `b` of list L: body.run<N>(rec0, rec1) consumes steps 0..N-1 of a PAIR of chunks (W dwords per step,
// 16 / W steps per chunk; a chunk is 16 dwords per row of the bundle), all N gathers requested before the first
// product -- a list of ten entries is one round trip, not a chunk of eight and a tail of two.  The records of the
// next pair are requested before the current one is consumed (the list ends with two spare chunks).
// (Non-temporal loads of the records were measured: 31.0 -> 37.2 ms on the 40000-state graph.)
template <int SPC, int I, class Body>
__device__ __forceinline__ void walk_tail(int n, uint32_t rec0, uint32_t rec1, Body &body) {
  if constexpr (I < 2 * SPC) {
    if (n == I)
      body.template run<I>(rec0, rec1);
    else
      walk_tail<SPC, I + 1>(n, rec0, rec1, body);
  }
}
// Sum over the row groups of a wave (every lane gets the total of its sequence): a bundle that holds ONE long row, its
// entries dealt round-robin to the groups (den_graph.cpp: hub rows), ends with it.  Fixed order of the adds.
template <int G>
__device__ __forceinline__ float groups_sum(float v) {
  if constexpr (G == 16) v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}

// Returns whether the bundle is such a shared row (sign bit of its step count).
template <int G, int W, class Body>
__device__ __forceinline__ bool walk(const SlabListDev &L, int b, const Lane<G> &ln, Body &body) {
  constexpr int SPC = 16 / W, CH = (64 / G) * 16;
  const int2 hd = L.head[b];
  int n = hd.y & 0x7fffffff;
  const uint32_t *r = L.rec + (size_t)hd.x * CH + ln.q * 16 + (ln.lane & 15);
  uint32_t rec0 = r[0], rec1 = r[CH];
  for (; n >= 2 * SPC; n -= 2 * SPC) {
    r += 2 * CH;
    const uint32_t nx0 = r[0], nx1 = r[CH];
    body.template run<2 * SPC>(rec0, rec1);
    rec0 = nx0;
    rec1 = nx1;
  }
  if (n > 0) walk_tail<SPC, 1>(n, rec0, rec1, body);
  if (hd.y < 0) body.sum = groups_sum<G>(body.sum);
  return hd.y < 0;
}

// record dword D of a step I of a pair of chunks (compile-time selection of the chunk's register)
template <int SPC, int W, int I, int D>
__device__ __forceinline__ uint32_t step_dword(uint32_t rec0, uint32_t rec1) {
  if constexpr (I < SPC)
    return rowb<W * I + D>(rec0);
  else
    return rowb<W * (I - SPC) + D>(rec1);
}

// sum += w * A[off]            entries {off, w}: tied graphs, both directions
struct GatherSum {
  const float *A;
  uint32_t j4;
  float sum;
  template <int N>
  __device__ __forceinline__ void run(uint32_t rec0, uint32_t rec1) {
    float g[N];
    static_for<0, N>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      g[i] = ld(A, step_dword<8, 2, i, 0>(rec0, rec1) + j4);
    });
    static_for<0, N>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      sum += __uint_as_float(step_dword<8, 2, i, 1>(rec0, rec1)) * g[i];
    });
  }
};

// sum += w * (A[a] + ca * pi) * (B[b] + cb)      entries {a, b, w, pi}: general graphs, all three lists
struct GatherSum2 {
  const float *A, *B;
  uint32_t j4;
  float ca, cb, sum;
  template <int N>
  __device__ __forceinline__ void run(uint32_t rec0, uint32_t rec1) {
    float ga[N], gb[N];
    static_for<0, N>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      ga[i] = ld(A, step_dword<4, 4, i, 0>(rec0, rec1) + j4);
      gb[i] = ld(B, step_dword<4, 4, i, 1>(rec0, rec1) + j4);
    });
    static_for<0, N>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      sum += __uint_as_float(step_dword<4, 4, i, 2>(rec0, rec1)) *
             (ga[i] + ca * __uint_as_float(step_dword<4, 4, i, 3>(rec0, rec1))) * (gb[i] + cb);
    });
  }
};

template <int G>
__device__ __forceinline__ SlabRow load_row(const SlabListDev &L, int b, uint32_t q) {
  const uint4 *src = reinterpret_cast<const uint4 *>(L.rows + ((size_t)b * (64 / G) + q));
  const uint4 lo = src[0], hi = src[1];
  SlabRow r;
  r.row = (int32_t)lo.x;
  r.n = (int32_t)lo.y;
  r.f_off = (int32_t)lo.z;
  r.s_off = (int32_t)lo.w;
  r.ws = __uint_as_float(hi.x);
  r.pi = __uint_as_float(hi.y);
  r.K = __uint_as_float(hi.z);
  r.pad = 0.f;
  return r;
}

// per-lane values (row q, sequence j) summed over the rows of a wave and the four waves: lanes 0..G-1 of wave 0
// return the block's total for sequence j
template <int G>
__device__ __forceinline__ float block_rows_sum(float v, float (*red)[64], int wave, uint32_t lane) {
  red[wave][lane] = v;
  __syncthreads();
  float t = 0.f;
  if (wave == 0 && lane < G) {
#pragma unroll
    for (int w = 0; w < 4; ++w)
#pragma unroll
      for (int q = 0; q < 64 / G; ++q) t += red[w][q * G + lane];
  }
  return t;
}

// ---- exp(y), transposed: E[frame][slab][pdf][G] ---------------------------------------------------------
// One wave = 64 pdfs x the slab's sequences (16 at a time), frames t0..t1-1; padding lanes (s >= S) get exp(0).
// sum_sq (the forward pass): sum(y^2) per sequence accumulated into this wave's own slot of part_y2, frame by frame.
template <int G>
__global__ __launch_bounds__(kBT) void slab_exp_kernel(const DenParams p, int t0, int t1, int sum_sq) {
  __shared__ float tile[4][64][17];
  __shared__ float sq[4][64][17];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int slab = blockIdx.y, ptile = blockIdx.x * 4 + wave, p0 = ptile * 64;
  if (p0 >= p.P) return;  // (no block-wide barrier below: LDS operations of one wave execute in order)
  const int pdf = p0 + lane;
  float *const slot = big_small(p).part_y2 + ((int64_t)slab * slab_pdf_tiles(p.P) + ptile) * G;
  float y2[G / 16];
#pragma unroll
  for (int part = 0; part < G / 16; ++part) y2[part] = (sum_sq && t0 > 0 && lane < 16) ? slot[part * 16 + lane] : 0.f;
  for (int t = t0; t < t1; ++t) {
    float *const out = p.big_expy + (p.big_exp_stride ? p.big_exp_stride * t : 0) + ((int64_t)slab * p.P + p0) * G;
#pragma unroll
    for (int part = 0; part < G / 16; ++part) {
      float yv[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int s = slab * G + part * 16 + i;
        yv[i] = (s < p.S && pdf < p.P) ? p.y[((int64_t)t * p.S + s) * p.y_stride + pdf] : 0.f;
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        tile[wave][lane][i] = big_exp(yv[i]);
        sq[wave][lane][i] = yv[i] * yv[i];
      }
      __builtin_amdgcn_wave_barrier();
      if (sum_sq && lane < 16) {
        float acc = 0.f;
        for (int i = 0; i < 64; ++i) acc += sq[wave][i][lane];
        y2[part] += acc;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int pl = (lane >> 2) + 16 * k, jq = (lane & 3) * 4;
        if (p0 + pl < p.P) {
          float4 v;
          v.x = tile[wave][pl][jq];
          v.y = tile[wave][pl][jq + 1];
          v.z = tile[wave][pl][jq + 2];
          v.w = tile[wave][pl][jq + 3];
          *reinterpret_cast<float4 *>(out + (int64_t)pl * G + part * 16 + jq) = v;
        }
      }
    }
  }
#pragma unroll
  for (int part = 0; part < G / 16; ++part)
    if (sum_sq && lane < 16) slot[part * 16 + lane] = y2[part];
}

// alpha_0 = pi for every sequence; asum_0 = sum(pi)   ([K] AlphaFirstFrame); gamma accumulators cleared
template <int G>
__global__ __launch_bounds__(kBT) void slab_alpha0_kernel(const DenParams p) {
  const int slabs = p.big_Sp / G;
  const int64_t n = (int64_t)slabs * p.H * G;
  const BigSmall sm = big_small(p);
  for (int64_t i = (int64_t)blockIdx.x * kBT + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBT)
    p.alpha_hist[i] = p.pi[(i / G) % p.H];
  if (p.big_gam) {
    const int64_t ng = (int64_t)slabs * p.P * G;
    for (int64_t i = (int64_t)blockIdx.x * kBT + threadIdx.x; i < ng; i += (int64_t)gridDim.x * kBT) p.big_gam[i] = 0u;
  }
  if (blockIdx.x == 0)
    for (int s = threadIdx.x; s < p.big_Sp; s += kBT) sm.asum[s] = p.big_sum_pi;
}

// sums part[rows][G] over the rows in a fixed order: NT / G groups of G lanes stride over the rows, then one
// group adds the partial sums (NT threads, red: NT floats).  Returns the total in threads 0..G-1.
template <int G, int NT = kRT>
__device__ __forceinline__ float slab_colsum(const float *part, int rows, float *red, int tid) {
  constexpr int NG = NT / G;
  const int r0 = tid / G, j = tid % G;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int r = r0;
  for (; r + 3 * NG < rows; r += 4 * NG) {
    a0 += part[(int64_t)r * G + j];
    a1 += part[(int64_t)(r + NG) * G + j];
    a2 += part[(int64_t)(r + 2 * NG) * G + j];
    a3 += part[(int64_t)(r + 3 * NG) * G + j];
  }
  for (; r < rows; r += NG) a0 += part[(int64_t)r * G + j];
  red[tid] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  float t = 0.f;
  if (tid < G)
    for (int g = 0; g < NG; ++g) t += red[g * G + tid];
  __syncthreads();
  return t;
}

// asum_t = sum over states of alpha_t: one block per slab
template <int G>
__global__ __launch_bounds__(kRT) void slab_asum_kernel(const DenParams p, int t) {
  __shared__ float red[kRT];
  const int slab = blockIdx.x, hb = p.big.hb;
  const BigSmall sm = big_small(p);
  const float tot = slab_colsum<G>(sm.part_a + (int64_t)slab * hb * G, p.big.in_blocks, red, threadIdx.x);
  if (threadIdx.x < G) sm.asum[(int64_t)t * p.big_Sp + slab * G + threadIdx.x] = tot;
}

// log-prob, 1/tot and beta_T after the last forward frame; sum of y^2 per sequence from the tile partials
template <int G>
__global__ __launch_bounds__(kBT) void slab_total_kernel(const DenParams p) {
  const int s = blockIdx.x * kBT + threadIdx.x;
  if (s >= p.big_Sp) return;
  const int Sp = p.big_Sp, slab = s / G, j = s % G;
  const BigSmall sm = big_small(p);
  // tot = sum_h alpha'_T(h) = asum_T * (1 + leaky * sum(pi))   ([K] ComputeTotLogLike)
  const float tot = sm.asum[(int64_t)p.T * Sp + s] * (1.0f + p.leaky * p.big_sum_pi);
  sm.inv_tot[s] = s < p.S ? 1.0f / tot : 0.f;
  // beta'_T = 1/tot for every state; beta_T = beta'_T + leaky * sum_h pi(h) beta'_T(h)   (padding lanes: 0)
  sm.bsum[(p.T & 1) * Sp + s] = s < p.S ? p.leaky * p.big_sum_pi * (1.0f / tot) : 0.f;
  if (s >= p.S) return;
  double y2 = 0.0;
  const int pb = slab_pdf_tiles(p.P);
  for (int b = 0; b < pb; ++b) y2 += (double)sm.part_y2[((int64_t)slab * pb + b) * G + j];
  p.seq_y2[s] = y2;
  double lp = (double)__logf(tot);
  for (int t = 0; t < p.T; ++t) lp += (double)__logf(sm.asum[(int64_t)t * Sp + s]);
  p.seq_logprob[s] = lp;
}

template <int G>
__global__ __launch_bounds__(kBT) void slab_beta_init_kernel(const DenParams p) {
  const BigSmall sm = big_small(p);
  const int slabs = p.big_Sp / G;
  float *B = p.big_beta + (int64_t)(p.T & 1) * p.H * p.big_Sp;
  const int64_t n = (int64_t)slabs * p.H * G;
  for (int64_t i = (int64_t)blockIdx.x * kBT + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBT) {
    const int s = (int)(i / ((int64_t)p.H * G)) * G + (int)(i % G);
    B[i] = sm.inv_tot[s];
  }
}

// bsum_t = leaky * sum_h pi(h) beta'_t(h): one block per slab.  At t == 0 also the two checks of
// [K] BetaGeneralFrameDebug(0) (after the frame's gamma: launch order)
template <int G>
__global__ __launch_bounds__(kRT) void slab_bsum_kernel(const DenParams p, int t) {
  __shared__ float red[kRT];
  const int slab = blockIdx.x, hb = p.big.hb, nb = p.big.out_blocks, pb = slab_pdf_tiles(p.P), tid = threadIdx.x;
  const BigSmall sm = big_small(p);
  const int s = slab * G + tid;
  const float b = slab_colsum<G>(sm.part_a + (int64_t)slab * hb * G, nb, red, tid);
  if (tid < G) sm.bsum[(t & 1) * p.big_Sp + s] = b;
  if (t == 0) {
    const float ab = slab_colsum<G>(sm.part_ab + (int64_t)slab * hb * G, nb, red, tid);
    const float gs = slab_colsum<G>(sm.part_g + (int64_t)slab * pb * G, pb, red, tid);
    if (tid < G && s < p.S) {
      p.seq_ab[s] = ab;
      p.seq_gsum[s] = gs;
    }
  }
}

// a tile of 64 pdfs x G sequences of gamma -> the derivative's rows; at t == 0 the tile's gamma sum per sequence
template <int G, bool ACCUM>
__device__ __forceinline__ void slab_deriv_tile(const DenParams &p, int t, int slab, int ptile, const float (*tile)[G + 1],
                                                int wave, int lane) {
  const int p0 = ptile * 64, pdf = p0 + lane;
#pragma unroll
  for (int i = 0; i < G / 4; ++i) {
    const int sl = wave * (G / 4) + i, sq = slab * G + sl;
    if (sq < p.S && pdf < p.P) {
      const int64_t row = (int64_t)t * p.S + sq;
      float o = p.deriv_weight * tile[lane][sl] - p.l2_scale * p.y[row * p.y_stride + pdf];
      float *d = p.deriv + row * p.deriv_stride + pdf;
      if (ACCUM) o += *d;
      *d = o;
    }
  }
  if (t == 0 && wave == 0 && lane < G) {
    float g = 0.f;
    for (int i = 0; i < 64; ++i) g += tile[i][lane];
    big_small(p).part_g[((int64_t)slab * slab_pdf_tiles(p.P) + ptile) * G + lane] = g;
  }
}

// ---- tied graphs: exp(y) applied per state, one row gather per arc and pass ---------------------------
//   forward : alpha_t(g) * asum_{t-1} = p(f(g)) * sum_in w * alpha'_{t-1}(src) + p(s(g)) * w_s(g) * alpha'_{t-1}(g)
//             with sum_in w * alpha'(src) = sum_in w * alpha(src) + leaky * asum * K(g), K(g) = sum_in w * pi(src)
//   backward: Y(g) = beta_{t+1}(g) * p_t(f(g));  beta'_t(h) * asum_t = sum_out w * Y(dst) + p_t(s(h)) * w_s(h) * beta_{t+1}(h)
//   gamma   : from per-state quantities (den_kernels.hip, tied path): self-loop occupation
//             w_s * beta_{t+1}(g) * p_t(s) * alpha'_t(g) / asum_t -> pdf s(g); forward-class occupation
//             beta_{t+1}(g) * (alpha_{t+1}(g) - selfpart) -> pdf f(g), alpha_{t+1} being the stored un-dashed value.
template <int G>
__global__ __launch_bounds__(kBT) void slab_fwd_tied_kernel(const DenParams p, int t) {
  __shared__ float red[4][64];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const Lane<G> ln(threadIdx.x & 63);
  const int Sp = p.big_Sp, slabs = Sp / G, hb = p.big.hb;
  const SlabBlock sb = slab_block(p.big.in_blocks, slabs);
  if (!sb.ok) return;
  const BigSmall sm = big_small(p);
  const int s = sb.slab * G + (int)ln.j;
  const float *prev = p.alpha_hist + ((int64_t)(t - 1) * slabs + sb.slab) * p.H * G;
  float *cur = p.alpha_hist + ((int64_t)t * slabs + sb.slab) * p.H * G;
  const float *E = p.big_expy + p.big_exp_stride * (t - 1) + (int64_t)sb.slab * p.P * G;
  const float asum_prev = sm.asum[(int64_t)(t - 1) * Sp + s];
  const float inv = 1.0f / asum_prev, cl_as = p.leaky * asum_prev;
  const SlabListDev &L = p.big.in;
  float part = 0.f;
#pragma unroll 1
  for (int k = 0; k < G / 4; ++k) {
    const int b = sb.blk * G + wave + 4 * k;
    if (b >= L.bundles) break;
    const SlabRow r = load_row<G>(L, b, ln.q);
    const bool valid = r.row >= 0;
    const uint32_t hoff = (valid ? (uint32_t)r.row * (G * 4u) : 0u) + ln.j4;
    const float own = ld(prev, hoff);
    const float ef = r.f_off >= 0 ? ld(E, (uint32_t)r.f_off + ln.j4) : 0.f;
    const float es = r.s_off >= 0 ? ld(E, (uint32_t)r.s_off + ln.j4) : 0.f;
    GatherSum body{prev, ln.j4, 0.f};
    walk<G, 2>(L, b, ln, body);
    const float a_self = own + cl_as * r.pi;
    const float v = (ef * (body.sum + cl_as * r.K) + es * (r.ws * a_self)) * inv;
    if (valid) {
      st(cur, hoff, v);
      part += v;
    }
  }
  const float tot = block_rows_sum<G>(part, red, wave, ln.lane);
  if (wave == 0 && ln.lane < G) sm.part_a[((int64_t)sb.slab * hb + sb.blk) * G + ln.lane] = tot;
}

// Y(g) = beta_{t+1}(g) * p_t(f(g)) for every state: one group of G lanes per state
template <int G>
__global__ __launch_bounds__(kBT) void slab_y_kernel(const DenParams p, int t) {
  const int Sp = p.big_Sp, slabs = Sp / G;
  const int hb = (p.H + 63) / 64;
  const SlabBlock sb = slab_block(hb, slabs);
  if (!sb.ok) return;
  const BigSmall sm = big_small(p);
  const uint32_t j = threadIdx.x % G;
  const float bs = sm.bsum[((t + 1) & 1) * Sp + sb.slab * G + j];
  const float *Bprev = p.big_beta + ((int64_t)((t + 1) & 1) * slabs + sb.slab) * p.H * G;
  const float *E = p.big_expy + p.big_exp_stride * t + (int64_t)sb.slab * p.P * G;
  float *Y = p.big_y + (int64_t)sb.slab * p.H * G;
  constexpr int RPI = kBT / G;  // states per iteration
  int f[64 / RPI];
  float bv[64 / RPI];
#pragma unroll
  for (int k = 0; k < 64 / RPI; ++k) {
    const int h = sb.blk * 64 + k * RPI + (int)(threadIdx.x / G), hc = h < p.H ? h : p.H - 1;
    f[k] = p.big.f_off[hc];
    bv[k] = Bprev[(int64_t)hc * G + j];
  }
#pragma unroll
  for (int k = 0; k < 64 / RPI; ++k) {
    const int h = sb.blk * 64 + k * RPI + (int)(threadIdx.x / G);
    if (h < p.H) Y[(int64_t)h * G + j] = (bv[k] + bs) * (f[k] >= 0 ? ld(E, (uint32_t)f[k] + j * 4) : 0.f);
  }
}

template <int G>
__global__ __launch_bounds__(kBT) void slab_bwd_tied_kernel(const DenParams p, int t) {
  __shared__ float red[2][4][64];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const Lane<G> ln(threadIdx.x & 63);
  const int Sp = p.big_Sp, slabs = Sp / G, hb = p.big.hb;
  const SlabBlock sb = slab_block(p.big.out_blocks, slabs);
  if (!sb.ok) return;
  const BigSmall sm = big_small(p);
  const int s = sb.slab * G + (int)ln.j;
  const int64_t slab_states = (int64_t)p.H * G;
  const float *Bprev = p.big_beta + ((int64_t)((t + 1) & 1) * slabs + sb.slab) * slab_states;
  float *Bcur = p.big_beta + ((int64_t)(t & 1) * slabs + sb.slab) * slab_states;
  const float *Y = p.big_y + (int64_t)sb.slab * slab_states;
  const float *E = p.big_expy + p.big_exp_stride * t + (int64_t)sb.slab * p.P * G;
  const float *alpha = p.alpha_hist + ((int64_t)t * slabs + sb.slab) * slab_states;
  const float *alpha_up = p.alpha_hist + ((int64_t)(t + 1) * slabs + sb.slab) * slab_states;  // un-dashed alpha_{t+1}
  uint32_t *gam = p.big_gam + (int64_t)sb.slab * p.P * G;
  const float asum_t = sm.asum[(int64_t)t * Sp + s];
  const float inv_as = 1.0f / asum_t, bs = sm.bsum[((t + 1) & 1) * Sp + s], cl_as = p.leaky * asum_t;
  const SlabListDev &L = p.big.out;
  float part = 0.f, part_ab = 0.f;
#pragma unroll 1
  for (int k = 0; k < G / 4; ++k) {
    const int b = sb.blk * G + wave + 4 * k;
    if (b >= L.bundles) break;
    const SlabRow r = load_row<G>(L, b, ln.q);
    const bool valid = r.row >= 0;
    const uint32_t hoff = (valid ? (uint32_t)r.row * (G * 4u) : 0u) + ln.j4;
#ifdef TC_ABL_NOGAM
    const float own = ld(Bprev, hoff), al = 0.5f, up = 0.25f;
#else
    const float own = ld(Bprev, hoff), al = ld(alpha, hoff), up = ld(alpha_up, hoff);
#endif
    const float es = r.s_off >= 0 ? ld(E, (uint32_t)r.s_off + ln.j4) : 0.f;
    GatherSum body{Y, ln.j4, 0.f};
    walk<G, 2>(L, b, ln, body);
    const float beta = own + bs;
    const float bp = (body.sum + es * r.ws * beta) * inv_as;
    const float cpi = p.leaky * r.pi;
    if (valid) {
      st(Bcur, hoff, bp);
      part += cpi * bp;
      if (t == 0) part_ab += (al + cpi * asum_t) * bp;
      // gamma of frame t from this state: its self-loop under pdf s, everything else entering it under pdf f
      // (one L2 atomic per row segment and pdf; two sequences packed into one 64-bit add cost the same)
      const float selfpart = es * r.ws * (al + cl_as * r.pi) * inv_as;
      int32_t qs, qf;
      const float occ_s = beta * selfpart * kGammaScale, occ_f = beta * fmaxf(up - selfpart, 0.f) * kGammaScale;
      asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(qs) : "v"(occ_s));
      asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(qf) : "v"(occ_f));
#ifndef TC_ABL_NOATOM
      if (r.s_off >= 0) atomicAdd(reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(gam) + (uint32_t)r.s_off + ln.j4), (uint32_t)qs);
      if (r.f_off >= 0) atomicAdd(reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(gam) + (uint32_t)r.f_off + ln.j4), (uint32_t)qf);
#else
      if (qs + qf == 0x12345) part += 1.f;
#endif
    }
  }
  red[0][wave][ln.lane] = part;
  red[1][wave][ln.lane] = part_ab;
  __syncthreads();
  if (wave == 0 && ln.lane < G) {
    float a = 0.f, ab = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w)
#pragma unroll
      for (int g = 0; g < 64 / G; ++g) {
        a += red[0][w][g * G + ln.lane];
        ab += red[1][w][g * G + ln.lane];
      }
    sm.part_a[((int64_t)sb.slab * hb + sb.blk) * G + ln.lane] = a;
    if (t == 0) sm.part_ab[((int64_t)sb.slab * hb + sb.blk) * G + ln.lane] = ab;
  }
}

// the frame's gamma accumulators -> derivative rows (and cleared for the next frame)
// Blocks behind the grid of pdf tiles (frames t > 0: launch_slab) take the frame's other small job, bsum_t of one slab
// each -- one dependent launch less per frame.
template <int G, bool ACCUM>
__global__ __launch_bounds__(kBT) void slab_gamma_out_kernel(const DenParams p, int t, int tile_blocks) {
  __shared__ float tile[64][G + 1];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, tid = threadIdx.x;
  const int slabs = p.big_Sp / G, pb = slab_pdf_tiles(p.P);
  if ((int)blockIdx.x >= tile_blocks) {
    const int slab = (int)blockIdx.x - tile_blocks, hb = p.big.hb;
    const BigSmall sm = big_small(p);
    const float b = slab_colsum<G, kBT>(sm.part_a + (int64_t)slab * hb * G, p.big.out_blocks, &tile[0][0], tid);
    if (tid < G) sm.bsum[(t & 1) * p.big_Sp + slab * G + tid] = b;
    return;
  }
  const SlabBlock sb = slab_block(pb, slabs);
  if (!sb.ok) return;
  const int p0 = sb.blk * 64;
#pragma unroll
  for (int k = 0; k < G / 16; ++k) {
    const int e = tid + k * kBT, pl = e / (G / 4), jq = (e % (G / 4)) * 4;  // 64 pdfs x G / 4 quads
    if (p0 + pl < p.P) {
      uint4 *src = reinterpret_cast<uint4 *>(p.big_gam + ((int64_t)sb.slab * p.P + p0 + pl) * G + jq);
      const uint4 g = *src;
      *src = make_uint4(0u, 0u, 0u, 0u);
      tile[pl][jq] = (float)g.x * kGammaInvScale;
      tile[pl][jq + 1] = (float)g.y * kGammaInvScale;
      tile[pl][jq + 2] = (float)g.z * kGammaInvScale;
      tile[pl][jq + 3] = (float)g.w * kGammaInvScale;
    } else {
      tile[pl][jq] = tile[pl][jq + 1] = tile[pl][jq + 2] = tile[pl][jq + 3] = 0.f;
    }
  }
  __syncthreads();
  slab_deriv_tile<G, ACCUM>(p, t, sb.slab, sb.blk, tile, wave, lane);
}

// ---- general graphs: two row gathers per arc ----------------------------------------------------------------
// forward frame t: alpha_t(h) = sum_in w * alpha'_{t-1}(src) * p_{t-1}(pdf) / asum_{t-1}   ([K] AlphaGeneralFrame)
template <int G>
__global__ __launch_bounds__(kBT) void slab_fwd_kernel(const DenParams p, int t) {
  __shared__ float red[4][64];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const Lane<G> ln(threadIdx.x & 63);
  const int Sp = p.big_Sp, slabs = Sp / G, hb = p.big.hb;
  const SlabBlock sb = slab_block(p.big.in_blocks, slabs);
  if (!sb.ok) return;
  const BigSmall sm = big_small(p);
  const int s = sb.slab * G + (int)ln.j;
  const float *prev = p.alpha_hist + ((int64_t)(t - 1) * slabs + sb.slab) * p.H * G;
  float *cur = p.alpha_hist + ((int64_t)t * slabs + sb.slab) * p.H * G;
  const float *E = p.big_expy + p.big_exp_stride * (t - 1) + (int64_t)sb.slab * p.P * G;
  const float asum_prev = sm.asum[(int64_t)(t - 1) * Sp + s];
  const float inv = 1.0f / asum_prev, cl_as = p.leaky * asum_prev;
  const SlabListDev &L = p.big.in;
  float part = 0.f;
#pragma unroll 1
  for (int k = 0; k < G / 4; ++k) {
    const int b = sb.blk * G + wave + 4 * k;
    if (b >= L.bundles) break;
    const SlabRow r = load_row<G>(L, b, ln.q);
    GatherSum2 body{prev, E, ln.j4, cl_as, 0.f, 0.f};
    walk<G, 4>(L, b, ln, body);
    const float v = body.sum * inv;
    if (r.row >= 0) {
      cur[(int64_t)r.row * G + ln.j] = v;
      part += v;
    }
  }
  const float tot = block_rows_sum<G>(part, red, wave, ln.lane);
  if (wave == 0 && ln.lane < G) sm.part_a[((int64_t)sb.slab * hb + sb.blk) * G + ln.lane] = tot;
}

// backward frame t: beta'_t(h) = sum_out w * beta_{t+1}(dst) * p_t(pdf) / asum_t   ([K] BetaDashGeneralFrame)
template <int G>
__global__ __launch_bounds__(kBT) void slab_bwd_kernel(const DenParams p, int t) {
  __shared__ float red[2][4][64];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const Lane<G> ln(threadIdx.x & 63);
  const int Sp = p.big_Sp, slabs = Sp / G, hb = p.big.hb;
  const SlabBlock sb = slab_block(p.big.out_blocks, slabs);
  if (!sb.ok) return;
  const BigSmall sm = big_small(p);
  const int s = sb.slab * G + (int)ln.j;
  const int64_t slab_states = (int64_t)p.H * G;
  const float *Bprev = p.big_beta + ((int64_t)((t + 1) & 1) * slabs + sb.slab) * slab_states;
  float *Bcur = p.big_beta + ((int64_t)(t & 1) * slabs + sb.slab) * slab_states;
  const float *E = p.big_expy + p.big_exp_stride * t + (int64_t)sb.slab * p.P * G;
  const float *alpha = p.alpha_hist + ((int64_t)t * slabs + sb.slab) * slab_states;
  const float asum_t = sm.asum[(int64_t)t * Sp + s];
  const float inv_as = 1.0f / asum_t, bs = sm.bsum[((t + 1) & 1) * Sp + s];
  const SlabListDev &L = p.big.out;
  float part = 0.f, part_ab = 0.f;
#pragma unroll 1
  for (int k = 0; k < G / 4; ++k) {
    const int b = sb.blk * G + wave + 4 * k;
    if (b >= L.bundles) break;
    const SlabRow r = load_row<G>(L, b, ln.q);
    GatherSum2 body{Bprev, E, ln.j4, bs, 0.f, 0.f};  // (entries carry pi = 1: B[dst] + bs)
    walk<G, 4>(L, b, ln, body);
    const float bp = body.sum * inv_as;
    if (r.row >= 0) {
      const int64_t at = (int64_t)r.row * G + ln.j;
      Bcur[at] = bp;
      const float cpi = p.leaky * r.pi;
      part += cpi * bp;
      if (t == 0) part_ab += (alpha[at] + cpi * asum_t) * bp;
    }
  }
  red[0][wave][ln.lane] = part;
  red[1][wave][ln.lane] = part_ab;
  __syncthreads();
  if (wave == 0 && ln.lane < G) {
    float a = 0.f, ab = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w)
#pragma unroll
      for (int g = 0; g < 64 / G; ++g) {
        a += red[0][w][g * G + ln.lane];
        ab += red[1][w][g * G + ln.lane];
      }
    sm.part_a[((int64_t)sb.slab * hb + sb.blk) * G + ln.lane] = a;
    if (t == 0) sm.part_ab[((int64_t)sb.slab * hb + sb.blk) * G + ln.lane] = ab;
  }
}

// gamma_t(pdf) = p_t(pdf) / asum_t * sum over the arcs carrying pdf of w * alpha'_t(src) * beta_{t+1}(dst): one
// group of G lanes per pdf, a block = a tile of 64 consecutive pdfs (its bundles sorted by length inside the tile)
// ([K] BetaDashGeneralFrame's log_nnet_output_deriv part)
template <int G, bool ACCUM>
__global__ __launch_bounds__(kBT) void slab_gamma_kernel(const DenParams p, int t) {
  __shared__ float tile[64][G + 1];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const Lane<G> ln(threadIdx.x & 63);
  const int Sp = p.big_Sp, slabs = Sp / G, pb = slab_pdf_tiles(p.P);
  const SlabBlock sb = slab_block(pb, slabs);
  if (!sb.ok) return;
  const BigSmall sm = big_small(p);
  const int s = sb.slab * G + (int)ln.j;
  const int64_t slab_states = (int64_t)p.H * G;
  const float *alpha = p.alpha_hist + ((int64_t)t * slabs + sb.slab) * slab_states;
  const float *Bprev = p.big_beta + ((int64_t)((t + 1) & 1) * slabs + sb.slab) * slab_states;
  const float *E = p.big_expy + p.big_exp_stride * t + (int64_t)sb.slab * p.P * G;
  const float asum_t = sm.asum[(int64_t)t * Sp + s];
  const float inv_as = 1.0f / asum_t, bs = sm.bsum[((t + 1) & 1) * Sp + s], cl_as = p.leaky * asum_t;
  const SlabListDev &L = p.big.pdf;
  if (sb.blk * 64 + 64 > p.P) {  // a tile's last pdfs beyond P have no row
    for (int e = threadIdx.x; e < 64 * (G + 1); e += kBT) (&tile[0][0])[e] = 0.f;
    __syncthreads();
  }
#pragma unroll 1
  for (int k = 0; k < G / 4; ++k) {
    const int b = sb.blk * G + wave + 4 * k;  // (the by-pdf list always has G bundles per tile)
    const SlabRow r = load_row<G>(L, b, ln.q);
    GatherSum2 body{alpha, Bprev, ln.j4, cl_as, bs, 0.f};
    walk<G, 4>(L, b, ln, body);
    if (r.row >= 0) tile[r.row - sb.blk * 64][ln.j] = body.sum * E[(int64_t)r.row * G + ln.j] * inv_as;
  }
  __syncthreads();
  slab_deriv_tile<G, ACCUM>(p, t, sb.slab, sb.blk, tile, wave, (int)ln.lane);
}

template <int G>
int launch_slab(const DenParams &p, int accumulate, hipStream_t stream) {
  const int Sp = p.big_Sp, slabs = Sp / G;
  const dim3 blk(kBT);
  const int pb = slab_pdf_tiles(p.P);
  const dim3 g_exp((pb + 3) / 4, slabs);
  const int g_in = slab_grid(p.big.in_blocks, slabs), g_out = slab_grid(p.big.out_blocks, slabs);
  const int g_pdfs = slab_grid(pb, slabs), g_y = slab_grid((p.H + 63) / 64, slabs);
  const int fill_blocks = (int)std::min<int64_t>(4096, ((int64_t)p.H * Sp + kBT - 1) / kBT);
  const bool tied = p.big.tied != 0;
  const bool exp_all = p.big_exp_stride != 0;
  hipLaunchKernelGGL(slab_alpha0_kernel<G>, dim3(fill_blocks), blk, 0, stream, p);
  if (exp_all) hipLaunchKernelGGL(slab_exp_kernel<G>, g_exp, blk, 0, stream, p, 0, p.T, 1);
  for (int t = 1; t <= p.T; ++t) {
    if (!exp_all) hipLaunchKernelGGL(slab_exp_kernel<G>, g_exp, blk, 0, stream, p, t - 1, t, 1);
    if (tied)
      hipLaunchKernelGGL(slab_fwd_tied_kernel<G>, dim3(g_in), blk, 0, stream, p, t);
    else
      hipLaunchKernelGGL(slab_fwd_kernel<G>, dim3(g_in), blk, 0, stream, p, t);
    hipLaunchKernelGGL(slab_asum_kernel<G>, dim3(slabs), dim3(kRT), 0, stream, p, t);
  }
  hipLaunchKernelGGL(slab_total_kernel<G>, dim3((Sp + kBT - 1) / kBT), blk, 0, stream, p);
  if (p.deriv) {
    hipLaunchKernelGGL(slab_beta_init_kernel<G>, dim3(fill_blocks), blk, 0, stream, p);
    for (int t = p.T - 1; t >= 0; --t) {
      if (!exp_all) hipLaunchKernelGGL(slab_exp_kernel<G>, g_exp, blk, 0, stream, p, t, t + 1, 0);
      if (tied) {
        hipLaunchKernelGGL(slab_y_kernel<G>, dim3(g_y), blk, 0, stream, p, t);
        hipLaunchKernelGGL(slab_bwd_tied_kernel<G>, dim3(g_out), blk, 0, stream, p, t);
        // (frame 0's column sums also need this launch's gamma sums: their own launch below)
        const int extra = t > 0 ? slabs : 0;
        if (accumulate)
          hipLaunchKernelGGL((slab_gamma_out_kernel<G, true>), dim3(g_pdfs + extra), blk, 0, stream, p, t, g_pdfs);
        else
          hipLaunchKernelGGL((slab_gamma_out_kernel<G, false>), dim3(g_pdfs + extra), blk, 0, stream, p, t, g_pdfs);
        if (t > 0) continue;
      } else {
        hipLaunchKernelGGL(slab_bwd_kernel<G>, dim3(g_out), blk, 0, stream, p, t);
        if (accumulate)
          hipLaunchKernelGGL((slab_gamma_kernel<G, true>), dim3(g_pdfs), blk, 0, stream, p, t);
        else
          hipLaunchKernelGGL((slab_gamma_kernel<G, false>), dim3(g_pdfs), blk, 0, stream, p, t);
      }
      hipLaunchKernelGGL(slab_bsum_kernel<G>, dim3(slabs), dim3(kRT), 0, stream, p, t);
    }
  }
  TC_HIP_CHECK(hipGetLastError());
  return TC_OK;
}

}  // namespace

// floats of p.big_small for this problem size (api.cpp sizes the workspace with it)
int64_t big_small_floats(int hb, int P, int T, int Sp) {
  const int64_t pb = slab_pdf_tiles(P);
  return ((int64_t)(T + 1) + 2 + 1 + 2 * hb + 2 * pb) * Sp;
}

int launch_den_big(const DenParams &p, int accumulate, hipStream_t stream) {
  return p.big.G == 32 ? launch_slab<32>(p, accumulate, stream) : launch_slab<16>(p, accumulate, stream);
}

}  // namespace tc
