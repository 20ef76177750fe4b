// Device helpers of the tied-graph kernels (den_tied_kernel.hip: the fused forward-backward; den_tied_split.hip:
// the two-CU form for small batches): LDS access by absolute address, buffer-descriptor global access, the
// 6-byte cell stream, the arc walk with its register-resident prefix, row commits.
#pragma once

#include "den_device.h"

namespace tc {

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) float lds_f;
typedef __attribute__((address_space(3))) uint32_t lds_u;
typedef __attribute__((address_space(3))) f4 lds_f4;
typedef __attribute__((address_space(3))) u4 lds_u4;

// LDS by absolute byte address: the kernel has no static __shared__, so the dynamic block starts at 0 and
// constant parts of an address fold into the ds instruction's immediate offset.
__device__ __forceinline__ float ldsf(uint32_t a) { return *reinterpret_cast<lds_f *>(a); }
__device__ __forceinline__ void ldsf_st(uint32_t a, float v) { *reinterpret_cast<lds_f *>(a) = v; }
__device__ __forceinline__ f4 lds4(uint32_t a) { return *reinterpret_cast<lds_f4 *>(a); }
__device__ __forceinline__ void lds4_st(uint32_t a, f4 v) { *reinterpret_cast<lds_f4 *>(a) = v; }
// ds_write_b128 base + compile-time offset, the offset pinned into the instruction's immediate field: left to
// the compiler, "16 * tid + constant" addresses of the per-frame tail become loop-invariant VGPRs of their own, and
// in the backward loop of the full kernel those are what it spills (reloaded behind a vmcnt(0) every frame).
__device__ __forceinline__ void lds4_st_at(uint32_t base, uint32_t off, f4 v) {
  asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(base), "v"(v), "i"(off) : "memory");
}
__device__ __forceinline__ u4 lds4u(uint32_t a) { return *reinterpret_cast<lds_u4 *>(a); }
__device__ __forceinline__ void lds_add_u32(uint32_t a, uint32_t v) {
  __hip_atomic_fetch_add(reinterpret_cast<lds_u *>(a), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Global memory through buffer descriptors: a wave-uniform 128-bit resource in SGPRs + one 32-bit VGPR
// byte offset (the same VGPR for every table: 16 * tid or 16 * lane) + an SGPR / immediate offset.  With
// flat addressing the compiler materialises (and, being loop-invariant, hoists) one 64-bit VGPR address per
// load -- registers this kernel needs for the resident stream.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void *ubase, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(ubase), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ u4 bld4u(rsrc_t r, uint32_t voff, uint32_t soff) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
}
__device__ __forceinline__ f4 bld4(rsrc_t r, uint32_t voff, uint32_t soff) {
  const u4 v = bld4u(r, voff, soff);
  return f4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
}
// ... with the cache policy chosen by the caller (0: default, 2: nt -- a row nobody reads again soon)
template <int AUX>
__device__ __forceinline__ f4 bld4_aux(rsrc_t r, uint32_t voff, uint32_t soff) {
  const u4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, AUX);
  return f4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
}
// Stores never put their offset into the SGPR soffset field.  Measured on gfx950 (this kernel's history rows, 2nd
// plane): `buffer_store_dwordx4 v[76:79], v106, s[40:43], s0 offen` followed directly by a VALU write of v76
// stored corrupted data -- the >64-bit store-data hazard.  The compiler pads that hazard with s_nop only when
// soffset is an immediate (it takes the hardware to be safe when soffset is a register), so the offset goes
// into the VGPR and soffset stays 0.
__device__ __forceinline__ void bst4(rsrc_t r, uint32_t voff, f4 v) {
#ifndef TC_STORE_AUX
#define TC_STORE_AUX 2  /* nt: streaming stores, -1 % at C3 */
#endif
  __builtin_amdgcn_raw_buffer_store_b128(u4{__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)},
                                         r, (int)voff, 0, TC_STORE_AUX);
}

// the same with the cache policy chosen by the caller (0: default, 2: nt)
template <int AUX>
__device__ __forceinline__ void bst4_aux(rsrc_t r, uint32_t voff, f4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(u4{__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)},
                                         r, (int)voff, 0, AUX);
}

__device__ __forceinline__ f4 mk4(float x) { return f4{x, x, x, x}; }
__device__ __forceinline__ float hsum(f4 v) { return (v.x + v.y) + (v.z + v.w); }
// exp(clamp(y, -30, 30)) ([K] later Kaldi: ApplyExpLimited; equal to the 22fbdd ApplyExp() for |y| < 30) with the
// clamp as one v_med3_f32.  That clamp would turn a NaN input into exp(-30); the sum of y^2, which this kernel
// forms anyway, carries every NaN / inf of the row into the sequence's log-prob (see seq_logprob below), so the
// objective still fails softly as in [K].
__device__ __forceinline__ float exp_med3(float x) { return __expf(__builtin_amdgcn_fmed3f(x, -30.0f, 30.0f)); }
__device__ __forceinline__ f4 exp4(f4 y) { return f4{exp_med3(y.x), exp_med3(y.y), exp_med3(y.z), exp_med3(y.w)}; }

// one row of y or of the derivative through its descriptor (num_records = the row's bytes: reads past the
// row return 0, writes past it are dropped): a 16-byte access when the caller's rows are 16-byte aligned,
// else four dword accesses
__device__ __forceinline__ f4 row_ld(rsrc_t r, uint32_t voff, int vec) {
  if (vec) return bld4(r, voff, 0);
  return f4{__uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, 0, 0)),
            __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, 4, 0)),
            __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, 8, 0)),
            __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, 12, 0))};
}
__device__ __forceinline__ void row_st(rsrc_t r, uint32_t voff, int vec, f4 v) {
  if (vec) {
    bst4(r, voff, v);
  } else {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v.x), r, (int)voff, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v.y), r, (int)voff, 4, 0);
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v.z), r, (int)voff, 8, 0);
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v.w), r, (int)voff, 12, 0);
  }
}

__device__ __forceinline__ float block_sum_a(float v, uint32_t red, int wave, uint32_t lane) {
  v = wave_sum(v);
  if (lane == 0) ldsf_st(red + 4u * (uint32_t)wave, v);
  __syncthreads();
  static_assert(kWaves == 16, "one DPP row holds the wave totals");
  float t = ldsf(red + 4u * (lane & 15u));
  t = dpp_add<0xB1>(t);
  t = dpp_add<0x4E>(t);
  t = dpp_add<0x124>(t);
  t = dpp_add<0x128>(t);
  return t;
}

__device__ __forceinline__ void gamma_add_a(uint32_t addr, float v) {
  int32_t q;  // floor(v + 0.5), v >= 0 (den_device.h: gamma_add)
  asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(q) : "v"(v));
  lds_add_u32(addr, (uint32_t)q);
}

// ---- the cell stream ---------------------------------------------------------------------------------
// A chunk is 8 cells of one lane: {w0..w3}, {w4..w7}, {off01, off23, off45, off67} with fp32 weights and
// 16-bit LDS byte offsets of the gathers; stored [chunk][3 blocks][lane]{16 bytes} (schedule_owner.cpp), so
// a chunk is three coalesced dwordx4 loads.
struct Chunk6 {
  u4 wa, wb, oc;
};

__device__ __forceinline__ void load_chunk(Chunk6 &q, rsrc_t stream, uint32_t lane16, int chunk) {
  const uint32_t so = (uint32_t)chunk * (3 * 64 * 16);
  q.wa = bld4u(stream, lane16, so);
  q.wb = bld4u(stream, lane16, so + 1024);
  q.oc = bld4u(stream, lane16, so + 2048);
}

// The unpacking of resident cells is loop-invariant over the frames; left to the compiler it is hoisted
// and every resident cell then holds a second register.  volatile asm pins it to its use.
__device__ __forceinline__ uint32_t lo16(uint32_t x) {
  uint32_t r;
  asm volatile("v_and_b32 %0, 0xffff, %1" : "=v"(r) : "v"(x));
  return r;
}
__device__ __forceinline__ uint32_t hi16(uint32_t x) {
  uint32_t r;
  asm volatile("v_lshrrev_b32 %0, 16, %1" : "=v"(r) : "v"(x));
  return r;
}

// Plane-wise form (den_tied_planes.hip, graphs beyond 16384 positions): the cell's 16-bit field is the POSITION of its
// source, and the byte offset is field << 2 -- word select and shift in one SDWA instruction, so the unpacking still
// costs one VALU instruction per cell.
__device__ __forceinline__ uint32_t lo16w(uint32_t x) {
  uint32_t r;
  const uint32_t two = 2u;
  asm volatile("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(r) : "s"(two), "v"(x));
  return r;
}
__device__ __forceinline__ uint32_t hi16w(uint32_t x) {
  uint32_t r;
  const uint32_t two = 2u;
  asm volatile("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(r) : "s"(two), "v"(x));
  return r;
}

// Where a finished row sum goes.  All 64 lanes of a wave are at the same row index k, so the row sums are
// kept [row][lane]: own rows of wave w at rows w * K + k, secondary rows (k >= K, hub states only) behind
// them.  A commit is then ds_write_addtid_b32 -- address = M0 + 4 * lane, no address VGPR (a VGPR that is only
// read in the rare commit block is the first thing the register allocator spills, and its reload would sit in
// the middle of the walk behind a vmcnt(0)), 256 contiguous bytes per wave, twice the rate of ds_write_b32 --
// and the cursor is one scalar.
struct RowCommit {
  uint32_t row;       // byte address of lane 0's slot of the current row
  uint32_t sec_row;   // ... of the wave's first secondary row
  int left;           // own rows still to come
  __device__ __forceinline__ void commit(float v) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tds_write_addtid_b32 %0" : : "v"(v), "s"(row) : "memory", "m0");
    --left;
    row = left == 0 ? sec_row : row + 256u;
  }
};

// acc(row) += w * SRC[off] over one chunk.  Row ends are wave-uniform bits of the schedule's mask word m
// (one word per two chunks; bit u: a row ends with the SECOND cell of pair u, bit 8 + u: with its FIRST
// cell), tested with s_bitcmp; the commit is the rare side of a scalar branch.  (Measured alternatives,
// profiles/microbench/walk_variants.hip: rows padded to quads with one test per quad and packed FMAs run 8 %
// faster per cell but need 14-21 % more cells; issuing the next chunk's gathers ahead of this chunk's sums
// gains nothing; without any row ends the same loop would run 1.6x faster -- the commits, two taken branches
// each, are what the rows cost.)
template <uint32_t SRC, int HALF>
__device__ __forceinline__ void do_chunk(const Chunk6 &q, uint32_t m, float &acc, RowCommit &rc) {
  uint32_t o[8];
  o[0] = lo16(q.oc.x);
  o[1] = hi16(q.oc.x);
  o[2] = lo16(q.oc.y);
  o[3] = hi16(q.oc.y);
  o[4] = lo16(q.oc.z);
  o[5] = hi16(q.oc.z);
  o[6] = lo16(q.oc.w);
  o[7] = hi16(q.oc.w);
  float a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
#ifdef TC_ABL_NOGATHER
    a[i] = __uint_as_float(o[i]);
#else
    a[i] = ldsf(SRC + o[i]);
#endif
  }
  const uint32_t w[8] = {q.wa.x, q.wa.y, q.wa.z, q.wa.w, q.wb.x, q.wb.y, q.wb.z, q.wb.w};
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    acc = fmaf(a[i], __uint_as_float(w[i]), acc);
    const int bit = (i & 1) ? 4 * HALF + i / 2 : 8 + 4 * HALF + i / 2;
    if (__builtin_expect((m >> bit) & 1u, 0)) {
      rc.commit(acc);
      acc = 0.f;
    }
  }
}

// the mask words of a wave's stream, word i in lane i (streams of up to 128 chunks: far beyond what fits on chip)
__device__ __forceinline__ uint32_t wave_masks(const ScheduleDev &sc, int wave, uint32_t lane) {
  const uint32_t i = lane < (uint32_t)sc.mask_stride ? lane : (uint32_t)sc.mask_stride - 1u;
  return sc.masks[(size_t)wave * sc.mask_stride + i];
}

// One walk of a wave's stream: RES resident chunks, then the rest through two register buffers in
// ping-pong (qa arrives preloaded with chunk RES when there is one; the stream is followed by readable
// padding, so the look-ahead loads need no guard).  The mask words come through the scalar cache.
// `after_chunk(i)` runs after resident chunk i; when nothing is resident, with i = -1 - n after the n-th pair of
// streamed chunks and finally with kWalkEnd: the hook through which the frame's global STORES are spread over the
// walk (see the kernels).
constexpr int kWalkEnd = -1000;  // last call of a walk's hook when nothing is resident: whatever is still pending

template <uint32_t SRC, int RES, class AfterChunk>
__device__ __forceinline__ void walk(const Chunk6 (&res)[RES > 0 ? RES : 1], Chunk6 &qa, rsrc_t sbase,
                                     uint32_t lane16, int nchunks, uint32_t vmask, RowCommit rc,
                                     AfterChunk after_chunk TC_WALK_ARG) {
  static_assert(RES % 2 == 0, "a mask word covers two chunks");
  // The wave's mask words sit in ONE vector register, word i in lane i (wave_masks() below), and come out through
  // v_readlane.  Read from memory where they were needed (one s_load_dword per two chunks), every second chunk lost the
  // counted waits on its gathers: scalar loads return out of order with LDS operations, so with one in flight the first
  // wait of the chunk is lgkmcnt(0) -- all eight gathers and a scalar-cache round trip before the first FMA.
  auto mk = [&](int i) { return (uint32_t)__builtin_amdgcn_readlane((int)vmask, i); };
  float acc = 0.f;
#ifdef TC_PHASE_STAMPS
  wst[2] = clock64();
#endif
#pragma unroll
  for (int i = 0; i < RES / 2; ++i) {
    const uint32_t m = mk(i);
    do_chunk<SRC, 0>(res[2 * i], m, acc, rc);
    after_chunk(2 * i);
    do_chunk<SRC, 1>(res[2 * i + 1], m, acc, rc);
    after_chunk(2 * i + 1);
  }
#ifdef TC_PHASE_STAMPS
  {
    __builtin_amdgcn_s_waitcnt(0xC07F); /* lgkmcnt(0) */
    const long long now = clock64();
    wst[0] += now - wst[2];  // resident part
    wst[2] = now;
  }
#endif
  // Streamed part.  Inside the loop both look-ahead loads are unconditional, so the compiler knows how many
  // loads are in flight when it waits for a buffer (a load behind a condition makes it fall back to vmcnt(0),
  // which serialises every chunk behind an L2 round trip); the last one or two chunks are peeled.
  Chunk6 qb;
  int c = RES;
  for (; c + 2 < nchunks; c += 2) {
    const uint32_t m = mk(c >> 1);
    load_chunk(qb, sbase, lane16, c + 1);
    do_chunk<SRC, 0>(qa, m, acc, rc);
    load_chunk(qa, sbase, lane16, c + 2);
    do_chunk<SRC, 1>(qb, m, acc, rc);
    if (RES == 0) after_chunk(-1 - (c >> 1));
  }
  if (c + 1 < nchunks) {
    const uint32_t m = mk(c >> 1);
    load_chunk(qb, sbase, lane16, c + 1);
    do_chunk<SRC, 0>(qa, m, acc, rc);
    do_chunk<SRC, 1>(qb, m, acc, rc);
  } else if (c < nchunks) {
    do_chunk<SRC, 0>(qa, mk(c >> 1), acc, rc);
  }
  if (RES == 0) after_chunk(kWalkEnd);
#ifdef TC_PHASE_STAMPS
  __builtin_amdgcn_s_waitcnt(0xC07F);
  wst[1] += clock64() - wst[2];  // streamed part
#endif
}

// Plane-wise form (den_tied_planes.hip): acc(row) += w * SRC[4 * position] over one chunk; m8 = the chunk's row-end byte
// (bit i: a row ends with cell i -- schedule_owner.cpp keeps one byte per chunk for this form).
template <uint32_t SRC>
__device__ __forceinline__ void chunk_pw(const Chunk6 &q, uint32_t m8, float &acc, RowCommit &rc) {
  uint32_t o[8];
  o[0] = lo16w(q.oc.x);
  o[1] = hi16w(q.oc.x);
  o[2] = lo16w(q.oc.y);
  o[3] = hi16w(q.oc.y);
  o[4] = lo16w(q.oc.z);
  o[5] = hi16w(q.oc.z);
  o[6] = lo16w(q.oc.w);
  o[7] = hi16w(q.oc.w);
  float a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = ldsf(SRC + o[i]);
  const uint32_t w[8] = {q.wa.x, q.wa.y, q.wa.z, q.wa.w, q.wb.x, q.wb.y, q.wb.z, q.wb.w};
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    acc = fmaf(a[i], __uint_as_float(w[i]), acc);
    if (__builtin_expect((m8 >> i) & 1u, 0)) {
      rc.commit(acc);
      acc = 0.f;
    }
  }
}

// The CU serves older waves first wherever waves contend, so the youngest wave of each SIMD finishes its
// walk last and every frame waits for it: during the walks the four wave generations run at issue
// priorities 0..3, youngest highest.
__device__ __forceinline__ void age_prio_on(int wave) {
  if (wave >= 12)
    __builtin_amdgcn_s_setprio(3);
  else if (wave >= 8)
    __builtin_amdgcn_s_setprio(2);
  else if (wave >= 4)
    __builtin_amdgcn_s_setprio(1);
}

// fs = forward-pdf*4 | self-loop-pdf*4 << 16 (LDS byte offsets into exp(y)), ws = self-loop probability
__device__ __forceinline__ float tied_alpha(uint32_t pb, uint32_t fs, float ws, float F, float a_self) {
  return fmaf(ldsf(pb + (fs & 0xffffu)), F, ldsf(pb + (fs >> 16)) * (ws * a_self));
}

// row sums of the four states of a thread's float4 in plane j: rows 4j .. 4j+3 of its wave
__device__ __forceinline__ f4 own_rows(uint32_t vrow, int j) {
  return f4{ldsf(vrow + 256u * (4 * j)), ldsf(vrow + 256u * (4 * j + 1)), ldsf(vrow + 256u * (4 * j + 2)),
            ldsf(vrow + 256u * (4 * j + 3))};
}

// Graphs with hub states: adds a secondary row's sum to its state's own row.  f = {position of the state (owned
// by this thread), logical slot Hs + 4 + 64 * e + lane of the secondary row} as the schedule builder numbers them.
__device__ __forceinline__ void fold_row(int2 f, uint32_t vrow, uint32_t aACC, int Hs, int K) {
  const int k = 4 * (f.x / (4 * kThreads)) + (f.x & 3);
  const uint32_t dst = vrow + 256u * (uint32_t)k;
  const uint32_t src = aACC + 256u * (uint32_t)(K * kWaves) + 4u * (uint32_t)(f.y - Hs - 4);
  ldsf_st(dst, ldsf(dst) + ldsf(src));
}

// ... plane-wise form: the four rows of the plane in progress are the wave's rows 0..3
__device__ __forceinline__ void fold_row_pw(int2 f, uint32_t vrow, uint32_t aACC, int Hs) {
  const uint32_t dst = vrow + 256u * (uint32_t)(f.x & 3);
  const uint32_t src = aACC + 256u * (uint32_t)(4 * kWaves) + 4u * (uint32_t)(f.y - Hs - 4);
  ldsf_st(dst, ldsf(dst) + ldsf(src));
}

constexpr uint32_t kPlane = 16u * kThreads;  // bytes between a thread's float4s of consecutive planes

// The frame sums asum_0..T of one sequence: behind the layout's off_asum in LDS, or -- an utterance too long for the LDS the
// graph leaves (DenLayout::asum_global) -- in the workspace.  Written by thread 0 in the forward phase, read by every
// thread of the SAME workgroup in the backward phase (a vector load: one address, one request).
struct AsumRow {
  uint32_t lds;
  float *g;  // null: LDS
  __device__ __forceinline__ AsumRow(const DenParams &p, int s)
      : lds(4u * (uint32_t)p.L.off_asum), g(p.L.asum_global ? p.asum_g + (int64_t)s * asum_stride(p.T) : nullptr) {}
  __device__ __forceinline__ void st(int t, float v) const {
    if (g)
      g[t] = v;
    else
      ldsf_st(lds + 4u * (uint32_t)t, v);
  }
  __device__ __forceinline__ float ld(int t) const {
    if (g) return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(make_rsrc(g + t, 4u), 0, 0, 0));
    return ldsf(lds + 4u * (uint32_t)t);
  }
};

}  // namespace

// every <JV, PV> instantiation of the tied kernels: CALL(JV, PV) for the layout's pair (den_layout.cpp)
#define TC_TIED_DISPATCH(CALL)                            \
  if (JV == kJvSmall && PV == kPvSmall) return CALL(kJvSmall, kPvSmall); \
  if (JV == kJvSmall && PV == kPvMid) return CALL(kJvSmall, kPvMid);     \
  if (JV == kJvSmall && PV == kPvLarge) return CALL(kJvSmall, kPvLarge); \
  if (JV == kJvMid && PV == kPvSmall) return CALL(kJvMid, kPvSmall);     \
  if (JV == kJvMid && PV == kPvMid) return CALL(kJvMid, kPvMid);         \
  if (JV == kJvMid && PV == kPvLarge) return CALL(kJvMid, kPvLarge);     \
  if (JV == kJvLarge && PV == kPvSmall) return CALL(kJvLarge, kPvSmall); \
  if (JV == kJvLarge && PV == kPvMid) return CALL(kJvLarge, kPvMid);     \
  if (JV == kJvLarge && PV == kPvLarge) return CALL(kJvLarge, kPvLarge);

}  // namespace tc
