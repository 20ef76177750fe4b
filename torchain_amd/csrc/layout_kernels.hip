// Layout conversion either side of the hot path: the reference's recipe hands
// chain_loss a (B, C, T) tensor and torchain/functions.py:118-125 (`to2d`) turns it into frame-major rows
// with x.permute(2,0,1).contiguous(); on the way back autograd undoes the permutation and
// functions.py:112 negates the gradient -- three more passes over a 629 MB tensor at C3.  These two
// kernels do the forward permutation in one pass and the backward permutation, the negation and the
// xent scale in one pass.  Both are plain HBM-bound transposes through LDS: a workgroup moves a
// [64 channels] x [<= 240 frames] tile of one sequence, reading and writing 256-byte (or longer) runs; the
// backward one reads its 2-D rows with 16-byte loads and writes 8-byte pairs where alignment allows.
#include "chain_internal.h"

namespace tc {

namespace {

constexpr int kTileC = 64, kTileTMax = 240, kLayoutThreads = 512;

// TO2D: in (B, C, T) contiguous -> out[(t*B + b) * out_stride + c];  !TO2D: the inverse, times `scale`
template <bool TO2D, bool VEC>
__global__ __launch_bounds__(kLayoutThreads) void layout_kernel(const float *__restrict__ in, float *__restrict__ out,
                                                                int B, int Cn, int T, int64_t stride2d, float scale) {
  extern __shared__ float tile[];  // kTileC x (frames of the tile | 1): sized by the launch, several blocks per CU
  const int c0 = blockIdx.x * kTileC, b = blockIdx.y, t0 = blockIdx.z * kTileTMax;
  const int tc = min(kTileTMax, T - t0), nc = min(kTileC, Cn - c0);
  const int pitch = tc | 1;  // odd pitch: the transposed accesses are bank-conflict-free
  const int tid = threadIdx.x;
  const float *bct = TO2D ? in : out;  // the (B, C, T) side
  (void)bct;
  if (TO2D) {
    // read runs of tc floats along t (for tc == T the whole tile is one contiguous block); dword accesses on both
    // sides: 16-byte row stores and 8-byte run loads were measured here and are slower (0.30 vs 0.26 ms at C3)
    for (int idx = tid; idx < nc * tc; idx += kLayoutThreads) {
      const int c = idx / tc, t = idx - c * tc;
      tile[c * pitch + t] = in[((int64_t)b * Cn + c0 + c) * T + t0 + t];
    }
    __syncthreads();
    const int cx = tid & (kTileC - 1);
    for (int t = tid / kTileC; t < tc; t += kLayoutThreads / kTileC)
      if (cx < nc) out[((int64_t)(t0 + t) * B + b) * stride2d + c0 + cx] = tile[cx * pitch + t];
  } else {
    // Read side: 16-byte loads when the 2-D rows are 16-byte aligned -- a wave then reads four 256-byte row
    // segments per instruction instead of one (the dword form left this kernel at 41 % of the HBM peak).
    if (VEC) {
      const int cq = (tid & 15) * 4, tr = tid >> 4;  // 16 threads per row segment, 32 rows per pass
      for (int t = tr; t < tc; t += kLayoutThreads / 16) {
        const float *src = in + ((int64_t)(t0 + t) * B + b) * stride2d + c0 + cq;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (cq + 3 < nc) {
          v = *reinterpret_cast<const float4 *>(src);
        } else {
          if (cq < nc) v.x = src[0];
          if (cq + 1 < nc) v.y = src[1];
          if (cq + 2 < nc) v.z = src[2];
        }
        tile[(cq + 0) * pitch + t] = v.x;
        tile[(cq + 1) * pitch + t] = v.y;
        tile[(cq + 2) * pitch + t] = v.z;
        tile[(cq + 3) * pitch + t] = v.w;
      }
    } else {
      const int cx = tid & (kTileC - 1);
      for (int t = tid / kTileC; t < tc; t += kLayoutThreads / kTileC)
        if (cx < nc) tile[cx * pitch + t] = in[((int64_t)(t0 + t) * B + b) * stride2d + c0 + cx];
    }
    __syncthreads();
    // Write side: runs of tc floats per channel; 8-byte stores when every run starts 8-byte aligned
    if (VEC && (T & 1) == 0 && (tc & 1) == 0) {
      const int half = tc / 2;
      for (int idx = tid; idx < nc * half; idx += kLayoutThreads) {
        const int c = idx / half, t = 2 * (idx - c * half);
        *reinterpret_cast<float2 *>(out + ((int64_t)b * Cn + c0 + c) * T + t0 + t) =
            make_float2(scale * tile[c * pitch + t], scale * tile[c * pitch + t + 1]);
      }
    } else {
      for (int idx = tid; idx < nc * tc; idx += kLayoutThreads) {
        const int c = idx / tc, t = idx - c * tc;
        out[((int64_t)b * Cn + c0 + c) * T + t0 + t] = scale * tile[c * pitch + t];
      }
    }
  }
}

// sum_{r,c} a[r][c] * b[r][c] in double: per-block partials in a fixed order, then one block adds them in
// index order -- deterministic.  [K] TraceMatMat(xent_output, xent_deriv, kTrans), the cross-entropy objective
// Kaldi's chain trainer reports (the reference leaves it as a TODO, torchain/functions.py:88-89).
constexpr int kTraceBlocks = 512, kTraceThreads = 256;
__global__ __launch_bounds__(kTraceThreads) void trace_partial_kernel(const float *__restrict__ a, int64_t a_stride,
                                                                      const float *__restrict__ b, int64_t b_stride,
                                                                      int64_t rows, int cols, double *partial) {
  __shared__ double sh[kTraceThreads];
  double acc = 0.0;
  for (int64_t r = blockIdx.x; r < rows; r += kTraceBlocks) {
    const float *ar = a + r * a_stride, *br = b + r * b_stride;
    float row = 0.f;
    for (int c = threadIdx.x; c < cols; c += kTraceThreads) row = fmaf(ar[c], br[c], row);
    acc += (double)row;
  }
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int w = kTraceThreads / 2; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = sh[0];
}
__global__ __launch_bounds__(64) void trace_final_kernel(const double *partial, double *out) {
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < kTraceBlocks; ++i) t += partial[i];
    *out = t;
  }
}

}  // namespace

int64_t trace_workspace_bytes() { return (int64_t)kTraceBlocks * 8; }

int launch_trace_mat_mat(const float *a, int64_t a_stride, const float *b, int64_t b_stride, int64_t rows, int cols,
                         double *partial, double *out, hipStream_t stream) {
  hipLaunchKernelGGL(trace_partial_kernel, dim3(kTraceBlocks), dim3(kTraceThreads), 0, stream, a, a_stride, b, b_stride,
                     rows, cols, partial);
  hipLaunchKernelGGL(trace_final_kernel, dim3(1), dim3(64), 0, stream, partial, out);
  TC_HIP_CHECK(hipGetLastError());
  return TC_OK;
}

int launch_layout(bool to2d, const float *in, float *out, int B, int Cn, int T, int64_t stride2d, float scale,
                  hipStream_t stream) {
  count_launch(kCntLayout);
  const dim3 grid((Cn + kTileC - 1) / kTileC, B, (T + kTileTMax - 1) / kTileTMax);
  const size_t lds = sizeof(float) * kTileC * ((size_t)(T < kTileTMax ? T : kTileTMax) | 1);
  // 16-byte path of the 2-D side: rows and base 16-byte aligned, whole float4s of channels in every tile
  const bool vec = stride2d % 4 == 0 && ((uintptr_t)(to2d ? out : in) & 15) == 0 && ((uintptr_t)(to2d ? in : out) & 7) == 0;
  if (to2d)
    hipLaunchKernelGGL((layout_kernel<true, false>), grid, dim3(kLayoutThreads), lds, stream, in, out, B, Cn, T, stride2d, scale);
  else if (vec)
    hipLaunchKernelGGL((layout_kernel<false, true>), grid, dim3(kLayoutThreads), lds, stream, in, out, B, Cn, T, stride2d, scale);
  else
    hipLaunchKernelGGL((layout_kernel<false, false>), grid, dim3(kLayoutThreads), lds, stream, in, out, B, Cn, T, stride2d, scale);
  TC_HIP_CHECK(hipGetLastError());
  return TC_OK;
}

}  // namespace tc
