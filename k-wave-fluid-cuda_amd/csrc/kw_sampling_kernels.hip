// kw_sampling_kernels.hip — sensor sampling kernels; replaces namespace OutputStreamsCudaKernels
// (OutputStreams/OutputStreamsCudaKernels.cu:83-126 index, :164-252 cuboid, :297-332 whole domain, :359-378 RMS).
// Results are bit-exact with the reference semantics: a gather, x*x accumulation, max or min — no reassociation.
#include "kw_internal.h"

namespace {

template<kw_reduce_op op> __device__ __forceinline__ void reduce(float* b, float v)
{
  switch (op)
  {
    case KW_OP_NONE: *b = v; break;
    case KW_OP_RMS: *b = __fmaf_rn(v, v, *b); break; // nvcc (-fmad=true default) emits one FMA for buf += v*v
    case KW_OP_MAX: *b = fmaxf(*b, v); break;
    case KW_OP_MIN: *b = fminf(*b, v); break;
  }
}

// one sample per lane; the mask is a 64-bit index stream read coalesced, the field read is a gather
template<kw_reduce_op op>
__global__ __launch_bounds__(256) void k_sample_index(float* __restrict__ buf, const float* __restrict__ src,
                                                       const uint64_t* __restrict__ mask, uint64_t n)
{
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
       i += static_cast<uint64_t>(gridDim.x) * blockDim.x)
    reduce<op>(&buf[i], src[mask[i]]);
}

// several reduce operators of ONE field over ONE mask in a single launch (e.g. -p --p_max --p_rms): the index and the
// gathered value are read once; per operator the arithmetic is exactly k_sample_index's
struct MultiSampleArgs { float* buf[4]; int32_t op[4]; int32_t n_ops; };
__global__ __launch_bounds__(256) void k_sample_index_multi(MultiSampleArgs a, const float* __restrict__ src,
                                                             const uint64_t* __restrict__ mask, uint64_t n)
{
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
       i += static_cast<uint64_t>(gridDim.x) * blockDim.x)
  {
    const float v = src[mask[i]];
#pragma unroll
    for (int o = 0; o < 4; o++)
    {
      if (o >= a.n_ops) break;
      float* b = a.buf[o] + i;
      switch (a.op[o])
      {
        case KW_OP_NONE: reduce<KW_OP_NONE>(b, v); break;
        case KW_OP_RMS: reduce<KW_OP_RMS>(b, v); break;
        case KW_OP_MAX: reduce<KW_OP_MAX>(b, v); break;
        default: reduce<KW_OP_MIN>(b, v); break;
      }
    }
  }
}

// cuboid-local x on threads, (y,z) rows on grid: no per-element division (the reference divides per element,
// OutputStreamsCudaKernels.cu:164-188); the output index is the same cuboid-local linear index.
template<kw_reduce_op op>
__global__ __launch_bounds__(256) void k_sample_cuboid(float* __restrict__ buf, const float* __restrict__ src,
                                                        uint32_t tlx, uint32_t tly, uint32_t tlz, uint32_t cx,
                                                        uint32_t cy, uint32_t cz, uint32_t nx, uint32_t ny, uint64_t n)
{
  const uint32_t lx = blockIdx.x * blockDim.x + threadIdx.x;
  if (lx >= cx) return;
  for (uint64_t row = blockIdx.y; row < static_cast<uint64_t>(cy) * cz; row += gridDim.y)
  {
    const uint32_t lz = static_cast<uint32_t>(row / cy);
    const uint32_t ly = static_cast<uint32_t>(row - static_cast<uint64_t>(lz) * cy);
    const uint64_t i  = row * cx + lx;
    if (i >= n) return;
    const uint64_t pos = (static_cast<uint64_t>(lz + tlz) * ny + (ly + tly)) * nx + (lx + tlx);
    reduce<op>(&buf[i], src[pos]);
  }
}

template<kw_reduce_op op>
__global__ __launch_bounds__(256) void k_sample_all(float* __restrict__ buf, const float* __restrict__ src, uint64_t n)
{
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
       i += static_cast<uint64_t>(gridDim.x) * blockDim.x)
    reduce<op>(&buf[i], src[i]);
}

__global__ __launch_bounds__(256) void k_post_rms(float* __restrict__ buf, float scale, uint64_t n)
{
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
       i += static_cast<uint64_t>(gridDim.x) * blockDim.x)
    buf[i] = sqrtf(buf[i] * scale);
}

// Compression sampling (config 5): gather + correlation with the windowed complex-exponential basis in one kernel.
// The reference gathers on the GPU and correlates on the CPU one step later (IndexOutputStream.cpp:373-470); the
// arithmetic per (point, harmonic) is the same: c1 += bE*x; c2 += bE_1*x; first saved frame: c2 += c1.
// c1 and c2 may alias (--no_overlap: mHostBuffer2 == mHostBuffer1, BaseOutputStream.cpp:246-249) -> no __restrict__.
__global__ __launch_bounds__(256) void k_sample_index_compress(float2* c1, float2* c2, const float* __restrict__ src,
                                                                const uint64_t* __restrict__ mask, uint64_t n,
                                                                uint32_t harmonics, const float2* __restrict__ bE,
                                                                const float2* __restrict__ bE_1, uint32_t b_size,
                                                                uint32_t step_local, int mirror)
{
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
       i += static_cast<uint64_t>(gridDim.x) * blockDim.x)
  {
    const float x = src[mask[i]];
    for (uint32_t h = 0; h < harmonics; h++)
    {
      const uint64_t ph = harmonics * i + h;
      const float2   b0 = bE[static_cast<size_t>(h) * b_size + step_local];
      const float2   b1 = bE_1[static_cast<size_t>(h) * b_size + step_local];
      float2 v1 = c1[ph];
      v1.x += b0.x * x;
      v1.y += b0.y * x;
      c1[ph] = v1;
      float2 v2 = c2[ph]; // after the store above: c2 may be c1
      v2.x += b1.x * x;
      v2.y += b1.y * x;
      if (mirror)
      {
        v2.x += v1.x;
        v2.y += v1.y;
      }
      c2[ph] = v2;
    }
  }
}

// I_avg_c accumulation from one emitted pair of coefficient frames (IndexOutputStream.cpp:315-339):
// iavg[i] += sum_h real(P * conj(U)) / 2
__global__ __launch_bounds__(256) void k_intensity_avg_c(float* __restrict__ iavg, const float2* __restrict__ P,
                                                          const float2* __restrict__ U, uint64_t n, uint32_t harmonics)
{
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
       i += static_cast<uint64_t>(gridDim.x) * blockDim.x)
  {
    float acc = iavg[i];
    for (uint32_t h = 0; h < harmonics; h++)
    {
      const float2 p = P[harmonics * i + h], u = U[harmonics * i + h];
      acc += (p.x * u.x + p.y * u.y) / 2.0f;
    }
    iavg[i] = acc;
  }
}

__global__ __launch_bounds__(256) void k_divide(float* __restrict__ buf, float divisor, uint64_t n)
{
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
       i += static_cast<uint64_t>(gridDim.x) * blockDim.x)
    buf[i] = buf[i] / divisor;
}

// time-averaged intensity of one sampled point: the products are added in step order, then divided by the step count
// (KSpaceFirstOrderSolver.cpp:1492-1513)
__global__ __launch_bounds__(256) void k_intensity_avg(float* __restrict__ iavg, const float* __restrict__ p,
                                                       const float* __restrict__ u, uint64_t steps, uint64_t n)
{
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
       i += static_cast<uint64_t>(gridDim.x) * blockDim.x)
  {
    float acc = 0.0f;
    for (uint64_t s = 0; s < steps; s++) acc += u[s * n + i] * p[s * n + i];
    iavg[i] = acc / static_cast<float>(steps);
  }
}

// Q = -(dIx/dx + dIy/dy [+ dIz/dz])   (KSpaceFirstOrderSolver.cpp:2014-2026)
__global__ __launch_bounds__(256) void k_q_term_sum(float* __restrict__ out, const float* __restrict__ a,
                                                    const float* __restrict__ b, const float* __restrict__ c, uint64_t n)
{
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
       i += static_cast<uint64_t>(gridDim.x) * blockDim.x)
    out[i] = (c != nullptr) ? -(a[i] + b[i] + c[i]) : -(a[i] + b[i]);
}

inline unsigned sampler_grid(const kw_ctx* ctx, uint64_t n)
{
  // CU count x 8 blocks, shrunk to the work size (reference: SM count x 8, CudaParameters.cpp:218-231)
  uint64_t g   = (n + 255) / 256;
  uint64_t cap = static_cast<uint64_t>(ctx->cu_count) * 8;
  if (g > cap) g = cap;
  if (g == 0) g = 1;
  return static_cast<unsigned>(g);
}

} // namespace

#define LAUNCH(kernel, grid, block, ...)                                                                               \
  do {                                                                                                                 \
    hipLaunchKernelGGL(kernel, grid, block, 0, ctx->stream, __VA_ARGS__);                                              \
    KW_LAUNCH_CHECK();                                                                                                 \
  } while (0)

#define DISPATCH_OP(op, K, grid, ...)                                                                                  \
  switch (op)                                                                                                          \
  {                                                                                                                    \
    case KW_OP_NONE: LAUNCH((K<KW_OP_NONE>), grid, dim3(256), __VA_ARGS__); break;                                     \
    case KW_OP_RMS: LAUNCH((K<KW_OP_RMS>), grid, dim3(256), __VA_ARGS__); break;                                       \
    case KW_OP_MAX: LAUNCH((K<KW_OP_MAX>), grid, dim3(256), __VA_ARGS__); break;                                       \
    case KW_OP_MIN: LAUNCH((K<KW_OP_MIN>), grid, dim3(256), __VA_ARGS__); break;                                       \
    default: kw_set_error("%s: unknown reduce operator %d", __func__, (int)op); return KW_ERR_INVALID;                 \
  }

extern "C" {

kw_status kw_sample_index(kw_ctx* ctx, kw_reduce_op op, float* buf, const float* src, const uint64_t* mask, uint64_t n)
{
  KW_CHECK_CTX(ctx);
  KW_PROF(ctx, "sample_index");
  if (n == 0) return KW_OK;
  KW_REQUIRE(buf && src && mask);
  DISPATCH_OP(op, k_sample_index, dim3(sampler_grid(ctx, n)), buf, src, mask, n);
  return KW_OK;
}

kw_status kw_sample_index_multi(kw_ctx* ctx, int n_ops, const kw_reduce_op* ops, float* const* bufs, const float* src,
                                const uint64_t* mask, uint64_t n)
{
  KW_CHECK_CTX(ctx);
  KW_PROF(ctx, "sample_index");
  KW_REQUIRE(n_ops >= 1 && n_ops <= 4 && ops && bufs);
  if (n == 0) return KW_OK;
  KW_REQUIRE(src && mask);
  MultiSampleArgs a{};
  a.n_ops = n_ops;
  for (int o = 0; o < n_ops; o++)
  {
    KW_REQUIRE(bufs[o] != nullptr && ops[o] >= KW_OP_NONE && ops[o] <= KW_OP_MIN);
    a.buf[o] = bufs[o];
    a.op[o]  = static_cast<int32_t>(ops[o]);
  }
  hipLaunchKernelGGL(k_sample_index_multi, dim3(sampler_grid(ctx, n)), dim3(256), 0, ctx->stream, a, src, mask, n);
  KW_LAUNCH_CHECK();
  return KW_OK;
}

kw_status kw_sample_cuboid(kw_ctx* ctx, kw_reduce_op op, float* buf, const float* src, const uint32_t tl[3],
                           const uint32_t br[3], const uint32_t size[3], uint64_t n)
{
  KW_CHECK_CTX(ctx);
  KW_PROF(ctx, "sample_cuboid");
  if (n == 0) return KW_OK;
  KW_REQUIRE(buf && src && tl && br && size);
  KW_REQUIRE(br[0] >= tl[0] && br[1] >= tl[1] && br[2] >= tl[2]);
  KW_REQUIRE(br[0] < size[0] && br[1] < size[1] && br[2] < size[2]);
  const uint32_t cx = br[0] - tl[0] + 1, cy = br[1] - tl[1] + 1, cz = br[2] - tl[2] + 1;
  KW_REQUIRE(n <= static_cast<uint64_t>(cx) * cy * cz);
  uint64_t rows = static_cast<uint64_t>(cy) * cz;
  if (rows > 65535) rows = 65535;
  const dim3 grid((cx + 255) / 256, static_cast<unsigned>(rows), 1);
  DISPATCH_OP(op, k_sample_cuboid, grid, buf, src, tl[0], tl[1], tl[2], cx, cy, cz, size[0], size[1], n);
  return KW_OK;
}

kw_status kw_sample_all(kw_ctx* ctx, kw_reduce_op op, float* buf, const float* src, uint64_t n)
{
  KW_CHECK_CTX(ctx);
  KW_PROF(ctx, "sample_all");
  if (n == 0) return KW_OK;
  KW_REQUIRE(buf && src);
  DISPATCH_OP(op, k_sample_all, dim3(sampler_grid(ctx, n)), buf, src, n);
  return KW_OK;
}

kw_status kw_sample_index_compress(kw_ctx* ctx, float* c1, float* c2, const float* src, const uint64_t* mask, uint64_t n,
                                   uint32_t harmonics, const float* bE, const float* bE_1, uint32_t b_size,
                                   uint32_t step_local, int mirror_first_half_frame)
{
  KW_CHECK_CTX(ctx);
  KW_PROF(ctx, "sample_index_compress");
  if (n == 0) return KW_OK;
  KW_REQUIRE(c1 && c2 && src && mask && bE && bE_1);
  KW_REQUIRE(harmonics >= 1 && b_size >= 3 && step_local < b_size);
  LAUNCH(k_sample_index_compress, dim3(sampler_grid(ctx, n)), dim3(256), (float2*)c1, (float2*)c2, src, mask, n,
         harmonics, (const float2*)bE, (const float2*)bE_1, b_size, step_local, mirror_first_half_frame);
  return KW_OK;
}

kw_status kw_intensity_avg_c_accumulate(kw_ctx* ctx, float* iavg, const float* frame_p, const float* frame_u, uint64_t n,
                                        uint32_t harmonics)
{
  KW_CHECK_CTX(ctx);
  KW_PROF(ctx, "intensity_avg_c_accumulate");
  if (n == 0) return KW_OK;
  KW_REQUIRE(iavg && frame_p && frame_u && harmonics >= 1);
  LAUNCH(k_intensity_avg_c, dim3(sampler_grid(ctx, n)), dim3(256), iavg, (const float2*)frame_p, (const float2*)frame_u,
         n, harmonics);
  return KW_OK;
}

kw_status kw_intensity_avg(kw_ctx* ctx, float* iavg, const float* p, const float* u, uint64_t steps, uint64_t n)
{
  KW_CHECK_CTX(ctx);
  KW_PROF(ctx, "intensity_avg");
  if (n == 0) return KW_OK;
  KW_REQUIRE(iavg && p && u && steps >= 1);
  LAUNCH(k_intensity_avg, dim3(sampler_grid(ctx, n)), dim3(256), iavg, p, u, steps, n);
  return KW_OK;
}

kw_status kw_q_term_sum(kw_ctx* ctx, float* out, const float* a, const float* b, const float* c, uint64_t n)
{
  KW_CHECK_CTX(ctx);
  KW_PROF(ctx, "q_term_sum");
  if (n == 0) return KW_OK;
  KW_REQUIRE(out && a && b);
  LAUNCH(k_q_term_sum, dim3(sampler_grid(ctx, n)), dim3(256), out, a, b, c, n);
  return KW_OK;
}

kw_status kw_divide(kw_ctx* ctx, float* buf, float divisor, uint64_t n)
{
  KW_CHECK_CTX(ctx);
  if (n == 0) return KW_OK;
  KW_REQUIRE(buf != nullptr);
  LAUNCH(k_divide, dim3(sampler_grid(ctx, n)), dim3(256), buf, divisor, n);
  return KW_OK;
}

kw_status kw_post_processing_rms(kw_ctx* ctx, float* buf, float scale, uint64_t n)
{
  KW_CHECK_CTX(ctx);
  KW_PROF(ctx, "post_processing_rms");
  if (n == 0) return KW_OK;
  KW_REQUIRE(buf);
  LAUNCH(k_post_rms, dim3(sampler_grid(ctx, n)), dim3(256), buf, scale, n);
  return KW_OK;
}

} // extern "C"
