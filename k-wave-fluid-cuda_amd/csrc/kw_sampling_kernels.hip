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
