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

// ---- --40-bit_complex: accumulators and frames held as 5-byte complex numbers (Compression/CompressHelper.cpp:224-389):
// byte 0 = sign(re) | sign(im) | top mantissa bit of re | of im | 4-bit shared exponent (biased by e: 138 pressure,
// 114 velocity); bytes 1-2 / 3-4 = the low 16 bits of the 17-bit mantissas (explicit leading one).  The reference
// decodes, adds and re-encodes the accumulator of every (point, harmonic) at every sampled step
// (IndexOutputStream.cpp:410-436) — the rounding of that round trip is part of the result, so it is done here too.
__device__ __forceinline__ uint32_t pack40_mantissa(uint32_t fraction, uint32_t shift)
{
  uint32_t m = fraction >> shift;
  if (m > 0u && m != (0x7FFFFFu >> shift)) m++; // round up unless the field is all ones
  m |= 1u << (23u - shift);                       // the float's hidden one becomes an explicit bit
  return m >> 1;
}
__device__ __forceinline__ void pack40(float2 v, uint8_t* out, int e)
{
  const uint32_t bR = __float_as_uint(v.x), bI = __float_as_uint(v.y);
  const uint32_t sR = bR >> 31, sI = bI >> 31;
  const int eR = static_cast<int>((bR & 0x7F800000u) >> 23) - e, eI = static_cast<int>((bI & 0x7F800000u) >> 23) - e;
  int      eS = max(eR, eI); // shared exponent: the larger one; the other mantissa is shifted down by the difference
  uint32_t shR = 6u + static_cast<uint32_t>(eS - eR), shI = 6u + static_cast<uint32_t>(eS - eI);
  if (eS < 0)
  { // below the range: denormalise against exponent 0
    shR += static_cast<uint32_t>(-eS);
    shI += static_cast<uint32_t>(-eS);
    eS = 0;
  }
  shR = min(shR, 23u);
  shI = min(shI, 23u);
  uint32_t mR = pack40_mantissa(bR & 0x007FFFFFu, shR), mI = pack40_mantissa(bI & 0x007FFFFFu, shI);
  if (eS > 0xF)
  { // above the range: saturate
    mR = mI = 0xFFFFu;
    eS = 0xF;
  }
  out[0] = static_cast<uint8_t>((sR << 7) | (sI << 6) | ((mR & 0x10000u) >> 11) | ((mI & 0x10000u) >> 12) | (static_cast<uint32_t>(eS) & 0xFu));
  out[1] = static_cast<uint8_t>(mR & 0xFFu);
  out[2] = static_cast<uint8_t>((mR >> 8) & 0xFFu);
  out[3] = static_cast<uint8_t>(mI & 0xFFu);
  out[4] = static_cast<uint8_t>((mI >> 8) & 0xFFu);
}
__device__ __forceinline__ float unpack40_component(uint32_t field17, uint32_t sign, int exponent)
{
  uint32_t m = field17 << 6;
  if (m == 0u) return __uint_as_float(sign << 31);
  const int index = 31 - __clz(static_cast<int>(m)); // position of the explicit leading one
  m <<= 23 - index;
  exponent -= 22 - index;
  return __uint_as_float((sign << 31) | (static_cast<uint32_t>(exponent) << 23) | (m & 0x007FFFFFu));
}
__device__ __forceinline__ float2 unpack40(const uint8_t* in, int e)
{
  const uint32_t head = in[0];
  const uint32_t mR = ((head & 0x20u) << 11) | (static_cast<uint32_t>(in[2]) << 8) | in[1];
  const uint32_t mI = ((head & 0x10u) << 12) | (static_cast<uint32_t>(in[4]) << 8) | in[3];
  const int ex = static_cast<int>(head & 0xFu) + e;
  return make_float2(unpack40_component(mR, head >> 7, ex), unpack40_component(mI, (head & 0x40u) >> 6, ex));
}

__global__ __launch_bounds__(256) void k_sample_index_compress_40b(uint8_t* c1, uint8_t* c2, const float* __restrict__ src,
                                                                    const uint64_t* __restrict__ mask, uint64_t n,
                                                                    uint32_t harmonics, const float2* __restrict__ bE,
                                                                    const float2* __restrict__ bE_1, uint32_t b_size,
                                                                    uint32_t step_local, int mirror, int no_overlap, int e)
{
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
       i += static_cast<uint64_t>(gridDim.x) * blockDim.x)
  {
    const float x = src[mask[i]];
    for (uint32_t h = 0; h < harmonics; h++)
    {
      const uint64_t ph = (harmonics * i + h) * 5u;
      const float2   b0 = bE[static_cast<size_t>(h) * b_size + step_local];
      const float2   b1 = bE_1[static_cast<size_t>(h) * b_size + step_local];
      float2 cc1 = unpack40(c1 + ph, e);
      if (no_overlap)
      { // IndexOutputStream.cpp:416-421: one buffer, cc1 += bE*x + bE_1*x
        cc1.x += b0.x * x + b1.x * x;
        cc1.y += b0.y * x + b1.y * x;
        pack40(cc1, c1 + ph, e);
        continue;
      }
      float2 cc2 = unpack40(c2 + ph, e);
      cc1.x += b0.x * x;
      cc1.y += b0.y * x;
      cc2.x += b1.x * x;
      cc2.y += b1.y * x;
      pack40(cc1, c1 + ph, e);
      if (mirror)
      { // :431-435: the first half frame mirrored onto the second buffer, from the unquantised sums
        cc2.x += cc1.x;
        cc2.y += cc1.y;
      }
      pack40(cc2, c2 + ph, e);
    }
  }
}

__global__ __launch_bounds__(256) void k_intensity_avg_c_40b(float* __restrict__ iavg, const uint8_t* __restrict__ P,
                                                              const uint8_t* __restrict__ U, uint64_t n, uint32_t harmonics,
                                                              int e_p, int e_u)
{ // IndexOutputStream.cpp:315-339 with the frames decoded first (:325-329)
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
       i += static_cast<uint64_t>(gridDim.x) * blockDim.x)
  {
    float acc = iavg[i];
    for (uint32_t h = 0; h < harmonics; h++)
    {
      const float2 p = unpack40(P + (harmonics * i + h) * 5u, e_p), u = unpack40(U + (harmonics * i + h) * 5u, e_u);
      acc += (p.x * u.x + p.y * u.y) / 2.0f;
    }
    iavg[i] = acc;
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

kw_status kw_sample_index_compress_40b(kw_ctx* ctx, void* c1, void* c2, const float* src, const uint64_t* mask, uint64_t n,
                                       uint32_t harmonics, const float* bE, const float* bE_1, uint32_t b_size,
                                       uint32_t step_local, int mirror_first_half_frame, int no_overlap, int max_exp)
{
  KW_CHECK_CTX(ctx);
  KW_PROF(ctx, "sample_index_compress_40b");
  if (n == 0) return KW_OK;
  KW_REQUIRE(c1 && (c2 || no_overlap) && src && mask && bE && bE_1);
  KW_REQUIRE(harmonics >= 1 && b_size >= 3 && step_local < b_size);
  LAUNCH(k_sample_index_compress_40b, dim3(sampler_grid(ctx, n)), dim3(256), (uint8_t*)c1, (uint8_t*)c2, src, mask, n,
         harmonics, (const float2*)bE, (const float2*)bE_1, b_size, step_local, mirror_first_half_frame, no_overlap, max_exp);
  return KW_OK;
}

kw_status kw_intensity_avg_c_accumulate_40b(kw_ctx* ctx, float* iavg, const void* frame_p, const void* frame_u, uint64_t n,
                                            uint32_t harmonics, int max_exp_p, int max_exp_u)
{
  KW_CHECK_CTX(ctx);
  KW_PROF(ctx, "intensity_avg_c_accumulate_40b");
  if (n == 0) return KW_OK;
  KW_REQUIRE(iavg && frame_p && frame_u && harmonics >= 1);
  LAUNCH(k_intensity_avg_c_40b, dim3(sampler_grid(ctx, n)), dim3(256), iavg, (const uint8_t*)frame_p, (const uint8_t*)frame_u,
         n, harmonics, max_exp_p, max_exp_u);
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
