// kw_fft.hip — rocFFT plan cache + exec; replaces the cuFFT wrapper of
// MatrixClasses/CufftComplexMatrix.cpp (plans :82-130 ND, :144-426 1-D; destroy :432-502; exec :508-692).
//
// Contract kept: single precision, unnormalised, forward sign -i, out-of-place, R2C keeps nx/2+1 bins along the
// fastest axis, plans are per-context "statics" shared by every complex matrix, errors -> KW_ERR_FFT with the
// failing routine named (reference: throwCufftException, :706-720).
#include "kw_internal.h"

static const char* rocfft_err(rocfft_status s)
{
  switch (s)
  {
    case rocfft_status_success: return "success";
    case rocfft_status_failure: return "rocFFT failure";
    case rocfft_status_invalid_arg_value: return "rocFFT invalid argument value";
    case rocfft_status_invalid_dimensions: return "rocFFT invalid dimensions";
    case rocfft_status_invalid_array_type: return "rocFFT invalid array type";
    case rocfft_status_invalid_strides: return "rocFFT invalid strides";
    case rocfft_status_invalid_distance: return "rocFFT invalid distance";
    case rocfft_status_invalid_offset: return "rocFFT invalid offset";
    case rocfft_status_invalid_work_buffer: return "rocFFT invalid work buffer";
    default: return "rocFFT unknown error";
  }
}

#define KW_FFT(call)                                                                                                   \
  do {                                                                                                                 \
    rocfft_status s_ = (call);                                                                                         \
    if (s_ != rocfft_status_success) {                                                                                 \
      kw_set_error("FFT error: %s in %s (%s:%d)", rocfft_err(s_), __func__, __FILE__, __LINE__);                       \
      return KW_ERR_FFT;                                                                                               \
    }                                                                                                                  \
  } while (0)

static void plan_free(kw_fft_plan& p)
{
  if (p.info) rocfft_execution_info_destroy(p.info);
  if (p.plan) rocfft_plan_destroy(p.plan);
  p = kw_fft_plan();
}

// rocfft_setup() once per process; never torn down while contexts may still hold plans
static kw_status ensure_setup(kw_ctx* ctx)
{
  static bool process_setup = false;
  if (!process_setup)
  {
    KW_FFT(rocfft_setup());
    process_setup = true;
  }
  ctx->fft_setup = true;
  return KW_OK;
}

// (re)allocate the shared work buffer and bind it + the stream to every plan
static kw_status bind_work(kw_ctx* ctx)
{
  kw_fft_plan* plans[] = { &ctx->r2c_3d, &ctx->c2r_3d, &ctx->r2c_1d[0], &ctx->r2c_1d[1], &ctx->r2c_1d[2],
                           &ctx->c2r_1d[0], &ctx->c2r_1d[1], &ctx->c2r_1d[2] };
  size_t need = 0;
  for (kw_fft_plan* p : plans)
    if (p->plan && p->work > need) need = p->work;
  if (need > ctx->fft_work_bytes)
  {
    KW_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->fft_work) KW_HIP(hipFree(ctx->fft_work));
    ctx->fft_work       = nullptr;
    ctx->fft_work_bytes = 0;
    KW_HIP(hipMalloc(&ctx->fft_work, need));
    ctx->fft_work_bytes = need;
  }
  for (kw_fft_plan* p : plans)
  {
    if (!p->plan) continue;
    if (!p->info) KW_FFT(rocfft_execution_info_create(&p->info));
    if (p->work > 0) KW_FFT(rocfft_execution_info_set_work_buffer(p->info, ctx->fft_work, ctx->fft_work_bytes));
    KW_FFT(rocfft_execution_info_set_stream(p->info, ctx->stream));
  }
  return KW_OK;
}

static kw_status make_plan(kw_fft_plan& p, rocfft_transform_type type, size_t dims, const size_t* lengths,
                           size_t batch, const size_t* in_strides, size_t in_dist, const size_t* out_strides,
                           size_t out_dist)
{
  plan_free(p);
  rocfft_plan_description desc = nullptr;
  const bool fwd = (type == rocfft_transform_type_real_forward);
  if (in_strides != nullptr)
  {
    KW_FFT(rocfft_plan_description_create(&desc));
    rocfft_status s = rocfft_plan_description_set_data_layout(
      desc, fwd ? rocfft_array_type_real : rocfft_array_type_hermitian_interleaved,
      fwd ? rocfft_array_type_hermitian_interleaved : rocfft_array_type_real, nullptr, nullptr, dims, in_strides,
      in_dist, dims, out_strides, out_dist);
    if (s != rocfft_status_success)
    {
      rocfft_plan_description_destroy(desc);
      KW_FFT(s);
    }
  }
  rocfft_status s = rocfft_plan_create(&p.plan, rocfft_placement_notinplace, type, rocfft_precision_single, dims,
                                       lengths, batch, desc);
  if (desc) rocfft_plan_description_destroy(desc);
  KW_FFT(s);
  KW_FFT(rocfft_plan_get_work_buffer_size(p.plan, &p.work));
  return KW_OK;
}

// X[k][i] *= divider * shift[k]   (series spectrum [kc][n], k along the step axis)
__global__ void k_series_shift(float2* __restrict__ spec, const float2* __restrict__ shift, uint64_t n, float divider)
{
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float2 s = shift[blockIdx.y];
  const float2 m = make_float2(divider * s.x, divider * s.y);
  float2&      v = spec[static_cast<uint64_t>(blockIdx.y) * n + i];
  v = make_float2(v.x * m.x - v.y * m.y, v.x * m.y + v.y * m.x);
}

extern "C" {

kw_status kw_fft_create_plans_3d(kw_ctx* ctx)
{
  KW_CHECK_CONSTS(ctx);
  KW_HIP(hipSetDevice(ctx->device));
  kw_status st = ensure_setup(ctx);
  if (st != KW_OK) return st;
  // rocFFT lengths are fastest-first; drop unit trailing dimensions (2-D planes / 1-D lines)
  size_t lengths[3] = { ctx->c.nx, ctx->c.ny, ctx->c.nz };
  size_t dims       = 3;
  while (dims > 1 && lengths[dims - 1] == 1) dims--;
  st = make_plan(ctx->r2c_3d, rocfft_transform_type_real_forward, dims, lengths, 1, nullptr, 0, nullptr, 0);
  if (st != KW_OK) return st;
  st = make_plan(ctx->c2r_3d, rocfft_transform_type_real_inverse, dims, lengths, 1, nullptr, 0, nullptr, 0);
  if (st != KW_OK) return st;
  return bind_work(ctx);
}

kw_status kw_fft_create_plans_1d(kw_ctx* ctx, int axis)
{
  KW_CHECK_CONSTS(ctx);
  KW_REQUIRE(axis >= 0 && axis <= 2);
  KW_HIP(hipSetDevice(ctx->device));
  kw_status st = ensure_setup(ctx);
  if (st != KW_OK) return st;
  const size_t nx = ctx->c.nx, ny = ctx->c.ny, nz = ctx->c.nz;
  if (axis == 0)
  {
    size_t len[1] = { nx };
    st = make_plan(ctx->r2c_1d[0], rocfft_transform_type_real_forward, 1, len, ny * nz, nullptr, 0, nullptr, 0);
    if (st != KW_OK) return st;
    st = make_plan(ctx->c2r_1d[0], rocfft_transform_type_real_inverse, 1, len, ny * nz, nullptr, 0, nullptr, 0);
  }
  else if (axis == 1)
  {
    // one plan per z-plane: nx lines of length ny, element stride nx, line distance 1 (executed nz times)
    size_t len[1] = { ny }, str[1] = { nx };
    st = make_plan(ctx->r2c_1d[1], rocfft_transform_type_real_forward, 1, len, nx, str, 1, str, 1);
    if (st != KW_OK) return st;
    st = make_plan(ctx->c2r_1d[1], rocfft_transform_type_real_inverse, 1, len, nx, str, 1, str, 1);
  }
  else
  {
    size_t len[1] = { nz }, str[1] = { nx * ny };
    st = make_plan(ctx->r2c_1d[2], rocfft_transform_type_real_forward, 1, len, nx * ny, str, 1, str, 1);
    if (st != KW_OK) return st;
    st = make_plan(ctx->c2r_1d[2], rocfft_transform_type_real_inverse, 1, len, nx * ny, str, 1, str, 1);
  }
  if (st != KW_OK) return st;
  return bind_work(ctx);
}

kw_status kw_time_shift_series(kw_ctx* ctx, float* series, const float* shift, uint64_t steps, uint64_t n)
{
  KW_CHECK_CTX(ctx);
  KW_REQUIRE(series != nullptr && shift != nullptr && steps >= 2 && n >= 1 && steps / 2 + 1 <= 65535u);
  KW_HIP(hipSetDevice(ctx->device));
  kw_status st = ensure_setup(ctx);
  if (st != KW_OK) return st;
  // transforms along the step axis: n lines of `steps` elements, element stride n, line distance 1
  const size_t kc = steps / 2 + 1;
  size_t len[1] = { static_cast<size_t>(steps) }, str[1] = { static_cast<size_t>(n) };
  kw_fft_plan fwd, inv;
  void *work = nullptr, *spec = nullptr;
  auto cleanup = [&]() {
    (void)hipStreamSynchronize(ctx->stream);
    plan_free(fwd);
    plan_free(inv);
    if (work) (void)hipFree(work);
    if (spec) (void)hipFree(spec);
  };
#define KW_TS(call) do { kw_status s__ = (call); if (s__ != KW_OK) { cleanup(); return s__; } } while (0)
  KW_TS(make_plan(fwd, rocfft_transform_type_real_forward, 1, len, n, str, 1, str, 1));
  KW_TS(make_plan(inv, rocfft_transform_type_real_inverse, 1, len, n, str, 1, str, 1));
  auto run = [&]() -> kw_status {
    const size_t wbytes = fwd.work > inv.work ? fwd.work : inv.work;
    if (wbytes) KW_HIP(hipMalloc(&work, wbytes));
    KW_HIP(hipMalloc(&spec, kc * n * sizeof(float2)));
    for (kw_fft_plan* p : { &fwd, &inv })
    {
      KW_FFT(rocfft_execution_info_create(&p->info));
      if (p->work > 0) KW_FFT(rocfft_execution_info_set_work_buffer(p->info, work, wbytes));
      KW_FFT(rocfft_execution_info_set_stream(p->info, ctx->stream));
    }
    void* a[1] = { series };
    void* b[1] = { spec };
    KW_FFT(rocfft_execute(fwd.plan, a, b, fwd.info));
    hipLaunchKernelGGL(k_series_shift, dim3(static_cast<unsigned>((n + 255) / 256), static_cast<unsigned>(kc), 1),
                       dim3(256), 0, ctx->stream, static_cast<float2*>(spec), reinterpret_cast<const float2*>(shift), n,
                       1.0f / static_cast<float>(steps));
    KW_LAUNCH_CHECK();
    KW_FFT(rocfft_execute(inv.plan, b, a, inv.info));
    return KW_OK;
  };
  st = run();
  cleanup();
#undef KW_TS
  return st;
}

kw_status kw_fft_destroy_plans(kw_ctx* ctx)
{
  KW_CHECK_CTX(ctx);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  plan_free(ctx->r2c_3d);
  plan_free(ctx->c2r_3d);
  for (int a = 0; a < 3; a++)
  {
    plan_free(ctx->r2c_1d[a]);
    plan_free(ctx->c2r_1d[a]);
  }
  if (ctx->fft_work) (void)hipFree(ctx->fft_work);
  ctx->fft_work       = nullptr;
  ctx->fft_work_bytes = 0;
  ctx->fft_setup = false;
  return KW_OK;
}

kw_status kw_fft_r2c_3d(kw_ctx* ctx, const float* in, float* out)
{
  KW_CHECK_CONSTS(ctx);
  KW_PROF(ctx, "fft_r2c_3d");
  KW_REQUIRE(in != nullptr && out != nullptr);
  if (!ctx->r2c_3d.plan) { kw_set_error("kw_fft_r2c_3d: plans not created (kw_fft_create_plans_3d)"); return KW_ERR_STATE; }
  void* ib[1] = { (void*)in };
  void* ob[1] = { (void*)out };
  KW_FFT(rocfft_execute(ctx->r2c_3d.plan, ib, ob, ctx->r2c_3d.info));
  return KW_OK;
}

kw_status kw_fft_c2r_3d(kw_ctx* ctx, float* in, float* out)
{
  KW_CHECK_CONSTS(ctx);
  KW_PROF(ctx, "fft_c2r_3d");
  KW_REQUIRE(in != nullptr && out != nullptr);
  if (!ctx->c2r_3d.plan) { kw_set_error("kw_fft_c2r_3d: plans not created (kw_fft_create_plans_3d)"); return KW_ERR_STATE; }
  void* ib[1] = { (void*)in };
  void* ob[1] = { (void*)out };
  KW_FFT(rocfft_execute(ctx->c2r_3d.plan, ib, ob, ctx->c2r_3d.info));
  return KW_OK;
}

kw_status kw_fft_r2c_1d(kw_ctx* ctx, int axis, const float* in, float* out)
{
  KW_CHECK_CONSTS(ctx);
  KW_PROF(ctx, "fft_r2c_1d");
  KW_REQUIRE(axis >= 0 && axis <= 2 && in != nullptr && out != nullptr);
  kw_fft_plan& p = ctx->r2c_1d[axis];
  if (!p.plan) { kw_set_error("kw_fft_r2c_1d: plan for axis %d not created", axis); return KW_ERR_STATE; }
  const size_t nx = ctx->c.nx, ny = ctx->c.ny, nz = ctx->c.nz;
  const size_t reps = (axis == 1) ? nz : 1;
  for (size_t z = 0; z < reps; z++)
  {
    void* ib[1] = { (void*)(in + z * nx * ny) };
    void* ob[1] = { (void*)(out + 2 * z * nx * (ny / 2 + 1)) };
    KW_FFT(rocfft_execute(p.plan, ib, ob, p.info));
  }
  return KW_OK;
}

kw_status kw_fft_c2r_1d(kw_ctx* ctx, int axis, float* in, float* out)
{
  KW_CHECK_CONSTS(ctx);
  KW_PROF(ctx, "fft_c2r_1d");
  KW_REQUIRE(axis >= 0 && axis <= 2 && in != nullptr && out != nullptr);
  kw_fft_plan& p = ctx->c2r_1d[axis];
  if (!p.plan) { kw_set_error("kw_fft_c2r_1d: plan for axis %d not created", axis); return KW_ERR_STATE; }
  const size_t nx = ctx->c.nx, ny = ctx->c.ny, nz = ctx->c.nz;
  const size_t reps = (axis == 1) ? nz : 1;
  for (size_t z = 0; z < reps; z++)
  {
    void* ib[1] = { (void*)(in + 2 * z * nx * (ny / 2 + 1)) };
    void* ob[1] = { (void*)(out + z * nx * ny) };
    KW_FFT(rocfft_execute(p.plan, ib, ob, p.info));
  }
  return KW_OK;
}

} // extern "C"
