// kw_context.hip — context, device selection, memory verbs, constants, events.
// Replaces CudaParameters::selectDevice / setUpDeviceConstants (Parameters/CudaParameters.cpp:81-177,238-288),
// CudaDeviceConstants::uploadDeviceConstants (Parameters/CudaDeviceConstants.cu:58-60) and the memory verbs of
// MatrixClasses/BaseFloatMatrix.cpp:77-80,124-168.
#include "kw_internal.h"

#include <map>
#include <string>

static thread_local char g_err[1024] = "";

void kw_set_error(const char* fmt, ...)
{
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" {

const char* kw_last_error(void) { return g_err; }

kw_status kw_init(int device_id, kw_ctx** out_ctx)
{
  if (out_ctx == nullptr) { kw_set_error("kw_init: out_ctx is NULL"); return KW_ERR_INVALID; }
  *out_ctx = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
  {
    kw_set_error("kw_init: no HIP device found");
    return KW_ERR_NO_DEVICE;
  }
  int dev = device_id;
  if (dev < 0)
  {
    // first device that accepts a context (reference: first free device, CudaParameters.cpp:95-123)
    dev = -1;
    for (int i = 0; i < n; i++)
    {
      if (hipSetDevice(i) == hipSuccess && hipFree(nullptr) == hipSuccess) { dev = i; break; }
    }
    if (dev < 0) { kw_set_error("kw_init: no free HIP device"); return KW_ERR_NO_DEVICE; }
  }
  else if (dev >= n)
  {
    kw_set_error("kw_init: device %d out of range (have %d)", dev, n);
    return KW_ERR_NO_DEVICE;
  }
  KW_HIP(hipSetDevice(dev));
  hipDeviceProp_t prop;
  KW_HIP(hipGetDeviceProperties(&prop, dev));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
  {
    // the code objects in this library are gfx950-only; fail loudly instead of at the first launch
    kw_set_error("kw_init: device %d is %s, this library is built for gfx950 (MI355X) only", dev, prop.gcnArchName);
    return KW_ERR_NO_DEVICE;
  }
  kw_ctx* ctx   = new kw_ctx();
  ctx->device   = dev;
  ctx->cu_count = prop.multiProcessorCount;
  hipError_t e  = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking);
  if (e != hipSuccess)
  {
    kw_set_error("GPU error: %s routine name: kw_init", hipGetErrorString(e));
    delete ctx;
    return KW_ERR_HIP;
  }
  ctx->stream = ctx->own_stream;
  if (hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->copy_fence, hipEventDisableTiming) != hipSuccess)
  {
    kw_set_error("GPU error: cannot create the copy stream (kw_init)");
    delete ctx;
    return KW_ERR_HIP;
  }
  *out_ctx    = ctx;
  return KW_OK;
}

kw_status kw_destroy(kw_ctx* ctx)
{
  if (ctx == nullptr) return KW_OK;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  kw_fft_destroy_plans(ctx);
  (void)kw_comm_destroy(ctx);
  if (ctx->copy_stream) { (void)hipStreamSynchronize(ctx->copy_stream); (void)hipStreamDestroy(ctx->copy_stream); }
  if (ctx->copy_fence) (void)hipEventDestroy(ctx->copy_fence);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
  delete ctx;
  return KW_OK;
}

kw_status kw_device_info_get(kw_ctx* ctx, kw_device_info* out)
{
  KW_CHECK_CTX(ctx);
  KW_REQUIRE(out != nullptr);
  hipDeviceProp_t prop;
  KW_HIP(hipGetDeviceProperties(&prop, ctx->device));
  memset(out, 0, sizeof(*out));
  snprintf(out->name, sizeof(out->name), "%s", prop.name);
  snprintf(out->arch, sizeof(out->arch), "%s", prop.gcnArchName);
  out->device_id      = ctx->device;
  out->compute_units  = prop.multiProcessorCount;
  out->wavefront_size = prop.warpSize;
  out->clock_mhz      = prop.clockRate / 1000;
  out->total_mem      = prop.totalGlobalMem;
  size_t fr = 0, tot = 0;
  KW_HIP(hipMemGetInfo(&fr, &tot));
  out->free_mem   = fr;
  out->lds_per_cu = prop.maxSharedMemoryPerMultiProcessor;
  out->l2_bytes   = (uint64_t)prop.l2CacheSize;
  return KW_OK;
}

kw_status kw_set_stream(kw_ctx* ctx, void* hip_stream)
{
  KW_CHECK_CTX(ctx);
  ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
  hipStream_t s = ctx->stream;
  kw_fft_plan* plans[] = { &ctx->r2c_3d, &ctx->c2r_3d, &ctx->r2c_1d[0], &ctx->r2c_1d[1], &ctx->r2c_1d[2],
                           &ctx->c2r_1d[0], &ctx->c2r_1d[1], &ctx->c2r_1d[2] };
  for (kw_fft_plan* p : plans)
    if (p->info && rocfft_execution_info_set_stream(p->info, s) != rocfft_status_success)
    {
      kw_set_error("kw_set_stream: rocfft_execution_info_set_stream failed");
      return KW_ERR_FFT;
    }
  return KW_OK;
}

void* kw_get_stream(kw_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

struct kw_graph
{
  hipGraph_t     graph = nullptr;
  hipGraphExec_t exec  = nullptr;
};

kw_status kw_graph_begin(kw_ctx* ctx)
{
  KW_CHECK_CTX(ctx);
  if (ctx->profiling) { kw_set_error("kw_graph_begin: not while per-call profiling is enabled"); return KW_ERR_STATE; }
  KW_HIP(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
  return KW_OK;
}

kw_status kw_graph_end(kw_ctx* ctx, kw_graph** out)
{
  KW_CHECK_CTX(ctx);
  KW_REQUIRE(out != nullptr);
  *out = nullptr;
  hipGraph_t g = nullptr;
  KW_HIP(hipStreamEndCapture(ctx->stream, &g));
  kw_graph* kg = new kw_graph();
  kg->graph    = g;
  const hipError_t e = hipGraphInstantiate(&kg->exec, g, nullptr, nullptr, 0);
  if (e != hipSuccess)
  {
    (void)hipGraphDestroy(g);
    delete kg;
    kw_set_error("kw_graph_end: hipGraphInstantiate: %s", hipGetErrorString(e));
    return KW_ERR_HIP;
  }
  *out = kg;
  return KW_OK;
}

kw_status kw_graph_launch(kw_ctx* ctx, kw_graph* g)
{
  KW_CHECK_CTX(ctx);
  KW_REQUIRE(g != nullptr && g->exec != nullptr);
  KW_HIP(hipGraphLaunch(g->exec, ctx->stream));
  return KW_OK;
}

kw_status kw_graph_destroy(kw_ctx* ctx, kw_graph* g)
{
  KW_CHECK_CTX(ctx);
  if (g == nullptr) return KW_OK;
  if (g->exec) (void)hipGraphExecDestroy(g->exec);
  if (g->graph) (void)hipGraphDestroy(g->graph);
  delete g;
  return KW_OK;
}

kw_status kw_sync(kw_ctx* ctx)
{
  KW_CHECK_CTX(ctx);
  KW_HIP(hipStreamSynchronize(ctx->stream));
  return kw_comm_check(ctx); // a P2P exchange that gave up waiting for a peer surfaces here (KW_ERR_COMM)
}

kw_status kw_get_tuning(kw_ctx* ctx, kw_tuning* out)
{
  KW_CHECK_CTX(ctx);
  KW_REQUIRE(out != nullptr);
  *out = ctx->tuning;
  return KW_OK;
}

kw_status kw_set_tuning(kw_ctx* ctx, const kw_tuning* tuning)
{
  KW_CHECK_CTX(ctx);
  KW_REQUIRE(tuning != nullptr && tuning->struct_bytes >= sizeof(uint32_t) && tuning->struct_bytes <= sizeof(kw_tuning));
  kw_tuning t = ctx->tuning; // fields beyond the caller's struct keep their values
  memcpy(&t, tuning, tuning->struct_bytes);
  t.struct_bytes = static_cast<uint32_t>(sizeof(kw_tuning));
  KW_REQUIRE(t.tail_chunks >= 0 && t.slab_chunks >= 1 && t.slab_chunks <= KW_XCHUNKS_MAX && t.slab_batch >= -1 && t.slab_batch <= 1);
  KW_REQUIRE(t.p2p_blocks_per_peer >= 1 && t.p2p_blocks_per_peer <= 8 && t.p2p_timeout_s > 0.f);
  ctx->tuning = t;
  return KW_OK;
}

kw_status kw_event_create(kw_ctx* ctx, void** out_event)
{
  KW_CHECK_CTX(ctx);
  KW_REQUIRE(out_event != nullptr);
  hipEvent_t ev;
  KW_HIP(hipEventCreate(&ev));
  *out_event = (void*)ev;
  return KW_OK;
}
kw_status kw_event_record(kw_ctx* ctx, void* event)
{
  KW_CHECK_CTX(ctx);
  KW_HIP(hipEventRecord((hipEvent_t)event, ctx->stream));
  return KW_OK;
}
kw_status kw_event_synchronize(kw_ctx* ctx, void* event)
{
  KW_CHECK_CTX(ctx);
  KW_HIP(hipEventSynchronize((hipEvent_t)event));
  return KW_OK;
}
kw_status kw_event_elapsed_ms(kw_ctx* ctx, void* start, void* stop, float* out_ms)
{
  KW_CHECK_CTX(ctx);
  KW_REQUIRE(out_ms != nullptr);
  KW_HIP(hipEventElapsedTime(out_ms, (hipEvent_t)start, (hipEvent_t)stop));
  return KW_OK;
}
kw_status kw_event_destroy(kw_ctx* ctx, void* event)
{
  KW_CHECK_CTX(ctx);
  KW_HIP(hipEventDestroy((hipEvent_t)event));
  return KW_OK;
}

// ---- profiling -----------------------------------------------------------------------------------------------------
int kw_profile_enabled(kw_ctx* ctx) { return (ctx != nullptr && ctx->profiling) ? 1 : 0; }

kw_status kw_profile_enable(kw_ctx* ctx, int on)
{
  KW_CHECK_CTX(ctx);
  ctx->profiling = (on != 0);
  return KW_OK;
}

kw_status kw_profile_collect(kw_ctx* ctx, kw_profile_entry* out, size_t capacity, size_t* n_out)
{
  KW_CHECK_CTX(ctx);
  KW_REQUIRE(n_out != nullptr);
  KW_HIP(hipStreamSynchronize(ctx->stream));
  std::map<std::string, std::pair<uint64_t, double>> agg;
  for (auto& r : ctx->prof)
  {
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess)
    {
      auto& a = agg[r.name];
      a.first++;
      a.second += ms;
    }
    (void)hipEventDestroy(r.e0);
    (void)hipEventDestroy(r.e1);
  }
  ctx->prof.clear();
  size_t n = 0;
  for (auto& kv : agg)
  {
    if (out != nullptr && n < capacity)
    {
      memset(&out[n], 0, sizeof(out[n]));
      snprintf(out[n].name, sizeof(out[n].name), "%s", kv.first.c_str());
      out[n].calls    = kv.second.first;
      out[n].total_ms = kv.second.second;
    }
    n++;
  }
  *n_out = n;
  return KW_OK;
}

// ---- memory --------------------------------------------------------------------------------------------------------
kw_status kw_malloc(kw_ctx* ctx, size_t bytes, void** out_dptr)
{
  KW_CHECK_CTX(ctx);
  KW_REQUIRE(out_dptr != nullptr);
  *out_dptr = nullptr;
  if (bytes == 0) return KW_OK;
  KW_HIP(hipSetDevice(ctx->device));
  KW_HIP(hipMalloc(out_dptr, bytes));
  return KW_OK;
}
kw_status kw_free(kw_ctx* ctx, void* dptr)
{
  KW_CHECK_CTX(ctx);
  if (dptr == nullptr) return KW_OK;
  KW_HIP(hipFree(dptr));
  return KW_OK;
}
kw_status kw_memcpy_h2d(kw_ctx* ctx, void* dst, const void* src, size_t bytes)
{
  KW_CHECK_CTX(ctx);
  if (bytes == 0) return KW_OK;
  KW_REQUIRE(dst != nullptr && src != nullptr);
  KW_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  KW_HIP(hipStreamSynchronize(ctx->stream)); // blocking like cudaMemcpy in BaseFloatMatrix::copyToDevice
  return KW_OK;
}
kw_status kw_memcpy_d2h(kw_ctx* ctx, void* dst, const void* src, size_t bytes)
{
  KW_CHECK_CTX(ctx);
  if (bytes == 0) return KW_OK;
  KW_REQUIRE(dst != nullptr && src != nullptr);
  KW_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  KW_HIP(hipStreamSynchronize(ctx->stream));
  return KW_OK;
}
kw_status kw_memcpy_d2h_async(kw_ctx* ctx, void* dst, const void* src, size_t bytes)
{
  KW_CHECK_CTX(ctx);
  if (bytes == 0) return KW_OK;
  KW_REQUIRE(dst != nullptr && src != nullptr);
  KW_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  return KW_OK;
}
kw_status kw_memcpy_d2h_overlapped(kw_ctx* ctx, void* dst, const void* src, size_t bytes, void* event)
{
  KW_CHECK_CTX(ctx);
  KW_REQUIRE(event != nullptr);
  if (bytes > 0)
  {
    KW_REQUIRE(dst != nullptr && src != nullptr);
    KW_HIP(hipEventRecord(ctx->copy_fence, ctx->stream));
    KW_HIP(hipStreamWaitEvent(ctx->copy_stream, ctx->copy_fence, 0));
    KW_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->copy_stream));
  }
  KW_HIP(hipEventRecord((hipEvent_t)event, (bytes > 0) ? ctx->copy_stream : ctx->stream));
  return KW_OK;
}
kw_status kw_memcpy_d2d(kw_ctx* ctx, void* dst, const void* src, size_t bytes)
{
  KW_CHECK_CTX(ctx);
  if (bytes == 0) return KW_OK;
  KW_REQUIRE(dst != nullptr && src != nullptr);
  KW_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
  return KW_OK;
}
kw_status kw_memset(kw_ctx* ctx, void* dptr, int value, size_t bytes)
{
  KW_CHECK_CTX(ctx);
  if (bytes == 0) return KW_OK;
  KW_REQUIRE(dptr != nullptr);
  KW_HIP(hipMemsetAsync(dptr, value, bytes, ctx->stream));
  return KW_OK;
}
kw_status kw_host_alloc(kw_ctx* ctx, size_t bytes, void** out_hptr)
{
  KW_CHECK_CTX(ctx);
  KW_REQUIRE(out_hptr != nullptr);
  *out_hptr = nullptr;
  if (bytes == 0) return KW_OK;
  KW_HIP(hipHostMalloc(out_hptr, bytes, hipHostMallocDefault));
  return KW_OK;
}
kw_status kw_host_free(kw_ctx* ctx, void* hptr)
{
  KW_CHECK_CTX(ctx);
  if (hptr == nullptr) return KW_OK;
  KW_HIP(hipHostFree(hptr));
  return KW_OK;
}

// ---- measured device-copy bandwidth (reported beside the 8 TB/s spec peak) ------------------------------
} // extern "C"

// Three shapes of the same float4 copy; the best rate of the three is what the box can do for a 1:1 read / write stream:
//   0  grid-stride loop, 16 blocks per CU      1  one float4 per thread (as many blocks as it takes)
//   2  four float4 per thread, all loads issued before the stores, streaming (non-temporal) accesses
typedef float kw_v4f __attribute__((ext_vector_type(4)));
template<int SHAPE> __global__ __launch_bounds__(256) void k_stream_copy(const kw_v4f* __restrict__ src, kw_v4f* __restrict__ dst, size_t n4)
{
  if (SHAPE == 0)
  {
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    for (size_t e = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; e < n4; e += stride) dst[e] = src[e];
  }
  else if (SHAPE == 1)
  {
    const size_t e = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (e < n4) dst[e] = src[e];
  }
  else
  {
    const size_t e0 = static_cast<size_t>(blockIdx.x) * (4u * 256u) + threadIdx.x;
    kw_v4f v[4];
#pragma unroll
    for (int i = 0; i < 4; i++) if (e0 + i * 256u < n4) v[i] = __builtin_nontemporal_load(src + e0 + i * 256u);
#pragma unroll
    for (int i = 0; i < 4; i++) if (e0 + i * 256u < n4) __builtin_nontemporal_store(v[i], dst + e0 + i * 256u);
  }
}

extern "C" {

kw_status kw_measure_copy_bandwidth(kw_ctx* ctx, size_t bytes, int reps, double* out_gbs)
{
  KW_CHECK_CTX(ctx);
  KW_REQUIRE(out_gbs != nullptr && bytes >= (1u << 20) && reps >= 1);
  KW_HIP(hipSetDevice(ctx->device));
  const size_t n4 = bytes / sizeof(float4);
  kw_v4f *src = nullptr, *dst = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  kw_status rc = KW_OK;
  auto cleanup = [&]() {
    if (src) (void)hipFree(src);
    if (dst) (void)hipFree(dst);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
  };
#define KW_BW(call)                                                                                                    \
  do {                                                                                                                 \
    const hipError_t e_ = (call);                                                                                      \
    if (e_ != hipSuccess) { kw_set_error("GPU error: %s routine name: %s", hipGetErrorString(e_), __func__);            \
                            cleanup(); return (e_ == hipErrorOutOfMemory) ? KW_ERR_ALLOC : KW_ERR_HIP; }                \
  } while (0)
  KW_BW(hipMalloc(reinterpret_cast<void**>(&src), n4 * sizeof(kw_v4f)));
  KW_BW(hipMalloc(reinterpret_cast<void**>(&dst), n4 * sizeof(kw_v4f)));
  KW_BW(hipMemsetAsync(src, 0, n4 * sizeof(kw_v4f), ctx->stream));
  KW_BW(hipMemsetAsync(dst, 0, n4 * sizeof(kw_v4f), ctx->stream));
  KW_BW(hipEventCreate(&e0));
  KW_BW(hipEventCreate(&e1));
  const dim3 block(256);
  const dim3 grids[3] = { dim3(static_cast<unsigned>(ctx->cu_count) * 16u), dim3(static_cast<unsigned>((n4 + 255) / 256)),
                          dim3(static_cast<unsigned>((n4 + 1023) / 1024)) };
  double best = 0.0;
  for (int shape = 0; shape < 3; shape++)
  {
    auto launch = [&]() {
      if (shape == 0) hipLaunchKernelGGL(k_stream_copy<0>, grids[0], block, 0, ctx->stream, src, dst, n4);
      else if (shape == 1) hipLaunchKernelGGL(k_stream_copy<1>, grids[1], block, 0, ctx->stream, src, dst, n4);
      else hipLaunchKernelGGL(k_stream_copy<2>, grids[2], block, 0, ctx->stream, src, dst, n4);
    };
    launch(); // untimed first pass
    KW_BW(hipEventRecord(e0, ctx->stream));
    for (int i = 0; i < reps; i++) launch();
    KW_BW(hipGetLastError());
    KW_BW(hipEventRecord(e1, ctx->stream));
    KW_BW(hipEventSynchronize(e1));
    float ms = 0.0f;
    KW_BW(hipEventElapsedTime(&ms, e0, e1));
    const double gbs = 2.0 * static_cast<double>(n4 * sizeof(kw_v4f)) * reps / (static_cast<double>(ms) * 1e-3) / 1e9;
    if (gbs > best) best = gbs;
  }
#undef KW_BW
  *out_gbs = best;
  cleanup();
  return rc;
}

// ---- constants -----------------------------------------------------------------------------------------------------
kw_status kw_set_constants(kw_ctx* ctx, const kw_constants* k)
{
  KW_CHECK_CTX(ctx);
  KW_REQUIRE(k != nullptr);
  KW_REQUIRE(k->nx >= 1 && k->ny >= 1 && k->nz >= 1);
  KW_REQUIRE((uint64_t)k->nx * k->ny * k->nz == (uint64_t)k->n_elements);
  KW_REQUIRE(k->nx_complex == k->nx / 2 + 1 && k->ny_complex == k->ny && k->nz_complex == k->nz);
  KW_REQUIRE((uint64_t)k->nx_complex * k->ny * k->nz == (uint64_t)k->n_elements_complex);
  KW_REQUIRE(k->velocity_source_mode <= 2 && k->pressure_source_mode <= 2);
  const bool dims_changed = ctx->have_consts && (ctx->c.nx != k->nx || ctx->c.ny != k->ny || ctx->c.nz != k->nz);
  if (dims_changed) kw_fft_destroy_plans(ctx);
  ctx->c           = *k;
  ctx->have_consts = true;
  return KW_OK;
}
kw_status kw_get_constants(kw_ctx* ctx, kw_constants* out)
{
  KW_CHECK_CONSTS(ctx);
  KW_REQUIRE(out != nullptr);
  *out = ctx->c;
  return KW_OK;
}

} // extern "C"
