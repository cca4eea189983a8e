// mall_residency.hip — does a stream of once-touched data evict a cache-resident working set from the 256 MB Infinity
// Cache, and do non-temporal accesses protect it?  (Tuning aid, not part of the library.)
//   hipcc --offload-arch=gfx950 -O3 -o mall_residency tools/micro/mall_residency.hip && ./mall_residency
// "scratch" = three 67.6 MB arrays (the spectral scratch of a 256^3 run), copied in place by a float4 kernel;
// "stream"  = 600 MB read + 200 MB written once (the state / medium arrays of an epilogue kernel), plain or non-temporal.
// Reported: time of the scratch pass right after (a) another scratch pass, (b) a plain stream, (c) a non-temporal stream.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float v4f __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_touch(v4f* p, size_t n4)
{
  const size_t e0 = static_cast<size_t>(blockIdx.x) * 1024u + threadIdx.x;
  v4f v[4];
#pragma unroll
  for (int i = 0; i < 4; i++) if (e0 + i * 256u < n4) v[i] = p[e0 + i * 256u];
#pragma unroll
  for (int i = 0; i < 4; i++) if (e0 + i * 256u < n4) { v[i].x += 1.0f; p[e0 + i * 256u] = v[i]; }
}

template<bool NT> __global__ __launch_bounds__(256) void k_stream(const v4f* __restrict__ a, const v4f* __restrict__ b,
                                                                  const v4f* __restrict__ c, v4f* __restrict__ d, size_t n4)
{
  const size_t e0 = static_cast<size_t>(blockIdx.x) * 1024u + threadIdx.x;
#pragma unroll
  for (int i = 0; i < 4; i++)
  {
    const size_t e = e0 + i * 256u;
    if (e >= n4) continue;
    v4f x, y, z;
    if (NT) { x = __builtin_nontemporal_load(a + e); y = __builtin_nontemporal_load(b + e); z = __builtin_nontemporal_load(c + e); }
    else { x = a[e]; y = b[e]; z = c[e]; }
    const v4f r = x * y + z;
    if (NT) __builtin_nontemporal_store(r, d + e);
    else d[e] = r;
  }
}

static float timed(hipEvent_t e0, hipEvent_t e1)
{
  float ms = 0.f;
  hipEventSynchronize(e1);
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f;
}

int main()
{
  const size_t scratch_bytes = 3ull * 67633152ull, stream_bytes = 200ull << 20;
  const size_t ns = scratch_bytes / 16, nt = stream_bytes / 16;
  v4f *s, *a, *b, *c, *d;
  hipMalloc(&s, scratch_bytes); hipMalloc(&a, stream_bytes); hipMalloc(&b, stream_bytes); hipMalloc(&c, stream_bytes); hipMalloc(&d, stream_bytes);
  hipMemset(s, 0, scratch_bytes); hipMemset(a, 0, stream_bytes); hipMemset(b, 0, stream_bytes); hipMemset(c, 0, stream_bytes); hipMemset(d, 0, stream_bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const dim3 gs(static_cast<unsigned>((ns + 1023) / 1024)), gt(static_cast<unsigned>((nt + 1023) / 1024)), blk(256);
  auto scratch_pass = [&]() {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_touch, gs, blk, 0, 0, s, ns);
    hipEventRecord(e1);
    return timed(e0, e1);
  };
  for (int rep = 0; rep < 3; rep++)
  {
    scratch_pass();
    const float warm = scratch_pass();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_stream<false>, gt, blk, 0, 0, a, b, c, d, nt);
    hipEventRecord(e1);
    const float t_plain = timed(e0, e1);
    const float after_plain = scratch_pass();
    scratch_pass();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_stream<true>, gt, blk, 0, 0, a, b, c, d, nt);
    hipEventRecord(e1);
    const float t_nt = timed(e0, e1);
    const float after_nt = scratch_pass();
    printf("scratch pass (203 MB in place): warm %.1f us (%.0f GB/s) | after plain stream (%.1f us, %.0f GB/s) %.1f us | after "
           "non-temporal stream (%.1f us, %.0f GB/s) %.1f us\n",
           warm, 2.0 * scratch_bytes / warm / 1e3, t_plain, 4.0 * stream_bytes / t_plain / 1e3, after_plain, t_nt,
           4.0 * stream_bytes / t_nt / 1e3, after_nt);
  }
  // out of place, ping-pong between two buffers of the same size: one array (67.6 MB each) and three (203 MB each)
  v4f* s2;
  hipMalloc(&s2, scratch_bytes);
  hipMemset(s2, 0, scratch_bytes);
  for (size_t bytes : { scratch_bytes / 3, scratch_bytes })
  {
    const size_t n = bytes / 16;
    const dim3 g(static_cast<unsigned>((n + 1023) / 1024));
    float best = 1e30f;
    for (int rep = 0; rep < 6; rep++)
    {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k_stream<false>, g, blk, 0, 0, (rep & 1) ? s2 : s, (rep & 1) ? s2 : s, (rep & 1) ? s2 : s, (rep & 1) ? s : s2, n);
      hipEventRecord(e1);
      const float t = timed(e0, e1);
      if (rep >= 2 && t < best) best = t;
    }
    float best_in = 1e30f;
    for (int rep = 0; rep < 6; rep++)
    {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k_touch, g, blk, 0, 0, s, n);
      hipEventRecord(e1);
      const float t = timed(e0, e1);
      if (rep >= 2 && t < best_in) best_in = t;
    }
    printf("%.1f MB: ping-pong between two buffers %.1f us (%.0f GB/s) | in place %.1f us (%.0f GB/s)\n", bytes / 1e6, best,
           2.0 * bytes / best / 1e3, best_in, 2.0 * bytes / best_in / 1e3);
  }
  return 0;
}
