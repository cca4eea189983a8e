// tile_rw.hip — memory patterns of the y / z line passes at grids whose spectra overflow the Infinity Cache (512^3):
// what does an in-place pass over 16-column x L-row tiles reach on its own, by tile shape, stride padding, block shape and
// tile order?  Build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/tile_rw tools/micro/tile_rw.hip ; run: tile_rw [n]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Geo { unsigned P, ny, nz, plane; }; // row pitch, rows, planes, plane pitch (complex elements)

// flat in-place float4
__global__ __launch_bounds__(256) void k_flat(float4* p, size_t n4)
{
  const size_t per = 16; // float4 per thread, contiguous per block: 256 * 16 * 16 B = 64 KB
  size_t base = (static_cast<size_t>(blockIdx.x) * 256 * per) + threadIdx.x;
  float4 v[per];
#pragma unroll
  for (int i = 0; i < (int)per; i++) v[i] = (base + i * 256 < n4) ? p[base + i * 256] : make_float4(0, 0, 0, 0);
#pragma unroll
  for (int i = 0; i < (int)per; i++) { v[i].x += 1.f; if (base + i * 256 < n4) p[base + i * 256] = v[i]; }
}

// MODE: 0 read+write, 1 read only, 2 write only.  AX: 1 lines along y (stride P), 2 lines along z (stride plane).
// block = 16 lanes x TJ threads per line; thread holds R rows (line length L = TJ * R); VEC complex per lane.
// one float4 per thread, in place (dst == src) or as a copy
__global__ __launch_bounds__(256) void k_flat1(const float4* __restrict__ src, float4* __restrict__ dst, size_t n4)
{
  const size_t e = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x;
  if (e < n4) { float4 v = src[e]; v.x += 1.f; dst[e] = v; }
}
__global__ __launch_bounds__(256) void k_flat1_inplace(float4* p, size_t n4)
{
  const size_t e = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x;
  if (e < n4) { float4 v = p[e]; v.x += 1.f; p[e] = v; }
}

template<int TJ, int R, int VEC, int MODE, int AX, bool YFAST> __global__ __launch_bounds__(16 * TJ) void k_tile(float2* a, Geo g, float* sink, float2* b = nullptr)
{
  float2* const dstp = (b != nullptr) ? b : a;
  typedef float vf __attribute__((ext_vector_type(2 * VEC)));
  const int c = threadIdx.x % 16, j = threadIdx.x / 16;
  unsigned bx = blockIdx.x, by = blockIdx.y;
  if (YFAST) { const unsigned b = by * gridDim.x + bx; by = b % gridDim.y; bx = b / gridDim.y; }
  const unsigned kx = bx * 16 * VEC + c * VEC;
  const size_t stride = (AX == 1) ? g.P : g.plane;
  const size_t base = ((AX == 1) ? static_cast<size_t>(by) * g.plane : static_cast<size_t>(by) * g.P) + kx + static_cast<size_t>(j) * stride;
  vf v[R];
  if (MODE != 2)
  {
#pragma unroll
    for (int n = 0; n < R; n++) v[n] = *reinterpret_cast<const vf*>(a + base + static_cast<size_t>(n) * TJ * stride);
  }
  else
  {
#pragma unroll
    for (int n = 0; n < R; n++) { v[n] = vf{}; v[n][0] = static_cast<float>(threadIdx.x + n); }
  }
  if (MODE == 1)
  {
    float s = 0.f;
#pragma unroll
    for (int n = 0; n < R; n++) s += v[n][0];
    if (s == 123.456f) sink[0] = s;
  }
  else
  {
#pragma unroll
    for (int n = 0; n < R; n++) { v[n][0] += 1.f; *reinterpret_cast<vf*>(dstp + base + static_cast<size_t>(n) * TJ * stride) = v[n]; }
  }
}

template<typename F> float time_ms(F launch, int reps = 8)
{
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  launch();
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < reps; i++) launch();
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  CHECK(hipGetLastError());
  return ms / reps;
}

int main(int argc, char** argv)
{
  const unsigned n = argc > 1 ? atoi(argv[1]) : 512;
  const unsigned Pbase = n / 2;
  float* sink;
  CHECK(hipMalloc(&sink, 64));
  printf("grid %u^3, rows of %u complex\n", n, Pbase);
  // plane pad (elements) and row pad
  const unsigned rowpads[] = { 0 };
  const unsigned planepads[] = { 0, 16 };
  for (unsigned rp : rowpads)
    for (unsigned pp : planepads)
    {
      Geo g{ Pbase + rp, n, n, (Pbase + rp) * n + pp };
      const size_t elems = static_cast<size_t>(g.plane) * n + 4096;
      float2 *a, *b2;
      CHECK(hipMalloc(&a, elems * sizeof(float2)));
      CHECK(hipMemset(a, 0, elems * sizeof(float2)));
      CHECK(hipMalloc(&b2, elems * sizeof(float2)));
      CHECK(hipMemset(b2, 0, elems * sizeof(float2)));
      const double bytes = 2.0 * Pbase * n * n * 8.0; // useful bytes read + written
      auto rep = [&](const char* name, float ms, double b) { printf("  rowpad %3u planepad %4u  %-34s %8.1f us  %6.2f TB/s\n", rp, pp, name, ms * 1e3, b / ms / 1e9); };
      if (pp == 0)
      {
        const size_t n4 = static_cast<size_t>(g.plane) * n / 2;
        rep("flat float4 in place", time_ms([&] { hipLaunchKernelGGL(k_flat, dim3((n4 + 4095) / 4096), dim3(256), 0, 0, reinterpret_cast<float4*>(a), n4); }), 2.0 * n4 * 16);
        rep("flat 1 float4/thread in place", time_ms([&] { hipLaunchKernelGGL(k_flat1_inplace, dim3((n4 + 255) / 256), dim3(256), 0, 0, reinterpret_cast<float4*>(a), n4); }), 2.0 * n4 * 16);
        rep("flat 1 float4/thread copy a->b", time_ms([&] { hipLaunchKernelGGL(k_flat1, dim3((n4 + 255) / 256), dim3(256), 0, 0, reinterpret_cast<const float4*>(a), reinterpret_cast<float4*>(b2), n4); }), 2.0 * n4 * 16);
        // y lines
        const dim3 gy(Pbase / 16, n);
        rep("y 16col x512, 256thr a->b", time_ms([&] { hipLaunchKernelGGL((k_tile<16, 32, 1, 0, 1, false>), gy, dim3(256), 0, 0, a, g, sink, b2); }), bytes);
        rep("y 16col x512, 128thr(8/line) r+w", time_ms([&] { hipLaunchKernelGGL((k_tile<8, 64, 1, 0, 1, false>), gy, dim3(128), 0, 0, a, g, sink, (float2*)nullptr); }), bytes);
        rep("y 16col x512, 1024thr(64/line) r+w", time_ms([&] { hipLaunchKernelGGL((k_tile<64, 8, 1, 0, 1, false>), gy, dim3(1024), 0, 0, a, g, sink, (float2*)nullptr); }), bytes);
        rep("y 16col x512, 256thr r+w", time_ms([&] { hipLaunchKernelGGL((k_tile<16, 32, 1, 0, 1, false>), gy, dim3(256), 0, 0, a, g, sink, (float2*)nullptr); }), bytes);
        rep("y 16col x512, 512thr r+w", time_ms([&] { hipLaunchKernelGGL((k_tile<32, 16, 1, 0, 1, false>), gy, dim3(512), 0, 0, a, g, sink, (float2*)nullptr); }), bytes);
        rep("y 32col x512, 512thr r+w", time_ms([&] { hipLaunchKernelGGL((k_tile<32, 16, 2, 0, 1, false>), dim3(Pbase / 32, n), dim3(512), 0, 0, a, g, sink, (float2*)nullptr); }), bytes);
        rep("y 16col read only", time_ms([&] { hipLaunchKernelGGL((k_tile<16, 32, 1, 1, 1, false>), gy, dim3(256), 0, 0, a, g, sink, (float2*)nullptr); }), bytes / 2);
        rep("y 16col write only", time_ms([&] { hipLaunchKernelGGL((k_tile<16, 32, 1, 2, 1, false>), gy, dim3(256), 0, 0, a, g, sink, (float2*)nullptr); }), bytes / 2);
      }
      const dim3 gz(Pbase / 16, n);
      rep("z 16col x512, 256thr r+w", time_ms([&] { hipLaunchKernelGGL((k_tile<16, 32, 1, 0, 2, false>), gz, dim3(256), 0, 0, a, g, sink, (float2*)nullptr); }), bytes);
      rep("z 16col x512, 512thr r+w", time_ms([&] { hipLaunchKernelGGL((k_tile<32, 16, 1, 0, 2, false>), gz, dim3(512), 0, 0, a, g, sink, (float2*)nullptr); }), bytes);
      rep("z 32col x512, 512thr r+w", time_ms([&] { hipLaunchKernelGGL((k_tile<32, 16, 2, 0, 2, false>), dim3(Pbase / 32, n), dim3(512), 0, 0, a, g, sink, (float2*)nullptr); }), bytes);
      rep("z 16col x512, 256thr a->b", time_ms([&] { hipLaunchKernelGGL((k_tile<16, 32, 1, 0, 2, false>), gz, dim3(256), 0, 0, a, g, sink, b2); }), bytes);
      rep("z 16col 256thr r+w, ky fastest", time_ms([&] { hipLaunchKernelGGL((k_tile<16, 32, 1, 0, 2, true>), gz, dim3(256), 0, 0, a, g, sink, (float2*)nullptr); }), bytes);
      rep("z 16col read only", time_ms([&] { hipLaunchKernelGGL((k_tile<16, 32, 1, 1, 2, false>), gz, dim3(256), 0, 0, a, g, sink, (float2*)nullptr); }), bytes / 2);
      rep("z 16col write only", time_ms([&] { hipLaunchKernelGGL((k_tile<16, 32, 1, 2, 2, false>), gz, dim3(256), 0, 0, a, g, sink, (float2*)nullptr); }), bytes / 2);
      CHECK(hipFree(a));
      CHECK(hipFree(b2));
    }
  return 0;
}
