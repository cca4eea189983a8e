// pk_rate.hip — issue rate of packed vs scalar fp32 VALU ops on gfx950 (tuning aid, not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -o pk_rate tools/micro/pk_rate.hip && ./pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float v2f __attribute__((ext_vector_type(2)));

template<int MODE> __global__ __launch_bounds__(256) void k(float* out, int iters, float s)
{
  // 8 independent accumulator pairs per lane: enough ILP to hide the VALU latency with 4 waves per SIMD
  v2f a[8];
#pragma unroll
  for (int i = 0; i < 8; i++) a[i] = v2f{ (float)threadIdx.x + i, 1.0f + i };
  const v2f m = { s, s * 0.5f };
  for (int it = 0; it < iters; it++)
  {
#pragma unroll
    for (int i = 0; i < 8; i++)
    {
      if (MODE == 0) { a[i].x = __builtin_fmaf(a[i].x, m.x, m.y); a[i].y = __builtin_fmaf(a[i].y, m.y, m.x); }     // 2 x v_fma_f32
      else if (MODE == 1) { asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(m)); }                     // 1 x v_pk_fma_f32
      else if (MODE == 2) { a[i].x = a[i].x + m.x; a[i].y = a[i].y + m.y; }                                           // 2 x v_add_f32
      else if (MODE == 3) { asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m)); }                         // 1 x v_pk_add_f32
      else if (MODE == 4) { asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "+v"(a[i]) : "v"(m)); }
    }
  }
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < 8; i++) r += a[i].x + a[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template<int MODE> double run(float* d, int iters)
{
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const dim3 grid(256 * 4), block(256); // 4 blocks per CU = 4 waves per SIMD
  hipLaunchKernelGGL(k<MODE>, grid, block, 0, 0, d, 10, 1.0001f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, grid, block, 0, 0, d, iters, 1.0001f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main()
{
  float* d; hipMalloc(&d, 256 * 4 * 256 * sizeof(float));
  const int iters = 20000;
  const char* names[] = { "2 x v_fma_f32   ", "1 x v_pk_fma_f32", "2 x v_add_f32   ", "1 x v_pk_add_f32", "1 x v_pk_mul_f32 (op_sel/neg)" };
  double ms[5] = { run<0>(d, iters), run<1>(d, iters), run<2>(d, iters), run<3>(d, iters), run<4>(d, iters) };
  for (int m = 0; m < 5; m++)
  {
    // lane-ops: 2 fp32 results per lane per loop body element in every mode
    const double lane_results = 2.0 * 8 * iters * 256.0 * 4 * 256;
    printf("%s  %8.3f ms   %7.2f T results/s\n", names[m], ms[m], lane_results / (ms[m] * 1e-3) / 1e12);
  }
  return 0;
}
