// kw_solver_kernels.hip — hand-written gfx950 field-update / spectral / source kernels.
// Replaces namespace SolverCudaKernels (KSpaceSolver/SolverCudaKernels.cu; API SolverCudaKernels.cuh:52-500).
//
// Design (MI355X-first, not a translation of the reference's 1-D grid-stride kernels):
//   * Real-space kernels: one wave64 per x-row segment, 16 B (float4) per lane, rows on gridDim.x.  The (y,z)
//     coordinates are wave-uniform, so pml_y[y] / pml_z[z] are one scalar each and no per-element %,/ is needed
//     (the reference recovers x,y,z from the flat index per element: Utils/CudaUtils.cuh:82-102).
//   * k-space kernels: the half-spectrum row length nx/2+1 is odd, so rows are not 16-B aligned; a z-plane is
//     processed flat with 16 B (two complex) per lane, one integer division per lane, z wave-uniform.
//   * Device constants travel as a by-value kernel argument (SGPRs) instead of a __constant__ symbol.
//   * Arithmetic association order follows the cited reference lines exactly.
// All kernels are bandwidth-bound (0.1-0.3 flop/B): no LDS tiling, no MFMA.
#include "kw_internal.h"

namespace {

constexpr int kWave         = 64;
constexpr int kRowsPerBlock = 4; // block = 4 waves = 4 rows

struct RowGeom
{
  dim3 grid, block;
};

// rows on grid.x (limit 2^31), x-chunks on grid.y
inline RowGeom row_geom(const kw_constants& c, int vec)
{
  RowGeom g;
  const uint32_t rows = c.ny * c.nz;
  g.block = dim3(kWave, kRowsPerBlock, 1);
  g.grid  = dim3((rows + kRowsPerBlock - 1) / kRowsPerBlock, (c.nx + kWave * vec - 1) / (kWave * vec), 1);
  return g;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

template<typename... P> inline bool all_aligned16(P... ptrs)
{
  bool ok = true;
  const void* arr[] = { static_cast<const void*>(ptrs)... };
  for (const void* p : arr) ok = ok && (p == nullptr || aligned16(p));
  return ok;
}

// ---- small vector helpers ------------------------------------------------------------------------------------------
template<int V> struct Vec;
template<> struct Vec<4>
{
  using T = float4;
  static __device__ __forceinline__ T load(const float* p) { return *reinterpret_cast<const float4*>(p); }
  static __device__ __forceinline__ void store(float* p, const T& v) { *reinterpret_cast<float4*>(p) = v; }
};
template<> struct Vec<1>
{
  using T = float;
  static __device__ __forceinline__ T load(const float* p) { return *p; }
  static __device__ __forceinline__ void store(float* p, const T& v) { *p = v; }
};

__device__ __forceinline__ float  get(const float& v, int) { return v; }
__device__ __forceinline__ float  get(const float4& v, int k) { return k == 0 ? v.x : k == 1 ? v.y : k == 2 ? v.z : v.w; }
__device__ __forceinline__ void   put(float& v, int, float s) { v = s; }
__device__ __forceinline__ void   put(float4& v, int k, float s)
{
  if (k == 0) v.x = s; else if (k == 1) v.y = s; else if (k == 2) v.z = s; else v.w = s;
}

__device__ __forceinline__ float2 cmul(float2 a, float2 b)
{ // cuCmulf semantics (Utils/CudaUtils.cuh:159-163)
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cscale(float2 a, float s) { return make_float2(a.x * s, a.y * s); }

// Row-space coordinates of this lane.  Returns false when out of range.
template<int V>
__device__ __forceinline__ bool row_coords(const kw_constants& c, uint32_t& x, uint32_t& y, uint32_t& z, size_t& i)
{
  const uint32_t row = blockIdx.x * kRowsPerBlock + threadIdx.y; // wave-uniform
  x                  = (blockIdx.y * kWave + threadIdx.x) * V;
  if (row >= c.ny * c.nz || x >= c.nx) return false;
  z = row / c.ny;
  y = row - z * c.ny;
  i = static_cast<size_t>(row) * c.nx + x;
  return true;
}

// =====================================================================================================================
// Velocity update — SolverCudaKernels.cu:184-215 (heterogeneous), :278-308 (homogeneous uniform)
// =====================================================================================================================
template<int V, bool kHetero>
__global__ __launch_bounds__(256) void k_compute_velocity(kw_constants c, float* __restrict__ ux, float* __restrict__ uy,
                                                           float* __restrict__ uz, const float* __restrict__ gx,
                                                           const float* __restrict__ gy, const float* __restrict__ gz,
                                                           const float* __restrict__ dx_, const float* __restrict__ dy_,
                                                           const float* __restrict__ dz_, const float* __restrict__ pmlx,
                                                           const float* __restrict__ pmly, const float* __restrict__ pmlz)
{
  uint32_t x, y, z;
  size_t   i;
  if (!row_coords<V>(c, x, y, z, i)) return;
  using VT = typename Vec<V>::T;
  const float ePmlY = pmly[y];
  const float ePmlZ = pmlz[z];
  const VT    ePmlX = Vec<V>::load(pmlx + x);
  VT vux = Vec<V>::load(ux + i), vuy = Vec<V>::load(uy + i), vuz = Vec<V>::load(uz + i);
  const VT vgx = Vec<V>::load(gx + i), vgy = Vec<V>::load(gy + i), vgz = Vec<V>::load(gz + i);
  if (kHetero)
  {
    const VT vdx = Vec<V>::load(dx_ + i), vdy = Vec<V>::load(dy_ + i), vdz = Vec<V>::load(dz_ + i);
#pragma unroll
    for (int k = 0; k < V; k++)
    {
      const float eIfftX = c.fft_divider * get(vgx, k) * get(vdx, k);
      const float eIfftY = c.fft_divider * get(vgy, k) * get(vdy, k);
      const float eIfftZ = c.fft_divider * get(vgz, k) * get(vdz, k);
      const float px     = get(ePmlX, k);
      put(vux, k, (get(vux, k) * px - eIfftX) * px);
      put(vuy, k, (get(vuy, k) * ePmlY - eIfftY) * ePmlY);
      put(vuz, k, (get(vuz, k) * ePmlZ - eIfftZ) * ePmlZ);
    }
  }
  else
  {
    const float dividerX = c.dt_rho0_sgx * c.fft_divider;
    const float dividerY = c.dt_rho0_sgy * c.fft_divider;
    const float dividerZ = c.dt_rho0_sgz * c.fft_divider;
#pragma unroll
    for (int k = 0; k < V; k++)
    {
      const float px = get(ePmlX, k);
      put(vux, k, (get(vux, k) * px - dividerX * get(vgx, k)) * px);
      put(vuy, k, (get(vuy, k) * ePmlY - dividerY * get(vgy, k)) * ePmlY);
      put(vuz, k, (get(vuz, k) * ePmlZ - dividerZ * get(vgz, k)) * ePmlZ);
    }
  }
  Vec<V>::store(ux + i, vux);
  Vec<V>::store(uy + i, vuy);
  Vec<V>::store(uz + i, vuz);
}

// =====================================================================================================================
// Density update — SolverCudaKernels.cu:1358-1393 (nonlinear), :1470-1497 (linear)
// =====================================================================================================================
template<int V, bool kNonlinear, bool kRho0Scalar>
__global__ __launch_bounds__(256) void k_compute_density(kw_constants c, float* __restrict__ rx, float* __restrict__ ry,
                                                          float* __restrict__ rz, const float* __restrict__ pmlx,
                                                          const float* __restrict__ pmly, const float* __restrict__ pmlz,
                                                          const float* __restrict__ dux, const float* __restrict__ duy,
                                                          const float* __restrict__ duz, const float* __restrict__ rho0)
{
  uint32_t x, y, z;
  size_t   i;
  if (!row_coords<V>(c, x, y, z, i)) return;
  using VT = typename Vec<V>::T;
  const float ePmlY = pmly[y];
  const float ePmlZ = pmlz[z];
  const VT    ePmlX = Vec<V>::load(pmlx + x);
  VT vrx = Vec<V>::load(rx + i), vry = Vec<V>::load(ry + i), vrz = Vec<V>::load(rz + i);
  const VT vdx = Vec<V>::load(dux + i), vdy = Vec<V>::load(duy + i), vdz = Vec<V>::load(duz + i);
  VT vr0{};
  if (!kRho0Scalar) vr0 = Vec<V>::load(rho0 + i);
#pragma unroll
  for (int k = 0; k < V; k++)
  {
    const float px    = get(ePmlX, k);
    const float eRhoX = get(vrx, k), eRhoY = get(vry, k), eRhoZ = get(vrz, k);
    if (kNonlinear)
    {
      const float eRho0     = kRho0Scalar ? c.rho0 : get(vr0, k);
      const float sumRhosDt = (2.0f * (eRhoX + eRhoY + eRhoZ) + eRho0) * c.dt;
      put(vrx, k, px * ((px * eRhoX) - sumRhosDt * get(vdx, k)));
      put(vry, k, ePmlY * ((ePmlY * eRhoY) - sumRhosDt * get(vdy, k)));
      put(vrz, k, ePmlZ * ((ePmlZ * eRhoZ) - sumRhosDt * get(vdz, k)));
    }
    else
    {
      const float dtRho0 = kRho0Scalar ? c.dt_rho0 : c.dt * get(vr0, k);
      put(vrx, k, px * (px * eRhoX - dtRho0 * get(vdx, k)));
      put(vry, k, ePmlY * (ePmlY * eRhoY - dtRho0 * get(vdy, k)));
      put(vrz, k, ePmlZ * (ePmlZ * eRhoZ - dtRho0 * get(vdz, k)));
    }
  }
  Vec<V>::store(rx + i, vrx);
  Vec<V>::store(ry + i, vry);
  Vec<V>::store(rz + i, vrz);
}

// =====================================================================================================================
// Flat real-space kernels (no coordinates needed): 1-D grid, V floats per lane
// =====================================================================================================================
template<int V> __device__ __forceinline__ bool flat_index(uint32_t n, size_t& i)
{
  const size_t t = (static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x) * V;
  i              = t;
  return t < n;
}

// SolverCudaKernels.cu:1577-1602
template<int V, bool kBonAScalar, bool kRho0Scalar>
__global__ __launch_bounds__(256) void k_pressure_terms_nonlinear(
  kw_constants c, float* __restrict__ densitySum, float* __restrict__ nonlinearTerm, float* __restrict__ velGradSum,
  const float* __restrict__ rx, const float* __restrict__ ry, const float* __restrict__ rz,
  const float* __restrict__ dux, const float* __restrict__ duy, const float* __restrict__ duz,
  const float* __restrict__ bona, const float* __restrict__ rho0)
{
  size_t i;
  if (!flat_index<V>(c.n_elements, i)) return;
  using VT = typename Vec<V>::T;
  const VT vrx = Vec<V>::load(rx + i), vry = Vec<V>::load(ry + i), vrz = Vec<V>::load(rz + i);
  const VT vdx = Vec<V>::load(dux + i), vdy = Vec<V>::load(duy + i), vdz = Vec<V>::load(duz + i);
  VT vb{}, vr0{};
  if (!kBonAScalar) vb = Vec<V>::load(bona + i);
  if (!kRho0Scalar) vr0 = Vec<V>::load(rho0 + i);
  VT o1, o2, o3;
#pragma unroll
  for (int k = 0; k < V; k++)
  {
    const float eBonA   = kBonAScalar ? c.b_on_a : get(vb, k);
    const float eRho0   = kRho0Scalar ? c.rho0 : get(vr0, k);
    const float eRhoSum = (get(vrx, k) + get(vry, k) + get(vrz, k));
    const float eDuSum  = (get(vdx, k) + get(vdy, k) + get(vdz, k));
    put(o1, k, eRhoSum);
    put(o2, k, ((eBonA * eRhoSum * eRhoSum) / (2.0f * eRho0)) + eRhoSum);
    put(o3, k, eRho0 * eDuSum);
  }
  Vec<V>::store(densitySum + i, o1);
  Vec<V>::store(nonlinearTerm + i, o2);
  Vec<V>::store(velGradSum + i, o3);
}

// SolverCudaKernels.cu:1724-1742
template<int V, bool kRho0Scalar>
__global__ __launch_bounds__(256) void k_pressure_terms_linear(
  kw_constants c, float* __restrict__ densitySum, float* __restrict__ velGradSum, const float* __restrict__ rx,
  const float* __restrict__ ry, const float* __restrict__ rz, const float* __restrict__ dux,
  const float* __restrict__ duy, const float* __restrict__ duz, const float* __restrict__ rho0)
{
  size_t i;
  if (!flat_index<V>(c.n_elements, i)) return;
  using VT = typename Vec<V>::T;
  const VT vrx = Vec<V>::load(rx + i), vry = Vec<V>::load(ry + i), vrz = Vec<V>::load(rz + i);
  const VT vdx = Vec<V>::load(dux + i), vdy = Vec<V>::load(duy + i), vdz = Vec<V>::load(duz + i);
  VT vr0{};
  if (!kRho0Scalar) vr0 = Vec<V>::load(rho0 + i);
  VT o1, o2;
#pragma unroll
  for (int k = 0; k < V; k++)
  {
    const float r0    = kRho0Scalar ? c.rho0 : get(vr0, k);
    put(o1, k, get(vrx, k) + get(vry, k) + get(vrz, k));
    const float duSum = get(vdx, k) + get(vdy, k) + get(vdz, k);
    put(o2, k, r0 * duSum);
  }
  Vec<V>::store(densitySum + i, o1);
  Vec<V>::store(velGradSum + i, o2);
}

// SolverCudaKernels.cu:1865-1879 (kNonlinearForm: first = nonlinearTerm) and :1966-1980 (first = densitySum)
template<int V, bool kC2Scalar, bool kTauEtaScalar>
__global__ __launch_bounds__(256) void k_sum_pressure_terms(kw_constants c, float* __restrict__ p,
                                                             const float* __restrict__ first,
                                                             const float* __restrict__ tauTerm,
                                                             const float* __restrict__ etaTerm,
                                                             const float* __restrict__ c2, const float* __restrict__ tau,
                                                             const float* __restrict__ eta)
{
  size_t i;
  if (!flat_index<V>(c.n_elements, i)) return;
  using VT = typename Vec<V>::T;
  const VT vf = Vec<V>::load(first + i), vt = Vec<V>::load(tauTerm + i), ve = Vec<V>::load(etaTerm + i);
  VT vc{}, vtau{}, veta{};
  if (!kC2Scalar) vc = Vec<V>::load(c2 + i);
  if (!kTauEtaScalar)
  {
    vtau = Vec<V>::load(tau + i);
    veta = Vec<V>::load(eta + i);
  }
  VT o;
#pragma unroll
  for (int k = 0; k < V; k++)
  {
    const float ec2  = kC2Scalar ? c.c2 : get(vc, k);
    const float etau = kTauEtaScalar ? c.absorb_tau : get(vtau, k);
    const float eeta = kTauEtaScalar ? c.absorb_eta : get(veta, k);
    put(o, k, ec2 * (get(vf, k) + (c.fft_divider * ((get(vt, k) * etau) - (get(ve, k) * eeta)))));
  }
  Vec<V>::store(p + i, o);
}

// SolverCudaKernels.cu:2067-2084
template<int V, bool kC2Scalar, bool kBonAScalar, bool kRho0Scalar>
__global__ __launch_bounds__(256) void k_sum_pressure_nonlinear_lossless(
  kw_constants c, float* __restrict__ p, const float* __restrict__ rx, const float* __restrict__ ry,
  const float* __restrict__ rz, const float* __restrict__ c2, const float* __restrict__ bona,
  const float* __restrict__ rho0)
{
  size_t i;
  if (!flat_index<V>(c.n_elements, i)) return;
  using VT = typename Vec<V>::T;
  const VT vrx = Vec<V>::load(rx + i), vry = Vec<V>::load(ry + i), vrz = Vec<V>::load(rz + i);
  VT vc{}, vb{}, vr0{};
  if (!kC2Scalar) vc = Vec<V>::load(c2 + i);
  if (!kBonAScalar) vb = Vec<V>::load(bona + i);
  if (!kRho0Scalar) vr0 = Vec<V>::load(rho0 + i);
  VT o;
#pragma unroll
  for (int k = 0; k < V; k++)
  {
    const float ec2    = kC2Scalar ? c.c2 : get(vc, k);
    const float eb     = kBonAScalar ? c.b_on_a : get(vb, k);
    const float er0    = kRho0Scalar ? c.rho0 : get(vr0, k);
    const float rhoSum = get(vrx, k) + get(vry, k) + get(vrz, k);
    put(o, k, ec2 * (rhoSum + (eb * (rhoSum * rhoSum) / (2.0f * er0))));
  }
  Vec<V>::store(p + i, o);
}

// SolverCudaKernels.cu:2224-2236
template<int V, bool kC2Scalar>
__global__ __launch_bounds__(256) void k_sum_pressure_linear_lossless(kw_constants c, float* __restrict__ p,
                                                                       const float* __restrict__ rx,
                                                                       const float* __restrict__ ry,
                                                                       const float* __restrict__ rz,
                                                                       const float* __restrict__ c2)
{
  size_t i;
  if (!flat_index<V>(c.n_elements, i)) return;
  using VT = typename Vec<V>::T;
  const VT vrx = Vec<V>::load(rx + i), vry = Vec<V>::load(ry + i), vrz = Vec<V>::load(rz + i);
  VT vc{};
  if (!kC2Scalar) vc = Vec<V>::load(c2 + i);
  VT o;
#pragma unroll
  for (int k = 0; k < V; k++)
  {
    const float ec2        = kC2Scalar ? c.c2 : get(vc, k);
    const float sumDensity = get(vrx, k) + get(vry, k) + get(vrz, k);
    put(o, k, ec2 * sumDensity);
  }
  Vec<V>::store(p + i, o);
}

// SolverCudaKernels.cu:864-884
template<int V, bool kC2Scalar>
__global__ __launch_bounds__(256) void k_add_initial_pressure_source(kw_constants c, float* __restrict__ p,
                                                                      float* __restrict__ rx, float* __restrict__ ry,
                                                                      float* __restrict__ rz,
                                                                      const float* __restrict__ p0,
                                                                      const float* __restrict__ c2)
{
  size_t i;
  if (!flat_index<V>(c.n_elements, i)) return;
  using VT = typename Vec<V>::T;
  const VT v = Vec<V>::load(p0 + i);
  VT vc{};
  if (!kC2Scalar) vc = Vec<V>::load(c2 + i);
  VT o, zero;
  // the initial density is split over the axes of the simulation: 3, or 2 when Nz == 1 (:873 dimScalingFactor)
  const float dimScalingFactor = (c.nz == 1) ? 2.0f : 3.0f;
#pragma unroll
  for (int k = 0; k < V; k++)
  {
    const float ec2 = kC2Scalar ? c.c2 : get(vc, k);
    put(o, k, get(v, k) / (dimScalingFactor * ec2));
    put(zero, k, 0.0f);
  }
  Vec<V>::store(p + i, v);
  Vec<V>::store(rx + i, o);
  Vec<V>::store(ry + i, o);
  Vec<V>::store(rz + i, (c.nz == 1) ? zero : o);
}

// SolverCudaKernels.cu:949-982
template<int V, bool kRho0Scalar>
__global__ __launch_bounds__(256) void k_compute_initial_velocity(kw_constants c, float* __restrict__ ux,
                                                                   float* __restrict__ uy, float* __restrict__ uz,
                                                                   const float* __restrict__ dx_,
                                                                   const float* __restrict__ dy_,
                                                                   const float* __restrict__ dz_)
{
  size_t i;
  if (!flat_index<V>(c.n_elements, i)) return;
  using VT = typename Vec<V>::T;
  VT vux = Vec<V>::load(ux + i), vuy = Vec<V>::load(uy + i), vuz = Vec<V>::load(uz + i);
  if (kRho0Scalar)
  {
    const float dividerX = c.fft_divider * 0.5f * c.dt_rho0_sgx;
    const float dividerY = c.fft_divider * 0.5f * c.dt_rho0_sgy;
    const float dividerZ = c.fft_divider * 0.5f * c.dt_rho0_sgz;
#pragma unroll
    for (int k = 0; k < V; k++)
    {
      put(vux, k, get(vux, k) * dividerX);
      put(vuy, k, get(vuy, k) * dividerY);
      put(vuz, k, get(vuz, k) * dividerZ);
    }
  }
  else
  {
    const float divider = c.fft_divider * 0.5f;
    const VT vdx = Vec<V>::load(dx_ + i), vdy = Vec<V>::load(dy_ + i), vdz = Vec<V>::load(dz_ + i);
#pragma unroll
    for (int k = 0; k < V; k++)
    {
      put(vux, k, get(vux, k) * (get(vdx, k) * divider));
      put(vuy, k, get(vuy, k) * (get(vdy, k) * divider));
      put(vuz, k, get(vuz, k) * (get(vdz, k) * divider));
    }
  }
  Vec<V>::store(ux + i, vux);
  Vec<V>::store(uy + i, vuy);
  Vec<V>::store(uz + i, vuz);
}

// SolverCudaKernels.cu:765-770 (one array) and :795-807 (three arrays)
template<int V, int kArrays>
__global__ __launch_bounds__(256) void k_add_scaled_source(kw_constants c, float* __restrict__ a0,
                                                            float* __restrict__ a1, float* __restrict__ a2,
                                                            const float* __restrict__ scaled)
{
  size_t i;
  if (!flat_index<V>(c.n_elements, i)) return;
  using VT = typename Vec<V>::T;
  const VT s  = Vec<V>::load(scaled + i);
  float*   arrs[3] = { a0, a1, a2 };
#pragma unroll
  for (int a = 0; a < kArrays; a++)
  {
    VT v = Vec<V>::load(arrs[a] + i);
#pragma unroll
    for (int k = 0; k < V; k++) put(v, k, get(v, k) + get(s, k));
    Vec<V>::store(arrs[a] + i, v);
  }
}

// SolverCudaKernels.cu:1285-1301: one x-row per block row (blockIdx.y = y + ny * z): y and z are block-uniform
__global__ __launch_bounds__(256) void k_velocity_gradient_shift_nonuniform(kw_constants c, float* __restrict__ dux,
                                                                             float* __restrict__ duy,
                                                                             float* __restrict__ duz,
                                                                             const float* __restrict__ nx,
                                                                             const float* __restrict__ ny,
                                                                             const float* __restrict__ nz)
{
  const uint32_t row = blockIdx.y;
  const uint32_t z   = row / c.ny;
  const uint32_t y   = row - z * c.ny;
  const float    ey = ny[y], ez = nz[z];
  for (uint32_t x = blockIdx.x * blockDim.x + threadIdx.x; x < c.nx; x += gridDim.x * blockDim.x)
  {
    const size_t i = static_cast<size_t>(row) * c.nx + x;
    dux[i] *= nx[x];
    duy[i] *= ey;
    duz[i] *= ez;
  }
}

// =====================================================================================================================
// k-space kernels.  Flat over one z-plane (blockIdx.y = z), P complex values per lane (P = 2 -> 16 B).
// =====================================================================================================================
template<int P> struct CVec;
template<> struct CVec<2>
{
  static __device__ __forceinline__ void load(const float* p, float2 (&v)[2])
  {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0]           = make_float2(t.x, t.y);
    v[1]           = make_float2(t.z, t.w);
  }
  static __device__ __forceinline__ void store(float* p, const float2 (&v)[2])
  {
    *reinterpret_cast<float4*>(p) = make_float4(v[0].x, v[0].y, v[1].x, v[1].y);
  }
  static __device__ __forceinline__ void loadr(const float* p, float (&v)[2])
  {
    const float2 t = *reinterpret_cast<const float2*>(p);
    v[0]           = t.x;
    v[1]           = t.y;
  }
};
template<> struct CVec<1>
{
  static __device__ __forceinline__ void load(const float* p, float2 (&v)[1]) { v[0] = *reinterpret_cast<const float2*>(p); }
  static __device__ __forceinline__ void store(float* p, const float2 (&v)[1]) { *reinterpret_cast<float2*>(p) = v[0]; }
  static __device__ __forceinline__ void loadr(const float* p, float (&v)[1]) { v[0] = *p; }
};

// coordinates of the P consecutive complex elements handled by this lane inside plane z
template<int P>
__device__ __forceinline__ bool plane_coords(const kw_constants& c, uint32_t (&x)[P], uint32_t (&y)[P], uint32_t& z,
                                             size_t& i)
{
  const uint32_t plane = c.nx_complex * c.ny;
  const uint32_t e0    = (blockIdx.x * blockDim.x + threadIdx.x) * P;
  if (e0 >= plane) return false;
  z    = blockIdx.y;
  y[0] = e0 / c.nx_complex;
  x[0] = e0 - y[0] * c.nx_complex;
#pragma unroll
  for (int k = 1; k < P; k++)
  {
    x[k] = x[k - 1] + 1;
    y[k] = y[k - 1];
    if (x[k] == c.nx_complex) { x[k] = 0; y[k]++; }
  }
  i = static_cast<size_t>(z) * plane + e0;
  return true;
}

// SolverCudaKernels.cu:1139-1157
template<int P>
__global__ __launch_bounds__(256) void k_pressure_gradient(kw_constants c, float* __restrict__ X, float* __restrict__ Y,
                                                            float* __restrict__ Z, const float* __restrict__ kappa,
                                                            const float2* __restrict__ ddx,
                                                            const float2* __restrict__ ddy,
                                                            const float2* __restrict__ ddz)
{
  uint32_t x[P], y[P], z;
  size_t   i;
  if (!plane_coords<P>(c, x, y, z, i)) return;
  float2 vx[P], ox[P], oy[P], oz[P];
  float  vk[P];
  CVec<P>::load(X + 2 * i, vx);
  CVec<P>::loadr(kappa + i, vk);
  const float2 eDdz = ddz[z];
#pragma unroll
  for (int k = 0; k < P; k++)
  {
    const float2 eKappa = cscale(vx[k], vk[k]);
    ox[k]               = cmul(eKappa, ddx[x[k]]);
    oy[k]               = cmul(eKappa, ddy[y[k]]);
    oz[k]               = cmul(eKappa, eDdz);
  }
  CVec<P>::store(X + 2 * i, ox);
  CVec<P>::store(Y + 2 * i, oy);
  CVec<P>::store(Z + 2 * i, oz);
}

// SolverCudaKernels.cu:1210-1239
template<int P>
__global__ __launch_bounds__(256) void k_velocity_gradient(kw_constants c, float* __restrict__ X, float* __restrict__ Y,
                                                            float* __restrict__ Z, const float* __restrict__ kappa,
                                                            const float2* __restrict__ ddx,
                                                            const float2* __restrict__ ddy,
                                                            const float2* __restrict__ ddz)
{
  uint32_t x[P], y[P], z;
  size_t   i;
  if (!plane_coords<P>(c, x, y, z, i)) return;
  float2 vx[P], vy[P], vz[P];
  float  vk[P];
  CVec<P>::load(X + 2 * i, vx);
  CVec<P>::load(Y + 2 * i, vy);
  CVec<P>::load(Z + 2 * i, vz);
  CVec<P>::loadr(kappa + i, vk);
  const float2 eDdz = ddz[z];
#pragma unroll
  for (int k = 0; k < P; k++)
  {
    const float eKappa = vk[k] * c.fft_divider;
    vx[k]              = cmul(cscale(vx[k], eKappa), ddx[x[k]]);
    vy[k]              = cmul(cscale(vy[k], eKappa), ddy[y[k]]);
    vz[k]              = cmul(cscale(vz[k], eKappa), eDdz);
  }
  CVec<P>::store(X + 2 * i, vx);
  CVec<P>::store(Y + 2 * i, vy);
  CVec<P>::store(Z + 2 * i, vz);
}

// flat complex index over the whole half-spectrum, P complex per lane
template<int P> __device__ __forceinline__ bool flat_cindex(uint32_t n, size_t& i)
{
  const size_t t = (static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x) * P;
  i              = t;
  return t < n;
}

// SolverCudaKernels.cu:1812-1820
template<int P>
__global__ __launch_bounds__(256) void k_absorbtion_term(kw_constants c, float* __restrict__ A, float* __restrict__ B,
                                                          const float* __restrict__ n1, const float* __restrict__ n2)
{
  size_t i;
  if (!flat_cindex<P>(c.n_elements_complex, i)) return;
  float2 va[P], vb[P];
  float  v1[P], v2[P];
  CVec<P>::load(A + 2 * i, va);
  CVec<P>::load(B + 2 * i, vb);
  CVec<P>::loadr(n1 + i, v1);
  CVec<P>::loadr(n2 + i, v2);
#pragma unroll
  for (int k = 0; k < P; k++)
  {
    va[k] = cscale(va[k], v1[k]);
    vb[k] = cscale(vb[k], v2[k]);
  }
  CVec<P>::store(A + 2 * i, va);
  CVec<P>::store(B + 2 * i, vb);
}

// SolverCudaKernels.cu:740-745
template<int P>
__global__ __launch_bounds__(256) void k_source_gradient(kw_constants c, float* __restrict__ S,
                                                          const float* __restrict__ sk)
{
  size_t i;
  if (!flat_cindex<P>(c.n_elements_complex, i)) return;
  float2 vs[P];
  float  vk[P];
  CVec<P>::load(S + 2 * i, vs);
  CVec<P>::loadr(sk + i, vk);
#pragma unroll
  for (int k = 0; k < P; k++) vs[k] = cscale(vs[k], vk[k] * c.fft_divider);
  CVec<P>::store(S + 2 * i, vs);
}

// SolverCudaKernels.cu:2617-2689.  Spectrum layout [oz][oy][ox] with the transformed axis shortened; one complex per
// lane over a 3-D grid (x on threads, y on grid.y, z on grid.z) — config-5 only, not a roofline kernel.
__global__ __launch_bounds__(256) void k_velocity_shift(float* __restrict__ T, const float2* __restrict__ shift, int axis,
                                                         uint32_t ox, uint32_t oy, uint32_t oz, float divider)
{
  const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t y = blockIdx.y;
  const uint32_t z = blockIdx.z;
  if (x >= ox) return;
  const size_t   i = (static_cast<size_t>(z) * oy + y) * ox + x;
  const uint32_t k = (axis == 0) ? x : (axis == 1) ? y : z;
  float2*        t = reinterpret_cast<float2*>(T) + i;
  *t               = cscale(cmul(*t, shift[k]), divider);
}

// =====================================================================================================================
// Scatter sources (tiny): 1-D grid over the source size
// =====================================================================================================================
// SolverCudaKernels.cu:463-471
__global__ void k_add_transducer_source(uint32_t n, float* __restrict__ ux, const uint64_t* __restrict__ index,
                                        const float* __restrict__ input, const uint64_t* __restrict__ delay, uint64_t t)
{
  for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += static_cast<size_t>(gridDim.x) * blockDim.x)
    ux[index[i]] += input[delay[i] + t];
}

// SolverCudaKernels.cu:504-528
__global__ void k_add_velocity_source(uint32_t n, uint32_t mode, uint32_t many, float* __restrict__ u,
                                      const float* __restrict__ input, const uint64_t* __restrict__ index, uint64_t t)
{
  const uint64_t index2D = (many == 0) ? t : t * n;
  for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += static_cast<size_t>(gridDim.x) * blockDim.x)
  {
    const float v = (many == 0) ? input[index2D] : input[index2D + i];
    if (mode == KW_SRC_DIRICHLET) u[index[i]] = v;
    else if (mode == KW_SRC_ADDITIVE_NO_CORRECTION) u[index[i]] += v;
  }
}

// SolverCudaKernels.cu:570-629
__global__ void k_add_pressure_source(uint32_t n, uint32_t mode, uint32_t many, float* __restrict__ rx,
                                      float* __restrict__ ry, float* __restrict__ rz, const float* __restrict__ input,
                                      const uint64_t* __restrict__ index, uint64_t t, bool is3D)
{
  const uint64_t index2D = (many == 0) ? t : t * n;
  for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += static_cast<size_t>(gridDim.x) * blockDim.x)
  {
    const float    v = (many == 0) ? input[index2D] : input[index2D + i];
    const uint64_t j = index[i];
    if (mode == KW_SRC_DIRICHLET) { rx[j] = v; ry[j] = v; if (is3D) rz[j] = v; }
    else if (mode == KW_SRC_ADDITIVE_NO_CORRECTION) { rx[j] += v; ry[j] += v; if (is3D) rz[j] += v; }
  }
}

// SolverCudaKernels.cu:679-697
__global__ void k_insert_source(uint64_t n, int many, float* __restrict__ scaled, const float* __restrict__ input,
                                const uint64_t* __restrict__ index, uint64_t t)
{
  const uint64_t index2D = many ? t * n : t;
  for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += static_cast<size_t>(gridDim.x) * blockDim.x)
    scaled[index[i]] = many ? input[index2D + i] : input[index2D];
}

inline dim3 grid1d(size_t work, int block = 256)
{
  size_t g = (work + block - 1) / block;
  if (g == 0) g = 1;
  return dim3(static_cast<unsigned>(g), 1, 1);
}

} // namespace

// =====================================================================================================================
// C-ABI wrappers
// =====================================================================================================================
#define LAUNCH(kernel, grid, block, ...)                                                                               \
  do {                                                                                                                 \
    hipLaunchKernelGGL(kernel, grid, block, 0, ctx->stream, __VA_ARGS__);                                              \
    KW_LAUNCH_CHECK();                                                                                                 \
  } while (0)

template<bool kNonlinear>
static kw_status density_impl(kw_ctx* ctx, float* rx, float* ry, float* rz, const float* pmlx, const float* pmly,
                              const float* pmlz, const float* dux, const float* duy, const float* duz,
                              const float* rho0)
{
  KW_CHECK_CONSTS(ctx);
  KW_PROF(ctx, kNonlinear ? "compute_density_nonlinear" : "compute_density_linear");
  KW_REQUIRE(rx && ry && rz && pmlx && pmly && pmlz && dux && duy && duz);
  const kw_constants& c = ctx->c;
  const bool v4 = (c.nx % 4 == 0) && all_aligned16(rx, ry, rz, dux, duy, duz, rho0, pmlx);
  if (v4)
  {
    RowGeom g = row_geom(c, 4);
    if (rho0) LAUNCH((k_compute_density<4, kNonlinear, false>), g.grid, g.block, c, rx, ry, rz, pmlx, pmly, pmlz, dux, duy, duz, rho0);
    else      LAUNCH((k_compute_density<4, kNonlinear, true>), g.grid, g.block, c, rx, ry, rz, pmlx, pmly, pmlz, dux, duy, duz, rho0);
  }
  else
  {
    RowGeom g = row_geom(c, 1);
    if (rho0) LAUNCH((k_compute_density<1, kNonlinear, false>), g.grid, g.block, c, rx, ry, rz, pmlx, pmly, pmlz, dux, duy, duz, rho0);
    else      LAUNCH((k_compute_density<1, kNonlinear, true>), g.grid, g.block, c, rx, ry, rz, pmlx, pmly, pmlz, dux, duy, duz, rho0);
  }
  return KW_OK;
}

extern "C" {

kw_status kw_compute_velocity(kw_ctx* ctx, float* ux, float* uy, float* uz, const float* gx, const float* gy,
                              const float* gz, const float* dx_, const float* dy_, const float* dz_,
                              const float* pmlx, const float* pmly, const float* pmlz)
{
  KW_CHECK_CONSTS(ctx);
  KW_PROF(ctx, "compute_velocity");
  KW_REQUIRE(ux && uy && uz && gx && gy && gz && pmlx && pmly && pmlz);
  const bool het = (dx_ != nullptr);
  KW_REQUIRE(het ? (dy_ != nullptr && dz_ != nullptr) : (dy_ == nullptr && dz_ == nullptr));
  const kw_constants& c = ctx->c;
  const bool v4 = (c.nx % 4 == 0) && all_aligned16(ux, uy, uz, gx, gy, gz, dx_, dy_, dz_, pmlx);
  if (v4)
  {
    RowGeom g = row_geom(c, 4);
    if (het) LAUNCH((k_compute_velocity<4, true>), g.grid, g.block, c, ux, uy, uz, gx, gy, gz, dx_, dy_, dz_, pmlx, pmly, pmlz);
    else     LAUNCH((k_compute_velocity<4, false>), g.grid, g.block, c, ux, uy, uz, gx, gy, gz, dx_, dy_, dz_, pmlx, pmly, pmlz);
  }
  else
  {
    RowGeom g = row_geom(c, 1);
    if (het) LAUNCH((k_compute_velocity<1, true>), g.grid, g.block, c, ux, uy, uz, gx, gy, gz, dx_, dy_, dz_, pmlx, pmly, pmlz);
    else     LAUNCH((k_compute_velocity<1, false>), g.grid, g.block, c, ux, uy, uz, gx, gy, gz, dx_, dy_, dz_, pmlx, pmly, pmlz);
  }
  return KW_OK;
}

kw_status kw_add_transducer_source(kw_ctx* ctx, float* ux, const uint64_t* index, const float* input,
                                   const uint64_t* delay, uint64_t t)
{
  KW_CHECK_CONSTS(ctx);
  KW_PROF(ctx, "add_transducer_source");
  const uint32_t n = ctx->c.velocity_source_size;
  if (n == 0) return KW_OK;
  KW_REQUIRE(ux && index && input && delay);
  LAUNCH(k_add_transducer_source, grid1d(n), dim3(256), n, ux, index, input, delay, t);
  return KW_OK;
}

kw_status kw_add_velocity_source(kw_ctx* ctx, float* u, const float* input, const uint64_t* index, uint64_t t)
{
  KW_CHECK_CONSTS(ctx);
  KW_PROF(ctx, "add_velocity_source");
  const uint32_t n = ctx->c.velocity_source_size;
  if (n == 0) return KW_OK;
  KW_REQUIRE(u && input && index);
  LAUNCH(k_add_velocity_source, grid1d(n), dim3(256), n, ctx->c.velocity_source_mode, ctx->c.velocity_source_many, u,
         input, index, t);
  return KW_OK;
}

kw_status kw_add_pressure_source(kw_ctx* ctx, float* rx, float* ry, float* rz, const float* input,
                                 const uint64_t* index, uint64_t t)
{
  KW_CHECK_CONSTS(ctx);
  KW_PROF(ctx, "add_pressure_source");
  const uint32_t n = ctx->c.pressure_source_size;
  if (n == 0) return KW_OK;
  KW_REQUIRE(rx && ry && rz && input && index);
  LAUNCH(k_add_pressure_source, grid1d(n), dim3(256), n, ctx->c.pressure_source_mode, ctx->c.pressure_source_many, rx,
         ry, rz, input, index, t, ctx->c.nz != 1); // 2-D (Nz == 1): rho_x, rho_y only (:588-599, :611-622)
  return KW_OK;
}

kw_status kw_insert_source_into_scaling_matrix(kw_ctx* ctx, float* scaled, const float* input, const uint64_t* index,
                                               uint64_t n, int many, uint64_t t)
{
  KW_CHECK_CONSTS(ctx);
  KW_PROF(ctx, "insert_source_into_scaling_matrix");
  if (n == 0) return KW_OK;
  KW_REQUIRE(scaled && input && index);
  LAUNCH(k_insert_source, grid1d(n), dim3(256), n, many, scaled, input, index, t);
  return KW_OK;
}

kw_status kw_compute_source_gradient(kw_ctx* ctx, float* S, const float* sk)
{
  KW_CHECK_CONSTS(ctx);
  KW_PROF(ctx, "compute_source_gradient");
  KW_REQUIRE(S && sk);
  const kw_constants& c = ctx->c;
  if ((c.n_elements_complex % 2 == 0) && all_aligned16(S) && ((reinterpret_cast<uintptr_t>(sk) & 7u) == 0))
    LAUNCH((k_source_gradient<2>), grid1d(c.n_elements_complex / 2), dim3(256), c, S, sk);
  else
    LAUNCH((k_source_gradient<1>), grid1d(c.n_elements_complex), dim3(256), c, S, sk);
  return KW_OK;
}

kw_status kw_add_velocity_scaled_source(kw_ctx* ctx, float* u, const float* scaled)
{
  KW_CHECK_CONSTS(ctx);
  KW_PROF(ctx, "add_velocity_scaled_source");
  KW_REQUIRE(u && scaled);
  const kw_constants& c = ctx->c;
  if (c.n_elements % 4 == 0 && all_aligned16(u, scaled))
    LAUNCH((k_add_scaled_source<4, 1>), grid1d(c.n_elements / 4), dim3(256), c, u, (float*)nullptr, (float*)nullptr, scaled);
  else
    LAUNCH((k_add_scaled_source<1, 1>), grid1d(c.n_elements), dim3(256), c, u, (float*)nullptr, (float*)nullptr, scaled);
  return KW_OK;
}

kw_status kw_add_pressure_scaled_source(kw_ctx* ctx, float* rx, float* ry, float* rz, const float* scaled)
{
  KW_CHECK_CONSTS(ctx);
  KW_PROF(ctx, "add_pressure_scaled_source");
  KW_REQUIRE(rx && ry && rz && scaled);
  const kw_constants& c = ctx->c;
  if (c.nz == 1)
  { // 2-D: rho_x, rho_y only (:795-807, simulationDimension == k2D)
    if (c.n_elements % 4 == 0 && all_aligned16(rx, ry, scaled))
      LAUNCH((k_add_scaled_source<4, 2>), grid1d(c.n_elements / 4), dim3(256), c, rx, ry, (float*)nullptr, scaled);
    else
      LAUNCH((k_add_scaled_source<1, 2>), grid1d(c.n_elements), dim3(256), c, rx, ry, (float*)nullptr, scaled);
    return KW_OK;
  }
  if (c.n_elements % 4 == 0 && all_aligned16(rx, ry, rz, scaled))
    LAUNCH((k_add_scaled_source<4, 3>), grid1d(c.n_elements / 4), dim3(256), c, rx, ry, rz, scaled);
  else
    LAUNCH((k_add_scaled_source<1, 3>), grid1d(c.n_elements), dim3(256), c, rx, ry, rz, scaled);
  return KW_OK;
}

kw_status kw_add_initial_pressure_source(kw_ctx* ctx, float* p, float* rx, float* ry, float* rz, const float* p0,
                                         const float* c2)
{
  KW_CHECK_CONSTS(ctx);
  KW_PROF(ctx, "add_initial_pressure_source");
  KW_REQUIRE(p && rx && ry && rz && p0);
  const kw_constants& c = ctx->c;
  const bool v4 = c.n_elements % 4 == 0 && all_aligned16(p, rx, ry, rz, p0, c2);
  const dim3 g  = v4 ? grid1d(c.n_elements / 4) : grid1d(c.n_elements);
  if (v4)
  {
    if (c2) LAUNCH((k_add_initial_pressure_source<4, false>), g, dim3(256), c, p, rx, ry, rz, p0, c2);
    else    LAUNCH((k_add_initial_pressure_source<4, true>), g, dim3(256), c, p, rx, ry, rz, p0, c2);
  }
  else
  {
    if (c2) LAUNCH((k_add_initial_pressure_source<1, false>), g, dim3(256), c, p, rx, ry, rz, p0, c2);
    else    LAUNCH((k_add_initial_pressure_source<1, true>), g, dim3(256), c, p, rx, ry, rz, p0, c2);
  }
  return KW_OK;
}

kw_status kw_compute_initial_velocity(kw_ctx* ctx, float* ux, float* uy, float* uz, const float* dx_,
                                      const float* dy_, const float* dz_)
{
  KW_CHECK_CONSTS(ctx);
  KW_PROF(ctx, "compute_initial_velocity");
  KW_REQUIRE(ux && uy && uz);
  const bool het = (dx_ != nullptr);
  KW_REQUIRE(het ? (dy_ != nullptr && dz_ != nullptr) : (dy_ == nullptr && dz_ == nullptr));
  const kw_constants& c = ctx->c;
  const bool v4 = c.n_elements % 4 == 0 && all_aligned16(ux, uy, uz, dx_, dy_, dz_);
  const dim3 g  = v4 ? grid1d(c.n_elements / 4) : grid1d(c.n_elements);
  if (v4)
  {
    if (het) LAUNCH((k_compute_initial_velocity<4, false>), g, dim3(256), c, ux, uy, uz, dx_, dy_, dz_);
    else     LAUNCH((k_compute_initial_velocity<4, true>), g, dim3(256), c, ux, uy, uz, dx_, dy_, dz_);
  }
  else
  {
    if (het) LAUNCH((k_compute_initial_velocity<1, false>), g, dim3(256), c, ux, uy, uz, dx_, dy_, dz_);
    else     LAUNCH((k_compute_initial_velocity<1, true>), g, dim3(256), c, ux, uy, uz, dx_, dy_, dz_);
  }
  return KW_OK;
}

static inline bool cplx_pairs_ok(const kw_constants& c) { return ((c.nx_complex * c.ny) % 2u) == 0; }

kw_status kw_compute_pressure_gradient(kw_ctx* ctx, float* X, float* Y, float* Z, const float* kappa,
                                       const float* ddx, const float* ddy, const float* ddz)
{
  KW_CHECK_CONSTS(ctx);
  KW_PROF(ctx, "compute_pressure_gradient");
  KW_REQUIRE(X && Y && Z && kappa && ddx && ddy && ddz);
  const kw_constants& c = ctx->c;
  const uint32_t plane  = c.nx_complex * c.ny;
  if (cplx_pairs_ok(c) && all_aligned16(X, Y, Z) && ((reinterpret_cast<uintptr_t>(kappa) & 7u) == 0))
    LAUNCH((k_pressure_gradient<2>), dim3((plane / 2 + 255) / 256, c.nz), dim3(256), c, X, Y, Z, kappa,
           (const float2*)ddx, (const float2*)ddy, (const float2*)ddz);
  else
    LAUNCH((k_pressure_gradient<1>), dim3((plane + 255) / 256, c.nz), dim3(256), c, X, Y, Z, kappa, (const float2*)ddx,
           (const float2*)ddy, (const float2*)ddz);
  return KW_OK;
}

kw_status kw_compute_velocity_gradient(kw_ctx* ctx, float* X, float* Y, float* Z, const float* kappa,
                                       const float* ddx, const float* ddy, const float* ddz)
{
  KW_CHECK_CONSTS(ctx);
  KW_PROF(ctx, "compute_velocity_gradient");
  KW_REQUIRE(X && Y && Z && kappa && ddx && ddy && ddz);
  const kw_constants& c = ctx->c;
  const uint32_t plane  = c.nx_complex * c.ny;
  if (cplx_pairs_ok(c) && all_aligned16(X, Y, Z) && ((reinterpret_cast<uintptr_t>(kappa) & 7u) == 0))
    LAUNCH((k_velocity_gradient<2>), dim3((plane / 2 + 255) / 256, c.nz), dim3(256), c, X, Y, Z, kappa,
           (const float2*)ddx, (const float2*)ddy, (const float2*)ddz);
  else
    LAUNCH((k_velocity_gradient<1>), dim3((plane + 255) / 256, c.nz), dim3(256), c, X, Y, Z, kappa, (const float2*)ddx,
           (const float2*)ddy, (const float2*)ddz);
  return KW_OK;
}

kw_status kw_compute_velocity_gradient_shift_nonuniform(kw_ctx* ctx, float* dux, float* duy, float* duz, const float* nx,
                                                        const float* ny, const float* nz)
{
  KW_CHECK_CONSTS(ctx);
  KW_PROF(ctx, "compute_velocity_gradient_shift_nonuniform");
  KW_REQUIRE(dux && duy && duz && nx && ny && nz);
  const kw_constants& c = ctx->c;
  KW_REQUIRE(static_cast<uint64_t>(c.ny) * c.nz <= 65535u * 65535u);
  // rows on grid.y would overflow 65535 for big grids: fold rows into grid.x/y = (x blocks, rows) only when they fit
  const uint32_t rows = c.ny * c.nz;
  if (rows <= 65535u)
    LAUNCH(k_velocity_gradient_shift_nonuniform, dim3((c.nx + 255) / 256, rows), dim3(256), c, dux, duy, duz, nx, ny, nz);
  else
  { // one launch per z-range of at most 65535 rows
    const uint32_t zstep = 65535u / c.ny;
    for (uint32_t z0 = 0; z0 < c.nz; z0 += zstep)
    {
      const uint32_t nzc = (z0 + zstep <= c.nz) ? zstep : c.nz - z0;
      const size_t   off = static_cast<size_t>(z0) * c.ny * c.nx;
      kw_constants cc = c;
      cc.nz = nzc;
      LAUNCH(k_velocity_gradient_shift_nonuniform, dim3((c.nx + 255) / 256, nzc * c.ny), dim3(256), cc, dux + off,
             duy + off, duz + off, nx, ny, nz + z0);
    }
  }
  return KW_OK;
}

kw_status kw_compute_density_nonlinear(kw_ctx* ctx, float* rx, float* ry, float* rz, const float* pmlx,
                                       const float* pmly, const float* pmlz, const float* dux, const float* duy,
                                       const float* duz, const float* rho0)
{
  return density_impl<true>(ctx, rx, ry, rz, pmlx, pmly, pmlz, dux, duy, duz, rho0);
}
kw_status kw_compute_density_linear(kw_ctx* ctx, float* rx, float* ry, float* rz, const float* pmlx, const float* pmly,
                                    const float* pmlz, const float* dux, const float* duy, const float* duz,
                                    const float* rho0)
{
  return density_impl<false>(ctx, rx, ry, rz, pmlx, pmly, pmlz, dux, duy, duz, rho0);
}

kw_status kw_compute_pressure_terms_nonlinear(kw_ctx* ctx, float* densitySum, float* nonlinearTerm, float* velGradSum,
                                              const float* rx, const float* ry, const float* rz, const float* dux,
                                              const float* duy, const float* duz, const float* bona, const float* rho0)
{
  KW_CHECK_CONSTS(ctx);
  KW_PROF(ctx, "compute_pressure_terms_nonlinear");
  KW_REQUIRE(densitySum && nonlinearTerm && velGradSum && rx && ry && rz && dux && duy && duz);
  const kw_constants& c = ctx->c;
  const bool v4 = c.n_elements % 4 == 0 && all_aligned16(densitySum, nonlinearTerm, velGradSum, rx, ry, rz, dux, duy, duz, bona, rho0);
  const dim3 g  = v4 ? grid1d(c.n_elements / 4) : grid1d(c.n_elements);
#define PT_NL(V, B, R) LAUNCH((k_pressure_terms_nonlinear<V, B, R>), g, dim3(256), c, densitySum, nonlinearTerm, velGradSum, rx, ry, rz, dux, duy, duz, bona, rho0)
  if (v4)
  {
    if (bona) { if (rho0) PT_NL(4, false, false); else PT_NL(4, false, true); }
    else      { if (rho0) PT_NL(4, true, false);  else PT_NL(4, true, true); }
  }
  else
  {
    if (bona) { if (rho0) PT_NL(1, false, false); else PT_NL(1, false, true); }
    else      { if (rho0) PT_NL(1, true, false);  else PT_NL(1, true, true); }
  }
#undef PT_NL
  return KW_OK;
}

kw_status kw_compute_pressure_terms_linear(kw_ctx* ctx, float* densitySum, float* velGradSum, const float* rx,
                                           const float* ry, const float* rz, const float* dux, const float* duy,
                                           const float* duz, const float* rho0)
{
  KW_CHECK_CONSTS(ctx);
  KW_PROF(ctx, "compute_pressure_terms_linear");
  KW_REQUIRE(densitySum && velGradSum && rx && ry && rz && dux && duy && duz);
  const kw_constants& c = ctx->c;
  const bool v4 = c.n_elements % 4 == 0 && all_aligned16(densitySum, velGradSum, rx, ry, rz, dux, duy, duz, rho0);
  const dim3 g  = v4 ? grid1d(c.n_elements / 4) : grid1d(c.n_elements);
  if (v4)
  {
    if (rho0) LAUNCH((k_pressure_terms_linear<4, false>), g, dim3(256), c, densitySum, velGradSum, rx, ry, rz, dux, duy, duz, rho0);
    else      LAUNCH((k_pressure_terms_linear<4, true>), g, dim3(256), c, densitySum, velGradSum, rx, ry, rz, dux, duy, duz, rho0);
  }
  else
  {
    if (rho0) LAUNCH((k_pressure_terms_linear<1, false>), g, dim3(256), c, densitySum, velGradSum, rx, ry, rz, dux, duy, duz, rho0);
    else      LAUNCH((k_pressure_terms_linear<1, true>), g, dim3(256), c, densitySum, velGradSum, rx, ry, rz, dux, duy, duz, rho0);
  }
  return KW_OK;
}

kw_status kw_compute_absorbtion_term(kw_ctx* ctx, float* A, float* B, const float* n1, const float* n2)
{
  KW_CHECK_CONSTS(ctx);
  KW_PROF(ctx, "compute_absorbtion_term");
  KW_REQUIRE(A && B && n1 && n2);
  const kw_constants& c = ctx->c;
  const bool p2 = (c.n_elements_complex % 2 == 0) && all_aligned16(A, B) &&
                  ((reinterpret_cast<uintptr_t>(n1) & 7u) == 0) && ((reinterpret_cast<uintptr_t>(n2) & 7u) == 0);
  if (p2) LAUNCH((k_absorbtion_term<2>), grid1d(c.n_elements_complex / 2), dim3(256), c, A, B, n1, n2);
  else    LAUNCH((k_absorbtion_term<1>), grid1d(c.n_elements_complex), dim3(256), c, A, B, n1, n2);
  return KW_OK;
}

static kw_status sum_terms_impl(kw_ctx* ctx, float* p, const float* first, const float* tauTerm, const float* etaTerm,
                                const float* c2, const float* tau, const float* eta)
{
  KW_CHECK_CONSTS(ctx);
  KW_PROF(ctx, "sum_pressure_terms");
  KW_REQUIRE(p && first && tauTerm && etaTerm);
  KW_REQUIRE((tau == nullptr) == (eta == nullptr));
  const kw_constants& c = ctx->c;
  const bool v4 = c.n_elements % 4 == 0 && all_aligned16(p, first, tauTerm, etaTerm, c2, tau, eta);
  const dim3 g  = v4 ? grid1d(c.n_elements / 4) : grid1d(c.n_elements);
#define ST(V, C, T) LAUNCH((k_sum_pressure_terms<V, C, T>), g, dim3(256), c, p, first, tauTerm, etaTerm, c2, tau, eta)
  if (v4)
  {
    if (c2) { if (tau) ST(4, false, false); else ST(4, false, true); }
    else    { if (tau) ST(4, true, false);  else ST(4, true, true); }
  }
  else
  {
    if (c2) { if (tau) ST(1, false, false); else ST(1, false, true); }
    else    { if (tau) ST(1, true, false);  else ST(1, true, true); }
  }
#undef ST
  return KW_OK;
}

kw_status kw_sum_pressure_terms_nonlinear(kw_ctx* ctx, float* p, const float* nonlinearTerm, const float* tauTerm,
                                          const float* etaTerm, const float* c2, const float* tau, const float* eta)
{
  return sum_terms_impl(ctx, p, nonlinearTerm, tauTerm, etaTerm, c2, tau, eta);
}
kw_status kw_sum_pressure_terms_linear(kw_ctx* ctx, float* p, const float* tauTerm, const float* etaTerm,
                                       const float* densitySum, const float* c2, const float* tau, const float* eta)
{
  return sum_terms_impl(ctx, p, densitySum, tauTerm, etaTerm, c2, tau, eta);
}

kw_status kw_sum_pressure_nonlinear_lossless(kw_ctx* ctx, float* p, const float* rx, const float* ry, const float* rz,
                                             const float* c2, const float* bona, const float* rho0)
{
  KW_CHECK_CONSTS(ctx);
  KW_PROF(ctx, "sum_pressure_nonlinear_lossless");
  KW_REQUIRE(p && rx && ry && rz);
  const kw_constants& c = ctx->c;
  const bool v4 = c.n_elements % 4 == 0 && all_aligned16(p, rx, ry, rz, c2, bona, rho0);
  const dim3 g  = v4 ? grid1d(c.n_elements / 4) : grid1d(c.n_elements);
#define SNL(V, C, B, R) LAUNCH((k_sum_pressure_nonlinear_lossless<V, C, B, R>), g, dim3(256), c, p, rx, ry, rz, c2, bona, rho0)
#define SNL_V(V)                                                                                                       \
  do {                                                                                                                 \
    if (c2) { if (bona) { if (rho0) SNL(V, false, false, false); else SNL(V, false, false, true); }                    \
              else      { if (rho0) SNL(V, false, true, false);  else SNL(V, false, true, true); } }                   \
    else    { if (bona) { if (rho0) SNL(V, true, false, false);  else SNL(V, true, false, true); }                     \
              else      { if (rho0) SNL(V, true, true, false);   else SNL(V, true, true, true); } }                    \
  } while (0)
  if (v4) SNL_V(4); else SNL_V(1);
#undef SNL_V
#undef SNL
  return KW_OK;
}

kw_status kw_sum_pressure_linear_lossless(kw_ctx* ctx, float* p, const float* rx, const float* ry, const float* rz,
                                          const float* c2)
{
  KW_CHECK_CONSTS(ctx);
  KW_PROF(ctx, "sum_pressure_linear_lossless");
  KW_REQUIRE(p && rx && ry && rz);
  const kw_constants& c = ctx->c;
  const bool v4 = c.n_elements % 4 == 0 && all_aligned16(p, rx, ry, rz, c2);
  const dim3 g  = v4 ? grid1d(c.n_elements / 4) : grid1d(c.n_elements);
  if (v4)
  {
    if (c2) LAUNCH((k_sum_pressure_linear_lossless<4, false>), g, dim3(256), c, p, rx, ry, rz, c2);
    else    LAUNCH((k_sum_pressure_linear_lossless<4, true>), g, dim3(256), c, p, rx, ry, rz, c2);
  }
  else
  {
    if (c2) LAUNCH((k_sum_pressure_linear_lossless<1, false>), g, dim3(256), c, p, rx, ry, rz, c2);
    else    LAUNCH((k_sum_pressure_linear_lossless<1, true>), g, dim3(256), c, p, rx, ry, rz, c2);
  }
  return KW_OK;
}

kw_status kw_compute_velocity_shift(kw_ctx* ctx, int axis, float* spectrum, const float* shift)
{
  KW_CHECK_CONSTS(ctx);
  KW_PROF(ctx, "compute_velocity_shift");
  KW_REQUIRE(axis >= 0 && axis <= 2 && spectrum && shift);
  const kw_constants& c = ctx->c;
  const uint32_t ox = (axis == 0) ? c.nx / 2 + 1 : c.nx;
  const uint32_t oy = (axis == 1) ? c.ny / 2 + 1 : c.ny;
  const uint32_t oz = (axis == 2) ? c.nz / 2 + 1 : c.nz;
  KW_REQUIRE(oy <= 65535u && oz <= 65535u);
  const float divider = (axis == 0) ? c.fft_divider_x : (axis == 1) ? c.fft_divider_y : c.fft_divider_z;
  LAUNCH(k_velocity_shift, dim3((ox + 255) / 256, oy, oz), dim3(256), spectrum, (const float2*)shift, axis, ox, oy, oz,
         divider);
  return KW_OK;
}

} // extern "C"
