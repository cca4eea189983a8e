// kw_fused.hip — the MI355X-fused spectral pipeline: hand-written 3-D FFT passes with the element-wise physics
// folded into them.  This is the fast path of the per-step loop; it computes exactly the stages of
// KSpaceFirstOrderSolver::computeVelocity / computeVelocityGradient / computeDensity* / computePressure*
// (KSpaceSolver/KSpaceFirstOrderSolver.cpp:2087-2245) with the arithmetic of the SolverCudaKernels.cu lines cited at
// each epilogue, but organised around memory traffic instead of around one kernel per MATLAB statement:
//
//   x-forward   real rows -> half-spectrum rows (two real rows ride one complex FFT)          R  -> C
//   y-pass      in-place complex FFT along y on 16-column tiles (128-B segments)              C <-> C
//   z-fused     forward FFT along z, spectral multiply (kappa*ddk, nabla, sourceKappa), inverse FFT along z,
//               all in registers/LDS — the spectrum never makes a round trip to HBM            C  -> C (x1..3)
//   y-pass^-1   in-place inverse along y
//   x-inverse   half-spectrum rows -> real rows + epilogue: velocity update / density update (+ pressure terms) /
//               pressure sum — the gradients never exist in HBM as separate arrays
//
// vs. the reference order (14 library FFTs + 7 element-wise kernels, each a full HBM round trip).
// Internal spectral scratch is private to the pipeline: rows are whole 16-complex tiles (128-B segments) — for even Nx
// exactly Nx/2 bins, the x-Nyquist bin of every row living in a compact side array behind them (tile_coord); kappa /
// nabla operators are imported once into the per-thread run layout their z-pass reads (load_op_run).
// Supported: each of Nx, Ny, Nz one of the lengths of KW_FUSED_LENGTHS (2^a 3^b 5^c with a two-factor split into 4- to
// 32-point register DFTs, 16 ... 1024; axes independent); anything else uses the rocFFT path (kw_fft.hip +
// kw_solver_kernels.hip).  Multi-GPU: Z-slabs, see "pipelined slab schedule" below and kw_comm.hip.
#include "kw_fft_device.h"
#include "kw_internal.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace kwfft;

// The x-inverse + epilogue kernels are the bulk of this file's compile time and code size (one kernel per epilogue
// variant and line length).  They are compiled in eight extra passes over this file, side by side with the main pass
// (build.py):
//   KW_FUSED_TU == 0  everything except those kernels;  1 / 7  the chained density epilogues of the lines below / from
//                  KW_LONG_LINES elements;  5 / 8  the density epilogues that store their terms, likewise;  2  the other
//                  epilogues;  3 / 6 / 4  the masked (TAIL) forms of 1 + 7 / 5 + 8 / 2, which only grids with a partial
//                  last x tile ever launch.
// Every pass is a code object of its own that the runtime loads at the first launch out of it: a run pays for the
// variants it uses.  The main pass reaches the others through the functions below (XinvArgs passed as an opaque pointer:
// the struct lives in this file's anonymous namespace).
#ifndef KW_FUSED_TU
#define KW_FUSED_TU 0
#endif
kw_status kw_fused_xinv_density(int chain, int terms, kw_ctx* ctx, int ncomp, const void* xinv_args, uint32_t tile0, uint32_t ntiles);
kw_status kw_fused_xinv_other(int epi, int chain, kw_ctx* ctx, int ncomp, const void* xinv_args, uint32_t tile0, uint32_t ntiles);
kw_status kw_fused_xinv_density_tail(int chain, int terms, kw_ctx* ctx, int ncomp, const void* xinv_args, uint32_t tile0, uint32_t ntiles);
kw_status kw_fused_xinv_density_plain(int chain, int terms, kw_ctx* ctx, int ncomp, const void* xinv_args, uint32_t tile0, uint32_t ntiles);
kw_status kw_fused_xinv_density_tail_plain(int chain, int terms, kw_ctx* ctx, int ncomp, const void* xinv_args, uint32_t tile0, uint32_t ntiles);
kw_status kw_fused_xinv_density_long(int chain, int terms, kw_ctx* ctx, int ncomp, const void* xinv_args, uint32_t tile0, uint32_t ntiles);
kw_status kw_fused_xinv_density_plain_long(int chain, int terms, kw_ctx* ctx, int ncomp, const void* xinv_args, uint32_t tile0, uint32_t ntiles);
#define KW_LONG_LINES 432 /* passes 7 / 8 hold the density epilogues of the x lines from this length on */
kw_status kw_fused_xinv_other_tail(int epi, int chain, kw_ctx* ctx, int ncomp, const void* xinv_args, uint32_t tile0, uint32_t ntiles);
// whole-plane tiles (small grids, see k_xinv): tile0 / ntiles count z-planes; in passes 1 / 2
kw_status kw_fused_xinv_density_plane(int chain, int terms, kw_ctx* ctx, int ncomp, const void* xinv_args, uint32_t tile0, uint32_t ntiles);
kw_status kw_fused_xinv_other_plane(int epi, int chain, kw_ctx* ctx, int ncomp, const void* xinv_args, uint32_t tile0, uint32_t ntiles);

namespace {

constexpr int NLMAX = 16; // widest tile: 16 complex = 128-B segments (also the row-pitch granule)
// x passes, rows per block = 2 * nl_x.  Chosen per line length from a measured matrix (every length from 160 up, both
// orientations of its factor pair, 8 / 10 / 12 / 16 line pairs; 4 / 8 / 16 from 96 to 144: profiles/r03_length_tuning.txt).
// What the matrix shows: a block should be a whole number of 4 waves or less — the 5- and 6-wave blocks of 18, 20 and 24
// threads per line at 16 line pairs are rarely co-resident (one wave more on one SIMD than on the others), 240^3: x-inverse
// kernels 1.4-1.5x faster at 12 line pairs — and the long lines want the small block's register budget.
// (tuning builds: -DKW_TUNE_NLX=<n> overrides the table, -DKW_TUNE_SWAP the orientation table in kw_fft_device.h)
#ifdef KW_TUNE_NLX
constexpr int nl_x(int) { return KW_TUNE_NLX; }
#else
constexpr int nl_x(int L)
{
  switch (L)
  {
    case 100: case 108: case 140: case 160: case 168: case 196: case 200: case 224: case 288: case 336: case 384: case 392: return 8;
    case 280: case 300: case 320: case 432: case 480: case 500: case 600: return 10;
    case 180: case 216: case 240: case 252: case 324: case 360: case 400: return 12;
    case 896: return 16;
    default: return L >= 400 ? 8 : 16;
  }
}
#endif
// Every fast-path length is a multiple of 4, so the Ny * Nz rows of a 3-D grid on one GPU are a multiple of 16 and x tiles of
// 8 line pairs (16 rows) always divide it: only 2-D grids and slabs with Ny * Nz(local) off a multiple of 16 ever need the
// masked (TAIL) kernels of such a length.  For the long lines added in round 3 — the most expensive kernels of the file to
// compile — those are not built; kw_fused_supported sends such a grid to the rocFFT path.
constexpr bool has_partial_x_tiles(int L)
{
  switch (L)
  {
    case 672: case 700: case 720: case 756: case 784: case 800: case 840: case 864: case 900: case 960: return false;
    default: return true;
  }
}
#define KW_NO_TAIL(LEN)                                                                                                \
  { kw_set_error("fused pipeline: rows of %d elements have no partial x tiles", LEN); return KW_ERR_INVALID; }

// y / z passes, columns per tile: 16 (128-B row segments) at every length.  500^3 measured with 8-column tiles (64-B
// segments) for the long lines: y passes 1.5x, z-fused 1.25x slower; 240^3 with 12- and 8-column tiles (4- and 3-wave
// blocks instead of 5): step 11 % and 15 % slower.
constexpr int nl_yz(int) { return 16; }
// ... except the z-fused kernels of 240- and 400-point lines (20 threads per line: 5-wave blocks, of which a CU mostly
// holds one): 32-column tiles — one 10-wave block, 256-B row segments — take 16 % / 10 % off them (the y-passes of the
// same lengths lose with 32 columns, every other length loses in both: profiles/r03_length_tuning.txt)
#ifdef KW_TUNE_NLZ
constexpr int nl_z(int) { return KW_TUNE_NLZ; }
#else
constexpr int nl_z(int L) { return (L == 240 || L == 400) ? 32 : 16; }
#endif
// tiles of nl_z columns over the P (a multiple of 16) columns of a row
constexpr uint32_t z_tiles(uint32_t P, uint32_t len) { return (P + static_cast<uint32_t>(nl_z(len)) - 1u) / static_cast<uint32_t>(nl_z(len)); }

constexpr int cmax(int a, int b) { return a > b ? a : b; }
// 20- to 32-point register DFTs next to a multi-array epilogue want more than 256 VGPRs, which leaves ONE wave per SIMD
// (nothing to hide a memory round trip behind): hold those kernels to the two-wave budget instead
constexpr int big_line_waves(int L) { return L >= 400 ? 2 : 1; }
// largest divisor of n that is <= want
constexpr int gq_pick(int n, int want) { return (n % want == 0) ? want : gq_pick(n, want - 1); }

template<int L, int NLV = nl_yz(L)> struct Geo
{
  static constexpr int R1 = Fac<L>::R1, R2 = Fac<L>::R2;
  static constexpr int NL      = NLV;
  static constexpr int TPL     = cmax(R1, R2);
  static constexpr int THREADS = NL * TPL;
  // y/z passes: LDS[k1][n2][c]; +16 complex per k1 block keeps the step-B reads conflict-free
  static constexpr int PADB    = NL;
  static constexpr int SF      = R2 * NL + PADB; // stride per k1 (forward and inverse use the same cells, see inverse_from_regs)
  static constexpr int LDSB    = R1 * SF;
  // x passes: LDS[c][k1][n2] (pitch LP per line) aliased with a natural-order line buffer (pitch ZP)
  static constexpr int LP0  = R1 * (R2 + 1);
  static constexpr int LP   = LP0 + ((16 - LP0 % 32) + 32) % 32;
  static constexpr int ZP   = L + 16;
  static constexpr int LDSX = NL * cmax(LP, ZP);
  // inter-step twiddles, LDS-resident: T[k][n] = W_L^(k*n), k < R1, n < R2, pitch TP (odd: conflict-free both ways)
  static constexpr int TP   = R2 + 1;
  static constexpr int TWN  = R1 * TP;
};

// "thread t takes part in a step of R threads per line": compile-time true when every thread of a line does (R == TPL),
// so that the common L = 256 kernels carry no divergent regions (and no phi-separated register sets) at all.
template<int L> using GeoX = Geo<L, nl_x(L)>; // geometry of the x passes

#define ACT(R, t) ((R) == G::TPL || (t) < (R))
// the same for the y / z kernels, whose threads are laid out t = j * NL + c: when NL * R is a whole number of waves the
// participants of a step are whole waves, and the test is made on a wave-uniform value (a scalar branch, no exec masking)
#define ACTW(R, j) ((R) == G::TPL || (((G::NL * (R)) % 64 == 0 && 64 % G::NL == 0) ? __builtin_amdgcn_readfirstlane(j) < (R) : (j) < (R)))

// Roles of the x kernels' threads.  A step with R participants per line (R2 in step A, R1 in step B) is taken by the FIRST
// NL * R threads of the block, R consecutive ones per line: with R < TPL (12 x 20, 12 x 18 ... factorisations) the idle
// threads are whole trailing waves instead of masked lanes inside every wave.  Everything crosses LDS between the steps,
// so the line a thread works on may change from step to step; with R == TPL this is the plain (c, f) = (t / TPL, t % TPL).
template<class G, int R> struct XRole
{
  int  c, f;
  bool on;
  __device__ __forceinline__ XRole()
  {
    c  = static_cast<int>(threadIdx.x) / R;
    f  = static_cast<int>(threadIdx.x) - c * R;
    on = (R == G::TPL) || static_cast<int>(threadIdx.x) < G::NL * R;
  }
};

// Block barrier that orders LDS traffic only.  __syncthreads() also drains every outstanding global load and store
// (vmcnt(0)) — here all cross-thread communication goes through LDS, and global stores / prefetched loads should stay in
// flight across the exchange steps.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// State and medium arrays are touched once per step and not again for a whole step (> 5 GB of traffic later): they are
// streamed past the caches (non-temporal), which leaves the 256 MB Infinity Cache to the spectral scratch that the very
// next kernel re-reads.  Measured: +5.4 % on the whole step against plain accesses; the same for the reduced operators
// (+2.6 %).  Spectra stay on plain accesses: marking their last-use reads non-temporal costs 0.6-2 % in every pass type.
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4(const float* p)
{
  const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st4(float* p, const float4& v)
{
  v4f t = { v.x, v.y, v.z, v.w };
  __builtin_nontemporal_store(t, reinterpret_cast<v4f*>(p));
}
__device__ __forceinline__ float ldop(const float* p) { return __builtin_nontemporal_load(p); }
#define LDOP(p) ldop(p) // reduced real operators (kappa, nabla): one or two reads per step each (+2.6 % measured)

// Reduced operators are stored for the z-pass that consumes them: [ky][kx tile of 16][q][j][c][V] — thread (c, j) of the
// block owning that tile finds the values of ITS bins (kz = j + R1*k2, k2 = q*V + r; 2 x 256 split lines: kz = 2j + h +
// 2*R1*k2 with run index 2*k2 + h) as RUN / V vectors of V floats, and the 64 lanes of a wave read one contiguous
// 64 * V * 4-byte piece per load instruction.  V = 4 when the per-thread run length allows, else 2 or 1 (V = 1 is
// plain [kz][c] order).
constexpr int op_vec(int run) { return (run % 4 == 0) ? 4 : (run % 2 == 0) ? 2 : 1; }
template<int RUN, int R1> __device__ __forceinline__ void load_op_run(float (&dst)[RUN], const float* __restrict__ op, uint32_t base)
{ // base: index (in floats) of vector q = 0 of this thread; consecutive q are R1 * 16 vectors apart
  constexpr int V = op_vec(RUN);
  typedef float vf __attribute__((ext_vector_type(V)));
#pragma unroll
  for (int q = 0; q < RUN / V; q++)
  {
    if constexpr (V == 1) dst[q] = LDOP(op + base + static_cast<uint32_t>(q * R1 * NLMAX));
    else
    {
      const vf t = __builtin_nontemporal_load(reinterpret_cast<const vf*>(op + base + static_cast<uint32_t>(q * R1 * NLMAX * V)));
#pragma unroll
      for (int r = 0; r < V; r++) dst[q * V + r] = t[r];
    }
  }
}


// Element e of a lane plus a wave-uniform element offset: "SGPR base + 32-bit lane offset" addressing.  The uniform part
// (array base + n * line stride) is 64-bit scalar arithmetic, the lane part one VGPR holding a byte offset — written
// this way the loads / stores of a tile cost no vector ALU work at all (as  p[lane + n * stride]  every access pays a
// v_add_u32 and a 64-bit v_lshl_add_u64: a sixth of the vector instructions of a z-pass, whose VALU is ~75 % busy).
// (the empty asm pins the uniform part to an SGPR pair: left alone, the compiler re-associates it into a chain of 64-bit
// vector adds)
// PIN = false: inside a region only some lanes of a wave enter (a step whose participants are not whole waves) the
// compiler may hold the base in vector registers, which the "s" constraint cannot take: there the base is read back
// from the first active lane instead (two v_readfirstlane; folded away when the value already is scalar).
template<bool PIN = true>
__device__ __forceinline__ float2 ld_uni(const float2* __restrict__ p, uint64_t uniform_elems, uint32_t lane_bytes)
{
  typedef float v2 __attribute__((ext_vector_type(2)));
  typedef const __attribute__((address_space(1))) char* gptr; // global address space survives the asm (else: flat loads)
  gptr b = (gptr)(p + uniform_elems);
  if constexpr (PIN) asm volatile("" : "+s"(b));
  else
  {
    const uint64_t bits = reinterpret_cast<uint64_t>(p + uniform_elems);
    const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(bits));
    const uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(bits >> 32));
    b = (gptr)((static_cast<uint64_t>(hi) << 32) | lo);
  }
  const v2 t = *(const __attribute__((address_space(1))) v2*)(b + lane_bytes);
  return make_float2(t.x, t.y);
}
template<bool PIN = true>
__device__ __forceinline__ void st_uni(float2* __restrict__ p, uint64_t uniform_elems, uint32_t lane_bytes, const float2& v)
{
  typedef float v2 __attribute__((ext_vector_type(2)));
  typedef __attribute__((address_space(1))) char* gptr;
  gptr b = (gptr)(p + uniform_elems);
  if constexpr (PIN) asm volatile("" : "+s"(b));
  else
  {
    const uint64_t bits = reinterpret_cast<uint64_t>(p + uniform_elems);
    const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(bits));
    const uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(bits >> 32));
    b = (gptr)((static_cast<uint64_t>(hi) << 32) | lo);
  }
  *(__attribute__((address_space(1))) v2*)(b + lane_bytes) = v2{ v.x, v.y };
}

// Tile coordinates of a block of a y- or z-pass.  Regular tiles: NL consecutive columns kx at one line position (the
// plane z of a y-pass, the row ky of a z-pass; blockIdx.y).
//
// Side tile.  A half-spectrum row has nxc = Nx/2 + 1 bins: for every Nx that is a multiple of 32 that is a whole number
// of 16-column tiles plus ONE bin, the x-Nyquist column.  Kept in the rows it costs a ninth tile of 16 lanes with one
// useful lane at 256^3 (11 % of every y- and z-pass; 20 % at 128^3).  The pipeline therefore stores that column apart,
// as a compact array N[line position][y] behind the main part (rows of Nx/2 bins, no padding at all), and the passes
// handle it with one extra tile index whose blocks take the column at NL line positions each: lane c <-> position
// NL * blockIdx.y + c.  Same kernel, same LDS traffic; the lanes of a side block read 8-B neighbours of one array.
// (Tried before, with the column left in the padded rows: every wave-level access then touches 64 different cache
// lines and those blocks become the critical path — y-passes +12 %, step -4 %.)
struct TileCoord
{
  uint32_t kx;    // true column index (operator lookups along kx)
  uint32_t col;   // column used for addressing: min(kx, last main column), 0 in a side block
  uint32_t pos;   // line position: z (y-pass) / ky (z-pass)
  uint32_t pitch; // row pitch of the region this block works in: P, or 1 (side array)
  uint32_t off;   // element offset of that region in the scratch array
  bool     valid, dead, side;
};
template<int NL>
__device__ __forceinline__ TileCoord tile_coord(uint32_t nxc, uint32_t P, uint32_t side_off, uint32_t npos, int c)
{
  TileCoord t;
  if (side_off != 0 && blockIdx.x == gridDim.x - 1)
  {
    const uint32_t p = blockIdx.y * NL + c;
    t.dead  = blockIdx.y * NL >= npos;
    t.kx    = nxc; // the bin after the nxc main columns
    t.col   = 0;
    t.valid = p < npos;
    t.pos   = min(p, npos - 1u); // lanes past the last position re-read it; their results are never stored
    t.pitch = 1;
    t.off   = side_off;
    t.side  = true;
  }
  else
  {
    t.kx    = blockIdx.x * NL + c;
    t.col   = min(t.kx, nxc - 1u); // pad lanes re-read the last column (no branch, no extra sector); never stored
    t.valid = t.kx < nxc;
    t.pos   = blockIdx.y;
    t.pitch = P;
    t.off   = 0;
    t.dead  = false;
    t.side  = false;
  }
  return t;
}

// ---- register-level steps -------------------------------------------------------------------------------------------
// Fill the block's twiddle table from the global one (tw[m] = exp(-2*pi*i*m/L)); the caller's next lds_barrier()
// publishes it.  Every use below is "one VGPR base + compile-time offset", so twiddles cost neither address registers
// nor global-memory requests.
template<int L> __device__ __forceinline__ void load_twiddles(float2* twl, const float2* __restrict__ tw)
{
  using G = Geo<L>;
  for (int e = threadIdx.x; e < G::R1 * G::R2; e += blockDim.x)
  {
    const int k = e / G::R2, n = e - k * G::R2;
    twl[k * G::TP + n] = tw[k * n];
  }
}

template<int L, int DIR> __device__ __forceinline__ void step_a(float2 (&v)[Fac<L>::R1], int n2, const float2* twl)
{
  Dft<Fac<L>::R1, DIR>::run(v);
#pragma unroll
  for (int k1 = 1; k1 < Fac<L>::R1; k1++) v[k1] = apply_tw<DIR>(v[k1], twl[k1 * Geo<L>::TP + n2]);
}

// =====================================================================================================================
// y-pass: in-place complex FFT along y (stride P) for tile (z = blockIdx.y, kx tile = blockIdx.x), array blockIdx.z
// =====================================================================================================================
// Row addressing of a y-line element.  Natural layout: row(z, ky) = z*ny + ky.  Packed layout (slab mode, the send /
// receive side of the all-to-all): row(z, ky) = ((ky / nyl) * nzl + z) * nyl + ky % nyl, i.e. one contiguous chunk
// [nzl][nyl][P] per peer rank.  Both are  q * qstride + (ky - q * nyl) + z * zmul  with q = ky / nyl taken as
// (ky * magic) >> 20, magic = 2^20 / nyl + 1: exact while ky * nyl < 2^20 (lines have at most 1024 elements); the natural
// layout has magic = 0, hence q = 0.
struct RowAddr
{
  uint32_t magic, nyl, qstride, zmul;
  uint32_t estride; // 1 for y-lines; ny for lines along z (probe / plain z transform)
  // 32-bit element indices throughout the pipeline: every scratch / field array has < 2^32 elements (checked in
  // kw_fused_supported), so addresses are "uniform base + 32-bit lane offset" and need no 64-bit VALU arithmetic
  __device__ __forceinline__ uint32_t row(uint32_t z, uint32_t ky) const
  {
    const uint32_t q = (ky * magic) >> 20;
    return q * qstride + (ky - q * nyl) * estride + z * zmul;
  }
};

struct PassArgs
{
  const float2* in[3];
  float2*       out[3];
  const float2* tw;
  uint32_t      nxc, P;
  uint32_t      PX;   // row pitch of the packed (exchange) side (= P unless rows travel without their padding)
  uint32_t      narr; // arrays per block (grid.z * narr arrays in the launch)
  uint32_t      z0;   // first plane of this launch (chunked plane-local passes)
  uint32_t      side_off; // element offset of the x-Nyquist side array, 0 = the column lives in the rows (see tile_coord)
  const float2* mul[3]; // per array: optional factor mul[ky] applied to the line before its transform (ddy of the gradient)
  RowAddr       ain, aout;
};

// PIN / POUT: packed (per-peer chunk) addressing on the input / output side; the natural layout is the cheap
// compile-time-strided case  element(k) = base + k * P.
template<int L, int DIR, bool PIN, bool POUT>
__global__ __launch_bounds__(Geo<L>::THREADS) void k_ypass(PassArgs a)
{
  using G = Geo<L>;
  constexpr int R1 = G::R1, R2 = G::R2;
  __shared__ float2 lds[G::LDSB];
  __shared__ float2 twl[G::TWN];
  load_twiddles<L>(twl, a.tw);
  const int      c     = threadIdx.x % G::NL;
  const int      j     = threadIdx.x / G::NL;
  const TileCoord tc   = tile_coord<G::NL>(a.nxc, a.P, a.side_off, gridDim.y, c);
  if (tc.dead) return;
  const uint32_t kx    = tc.side ? 0u : tc.kx; // addressing column (side blocks: the one column of the side array)
  const bool     valid = tc.valid;
  const uint32_t kxl   = tc.col;
  const uint32_t z     = tc.pos + a.z0;
  const uint32_t Pe    = tc.pitch, soff = tc.off; // row pitch / offset of the region (main rows, or the side array)
  const uint32_t PXe   = tc.side ? 1u : a.PX;     // the same on the packed (exchange) side: per-peer chunks of the side array
  const uint32_t arr0  = blockIdx.z * a.narr; // each block takes a.narr arrays back to back (next one's lines prefetched)

  auto load_lines = [&](float2 (&v)[R1], const float2* __restrict__ Sin) {
    if (PIN)
    {
#pragma unroll
      for (int n1 = 0; n1 < R1; n1++) v[n1] = Sin[a.ain.row(z, n1 * R2 + j) * PXe + kxl + soff];
    }
    else
    {
      uint32_t b = (z * a.ain.zmul + j * a.ain.estride) * Pe + kxl + soff; // estride = 1 (y lines) or ny (z probe)
      asm volatile("" : "+v"(b));
      const uint32_t step = R2 * a.ain.estride * Pe;
#pragma unroll
      for (int n1 = 0; n1 < R1; n1++)
        v[n1] = ld_uni(Sin, static_cast<uint64_t>(n1) * step, b * static_cast<uint32_t>(sizeof(float2)));
    }
  };

  float2 v[R1];
  if (ACTW(R2, j)) load_lines(v, a.in[arr0]);
  lds_barrier(); // twiddle table visible (the line loads stay in flight across it)
#pragma unroll 1
  for (uint32_t ia = 0; ia < a.narr; ia++)
  {
    if (ACTW(R2, j))
    {
      const float2* __restrict__ m = a.mul[arr0 + ia];
      if (m != nullptr)
      {
#pragma unroll
        for (int n1 = 0; n1 < R1; n1++) v[n1] = cmulf(v[n1], m[n1 * R2 + j]);
      }
      step_a<L, DIR>(v, j, twl);
#pragma unroll
      for (int k1 = 0; k1 < R1; k1++) lds[k1 * G::SF + j * G::NL + c] = v[k1];
      if (ia + 1 < a.narr) load_lines(v, a.in[arr0 + ia + 1]);
    }
    lds_barrier();
    if (ACTW(R1, j))
    {
      float2 w[R2];
#pragma unroll
      for (int n2 = 0; n2 < R2; n2++) w[n2] = lds[j * G::SF + n2 * G::NL + c];
      Dft<R2, DIR>::run(w);
      if (valid)
      {
        float2* __restrict__ Sout = a.out[arr0 + ia];
        if (POUT)
        {
#pragma unroll
          for (int k2 = 0; k2 < R2; k2++) Sout[a.aout.row(z, j + R1 * k2) * PXe + kx + soff] = w[k2];
        }
        else
        {
          uint32_t b = (z * a.aout.zmul + j * a.aout.estride) * Pe + kx + soff;
          asm volatile("" : "+v"(b));
          const uint32_t step = R1 * a.aout.estride * Pe;
#pragma unroll
          for (int k2 = 0; k2 < R2; k2++) st_uni(Sout, static_cast<uint64_t>(k2) * step, b * static_cast<uint32_t>(sizeof(float2)), w[k2]);
        }
      }
    }
    if (ia + 1 < a.narr) lds_barrier(); // exchange buffer reused by the next array
  }
}

// =====================================================================================================================
// z-fused: forward along z, spectral multiply, inverse along z.  Tile (y = blockIdx.y, kx tile = blockIdx.x).
// =====================================================================================================================
// Z_SHIFT: plain "transform along a line, multiply bin k by a complex H[k], transform back" on lines with arbitrary
// strides — the half-cell shift of a staggered velocity (computeVelocityShiftInY / InZ) done on pairs of real
// x-neighbours packed as one complex value
enum ZMode { Z_PGRAD = 0, Z_VGRAD = 1, Z_ABSORB = 2, Z_SOURCE = 3, Z_SHIFT = 4 };

struct ZArgs
{
  const float2* in[3];
  float2*       out[3];
  const float*  op[2];  // padded reduced real operators: kappa (PGRAD/VGRAD), nabla1/nabla2 (ABSORB), sourceKappa
  const float2* dd[3];  // ddx[kx], ddy[ky], ddz[kz]
  const float2* tw;
  float         divider;
  uint32_t      nxc, P, ny, nz;
  uint32_t      Pop;  // row pitch of the operator arrays (always padded; P is the unpadded exchange pitch in slab mode)
  uint32_t      arr0; // first array of this launch
  uint32_t      narr; // arrays processed back to back by each block (VGRAD / ABSORB)
  uint32_t      ky0;  // global ky of local row 0 (slab mode: rank * ny/nranks); ny above = number of LOCAL rows
  uint32_t      lstride, bstride; // Z_SHIFT: element stride along a line, offset per blockIdx.y (both in complex units)
  uint32_t      side_off;    // element offset of the x-Nyquist side array in the scratch arrays (0: none; see tile_coord)
  uint32_t      op_side_off; // float offset of the side column's values in the imported operators
  uint32_t      axis_of[3];  // VGRAD: derivative axis of array i (0 ddx[kx], 1 ddy[ky], 2 along the line: dd[2][k]); 3-D: 0 1 2
};

// inverse along the line, started from the step-B register layout (thread (c,k1) holds X[k1 + R1*k2]); result:
// thread (c,q1) holds x[q1 + R2*q2], q2 < R1 — returned in v.
// The exchange runs the forward one backwards over the same cells: in the forward transform thread n2 writes the cells
// (k1, n2) for all k1 ("its column") and thread k1 reads (k1, n2) for all n2 ("its row"); here thread k1 writes its row
// — cells nobody else has read — and thread q1 reads its column.  A thread therefore only ever overwrites what it read
// last itself: no barrier is needed between the forward read and this write, nor between this read and the next
// forward write.
// COLWRITE (square factorisations only, used by the 2 x 256 split kernels): the transposed cell assignment — thread k1
// writes its column and thread q1 reads its row — for an exchange that follows one whose read was by columns.
template<int L, bool COLWRITE = false, int NLV = nl_yz(L)>
__device__ __forceinline__ void inverse_from_regs(float2 (&w)[Fac<L>::R2], float2 (&v)[Fac<L>::R1], float2* lds, int c,
                                                  int j, const float2* twl)
{
  using G = Geo<L, NLV>;
  constexpr int R1 = G::R1, R2 = G::R2;
  static_assert(!COLWRITE || R1 == R2, "transposed exchange needs a square factorisation");
  if (ACTW(R1, j))
  {
    Dft<R2, kInv>::run(w);
#pragma unroll
    for (int q1 = 0; q1 < R2; q1++)
    {
      const float2 t = (q1 == 0) ? w[0] : apply_tw<kInv>(w[q1], twl[j * G::TP + q1]);
      if (COLWRITE) lds[q1 * G::SF + j * G::NL + c] = t;
      else lds[j * G::SF + q1 * G::NL + c] = t;
    }
  }
  lds_barrier();
  if (ACTW(R2, j))
  {
#pragma unroll
    for (int k1 = 0; k1 < R1; k1++)
      v[k1] = COLWRITE ? lds[j * G::SF + k1 * G::NL + c] : lds[k1 * G::SF + j * G::NL + c];
    Dft<R1, kInv>::run(v);
  }
}

// One block owns the z-lines of tile (ky = blockIdx.y, kx tile = blockIdx.x) of `narr` arrays (VGRAD: the three velocity
// spectra, ABSORB: the two pressure terms), processed back to back: the lines of array i+1 are in flight while array i
// is transformed, and kappa is fetched once for all three velocity components.  PGRAD has one input and three outputs.
template<int L, int MODE> __global__ __launch_bounds__((Geo<L, nl_z(L)>::THREADS), big_line_waves(L)) void k_zfused(ZArgs a)
{
  using G = Geo<L, nl_z(L)>;
  constexpr int R1 = G::R1, R2 = G::R2;
  constexpr bool WA = (R2 == G::TPL) || ((G::NL * R2) % 64 == 0 && 64 % G::NL == 0); // step-A participants: whole waves
  __shared__ float2 lds[G::LDSB];
  __shared__ float2 twl[G::TWN];
  load_twiddles<L>(twl, a.tw);
  const int      c      = threadIdx.x % G::NL;
  const int      j      = threadIdx.x / G::NL;
  const TileCoord tc    = tile_coord<G::NL>(a.nxc, a.P, (MODE == Z_SHIFT) ? 0u : a.side_off, gridDim.y, c);
  if (tc.dead) return;
  const uint32_t ky     = tc.pos;
  const bool     valid  = tc.valid;
  const uint32_t zstr   = (MODE == Z_SHIFT) ? a.lstride : a.ny * tc.pitch;
  const uint32_t bstr   = (MODE == Z_SHIFT) ? a.bstride : tc.pitch;
  const uint32_t kxl    = tc.kx < a.nxc ? tc.kx : (tc.side ? a.nxc : a.nxc - 1u); // true column (ddx lookups)
  const uint32_t base   = ky * bstr + (tc.side ? 0u : tc.kx) + tc.off;
  const uint32_t basel  = ky * bstr + tc.col + tc.off;
  // operators live in a tile-blocked layout [ky][kx tile][kz][16]: the 16 x nz values of this block's tile are one
  // contiguous run (k_import_reduced), instead of 64-B pieces a whole plane apart
  constexpr int  OPV    = op_vec(R2);
  // side blocks: the side column's operator values are stored like one more row of tiles whose columns are the ky
  const uint32_t opbase = tc.side ? a.op_side_off + (((ky / NLMAX) * (R2 / OPV) * R1 + j) * NLMAX + ky % NLMAX) * OPV
                                  : (((ky * (a.Pop / NLMAX) + tc.col / NLMAX) * (R2 / OPV) * R1 + j) * NLMAX + tc.col % NLMAX) * OPV;
  constexpr bool MULTI  = (MODE == Z_VGRAD || MODE == Z_ABSORB);
  const uint32_t arr0   = MULTI ? a.arr0 + blockIdx.z * a.narr : 0; // (grid.z > 1: small grids, one array per block)
  const uint32_t narr   = MULTI ? a.narr : 1;
  // PGRAD: the x- and y-gradients share one z-inverse — ddx(kx), ddy(ky) do not depend on kz, so they are applied after
  // the way back (y-pass / x-pass); only Q = F_z^-1{kappa X} and G_z = F_z^-1{ddz kappa X} leave this kernel
  constexpr int  NOUT   = (MODE == Z_PGRAD) ? 2 : 1;

  float2 v[R1];
  const float2* __restrict__ in0 = a.in[arr0]; // (array pointers are picked outside the thread-dependent regions: SGPR bases)
  if (ACTW(R2, j))
  {
    const float2* __restrict__ in = in0;
    const uint32_t lb = (basel + static_cast<uint32_t>(j) * zstr) * static_cast<uint32_t>(sizeof(float2));
#pragma unroll
    for (int n1 = 0; n1 < R1; n1++)
      v[n1] = ld_uni<WA>(in, static_cast<uint64_t>(n1 * R2) * zstr, lb);
  }
  // spectral operator of the elements this thread will hold after the forward transform (kz = j + R1*k2)
  float kap[R2];
  if (MODE != Z_SHIFT && ACTW(R1, j))
  {
    load_op_run<R2, R1>(kap, a.op[(MODE == Z_ABSORB) ? arr0 : 0], opbase);
  }
  lds_barrier(); // twiddle table visible (the loads above stay in flight across it)

#pragma unroll 1
  for (uint32_t ia = 0; ia < narr; ia++)
  {
    const uint32_t arr = arr0 + ia;
    const float2* __restrict__ in_next = a.in[min(arr + 1u, arr0 + narr - 1u)];
    if (ACTW(R2, j))
    {
      step_a<L, kFwd>(v, j, twl);
#pragma unroll
      for (int k1 = 0; k1 < R1; k1++) lds[k1 * G::SF + j * G::NL + c] = v[k1];
    }
    // next array's lines: in flight during this array's two transforms
    if (MULTI && ia + 1 < narr && ACTW(R2, j))
    {
      const float2* __restrict__ in = in_next;
      uint32_t lb = basel + static_cast<uint32_t>(j) * zstr;
      asm volatile("" : "+v"(lb)); // per-iteration address arithmetic instead of 16 loop-invariant address registers
#pragma unroll
      for (int n1 = 0; n1 < R1; n1++)
        v[n1] = ld_uni<WA>(in, static_cast<uint64_t>(n1 * R2) * zstr, lb * static_cast<uint32_t>(sizeof(float2)));
    }
    lds_barrier();
    float2 X[R2];
    if (ACTW(R1, j))
    {
#pragma unroll
      for (int n2 = 0; n2 < R2; n2++) X[n2] = lds[j * G::SF + n2 * G::NL + c];
      Dft<R2, kFwd>::run(X);
      //   Z_PGRAD  SolverCudaKernels.cu:1149-1155  e = X*kappa;            out_d = e (x) dd_d_pos
      //   Z_VGRAD  :1220-1236                      e = X*(kappa*divider);  out   = e (x) dd_neg of this array's own axis
      //   Z_ABSORB :1817-1818                      out = X*nabla
      //   Z_SOURCE :742-744                        out = X*(sourceKappa*divider)
      if (MODE == Z_SHIFT)
      { // SolverCudaKernels.cu:2617-2710: bin k times shift[k] / N (both folded into H by the host)
#pragma unroll
        for (int k2 = 0; k2 < R2; k2++) X[k2] = cmulf(X[k2], a.dd[2][j + R1 * k2]);
      }
      else
      {
#pragma unroll
        for (int k2 = 0; k2 < R2; k2++)
        {
          float sc = kap[k2];
          if (MODE == Z_VGRAD || MODE == Z_SOURCE) sc *= a.divider;
          X[k2] = make_float2(X[k2].x * sc, X[k2].y * sc);
        }
      }
      if (MODE == Z_ABSORB && ia + 1 < narr)
      {
        uint32_t lb = opbase;
        asm volatile("" : "+v"(lb));
        load_op_run<R2, R1>(kap, a.op[arr + 1], lb);
      }
    }

#pragma unroll 1
    for (int o = 0; o < NOUT; o++)
    {
      float2 w[R2];
      if (ACTW(R1, j))
      {
        if (MODE == Z_PGRAD || MODE == Z_VGRAD)
        {
          const uint32_t axis = (MODE == Z_PGRAD) ? (o == 0 ? 3u : 2u) : a.axis_of[arr];
          if (axis == 3)
          {
#pragma unroll
            for (int k2 = 0; k2 < R2; k2++) w[k2] = X[k2];
          }
          else if (axis == 2)
          {
            uint32_t jz = j;
            asm volatile("" : "+v"(jz)); // keeps the 16 ddz loads inside this pass instead of hoisted registers
#pragma unroll
            for (int k2 = 0; k2 < R2; k2++) w[k2] = cmulf(X[k2], a.dd[2][jz + R1 * k2]);
          }
          else
          {
            const float2 dxy = (axis == 0) ? a.dd[0][kxl] : a.dd[1][ky + a.ky0];
#pragma unroll
            for (int k2 = 0; k2 < R2; k2++) w[k2] = cmulf(X[k2], dxy);
          }
        }
        else
        {
#pragma unroll
          for (int k2 = 0; k2 < R2; k2++) w[k2] = X[k2];
        }
      }
      float2 r[R1];
      inverse_from_regs<L, false, G::NL>(w, r, lds, c, j, twl);
      float2* __restrict__ out = a.out[(MODE == Z_PGRAD) ? 2 * o : arr];
      if (ACTW(R2, j) && valid)
      {
        uint32_t ob = base + static_cast<uint32_t>(j) * zstr;
        asm volatile("" : "+v"(ob)); // recomputed per output: 16 hoisted 64-bit addresses cost an occupancy step
#pragma unroll
        for (int q2 = 0; q2 < R1; q2++) st_uni<WA>(out, static_cast<uint64_t>(R2 * q2) * zstr, ob * static_cast<uint32_t>(sizeof(float2)), r[q2]);
      }
      // the next output's inverse writes rows while other threads may still read their columns: barrier; the next
      // array's forward transform writes the column this thread has just read: none
      if (o + 1 < NOUT) lds_barrier();
    }
  }
}

// =====================================================================================================================
// Lines of 512 = 2 x 256.  One radix-2 decimation-in-frequency stage in registers around two 256-point four-step
// transforms that use the block's exchange buffer one after the other:
//   forward / standalone (either sign):  a[n] = x[n] + x[n+H],  b[n] = (x[n] - x[n+H]) * W_L^(DIR*n),
//                                        X[2k] = F_H(a)[k],     X[2k+1] = F_H(b)[k]
//   inverse started from spectrum registers:  a = F_H^-1(X[2k]),  b = F_H^-1(X[2k+1]),
//                                        x[n] = a[n] + W_L^(+n) b[n],   x[n+H] = a[n] - W_L^(+n) b[n]
// 16 lines x 16 threads with 32 elements per thread: the tile width (128-B segments), block size and LDS footprint of the
// 256-point kernels, instead of 32-point register DFTs whose register demand leaves one or two waves per SIMD.
// =====================================================================================================================
template<int L> __device__ __forceinline__ void load_twiddles_split(float2* twl, float2* tw2, const float2* __restrict__ tw)
{
  using G = Geo<L / 2>;
  for (int e = threadIdx.x; e < G::R1 * G::R2; e += G::THREADS)
  {
    const int k = e / G::R2, n = e - k * G::R2;
    twl[k * G::TP + n] = tw[2 * k * n]; // W_H^(kn) = W_L^(2kn)
  }
  for (int e = threadIdx.x; e < L / 2; e += G::THREADS) tw2[e] = tw[e];
}

// H-point transform of the step-A registers v (thread (c, n2 = j) holds x[n1*R2 + j]); thread (c, k1 = j) ends with
// X[j + R1*k2] in w.  Default cell assignment: the thread writes its column of the exchange buffer and reads its row;
// ROWWRITE: the transposed one (writes its row, reads its column) — for a transform that follows an exchange whose
// read was by rows, so that every thread overwrites only what it read itself and no barrier is needed in between
// (see inverse_from_regs).  Otherwise the buffer must be free on entry.
template<int H, int DIR, bool ROWWRITE = false>
__device__ __forceinline__ void line_fft(float2 (&v)[Fac<H>::R1], float2 (&w)[Fac<H>::R2], float2* lds, int c, int j,
                                         const float2* twl)
{
  using G = Geo<H>;
  static_assert(!ROWWRITE || G::R1 == G::R2, "transposed exchange needs a square factorisation");
  step_a<H, DIR>(v, j, twl);
#pragma unroll
  for (int k1 = 0; k1 < G::R1; k1++) lds[ROWWRITE ? j * G::SF + k1 * G::NL + c : k1 * G::SF + j * G::NL + c] = v[k1];
  lds_barrier();
#pragma unroll
  for (int n2 = 0; n2 < G::R2; n2++) w[n2] = lds[ROWWRITE ? n2 * G::SF + j * G::NL + c : j * G::SF + n2 * G::NL + c];
  Dft<G::R2, DIR>::run(w);
}

template<int L, int DIR, bool PIN, bool POUT>
__global__ __launch_bounds__(Geo<L / 2>::THREADS) void k_ypass_split(PassArgs a)
{
  constexpr int H = L / 2;
  using G = Geo<H>;
  constexpr int R1 = G::R1, R2 = G::R2;
  static_assert(R1 == R2 && G::TPL == R1 && H == R1 * R2, "split lines are built on the balanced 256-point transform");
  __shared__ float2 lds[G::LDSB];
  __shared__ float2 twl[G::TWN];
  __shared__ float2 tw2[H];
  load_twiddles_split<L>(twl, tw2, a.tw);
  const int      c     = threadIdx.x % G::NL;
  const int      j     = threadIdx.x / G::NL;
  const TileCoord tc   = tile_coord<G::NL>(a.nxc, a.P, a.side_off, gridDim.y, c);
  if (tc.dead) return;
  const uint32_t kx    = tc.side ? 0u : tc.kx;
  const bool     valid = tc.valid;
  const uint32_t kxl   = tc.col;
  const uint32_t z     = tc.pos + a.z0;
  const uint32_t Pe    = tc.pitch, soff = tc.off;
  const uint32_t PXe   = tc.side ? 1u : a.PX;
  const float2* __restrict__ Sin = a.in[blockIdx.z];
  float2* __restrict__ Sout      = a.out[blockIdx.z];

  float2 va[R1], vb[R1];
  if (PIN)
  {
#pragma unroll
    for (int n1 = 0; n1 < R1; n1++)
    {
      va[n1] = Sin[a.ain.row(z, n1 * R2 + j) * PXe + kxl + soff];
      vb[n1] = Sin[a.ain.row(z, H + n1 * R2 + j) * PXe + kxl + soff];
    }
  }
  else
  {
    const uint32_t b    = (z * a.ain.zmul + j * a.ain.estride) * Pe + kxl + soff;
    const uint32_t step = R2 * a.ain.estride * Pe;
#pragma unroll
    for (int n1 = 0; n1 < R1; n1++)
    {
      va[n1] = ld_uni(Sin, static_cast<uint64_t>(n1) * step, b * static_cast<uint32_t>(sizeof(float2)));
      vb[n1] = ld_uni(Sin, static_cast<uint64_t>(R1 + n1) * step, b * static_cast<uint32_t>(sizeof(float2)));
    }
  }
  lds_barrier(); // twiddle tables visible (the line loads stay in flight across it)
  const float2* __restrict__ m = a.mul[blockIdx.z];
#pragma unroll
  for (int n1 = 0; n1 < R1; n1++)
  {
    float2 lo = va[n1], hi = vb[n1];
    if (m != nullptr) { lo = cmulf(lo, m[n1 * R2 + j]); hi = cmulf(hi, m[H + n1 * R2 + j]); }
    va[n1] = cadd(lo, hi);
    vb[n1] = apply_tw<DIR>(csub(lo, hi), tw2[n1 * R2 + j]);
  }
  float2 wa[R2], wb[R2];
  line_fft<H, DIR>(va, wa, lds, c, j, twl);
  line_fft<H, DIR, true>(vb, wb, lds, c, j, twl); // writes the rows the first transform just read: no barrier
  if (valid)
  {
    if (POUT)
    {
#pragma unroll
      for (int k2 = 0; k2 < R2; k2++)
      {
        Sout[a.aout.row(z, 2 * (j + R1 * k2)) * PXe + kx + soff]     = wa[k2];
        Sout[a.aout.row(z, 2 * (j + R1 * k2) + 1) * PXe + kx + soff] = wb[k2];
      }
    }
    else
    {
      const uint32_t b    = (z * a.aout.zmul + 2 * j * a.aout.estride) * Pe + kx + soff;
      const uint32_t one  = a.aout.estride * Pe;
      const uint32_t step = 2 * R1 * one;
#pragma unroll
      for (int k2 = 0; k2 < R2; k2++)
      {
        st_uni(Sout, static_cast<uint64_t>(k2) * step, b * static_cast<uint32_t>(sizeof(float2)), wa[k2]);
        st_uni(Sout, static_cast<uint64_t>(k2) * step + one, b * static_cast<uint32_t>(sizeof(float2)), wb[k2]);
      }
    }
  }
}

template<int L, int MODE> __global__ __launch_bounds__(Geo<L / 2>::THREADS) void k_zfused_split(ZArgs a)
{
  constexpr int H = L / 2;
  using G = Geo<H>;
  constexpr int R1 = G::R1, R2 = G::R2;
  static_assert(R1 == R2 && G::TPL == R1 && H == R1 * R2, "split lines are built on the balanced 256-point transform");
  __shared__ float2 lds[G::LDSB];
  __shared__ float2 twl[G::TWN];
  __shared__ float2 tw2[H];
  load_twiddles_split<L>(twl, tw2, a.tw);
  const int      c      = threadIdx.x % G::NL;
  const int      j      = threadIdx.x / G::NL;
  const TileCoord tc    = tile_coord<G::NL>(a.nxc, a.P, a.side_off, gridDim.y, c);
  if (tc.dead) return;
  const uint32_t ky     = tc.pos;
  const bool     valid  = tc.valid;
  const uint32_t zstr   = a.ny * tc.pitch;
  const uint32_t kxl    = tc.kx < a.nxc ? tc.kx : (tc.side ? a.nxc : a.nxc - 1u); // true column (ddx lookups)
  const uint32_t base   = ky * tc.pitch + (tc.side ? 0u : tc.kx) + tc.off;
  const uint32_t basel  = ky * tc.pitch + tc.col + tc.off;
  // operators live in a tile-blocked layout [ky][kx tile][kz][16]: the 16 x nz values of this block's tile are one
  // contiguous run (k_import_reduced), instead of 64-B pieces a whole plane apart
  constexpr int  OPV    = op_vec(2 * R2); // run of this thread: bins 2*(j + R1*k2) + h at run index 2*k2 + h
  const uint32_t opbase = tc.side ? a.op_side_off + (((ky / NLMAX) * (2 * R2 / OPV) * R1 + j) * NLMAX + ky % NLMAX) * OPV
                                  : (((ky * (a.Pop / NLMAX) + tc.col / NLMAX) * (2 * R2 / OPV) * R1 + j) * NLMAX + tc.col % NLMAX) * OPV;
  constexpr bool MULTI  = (MODE == Z_VGRAD || MODE == Z_ABSORB);
  const uint32_t arr    = MULTI ? a.arr0 + blockIdx.z : 0;
  // PGRAD: the x- and y-gradients share one z-inverse — ddx(kx), ddy(ky) do not depend on kz, so they are applied after
  // the way back (y-pass / x-pass); only Q = F_z^-1{kappa X} and G_z = F_z^-1{ddz kappa X} leave this kernel
  constexpr int  NOUT   = (MODE == Z_PGRAD) ? 2 : 1;

  float2 Xa[R2], Xb[R2]; // after the forward transform: X[2*(j + R1*k2)], X[2*(j + R1*k2) + 1]
  {
    float2 va[R1], vb[R1];
    const float2* __restrict__ in = a.in[arr];
    const uint32_t lb = basel + static_cast<uint32_t>(j) * zstr;
#pragma unroll
    for (int n1 = 0; n1 < R1; n1++)
    {
      va[n1] = ld_uni(in, static_cast<uint64_t>(n1 * R2) * zstr, lb * static_cast<uint32_t>(sizeof(float2)));
      vb[n1] = ld_uni(in, static_cast<uint64_t>(H + n1 * R2) * zstr, lb * static_cast<uint32_t>(sizeof(float2)));
    }
    lds_barrier(); // twiddle tables visible
#pragma unroll
    for (int n1 = 0; n1 < R1; n1++)
    {
      const float2 lo = va[n1], hi = vb[n1];
      va[n1] = cadd(lo, hi);
      vb[n1] = apply_tw<kFwd>(csub(lo, hi), tw2[n1 * R2 + j]);
    }
    // four exchanges per line and output (two forward, two inverse halves), each writing exactly the cells its thread
    // read in the one before (column / row / column / row ...): the only barriers left are the ones inside them
    line_fft<H, kFwd>(va, Xa, lds, c, j, twl);
    line_fft<H, kFwd, true>(vb, Xb, lds, c, j, twl);
  }
  { // spectral operator (see k_zfused for the reference lines), kz = 2*(j + R1*k2) (+1)
    float run[2 * R2];
    load_op_run<2 * R2, R1>(run, a.op[(MODE == Z_ABSORB) ? arr : 0], opbase);
#pragma unroll
    for (int k2 = 0; k2 < R2; k2++)
    {
      float sa = run[2 * k2];
      float sb = run[2 * k2 + 1];
      if (MODE == Z_VGRAD || MODE == Z_SOURCE) { sa *= a.divider; sb *= a.divider; }
      Xa[k2] = make_float2(Xa[k2].x * sa, Xa[k2].y * sa);
      Xb[k2] = make_float2(Xb[k2].x * sb, Xb[k2].y * sb);
    }
  }
#pragma unroll 1
  for (int o = 0; o < NOUT; o++)
  {
    const uint32_t axis = (MODE == Z_PGRAD) ? (o == 0 ? 3u : 2u) : a.axis_of[arr];
    float2 ra[R1], rb[R1];
#pragma unroll
    for (int half = 0; half < 2; half++)
    {
      float2 w[R2];
      if (MODE == Z_PGRAD || MODE == Z_VGRAD)
      {
        if (axis == 3)
        {
#pragma unroll
          for (int k2 = 0; k2 < R2; k2++) w[k2] = half ? Xb[k2] : Xa[k2];
        }
        else if (axis == 2)
        {
          uint32_t jz = 2 * j + half;
          asm volatile("" : "+v"(jz));
#pragma unroll
          for (int k2 = 0; k2 < R2; k2++) w[k2] = cmulf(half ? Xb[k2] : Xa[k2], a.dd[2][jz + 2 * R1 * k2]);
        }
        else
        {
          const float2 dxy = (axis == 0) ? a.dd[0][kxl] : a.dd[1][ky + a.ky0];
#pragma unroll
          for (int k2 = 0; k2 < R2; k2++) w[k2] = cmulf(half ? Xb[k2] : Xa[k2], dxy);
        }
      }
      else
      {
#pragma unroll
        for (int k2 = 0; k2 < R2; k2++) w[k2] = half ? Xb[k2] : Xa[k2];
      }
      if (half == 0) inverse_from_regs<H, true>(w, ra, lds, c, j, twl); // after a read by columns: write columns
      else inverse_from_regs<H>(w, rb, lds, c, j, twl);                  // after a read by rows: write rows
    }
    if (valid)
    {
      float2* __restrict__ out = a.out[(MODE == Z_PGRAD) ? 2 * o : arr];
      uint32_t ob = base + static_cast<uint32_t>(j) * zstr;
      asm volatile("" : "+v"(ob));
#pragma unroll
      for (int q2 = 0; q2 < R1; q2++)
      { // n = j + R2*q2
        const float2 t = apply_tw<kInv>(rb[q2], tw2[j + R2 * q2]);
        st_uni(out, static_cast<uint64_t>(R2 * q2) * zstr, ob * static_cast<uint32_t>(sizeof(float2)), cadd(ra[q2], t));
        st_uni(out, static_cast<uint64_t>(H + R2 * q2) * zstr, ob * static_cast<uint32_t>(sizeof(float2)), csub(ra[q2], t));
      }
    }
  }
}

// =====================================================================================================================
// x-forward: two real rows per complex line -> two half-spectrum rows
// =====================================================================================================================
struct XfwdArgs
{
  const float*  in[3];
  float2*       out[3];
  const float2* tw;
  uint32_t      nx, P;
  uint32_t      nrows, tile0; // rows of the grid (ny * nz); first tile of this launch
  uint32_t      side_off;     // element offset of the x-Nyquist side array in out[] (0: the bin stays in its row)
};

// The x kernels work on tiles of 2 * NL rows.  A grid whose row count ny * nz is not a whole number of tiles ends in a
// partial tile: that one tile is launched on its own with TAIL = true — row loads clamp to the last row, every store is
// predicated on the row — so the full tiles keep their unmasked kernels.

// forward line FFT of this block's 16 complex lines (= 32 real rows) from the step-A registers v (of role XRole<G, R2>),
// split into the two half-spectra and stored to rows tile_row0.. of `out`.  Called by every thread of the block; the
// exchange buffer must be free on entry and is free again on exit (trailing barrier).
// NLX: line pairs per block.  PLANE (the tile is a whole z-plane, see k_xinv): the half-spectrum rows go to the block's
// plane buffer in LDS ([row][L / 2 + 1], `out`) instead of to the scratch array.
template<int L, bool TAIL = false, int NLX = nl_x(L), bool PLANE = false>
__device__ __forceinline__ void xfwd_tail(float2 (&v)[Fac<L>::R1], float2* lds,
                                          const float2* tw, float2* __restrict__ out, uint32_t P, uint32_t tile,
                                          uint32_t nrows = 0, uint32_t side_off = 0)
{
  using G = Geo<L, NLX>;
  constexpr int R1 = G::R1, R2 = G::R2, HALF = L / 2 + 1;
  const XRole<G, R2> sa; // v: the step-A registers of line sa.c, column sa.f
  const XRole<G, R1> sb;
  if (sa.on)
  {
    step_a<L, kFwd>(v, sa.f, tw);
#pragma unroll
    for (int k1 = 0; k1 < R1; k1++) lds[sa.c * G::LP + k1 * (R2 + 1) + sa.f] = v[k1];
  }
  lds_barrier();
  float2 w[R2];
  if (sb.on)
  {
#pragma unroll
    for (int n2 = 0; n2 < R2; n2++) w[n2] = lds[sb.c * G::LP + sb.f * (R2 + 1) + n2];
    Dft<R2, kFwd>::run(w);
  }
  lds_barrier();
  if (sb.on)
  {
#pragma unroll
    for (int k2 = 0; k2 < R2; k2++) lds[sb.c * G::ZP + sb.f + R1 * k2] = w[k2];
  }
  lds_barrier();
  const uint32_t tile_row0 = tile * G::NL * 2;
  for (int e = threadIdx.x; e < G::NL * HALF; e += G::THREADS)
  {
    const int    cc = e / HALF;
    const int    k  = e - cc * HALF;
    const float2 zk = lds[cc * G::ZP + k];
    const float2 zn = lds[cc * G::ZP + (k == 0 ? 0 : L - k)];
    const float2 xa = make_float2(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y));
    const float2 xb = make_float2(0.5f * (zk.y + zn.y), 0.5f * (zn.x - zk.x));
    if (PLANE)
    {
      out[(2 * cc) * HALF + k]     = xa;
      out[(2 * cc + 1) * HALF + k] = xb;
      continue;
    }
    const uint32_t r = tile_row0 + 2 * cc;
    // the x-Nyquist bin of a row goes to the side array N[row] when the pipeline keeps that column apart (tile_coord)
    const bool     ny_bin = (side_off != 0) && (k == L / 2);
    const uint32_t ia = ny_bin ? side_off + r : r * P + k;
    const uint32_t ib = ny_bin ? ia + 1u : ia + P;
    if (!TAIL || r < nrows) out[ia] = xa;
    if (!TAIL || r + 1 < nrows) out[ib] = xb;
  }
  lds_barrier();
}

template<int L, bool TAIL = false> __global__ __launch_bounds__(GeoX<L>::THREADS) void k_xfwd(XfwdArgs a)
{
  using G = GeoX<L>;
  constexpr int R1 = G::R1, R2 = G::R2;
  __shared__ float2 lds[G::LDSX];
  __shared__ float2 twl[G::TWN];
  load_twiddles<L>(twl, a.tw);
  const XRole<G, R2> sa;
  const float* __restrict__ in = a.in[blockIdx.y];
  const uint32_t tile = blockIdx.x + a.tile0;
  const uint32_t row0 = (tile * G::NL + sa.c) * 2;
  float2 v[R1];
  if (sa.on)
  {
    const float* __restrict__ ra = in + (TAIL ? min(row0, a.nrows - 1u) : row0) * L;
    const float* __restrict__ rb = TAIL ? in + min(row0 + 1u, a.nrows - 1u) * L : ra + L;
#pragma unroll
    for (int n1 = 0; n1 < R1; n1++) v[n1] = make_float2(ra[n1 * R2 + sa.f], rb[n1 * R2 + sa.f]);
  }
  lds_barrier(); // twiddle table visible
  xfwd_tail<L, TAIL>(v, lds, twl, a.out[blockIdx.y], a.P, tile, a.nrows, a.side_off);
}

// =====================================================================================================================
// x-inverse + epilogue
// =====================================================================================================================
enum Epi { EPI_STORE = 0, EPI_VELOCITY = 1, EPI_INITVEL = 2, EPI_DENSITY = 3, EPI_PSUM = 4 };

struct XinvArgs
{
  const float2* in[3];
  const float2* tw;
  kw_constants  c;
  uint32_t      P;
  // EPI_STORE: out[0]; EPI_VELOCITY/INITVEL: u[3], dtrho[3] (NULL -> scalar), pml[3]
  // EPI_DENSITY: rho[3] in/out, pml[3], rho0, bona, du[3] (optional store), t[3] terms outputs, flags
  // EPI_PSUM: p, first, c2, tau, eta
  float*        out[3];
  const float*  m0[3]; // dtrho / (rho0, bona, -) / (first, c2, -)
  const float*  m1[3]; // pml / (tau, eta, -)
  float*        aux[3]; // du stores / -
  float*        t[3];  // pressure-term outputs
  int           nonlinear;
  int           terms; // 0 none, 1 linear (t0 = sum rho, t1 = rho0*sum du), 2 nonlinear (t0, t1 = nonlinear term, t2)
  uint32_t      comp0; // first component of this launch (per-array launches)
  float2*       fout[3]; // CHAIN: where the forward x-transform of the epilogue's result goes (scratch rows)
  uint32_t      tile0;   // first 2*NL-row tile of this launch (chunked plane-local passes)
  const float2* mulx[3]; // per component: optional factor mulx[kx] applied to the rows before the inverse (ddx of the gradient)
  uint32_t      nrows;      // rows of the grid (ny * nz): bounds the partial last tile (TAIL kernels)
  uint32_t      side_off;   // element offset of the x-Nyquist side array in in[] / fout[] (0: none)
  const float2* ymul[3];    // PLANE kernels: optional factor ymul[ky] applied to array i before its y-inverse (ddy of the gradient)
};

// standalone inverse of one array for this block's 32 rows; the thread of role sb = XRole<G, R1> ends with
// x[sb.f + R1*k2] of rows (2 sb.c, 2 sb.c + 1)
// PLANE: the rows come from the block's plane buffer in LDS ([row][L / 2 + 1], `src`), where the y-inverse left them
template<int L, int NGRP = 1, bool TAIL = false, int NLX = nl_x(L), bool PLANE = false>
__device__ __forceinline__ void xinv_lines(const float2* __restrict__ src, uint32_t P, float2* lds,
                                           const float2* tw, float2 (&w)[Fac<L>::R2], uint32_t tile,
                                           const float2* __restrict__ mulx = nullptr, uint32_t nrows = 0,
                                           uint32_t side_off = 0)
{
  using G = Geo<L, NLX>;
  constexpr int R1 = G::R1, R2 = G::R2, HALF = L / 2 + 1;
  const uint32_t tile_row0 = tile * G::NL * 2;
  // all row loads of the tile are requested before the first one is consumed (a load-use loop would expose the memory
  // latency once per iteration); the last, partial iteration re-reads the final element instead of being predicated
  constexpr int NE = (G::NL * HALF + G::THREADS - 1) / G::THREADS;
  constexpr int NG = (NE + NGRP - 1) / NGRP; // iterations per group (NGRP > 1: fewer loads in flight, fewer registers)
#pragma unroll
  for (int g0 = 0; g0 < NE; g0 += NG)
  {
    float2 A[NG], B[NG];
#pragma unroll
    for (int it = 0; it < NG; it++)
    {
      const int e  = min(static_cast<int>(threadIdx.x) + (g0 + it) * G::THREADS, G::NL * HALF - 1);
      const int cc = e / HALF;
      const int k  = e - cc * HALF;
      const uint32_t r = tile_row0 + 2 * cc;
      const uint32_t ra = TAIL ? min(r, nrows - 1u) : r, rb = TAIL ? min(r + 1u, nrows - 1u) : r + 1u;
      if (PLANE)
      {
        A[it] = src[(2 * cc) * HALF + k];
        B[it] = src[(2 * cc + 1) * HALF + k];
        continue;
      }
      const bool     ny_bin = (side_off != 0) && (k == L / 2); // the x-Nyquist bin lives in the side array N[row]
      A[it] = src[ny_bin ? side_off + ra : ra * P + k];
      B[it] = src[ny_bin ? side_off + rb : rb * P + k];
    }
#pragma unroll
    for (int it = 0; it < NG; it++)
    {
      const int e = static_cast<int>(threadIdx.x) + (g0 + it) * G::THREADS;
      if (g0 + it < NE && e < G::NL * HALF)
      {
        const int cc = e / HALF;
        const int k  = e - cc * HALF;
        float2 a = A[it], b = B[it];
        if (mulx != nullptr) { const float2 d = mulx[k]; a = cmulf(a, d); b = cmulf(b, d); }
        if (k == 0 || k == L / 2) { a.y = 0.f; b.y = 0.f; } // C2R ignores the imaginary part of DC / Nyquist
        lds[cc * G::ZP + k] = make_float2(a.x - b.y, a.y + b.x);
        if (k != 0 && k != L / 2) lds[cc * G::ZP + L - k] = make_float2(a.x + b.y, b.x - a.y);
      }
    }
  }
  lds_barrier();
  const XRole<G, R2> sa;
  const XRole<G, R1> sb;
  float2 v[R1];
  if (sa.on)
  {
#pragma unroll
    for (int n1 = 0; n1 < R1; n1++) v[n1] = lds[sa.c * G::ZP + n1 * R2 + sa.f];
  }
  lds_barrier();
  if (sa.on)
  {
    step_a<L, kInv>(v, sa.f, tw);
#pragma unroll
    for (int k1 = 0; k1 < R1; k1++) lds[sa.c * G::LP + k1 * (R2 + 1) + sa.f] = v[k1];
  }
  lds_barrier();
  if (sb.on)
  {
#pragma unroll
    for (int n2 = 0; n2 < R2; n2++) w[n2] = lds[sb.c * G::LP + sb.f * (R2 + 1) + n2];
    Dft<R2, kInv>::run(w);
  }
  lds_barrier();
}

__device__ __forceinline__ float  f4get(const float4& v, int k) { return k == 0 ? v.x : k == 1 ? v.y : k == 2 ? v.z : v.w; }
__device__ __forceinline__ void   f4put(float4& v, int k, float s)
{
  if (k == 0) v.x = s; else if (k == 1) v.y = s; else if (k == 2) v.z = s; else v.w = s;
}


// ---- small grids: a whole z-plane per block -------------------------------------------------------------------------------
// When a half-spectrum plane [L][L / 2 + 1] (Nx == Ny == L <= 64) fits the LDS next to the x kernels' line buffers, the
// block of an x-inverse + epilogue kernel takes a whole plane as its tile and does the plane's y transforms itself: the
// y-inverse of every input array before its x-inverse, the y-forward of every chained array after its x-forward — same
// small DFTs, same twiddles, same order of operations as k_ypass (bit-identical results), but the stage's tail is ONE
// launch instead of three.  32^3 and 64^3 are bound by launches and by kernels that are over before the chip has filled
// (13 kernels of 5-15 us per step): this takes six of them away and the spectra of a stage's tail never leave the CU.
// plane buffer Y[ky][kx], kx < L / 2 + 1 (the x-Nyquist bin sits in its row here: no side array in LDS)
template<int L> __device__ __forceinline__ void plane_load(float2* Y, const float2* __restrict__ src, uint32_t plane, uint32_t P,
                                                           uint32_t side_off, const float2* __restrict__ mul, int threads)
{ // rows plane*L .. of the scratch array (+ their side-array bins) -> Y, times mul[ky] where given (ddy of the gradient)
  constexpr int HALF = L / 2 + 1;
  const uint32_t row0 = plane * L;
  if (side_off != 0)
  { // rows of exactly L / 2 bins
    for (int e = threadIdx.x; e < L * (L / 2); e += threads)
    {
      const int ky = e / (L / 2), k = e % (L / 2);
      float2 v = src[(row0 + ky) * P + k];
      if (mul != nullptr) v = cmulf(v, mul[ky]);
      Y[ky * HALF + k] = v;
    }
    for (int ky = threadIdx.x; ky < L; ky += threads)
    {
      float2 v = src[side_off + row0 + ky];
      if (mul != nullptr) v = cmulf(v, mul[ky]);
      Y[ky * HALF + L / 2] = v;
    }
  }
  else
  {
    for (int e = threadIdx.x; e < L * HALF; e += threads)
    {
      const int ky = e / HALF, k = e - ky * HALF;
      float2 v = src[(row0 + ky) * P + k];
      if (mul != nullptr) v = cmulf(v, mul[ky]);
      Y[e] = v;
    }
  }
}
template<int L> __device__ __forceinline__ void plane_store(const float2* Y, float2* __restrict__ dst, uint32_t plane, uint32_t P,
                                                            uint32_t side_off, int threads)
{
  constexpr int HALF = L / 2 + 1;
  const uint32_t row0 = plane * L;
  if (side_off != 0)
  {
    for (int e = threadIdx.x; e < L * (L / 2); e += threads)
    {
      const int ky = e / (L / 2), k = e % (L / 2);
      dst[(row0 + ky) * P + k] = Y[ky * HALF + k];
    }
    for (int ky = threadIdx.x; ky < L; ky += threads) dst[side_off + row0 + ky] = Y[ky * HALF + L / 2];
  }
  else
  {
    for (int e = threadIdx.x; e < L * HALF; e += threads)
    {
      const int ky = e / HALF, k = e - ky * HALF;
      dst[(row0 + ky) * P + k] = Y[e];
    }
  }
}
// in-place transform of the L / 2 + 1 columns of Y along y: the four-step of k_ypass with the plane buffer itself as the
// exchange buffer.  (L / 2) * TPL threads: thread (c, j) of the main columns, then the first TPL threads once more for the
// x-Nyquist column.  Y must be complete on entry (barrier before); complete again on exit (trailing barrier).
template<int L, int DIR> __device__ __forceinline__ void plane_yfft(float2* Y, const float2* twl)
{
  using G = Geo<L, L / 2>;
  constexpr int R1 = G::R1, R2 = G::R2, HALF = L / 2 + 1, TPL = G::TPL;
#pragma unroll 1
  for (int pass = 0; pass < 2; pass++)
  {
    const bool on = (pass == 0) || (threadIdx.x < TPL);
    const int  c  = (pass == 0) ? static_cast<int>(threadIdx.x) % (L / 2) : L / 2;
    const int  j  = (pass == 0) ? static_cast<int>(threadIdx.x) / (L / 2) : static_cast<int>(threadIdx.x);
    if (on && ACT(R2, j))
    { // step A on rows n1 * R2 + j; the results go back to the same cells (index k1 in place of n1)
      float2 v[R1];
#pragma unroll
      for (int n1 = 0; n1 < R1; n1++) v[n1] = Y[(n1 * R2 + j) * HALF + c];
      step_a<L, DIR>(v, j, twl);
#pragma unroll
      for (int k1 = 0; k1 < R1; k1++) Y[(k1 * R2 + j) * HALF + c] = v[k1];
    }
    lds_barrier();
    float2 w[R2];
    if (on && ACT(R1, j))
    {
#pragma unroll
      for (int n2 = 0; n2 < R2; n2++) w[n2] = Y[(j * R2 + n2) * HALF + c];
      Dft<R2, DIR>::run(w);
    }
    lds_barrier(); // every cell has been read before the natural-order results overwrite them
    if (on && ACT(R1, j))
    {
#pragma unroll
      for (int k2 = 0; k2 < R2; k2++) Y[(j + R1 * k2) * HALF + c] = w[k2];
    }
    lds_barrier();
  }
}
// (128: one 1024-thread block per CU, held to 128 VGPRs — measured 14 % slower than the three-launch form; 32 and 64 gain)
constexpr bool plane_len(int L) { return L == 32 || L == 64; }

// The inverse leaves each thread with x = f + R1*k2 of two rows — a 64-B-segment pattern.  The results are restaged
// through LDS as a plain real tile [32 rows][L] and re-read as float4 in a row-contiguous mapping, so that every
// epilogue access to the state / medium arrays is a 16-B-per-lane coalesced access.
// (the 256-point density epilogue sits two registers above the 3-waves-per-SIMD step: ask the allocator for that step)
// TERMS: compile-time value of a.terms for the density epilogue (one specialised kernel per pressure-term mode)
// PLANE: the tile is a whole z-plane (L / 2 line pairs) and the kernel does the plane's y transforms too — see above.
template<int L, int EPI, bool CHAIN, int TERMS = 0, bool TAIL = false, bool PLANE = false>
__global__ __launch_bounds__((Geo<L, PLANE ? L / 2 : nl_x(L)>::THREADS), (EPI == EPI_DENSITY && L == 256) ? 3 : big_line_waves(L)) void k_xinv(XinvArgs a)
{
  constexpr int terms = TERMS;
  constexpr int NLX = PLANE ? L / 2 : nl_x(L); // (the registers a thread needs follow from L / TPL, not from the lines per block)
  using G = Geo<L, NLX>;
  constexpr int R1 = G::R1, R2 = G::R2;
  constexpr int NA  = (EPI == EPI_DENSITY) ? 3 : (EPI == EPI_PSUM) ? 2 : 1;
  constexpr int RP  = L + 8;                       // real-tile row pitch (floats): conflict-free 4-B scatter
  constexpr int Q4  = L / 4;                       // float4 per row
  constexpr int TOT4 = 2 * G::NL * Q4;              // float4 of the real tile
  constexpr int NQ  = (TOT4 + G::THREADS - 1) / G::THREADS;  // float4 per thread
  // L / 2 not a multiple of the threads per line (25 x 28, 27 x 28, 25 x 32, 27 x 32): the tile does not divide by the block;
  // the surplus lanes of the last round read the tile's last float4 (XE) and store nothing (XE_OK)
  constexpr bool QPART = (TOT4 % G::THREADS) != 0;
#define XE(q) (QPART ? min(static_cast<int>(threadIdx.x) + (q) * G::THREADS, TOT4 - 1) : static_cast<int>(threadIdx.x) + (q) * G::THREADS)
#define XE_OK(q) (!QPART || static_cast<int>(threadIdx.x) + (q) * G::THREADS < TOT4)
  __shared__ float2 lds[G::LDSX];
  __shared__ float2 twl[G::TWN];
  __shared__ float2 Yp[PLANE ? L * (L / 2 + 1) : 1]; // PLANE: the plane's half-spectrum between its y and x transforms
  load_twiddles<L>(twl, a.tw); // published by the first barrier of xinv_lines (PLANE: of the plane's y transform)
  float* ldsr = reinterpret_cast<float*>(lds);
  const XRole<G, R2> sa; // (see XRole: who takes part in the R2- and in the R1-participant steps)
  const XRole<G, R1> sb;
  const uint32_t comp = (NA == 1) ? blockIdx.y + a.comp0 : 0; // component / array index for single-array epilogues
  const uint32_t tile = blockIdx.x + a.tile0;
  float4 res[NA][NQ];
  constexpr int NF = CHAIN ? ((EPI == EPI_DENSITY) ? 2 : 1) : 1; // chained forward transforms
  // rows to chain: kept in registers, except the first of the density epilogue's two, which goes straight into the
  // real tile in LDS (free once the last inverse has been read out) — 32 registers less at the kernel's widest point
  constexpr bool FW0_IN_LDS = CHAIN && (EPI == EPI_DENSITY);
  float4 fw[FW0_IN_LDS ? 1 : NF][NQ];
#pragma unroll
  for (int i = 0; i < NA; i++)
  {
    float2 w[R2];
    if constexpr (PLANE)
    { // this array's plane: scratch -> LDS (x ddy[ky] for the y-gradient), inverse along y, then rows out of LDS
      const uint32_t ia = (NA == 1) ? comp : i;
      plane_load<L>(Yp, a.in[ia], tile, a.P, a.side_off, a.ymul[ia], G::THREADS);
      lds_barrier();
      plane_yfft<L, kInv>(Yp, twl);
      xinv_lines<L, 1, false, NLX, true>(Yp, a.P, lds, twl, w, tile, (NA == 1) ? a.mulx[comp] : nullptr);
    }
    else
    xinv_lines<L, (EPI == EPI_DENSITY) ? 2 : 1, TAIL, NLX>(a.in[(NA == 1) ? comp : i], a.P, lds, twl, w, tile,
                                                            (NA == 1) ? a.mulx[comp] : nullptr, a.nrows, a.side_off); // ends with a barrier
    if (sb.on)
    {
#pragma unroll
      for (int k2 = 0; k2 < R2; k2++)
      {
        ldsr[(2 * sb.c) * RP + sb.f + R1 * k2]     = w[k2].x;
        ldsr[(2 * sb.c + 1) * RP + sb.f + R1 * k2] = w[k2].y;
      }
    }
    lds_barrier();
#pragma unroll
    for (int q = 0; q < NQ; q++)
    {
      const int e   = XE(q);
      const int row = e / Q4;
      const int x4  = e - row * Q4;
      res[i][q]     = *reinterpret_cast<const float4*>(&ldsr[row * RP + 4 * x4]);
    }
    lds_barrier();
  }

  const kw_constants& k = a.c;
  const uint32_t tile_row0 = tile * G::NL * 2;
  // The operands of the epilogue are requested for a group of GQ float4 per thread before the first one is used: a
  // load - use - store loop per float4 exposes one memory round trip each time (and vmcnt also waits for the stores
  // issued before the loads).  GQ is bounded by the register budget of each epilogue.
  // measured on one box: density 1 (2 spills registers: -1 %), pressure sum 2 (1 gives a fourth wave but -1.4 %),
  // velocity 4 (8: -1 %)
  constexpr int GQ = gq_pick(NQ, (EPI == EPI_DENSITY) ? 1 : (EPI == EPI_PSUM) ? 2 : 4);
  // power-of-two rows: x is the same for every float4 of a thread (one PML-x load per thread); the 3 * 2^m rows whose
  // float4 count does not divide the block take x (and the PML-x operand) per float4
  constexpr bool XFIX = (G::THREADS % Q4 == 0);
  const uint32_t xfix = 4u * (threadIdx.x % Q4);
  float4 pmlx4 = make_float4(1.f, 1.f, 1.f, 1.f);
  if (XFIX && ((EPI == EPI_VELOCITY && comp == 0) || EPI == EPI_DENSITY)) pmlx4 = ld4(a.m1[0] + xfix);
  const bool hetRho0 = (EPI == EPI_DENSITY) && (a.m0[0] != nullptr);
  const bool hetBonA = (EPI == EPI_DENSITY) && (terms == 2 || (terms == 3 && a.nonlinear)) && (a.m0[1] != nullptr);
  const bool hetC2   = (EPI == EPI_DENSITY) && (terms == 3) && (a.m0[2] != nullptr);
#pragma unroll
  for (int q0 = 0; q0 < NQ; q0 += GQ)
  {
    float4 op0[GQ], op1[GQ], op2[GQ], op3[GQ], op4[GQ], op5[GQ], opx[GQ];
    float  sy[GQ], sz[GQ];
#pragma unroll
    for (int g = 0; g < GQ; g++)
    {
      const int      e   = XE(q0 + g);
      const uint32_t r   = TAIL ? min(tile_row0 + e / Q4, a.nrows - 1u) : tile_row0 + e / Q4; // operands of a masked row: the last row's
      const uint32_t z   = r / k.ny;
      const uint32_t y   = r - z * k.ny;
      const uint32_t x   = XFIX ? xfix : 4u * (e % Q4);
      const uint32_t i   = r * L + x;
      if (!XFIX && ((EPI == EPI_VELOCITY && comp == 0) || EPI == EPI_DENSITY)) opx[g] = ld4(a.m1[0] + x);
      if (EPI == EPI_VELOCITY)
      {
        op0[g] = ld4(a.out[comp] + i);
        if (a.m0[comp] != nullptr) op1[g] = ld4(a.m0[comp] + i);
        if (comp == 1) sy[g] = a.m1[1][y];
        if (comp == 2) sy[g] = a.m1[2][z];
      }
      else if (EPI == EPI_INITVEL)
      {
        if (a.m0[comp] != nullptr) op1[g] = ld4(a.m0[comp] + i);
      }
      else if (EPI == EPI_DENSITY)
      {
        op0[g] = ld4(a.out[0] + i);
        op1[g] = ld4(a.out[1] + i);
        op2[g] = ld4(a.out[2] + i);
        if (hetRho0) op3[g] = ld4(a.m0[0] + i);
        if (hetBonA) op4[g] = ld4(a.m0[1] + i);
        if (hetC2) op5[g] = ld4(a.m0[2] + i);
        sy[g] = a.m1[1][y];
        sz[g] = a.m1[2][z];
      }
      else if (EPI == EPI_PSUM)
      {
        op0[g] = ld4(a.m0[0] + i);
        if (a.m0[1] != nullptr) op1[g] = ld4(a.m0[1] + i);
        if (a.m1[0] != nullptr) op2[g] = ld4(a.m1[0] + i);
        if (a.m1[1] != nullptr) op3[g] = ld4(a.m1[1] + i);
      }
    }
#pragma unroll
    for (int g = 0; g < GQ; g++)
    {
      const int      q = q0 + g;
      const int      e = XE(q);
      const uint32_t r = tile_row0 + e / Q4;
      const uint32_t x = XFIX ? xfix : 4u * (e % Q4);
      const uint32_t i = r * L + x;
      const bool     e_ok   = XE_OK(q);                          // (the chained rows in LDS are predicated on it)
      const bool     row_ok = (!TAIL || r < a.nrows) && e_ok;     // every global store below is predicated on it
      const float4   pmx = XFIX ? pmlx4 : opx[g];
      if (EPI == EPI_STORE)
      {
        if (row_ok) st4(a.out[comp] + i, res[0][q]);
      }
      else if (EPI == EPI_VELOCITY)
      { // SolverCudaKernels.cu:199-212 (heterogeneous) / :287-305 (homogeneous)
        float4       vu   = op0[g];
        const float4 gr   = res[0][q];
        const float4 pml4 = (comp == 0) ? pmx : make_float4(sy[g], sy[g], sy[g], sy[g]);
        if (a.m0[comp] != nullptr)
        {
          const float4 d = op1[g];
#pragma unroll
          for (int t = 0; t < 4; t++)
          {
            const float ee = k.fft_divider * f4get(gr, t) * f4get(d, t);
            const float pm = f4get(pml4, t);
            f4put(vu, t, (f4get(vu, t) * pm - ee) * pm);
          }
        }
        else
        {
          const float dtr     = (comp == 0) ? k.dt_rho0_sgx : (comp == 1) ? k.dt_rho0_sgy : k.dt_rho0_sgz;
          const float divider = dtr * k.fft_divider;
#pragma unroll
          for (int t = 0; t < 4; t++)
          {
            const float pm = f4get(pml4, t);
            f4put(vu, t, (f4get(vu, t) * pm - divider * f4get(gr, t)) * pm);
          }
        }
        if (row_ok) st4(a.out[comp] + i, vu);
        if constexpr (CHAIN) fw[0][q] = vu;
      }
      else if (EPI == EPI_INITVEL)
      { // :957-980: u = ifft * (dtRho0Sg * (fftDivider*0.5)) | u = ifft * (fftDivider*0.5*dtRho0Sg)
        const float4 gr = res[0][q];
        float4       o;
        if (a.m0[comp] != nullptr)
        {
          const float4 d = op1[g];
#pragma unroll
          for (int t = 0; t < 4; t++) f4put(o, t, f4get(gr, t) * (f4get(d, t) * (k.fft_divider * 0.5f)));
        }
        else
        {
          const float dtr = (comp == 0) ? k.dt_rho0_sgx : (comp == 1) ? k.dt_rho0_sgy : k.dt_rho0_sgz;
#pragma unroll
          for (int t = 0; t < 4; t++) f4put(o, t, f4get(gr, t) * (k.fft_divider * 0.5f * dtr));
        }
        if (row_ok) st4(a.out[comp] + i, o);
      }
      else if (EPI == EPI_DENSITY)
      { // :1368-1392 (nonlinear) / :1480-1496 (linear); du already carries fftDivider (applied in k-space, :1220)
        const float4 dux = res[0][q], duy = res[1][q], duz = res[2][q];
        const float  py = sy[g], pz = sz[g];
        const float4 rx = op0[g], ry = op1[g], rz = op2[g];
        const float4 r04 = hetRho0 ? op3[g] : make_float4(k.rho0, k.rho0, k.rho0, k.rho0);
        float4 nrx, nry, nrz;
#pragma unroll
        for (int t = 0; t < 4; t++)
        {
          const float px = f4get(pmx, t);
          const float r0 = f4get(r04, t);
          const float erx = f4get(rx, t), ery = f4get(ry, t), erz = f4get(rz, t);
          if (a.nonlinear)
          {
            const float sumRhosDt = (2.0f * (erx + ery + erz) + r0) * k.dt;
            f4put(nrx, t, px * ((px * erx) - sumRhosDt * f4get(dux, t)));
            f4put(nry, t, py * ((py * ery) - sumRhosDt * f4get(duy, t)));
            f4put(nrz, t, pz * ((pz * erz) - sumRhosDt * f4get(duz, t)));
          }
          else
          {
            const float dtRho0 = hetRho0 ? k.dt * r0 : k.dt_rho0;
            f4put(nrx, t, px * (px * erx - dtRho0 * f4get(dux, t)));
            f4put(nry, t, py * (py * ery - dtRho0 * f4get(duy, t)));
            f4put(nrz, t, pz * (pz * erz - dtRho0 * f4get(duz, t)));
          }
        }
        if (row_ok) st4(a.out[0] + i, nrx);
        if (row_ok) st4(a.out[1] + i, nry);
        if (row_ok) st4(a.out[2] + i, nrz);
        if (a.aux[0] != nullptr)
        {
          if (row_ok) st4(a.aux[0] + i, dux);
          if (row_ok) st4(a.aux[1] + i, duy);
          if (row_ok) st4(a.aux[2] + i, duz);
        }
        if (terms == 2)
        { // :1588-1601 with the updated densities
          const float4 b4 = hetBonA ? op4[g] : make_float4(k.b_on_a, k.b_on_a, k.b_on_a, k.b_on_a);
          float4 o0, o1, o2;
#pragma unroll
          for (int t = 0; t < 4; t++)
          {
            const float eBonA   = f4get(b4, t);
            const float r0      = f4get(r04, t);
            const float eRhoSum = (f4get(nrx, t) + f4get(nry, t) + f4get(nrz, t));
            const float eDuSum  = (f4get(dux, t) + f4get(duy, t) + f4get(duz, t));
            f4put(o0, t, eRhoSum);
            f4put(o1, t, ((eBonA * eRhoSum * eRhoSum) / (2.0f * r0)) + eRhoSum);
            f4put(o2, t, r0 * eDuSum);
          }
          if (row_ok) st4(a.t[1] + i, o1); // the nonlinear term is read again by the pressure sum (a stage later: cached or not, same time)
          if constexpr (CHAIN) { if (e_ok) *reinterpret_cast<float4*>(&ldsr[(e / Q4) * RP + x]) = o2; fw[0][q] = o0; }
          else { if (row_ok) st4(a.t[0] + i, o0); if (row_ok) st4(a.t[2] + i, o2); }
        }
        else if (terms == 3)
        { // lossless equation of state on the updated densities: sumPressureNonlinearLossless (:2067-2084) /
          // sumPressureLinearLossless (:2224-2236); the new p is chained like the pressure sum's
          const float4 b4  = hetBonA ? op4[g] : make_float4(k.b_on_a, k.b_on_a, k.b_on_a, k.b_on_a);
          const float4 c24 = hetC2 ? op5[g] : make_float4(k.c2, k.c2, k.c2, k.c2);
          float4 pn;
#pragma unroll
          for (int t = 0; t < 4; t++)
          {
            const float rhoSum = f4get(nrx, t) + f4get(nry, t) + f4get(nrz, t);
            if (a.nonlinear) f4put(pn, t, f4get(c24, t) * (rhoSum + (f4get(b4, t) * (rhoSum * rhoSum) / (2.0f * f4get(r04, t)))));
            else f4put(pn, t, f4get(c24, t) * rhoSum);
          }
          if (row_ok) st4(a.t[0] + i, pn);
          if constexpr (CHAIN) { if (e_ok) *reinterpret_cast<float4*>(&ldsr[(e / Q4) * RP + x]) = pn; }
        }
        else if (terms == 1)
        { // :1733-1741
          float4 o0, o1;
#pragma unroll
          for (int t = 0; t < 4; t++)
          {
            f4put(o0, t, f4get(nrx, t) + f4get(nry, t) + f4get(nrz, t));
            const float duSum = f4get(dux, t) + f4get(duy, t) + f4get(duz, t);
            f4put(o1, t, f4get(r04, t) * duSum);
          }
          if (row_ok) st4(a.t[0] + i, o0); // the density sum is read again by the pressure sum
          if constexpr (CHAIN) { if (e_ok) *reinterpret_cast<float4*>(&ldsr[(e / Q4) * RP + x]) = o1; fw[0][q] = o0; }
          else if (row_ok) st4(a.t[1] + i, o1);
        }
      }
      else if (EPI == EPI_PSUM)
      { // :1877 / :1978: p = c2*(first + (fftDivider*((tauTerm*tau) - (etaTerm*eta))))
        const float4 tt = res[0][q], et = res[1][q];
        const float4 fi = op0[g];
        const float4 c24  = (a.m0[1] != nullptr) ? op1[g] : make_float4(k.c2, k.c2, k.c2, k.c2);
        const float4 tau4 = (a.m1[0] != nullptr) ? op2[g] : make_float4(k.absorb_tau, k.absorb_tau, k.absorb_tau, k.absorb_tau);
        const float4 eta4 = (a.m1[1] != nullptr) ? op3[g] : make_float4(k.absorb_eta, k.absorb_eta, k.absorb_eta, k.absorb_eta);
        float4 o;
#pragma unroll
        for (int t = 0; t < 4; t++)
          f4put(o, t, f4get(c24, t) * (f4get(fi, t) + (k.fft_divider * ((f4get(tt, t) * f4get(tau4, t)) - (f4get(et, t) * f4get(eta4, t))))));
        if (row_ok) st4(a.out[0] + i, o);
        if constexpr (CHAIN) fw[0][q] = o;
      }
    }
  }

  // ---- chained forward x-transform of what the epilogue just produced (rows are still in registers): the consumer
  // stage finds the spectra in the scratch arrays and skips its own x-forward pass and the HBM round trip ----
  if constexpr (CHAIN)
  {
    constexpr int R1c = G::R1, R2c = G::R2;
#pragma unroll
    for (int jf = 0; jf < NF; jf++)
    {
      if (EPI == EPI_DENSITY && jf == 1 && terms == 3) break; // lossless: only p is chained
      if (!(FW0_IN_LDS && jf == 0))
      {
#pragma unroll
        for (int q = 0; q < NQ; q++)
        {
          const int e   = XE(q);
          const int row = e / Q4;
          const int x4  = e - row * Q4;
          if (XE_OK(q)) *reinterpret_cast<float4*>(&ldsr[row * RP + 4 * x4]) = fw[FW0_IN_LDS ? 0 : jf][q];
        }
      }
      lds_barrier();
      float2 v[R1c];
      if (sa.on)
      {
#pragma unroll
        for (int n1 = 0; n1 < R1c; n1++)
          v[n1] = make_float2(ldsr[(2 * sa.c) * RP + n1 * R2c + sa.f], ldsr[(2 * sa.c + 1) * RP + n1 * R2c + sa.f]);
      }
      lds_barrier(); // the real tile aliases the exchange buffer
      if constexpr (PLANE)
      { // rows into the plane buffer, forward along y, plane -> scratch (the consumer's z-pass comes next)
        xfwd_tail<L, false, NLX, true>(v, lds, twl, Yp, a.P, tile);
        plane_yfft<L, kFwd>(Yp, twl);
        plane_store<L>(Yp, a.fout[(NA == 1) ? comp : jf], tile, a.P, a.side_off, G::THREADS);
        lds_barrier(); // the plane buffer is reused by the second chained array
      }
      else
      xfwd_tail<L, TAIL, NLX>(v, lds, twl, a.fout[(NA == 1) ? comp : jf], a.P, tile, a.nrows, a.side_off);
    }
  }
}

// Half-cell shift along x (computeVelocityShiftInX, SolverCudaKernels.cu:2617-2640): rows 2c and 2c+1 travel as the
// real and imaginary part of one complex line — the filter H is Hermitian with a real Nyquist bin (what the reference's
// R2C -> multiply -> C2R applies to each row), so it acts on both parts independently.  One read and one write of the
// array; loads as in k_xfwd, stores as coalesced float4 through the real tile like the x-inverse epilogues.
struct XshiftArgs
{
  const float*  in;
  float*        out;
  const float2* tw;
  const float2* H; // L complex: shift[k] / L on 0 < k < L/2, conjugate above, real parts at k = 0 and L/2
  uint32_t      nrows, tile0;
};

template<int L, bool TAIL = false> __global__ __launch_bounds__(GeoX<L>::THREADS) void k_xshift(XshiftArgs a)
{
  using G = GeoX<L>;
  constexpr int R1 = G::R1, R2 = G::R2;
  constexpr int RP = L + 8, Q4 = L / 4, TOT4 = 2 * G::NL * Q4, NQ = (TOT4 + G::THREADS - 1) / G::THREADS;
  constexpr bool QPART = (TOT4 % G::THREADS) != 0; // (see k_xinv)
  __shared__ float2 lds[G::LDSX];
  __shared__ float2 twl[G::TWN];
  load_twiddles<L>(twl, a.tw);
  float* ldsr = reinterpret_cast<float*>(lds);
  const XRole<G, R2> sa;
  const XRole<G, R1> sb;
  const uint32_t tile_row0 = (blockIdx.x + a.tile0) * G::NL * 2;
  float2 v[R1];
  if (sa.on)
  {
    const uint32_t row0 = tile_row0 + 2 * sa.c;
    const float* __restrict__ ra = a.in + (TAIL ? min(row0, a.nrows - 1u) : row0) * L;
    const float* __restrict__ rb = TAIL ? a.in + min(row0 + 1u, a.nrows - 1u) * L : ra + L;
#pragma unroll
    for (int n1 = 0; n1 < R1; n1++) v[n1] = make_float2(ra[n1 * R2 + sa.f], rb[n1 * R2 + sa.f]);
  }
  lds_barrier(); // twiddle table visible
  if (sa.on)
  {
    step_a<L, kFwd>(v, sa.f, twl);
#pragma unroll
    for (int k1 = 0; k1 < R1; k1++) lds[sa.c * G::LP + k1 * (R2 + 1) + sa.f] = v[k1];
  }
  lds_barrier();
  float2 w[R2];
  if (sb.on)
  {
#pragma unroll
    for (int n2 = 0; n2 < R2; n2++) w[n2] = lds[sb.c * G::LP + sb.f * (R2 + 1) + n2];
    Dft<R2, kFwd>::run(w);
#pragma unroll
    for (int k2 = 0; k2 < R2; k2++) w[k2] = cmulf(w[k2], a.H[sb.f + R1 * k2]);
  }
  lds_barrier();
  if (sb.on)
  { // natural-order spectrum of the line, the starting point of the inverse (as in xinv_lines)
#pragma unroll
    for (int k2 = 0; k2 < R2; k2++) lds[sb.c * G::ZP + sb.f + R1 * k2] = w[k2];
  }
  lds_barrier();
  if (sa.on)
  {
#pragma unroll
    for (int n1 = 0; n1 < R1; n1++) v[n1] = lds[sa.c * G::ZP + n1 * R2 + sa.f];
  }
  lds_barrier();
  if (sa.on)
  {
    step_a<L, kInv>(v, sa.f, twl);
#pragma unroll
    for (int k1 = 0; k1 < R1; k1++) lds[sa.c * G::LP + k1 * (R2 + 1) + sa.f] = v[k1];
  }
  lds_barrier();
  if (sb.on)
  {
#pragma unroll
    for (int n2 = 0; n2 < R2; n2++) w[n2] = lds[sb.c * G::LP + sb.f * (R2 + 1) + n2];
    Dft<R2, kInv>::run(w);
  }
  lds_barrier();
  if (sb.on)
  {
#pragma unroll
    for (int k2 = 0; k2 < R2; k2++)
    {
      ldsr[(2 * sb.c) * RP + sb.f + R1 * k2]     = w[k2].x;
      ldsr[(2 * sb.c + 1) * RP + sb.f + R1 * k2] = w[k2].y;
    }
  }
  lds_barrier();
#pragma unroll
  for (int q = 0; q < NQ; q++)
  {
    const int e   = XE(q);
    const int row = e / Q4;
    const int x4  = e - row * Q4;
    if ((!TAIL || tile_row0 + row < a.nrows) && XE_OK(q))
      st4(a.out + (tile_row0 + row) * L + 4 * x4, *reinterpret_cast<const float4*>(&ldsr[row * RP + 4 * x4]));
  }
}

// import of a reduced real operator into the layout the z-pass reads (see load_op_run):
// dst[ky][kx tile][q][j][c][V] <- src[kz][ky][nxc], kz = j + r1*(q*V + r)  (split lines: 2j + h + 2*r1*k2, run index 2*k2 + h)
// (rows = nyl local ky, nzg planes: the transposed operators of slab mode have the same form)
// nxc: columns of the source rows; nxm <= nxc: columns that go into the row tiles (nxc - 1 when the x-Nyquist column is
// kept apart: k_import_reduced_side stores that one)
// One block per (ky, kx tile): the 16 x nzg values of the tile are gathered as 64-B row pieces (16 kx of one kz) into
// LDS and leave as the tile's contiguous run in destination order.  (Set-up only.  In a rocprof table the first launch of
// this kernel is charged ~58 ms — it is the first dispatch out of this file's code object; the others take ~56 us.)
__global__ __launch_bounds__(256) void k_import_reduced(float* __restrict__ dst, const float* __restrict__ src, uint32_t nxc,
                                                        uint32_t nxm, uint32_t P, uint32_t nyl, uint32_t nzg, uint32_t r1,
                                                        uint32_t vec, uint32_t split)
{
  __shared__ float tile[NLMAX * 1024];
  const uint32_t nt = P / NLMAX, t = blockIdx.x % nt, ky = blockIdx.x / nt;
  const uint32_t n = NLMAX * nzg;
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x)
  {
    const uint32_t kz = i / NLMAX, kx = t * NLMAX + i % NLMAX;
    tile[i] = (kx < nxm) ? src[(static_cast<size_t>(kz) * nyl + ky) * nxc + kx] : 0.f;
  }
  __syncthreads();
  float* __restrict__ out = dst + static_cast<size_t>(blockIdx.x) * n;
  for (uint32_t p = threadIdx.x; p < n; p += blockDim.x)
  {
    uint32_t r = p;
    const uint32_t v  = r % vec; r /= vec;
    const uint32_t c  = r % NLMAX; r /= NLMAX;
    const uint32_t j  = r % r1;
    const uint32_t q  = r / r1;
    const uint32_t ri = q * vec + v; // position in the thread's run
    const uint32_t kz = split ? 2u * j + (ri & 1u) + 2u * r1 * (ri >> 1) : j + r1 * ri;
    out[p] = tile[kz * NLMAX + c];
  }
}
// the side column (kx = nxc - 1) in the same per-thread run layout, its 16-wide tiles running over ky:
// dst[ky tile][q][j][c][V] <- src[kz][ky = 16 * tile + c][nxc - 1]
__global__ void k_import_reduced_side(float* __restrict__ dst, const float* __restrict__ src, uint32_t nxc, uint32_t nyl,
                                      uint32_t nzg, size_t total, uint32_t r1, uint32_t vec, uint32_t split)
{
  const uint32_t nq = nzg / (r1 * vec);
  for (size_t e = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; e < total;
       e += static_cast<size_t>(gridDim.x) * blockDim.x)
  {
    size_t         r  = e;
    const uint32_t r4 = static_cast<uint32_t>(r % vec); r /= vec;
    const uint32_t c  = static_cast<uint32_t>(r % NLMAX); r /= NLMAX;
    const uint32_t j  = static_cast<uint32_t>(r % r1); r /= r1;
    const uint32_t q  = static_cast<uint32_t>(r % nq); r /= nq;
    const uint32_t ky = static_cast<uint32_t>(r) * NLMAX + c;
    const uint32_t ri = q * vec + r4;
    const uint32_t kz = split ? 2u * j + (ri & 1u) + 2u * r1 * (ri >> 1) : j + r1 * ri;
    dst[e] = (ky < nyl) ? src[(static_cast<size_t>(kz) * nyl + ky) * nxc + (nxc - 1u)] : 0.f;
  }
}

// ---- tuning probes (kw_fused_probe): the memory pattern of a line pass without its arithmetic ---------------------
__global__ void k_probe_copy4(float4* __restrict__ p, size_t n4)
{
  for (size_t e = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; e < n4; e += static_cast<size_t>(gridDim.x) * blockDim.x)
  {
    float4 v = p[e];
    v.x += 1.f;
    p[e] = v;
  }
}
// LEVEL 0: tile loads + tile stores only; 1: plus the LDS exchange (no DFTs)
template<int L, int LEVEL> __global__ __launch_bounds__(Geo<L>::THREADS) void k_probe_tile(PassArgs a)
{
  using G = Geo<L>;
  constexpr int R1 = G::R1, R2 = G::R2;
  __shared__ float2 lds[G::LDSB];
  const int      c   = threadIdx.x % G::NL;
  const int      j   = threadIdx.x / G::NL;
  const uint32_t kx  = blockIdx.x * G::NL + c;
  const uint32_t kxl = min(kx, a.nxc - 1u);
  const uint32_t z   = blockIdx.y;
  float2* __restrict__ S = a.out[blockIdx.z];
  float2 v[R1], w[R2];
  const uint32_t b = (z * a.ain.zmul + j * a.ain.estride) * a.P;
#pragma unroll
  for (int n1 = 0; n1 < R1; n1++) v[n1] = S[b + kxl + n1 * (R2 * a.ain.estride * a.P)];
  if (LEVEL >= 1)
  {
#pragma unroll
    for (int k1 = 0; k1 < R1; k1++) lds[k1 * G::SF + j * G::NL + c] = v[k1];
    lds_barrier();
#pragma unroll
    for (int n2 = 0; n2 < R2; n2++) w[n2] = lds[j * G::SF + n2 * G::NL + c];
  }
  else
  {
#pragma unroll
    for (int n2 = 0; n2 < R2; n2++) w[n2] = make_float2(v[n2 % R1].x + 1.f, v[n2 % R1].y);
  }
  if (kx < a.nxc)
  {
#pragma unroll
    for (int k2 = 0; k2 < R2; k2++) S[b + kx + k2 * (R1 * a.ain.estride * a.P)] = w[k2];
  }
}

// tile loads + stores with 256-B row segments (two complex = one float4 per lane, 16 lanes per row, 32 columns per tile)
template<int L> __global__ __launch_bounds__(256) void k_probe_tile_wide(PassArgs a)
{
  constexpr int R = 16;                    // rows per thread
  const int      c   = threadIdx.x % 16;   // float4 column within the 32-column tile
  const int      j   = threadIdx.x / 16;
  const uint32_t kx  = blockIdx.x * 32 + 2 * c;
  const uint32_t kxl = min(kx, (a.P - 2u));
  const uint32_t z   = blockIdx.y;
  float4* __restrict__ S = reinterpret_cast<float4*>(a.out[0]);
  const uint32_t b = ((z * a.ain.zmul + j * a.ain.estride) * a.P + kxl) / 2;
  const uint32_t step = (R * a.ain.estride * a.P) / 2;
  float4 v[R];
#pragma unroll
  for (int n1 = 0; n1 < R; n1++) v[n1] = S[b + n1 * step];
  if (kx < a.nxc)
  {
#pragma unroll
    for (int n1 = 0; n1 < R; n1++) { v[n1].x += 1.f; S[b + n1 * step] = v[n1]; }
  }
}

// tile loads + stores of lines of 16*R elements, 256 threads, VEC complex per lane (16 lanes per row: 128-B or 256-B
// row segments); XCD: blocks that follow each other in the logical tile order run on the same XCD (same L2 / TLB)
template<int R, int VEC, bool XCD> __global__ __launch_bounds__(256) void k_probe_tile_rt(PassArgs a)
{
  typedef float vf __attribute__((ext_vector_type(2 * VEC)));
  uint32_t bx = blockIdx.x, by = blockIdx.y;
  if (XCD)
  {
    const uint32_t nb = gridDim.x * gridDim.y, b = by * gridDim.x + bx;
    const uint32_t l = (b % 8u) * (nb / 8u) + b / 8u; // nb % 8 == 0 checked by the host
    bx = l % gridDim.x;
    by = l / gridDim.x;
  }
  const int      c   = threadIdx.x % 16;
  const int      j   = threadIdx.x / 16;
  const uint32_t kx  = bx * (16 * VEC) + VEC * c;
  const uint32_t kxl = min(kx, a.P - VEC);
  vf* __restrict__ S = reinterpret_cast<vf*>(a.out[0]);
  const uint32_t b0   = ((by * a.ain.zmul + j * a.ain.estride) * a.P + kxl) / VEC;
  const uint32_t step = (16 * a.ain.estride * a.P) / VEC;
  vf v[R];
#pragma unroll
  for (int n1 = 0; n1 < R; n1++) v[n1] = S[b0 + n1 * step];
  if (kx < a.nxc)
  {
#pragma unroll
    for (int n1 = 0; n1 < R; n1++) { v[n1].x += 1.f; S[b0 + n1 * step] = v[n1]; }
  }
}

// memory pattern of k_xinv<velocity, chain> without its arithmetic: per block 32 spectrum rows in, 32 rows of two real
// arrays in (float4), one real array out, 32 spectrum rows out
template<int L> __global__ __launch_bounds__(GeoX<L>::THREADS) void k_probe_xinv(XinvArgs a)
{
  using G = GeoX<L>;
  constexpr int HALF = L / 2 + 1, Q4 = L / 4, NQ = (2 * G::NL * Q4) / G::THREADS;
  const uint32_t comp = blockIdx.y;
  const uint32_t tile_row0 = blockIdx.x * G::NL * 2;
  const float2* __restrict__ src = a.in[comp];
  float2 acc = make_float2(0.f, 0.f);
  for (int e = threadIdx.x; e < G::NL * HALF; e += G::THREADS)
  {
    const int cc = e / HALF, k = e - cc * HALF;
    const uint32_t r = tile_row0 + 2 * cc;
    const float2 A = src[r * a.P + k], B = src[(r + 1) * a.P + k];
    acc.x += A.x + B.x; acc.y += A.y + B.y;
  }
  float4 keep[NQ];
#pragma unroll
  for (int q = 0; q < NQ; q++)
  {
    const uint32_t i = blockIdx.x * (2 * G::NL * L) + 4u * (threadIdx.x + q * G::THREADS);
    const float4 u = ld4(a.out[comp] + i), d = ld4(a.m0[comp] + i);
    keep[q] = make_float4(u.x + d.x + acc.x, u.y + d.y, u.z + d.z, u.w + d.w + acc.y);
    st4(a.out[comp] + i, keep[q]);
  }
  float2* __restrict__ dst = a.fout[comp];
  for (int e = threadIdx.x; e < G::NL * HALF; e += G::THREADS)
  {
    const int cc = e / HALF, k = e - cc * HALF;
    const uint32_t r = tile_row0 + 2 * cc;
    dst[r * a.P + k]       = make_float2(keep[0].x, acc.y);
    dst[(r + 1) * a.P + k] = make_float2(keep[NQ - 1].y, acc.x);
  }
}

// ---- host side ------------------------------------------------------------------------------------------------------
// line lengths with a two-factor register decomposition L = R1 * R2, R1, R2 in {4 ... 32} with at most one odd prime
// power (3, 9, 27, 5, 25, 7) each: 2^m, 3 * 2^m, 9 * 2^m, 27 * 2^m, 81 * 4, 5 * 2^m, 15 * 2^m, 25 * 2^m, 75 * 2^m, 125 * 4,
// 45 * 2^m (180, 360), 135 * 4 (540), 7 * 2^m (112 ... 896), 21 * 2^m (168, 336, 672), 35 * 2^m (140, 280, 560), 49 * 2^m
// (196, 392, 784), 63 * 2^m (252, 504), 105 * 2^m (420, 840), 45 * 16 (720), 225 * 4 (900), 15 * 64 (960: 30 x 32, the
// 15 * 2^m register DFTs), 175 * 4 (700), 189 * 4 (756), 25 * 32 (800), 27 * 32 (864); the x kernels need L % 4 == 0 (where
// L / 2 is no multiple of the threads per line their float4 tile ends in a partial round: QPART in k_xinv)
#ifdef KW_FUSED_ONLY /* tuning builds: one line length only (-DKW_FUSED_ONLY=256), compiles in seconds */
#define KW_FUSED_LENGTHS(X) X(KW_FUSED_ONLY)
#else
#define KW_FUSED_LENGTHS_SHORT(X) X(16) X(32) X(48) X(64) X(72) X(80) X(96) X(100) X(108) X(112) X(120) X(128) X(140)  \
  X(144) X(160) X(168) X(180) X(192) X(196) X(200) X(216) X(224) X(240) X(252) X(256) X(280) X(288) X(300) X(320) X(324)  \
  X(336) X(360) X(384) X(392) X(400) X(420)
#define KW_FUSED_LENGTHS_LONG(X) X(432) X(448) X(480) X(500) X(504) X(512) X(540) X(560) X(576) X(600) X(640) X(648)     \
  X(672) X(700) X(720) X(756) X(768) X(784) X(800) X(840) X(864) X(896) X(900) X(960) X(1024)
#if KW_FUSED_TU == 1 || KW_FUSED_TU == 5
#define KW_FUSED_LENGTHS(X) KW_FUSED_LENGTHS_SHORT(X)
#elif KW_FUSED_TU == 7 || KW_FUSED_TU == 8
#define KW_FUSED_LENGTHS(X) KW_FUSED_LENGTHS_LONG(X)
#else
#define KW_FUSED_LENGTHS(X) KW_FUSED_LENGTHS_SHORT(X) KW_FUSED_LENGTHS_LONG(X)
#endif
#endif
bool supported_len(uint32_t n)
{
#define X(LEN) if (n == LEN) return true;
  KW_FUSED_LENGTHS(X)
#undef X
  return false;
}

// dispatch on a runtime line length; the call sites define the per-length launch macro under the name M
#define KW_LEN_CASE(LEN) case LEN: M(LEN); break;
#define KW_LEN_SWITCH(len, MACRO)                                                                                      \
  switch (len)                                                                                                         \
  {                                                                                                                    \
    KW_FUSED_LENGTHS(KW_LEN_CASE)                                                                                      \
    default: kw_set_error("fused pipeline: unsupported length %u", (unsigned)(len)); return KW_ERR_INVALID;            \
  }

#define LAUNCH(kernel, grid, block, ...)                                                                               \
  do {                                                                                                                 \
    hipLaunchKernelGGL(kernel, grid, block, 0, ctx->stream, __VA_ARGS__);                                              \
    KW_LAUNCH_CHECK();                                                                                                 \
  } while (0)

#define KW_FUSED_READY(ctx)                                                                                            \
  do {                                                                                                                 \
    KW_CHECK_CONSTS(ctx);                                                                                              \
    if (!(ctx)->fused.ready) { kw_set_error("%s: kw_fused_create has not been called", __func__); return KW_ERR_STATE; } \
  } while (0)

#define KW_TRY(call) KW_TRY_STATUS(call)

#if KW_FUSED_TU == 0
kw_status launch_xfwd(kw_ctx* ctx, int narr, const float* const* in, float2* const* out)
{
  const kw_constants& c = ctx->c;
  static const char* const names[3] = { "k_xfwd[1]", "k_xfwd[2]", "k_xfwd[3]" };
  KW_PROF(ctx, names[narr - 1]);
  XfwdArgs a{};
  for (int i = 0; i < narr; i++) { a.in[i] = in[i]; a.out[i] = out[i]; }
  a.tw = ctx->fused.tw[0];
  a.nx = c.nx;
  a.P  = ctx->fused.P;
  a.side_off = ctx->fused.side_off;
  a.nrows = c.ny * c.nz;
  const uint32_t rows_per_tile = 2u * static_cast<uint32_t>(nl_x(c.nx)), full = a.nrows / rows_per_tile;
  if (full > 0)
  {
    const dim3 grid(full, narr, 1);
#define M(LEN) LAUNCH((k_xfwd<LEN, false>), grid, dim3(GeoX<LEN>::THREADS), a)
    KW_LEN_SWITCH(c.nx, M)
#undef M
  }
  if (a.nrows % rows_per_tile != 0)
  { // the partial last tile, masked
    a.tile0 = full;
    const dim3 grid(1, narr, 1);
#define M(LEN) if constexpr (!has_partial_x_tiles(LEN)) KW_NO_TAIL(LEN) else LAUNCH((k_xfwd<LEN, true>), grid, dim3(GeoX<LEN>::THREADS), a)
    KW_LEN_SWITCH(c.nx, M)
#undef M
  }
  return KW_OK;
}

// y-pass.  pack_out / pack_in select the packed (per-peer-chunk) row layout on that side; with one rank both layouts
// coincide and the pass may run in place.
// mul: optional per-array factor (see PassArgs).  ordered: the arrays must be taken in list order by one block (a later
// one overwrites, in place, the input of an earlier one) — true for the pressure-gradient pair below.
kw_status launch_ypass(kw_ctx* ctx, int dir, int narr, float2* const* in, float2* const* out, bool pack_in, bool pack_out,
                       uint32_t z0 = 0, uint32_t nzc = 0, const float2* const* mul = nullptr, bool ordered = false)
{
  const kw_constants& c = ctx->c;
  const auto& f = ctx->fused;
  if (ordered && narr > 1 && c.ny == 512 && f.split512)
  { // one array per block in these kernels: order by launch instead
    KW_TRY(launch_ypass(ctx, dir, 1, in, out, pack_in, pack_out, z0, nzc, mul, false));
    return launch_ypass(ctx, dir, narr - 1, in + 1, out + 1, pack_in, pack_out, z0, nzc, mul ? mul + 1 : nullptr, true);
  }
  static const char* const names[3][3] = { { "k_ypass_fwd[1]", "k_ypass_fwd[2]", "k_ypass_fwd[3]" },
                                           { "k_ypass_inv[1]", "k_ypass_inv[2]", "k_ypass_inv[3]" },
                                           { "k_ypass_inv_pgrad[1]", "k_ypass_inv_pgrad[2]", "k_ypass_inv_pgrad[3]" } };
  KW_PROF(ctx, names[dir < 0 ? 0 : (mul != nullptr ? 2 : 1)][narr - 1]);
  PassArgs a{};
  for (int i = 0; i < narr; i++) { a.in[i] = in[i]; a.out[i] = out[i]; a.mul[i] = mul ? mul[i] : nullptr; }
  a.tw  = f.tw[1];
  a.nxc = f.nxm;
  a.P   = f.P;
  a.PX  = f.PX;
  a.side_off = f.side_off; // (packed sides: the side array travels as per-peer chunks [nz local][nyl] behind the row chunks)
  const uint32_t side_tile = (a.side_off != 0) ? 1u : 0u; // one more tile index: the blocks of the x-Nyquist side array
  const RowAddr natural{0u, 0u, 0u, c.ny, 1u};
  const RowAddr packed{(1u << 20) / f.nyl + 1u, f.nyl, c.nz * f.nyl, f.nyl, 1u};
  a.ain  = pack_in ? packed : natural;
  a.aout = pack_out ? packed : natural;
  a.narr = narr; // each block walks the arrays of the launch (the next one's lines requested before the current transform)
  { // ... unless the launch would not even fill the chip twice (small grids): then one array per block — three times the
    // blocks, a third of each block's life — is worth more than the prefetch (128^3: y-passes of three arrays 15 -> 13 us,
    // step +3 %).  Not for `ordered` lists (in-place hazards).
    const uint32_t blocks = (f.P / nl_yz(c.ny) + side_tile) * (nzc ? nzc : c.nz);
    if (!ordered && narr > 1 && blocks < 8u * static_cast<uint32_t>(ctx->cu_count)) a.narr = 1;
  }
  a.z0   = z0;
  if (c.ny == 512 && f.split512)
  { // 2 x 256 lines: 16-column tiles, one array per block
    a.narr = 1;
    const dim3 g(f.P / NLMAX + side_tile, nzc ? nzc : c.nz, narr), b(Geo<256>::THREADS);
    if (dir < 0) { if (pack_out) LAUNCH((k_ypass_split<512, kFwd, false, true>), g, b, a);
                   else LAUNCH((k_ypass_split<512, kFwd, false, false>), g, b, a); }
    else         { if (pack_in) LAUNCH((k_ypass_split<512, kInv, true, false>), g, b, a);
                   else LAUNCH((k_ypass_split<512, kInv, false, false>), g, b, a); }
    return KW_OK;
  }
  const dim3 grid(f.P / nl_yz(c.ny) + side_tile, nzc ? nzc : c.nz, narr / a.narr);
  // forward: natural in, natural or packed out; inverse: natural or packed in, natural out
#define M(LEN)                                                                                                         \
  if (dir < 0) { if (pack_out) LAUNCH((k_ypass<LEN, kFwd, false, true>), grid, dim3(Geo<LEN>::THREADS), a);           \
                 else LAUNCH((k_ypass<LEN, kFwd, false, false>), grid, dim3(Geo<LEN>::THREADS), a); }                  \
  else         { if (pack_in) LAUNCH((k_ypass<LEN, kInv, true, false>), grid, dim3(Geo<LEN>::THREADS), a);            \
                 else LAUNCH((k_ypass<LEN, kInv, false, false>), grid, dim3(Geo<LEN>::THREADS), a); }
  KW_LEN_SWITCH(c.ny, M)
#undef M
  return KW_OK;
}

// z-fused on the (possibly transposed) spectra: lines of nz_global elements, nyl local rows, stride nyl*P
template<int MODE> kw_status launch_zfused(kw_ctx* ctx, int narr, ZArgs a)
{
  const kw_constants& c = ctx->c;
  const auto& f = ctx->fused;
  static const char* const names[4][3] = { { "k_zfused_pgrad", "k_zfused_pgrad", "k_zfused_pgrad" },
                                           { "k_zfused_vgrad[1]", "k_zfused_vgrad[2]", "k_zfused_vgrad[3]" },
                                           { "k_zfused_absorb[1]", "k_zfused_absorb[2]", "k_zfused_absorb[3]" },
                                           { "k_zfused_source", "k_zfused_source", "k_zfused_source" } };
  KW_PROF(ctx, names[MODE][narr - 1]);
  if (f.pipelined)
  { // the transposed spectra live in r[] (callers name the arrays by their s[] slots)
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++)
      {
        if (a.in[i] == f.s[j]) a.in[i] = f.r[j];
        if (a.out[i] == f.s[j]) a.out[i] = f.r[j];
      }
  }
  a.tw      = f.tw[2];
  a.divider = c.fft_divider;
  a.nxc     = f.nxm;
  // 2-D (Nz == 1): the "z" pass runs along y — the y-derivative is the one along the line, there is no third array
  a.axis_of[0] = 0; a.axis_of[1] = f.two_d ? 2u : 1u; a.axis_of[2] = 2;
  if (f.two_d) a.dd[2] = a.dd[1]; // ... whose derivative vector is ddy
  a.side_off    = f.side_off;
  a.op_side_off = f.side_off; // the imported operators hold the side column's values behind the P * ny * nz of the row tiles
  const uint32_t side_tile = (f.side_off != 0) ? 1u : 0u;
  a.P       = f.slab ? f.PX : f.P; // slab mode: the z-pass works on the exchanged rows in place
  a.Pop     = f.P;
  a.ny      = f.nyl;
  a.nz      = f.nz_global;
  a.ky0     = f.rank * f.nyl;
  a.narr    = narr;
  if (f.nz_global == 512 && f.split512)
  {
    LAUNCH((k_zfused_split<512, MODE>), dim3(f.P / NLMAX + side_tile, f.nyl, narr), dim3(Geo<256>::THREADS), a);
    return KW_OK;
  }
  // smallest grids (fewer blocks than two per CU): one array per block instead of the arrays back to back (64^3: z-fused
  // kernels 10.6 -> 8.8 us, step +4 %; at 128^3, 640 blocks, the in-block walk with its prefetch is still the faster form)
  uint32_t split = 1;
  if ((MODE == Z_VGRAD || MODE == Z_ABSORB) && narr > 1 &&
      (z_tiles(f.P, f.nz_global) + side_tile) * f.nyl < 2u * static_cast<uint32_t>(ctx->cu_count))
  {
    split  = static_cast<uint32_t>(narr);
    a.narr = 1;
  }
  const dim3 grid(z_tiles(f.P, f.nz_global) + side_tile, f.nyl, split);
#define M(LEN) LAUNCH((k_zfused<LEN, MODE>), grid, dim3((Geo<LEN, nl_z(LEN)>::THREADS)), a)
  KW_LEN_SWITCH(f.nz_global, M)
#undef M
  return KW_OK;
}

// main pass: the full tiles go to the pass that holds this epilogue's kernels, the partial last tile of the grid (if any;
// only ever in the last chunk: chunks are whole tiles otherwise) to the pass that holds their masked forms
template<int EPI, bool CHAIN = false, int TERMS = 0>
kw_status launch_xinv(kw_ctx* ctx, int ncomp, XinvArgs a, uint32_t z0 = 0, uint32_t nzc = 0, bool plane = false)
{
  const kw_constants& c = ctx->c;
  static const char* const names[5][2] = { { "k_xinv_store", "k_xinv_store" }, { "k_xinv_velocity", "k_xinv_velocity_chain" },
                                           { "k_xinv_initvel", "k_xinv_initvel" }, { "k_xinv_density", "k_xinv_density_chain" },
                                           { "k_xinv_psum", "k_xinv_psum_chain" } };
  KW_PROF(ctx, names[EPI][CHAIN ? 1 : 0]);
  a.tw = ctx->fused.tw[0];
  a.c  = c;
  a.P  = ctx->fused.P;
  a.side_off = ctx->fused.side_off;
  a.nrows = c.ny * c.nz;
  if (plane) // one block per z-plane; the kernel does the plane's y transforms as well
    return EPI == EPI_DENSITY ? kw_fused_xinv_density_plane(CHAIN ? 1 : 0, TERMS, ctx, ncomp, &a, z0, nzc ? nzc : c.nz)
                              : kw_fused_xinv_other_plane(EPI, CHAIN ? 1 : 0, ctx, ncomp, &a, z0, nzc ? nzc : c.nz);
  const uint32_t rows_per_tile = 2u * static_cast<uint32_t>(nl_x(c.nx));
  const uint32_t rows = c.ny * (nzc ? nzc : c.nz), full = rows / rows_per_tile;
  const uint32_t tile0 = z0 * c.ny / rows_per_tile; // chunked launches start on tile boundaries (plane_local_tail)
  if (full > 0)
    KW_TRY(EPI == EPI_DENSITY ? kw_fused_xinv_density(CHAIN ? 1 : 0, TERMS, ctx, ncomp, &a, tile0, full)
                              : kw_fused_xinv_other(EPI, CHAIN ? 1 : 0, ctx, ncomp, &a, tile0, full));
  if (rows % rows_per_tile != 0)
    KW_TRY(EPI == EPI_DENSITY ? kw_fused_xinv_density_tail(CHAIN ? 1 : 0, TERMS, ctx, ncomp, &a, tile0 + full, 1)
                              : kw_fused_xinv_other_tail(EPI, CHAIN ? 1 : 0, ctx, ncomp, &a, tile0 + full, 1));
  return KW_OK;
}
#else
template<int EPI, bool CHAIN, int TERMS, bool TAIL>
kw_status launch_xinv_impl(kw_ctx* ctx, int ncomp, XinvArgs a, uint32_t tile0, uint32_t ntiles)
{
  a.tile0 = tile0;
  const dim3 grid(ntiles, ncomp, 1);
#define M(LEN) if constexpr (TAIL && !has_partial_x_tiles(LEN)) KW_NO_TAIL(LEN) else LAUNCH((k_xinv<LEN, EPI, CHAIN, TERMS, TAIL>), grid, dim3(GeoX<LEN>::THREADS), a)
  KW_LEN_SWITCH(ctx->c.nx, M)
#undef M
  return KW_OK;
}
#if KW_FUSED_TU == 1 || KW_FUSED_TU == 2
template<int EPI, bool CHAIN, int TERMS>
kw_status launch_xinv_plane_impl(kw_ctx* ctx, int ncomp, XinvArgs a, uint32_t plane0, uint32_t nplanes)
{
  a.tile0 = plane0;
  const dim3 grid(nplanes, ncomp, 1);
#define MP(LEN) LAUNCH((k_xinv<LEN, EPI, CHAIN, TERMS, false, true>), grid, dim3((Geo<LEN, LEN / 2>::THREADS)), a)
  switch (ctx->c.nx)
  {
#if !defined(KW_FUSED_ONLY) || KW_FUSED_ONLY == 32
    case 32: MP(32); break;
#endif
#if !defined(KW_FUSED_ONLY) || KW_FUSED_ONLY == 64
    case 64: MP(64); break;
#endif
    default: kw_set_error("fused pipeline: no whole-plane kernels for rows of %u", ctx->c.nx); return KW_ERR_INVALID;
  }
#undef MP
  return KW_OK;
}
#endif
#endif

#if KW_FUSED_TU == 0

// split-phase exchange of scratch array `slot`: start is ordered after the work enqueued so far; wait orders later
// work after its completion.  Without asynchronous callbacks, start is the blocking exchange and wait a no-op.
kw_status xstart_bytes(kw_ctx* ctx, int slot, void* send, void* recv, size_t bytes_per_peer)
{
  const auto& f = ctx->fused;
  if (f.exchange_piece != nullptr)
  {
    const int rc = f.exchange_piece(f.exchange_user, send, recv, bytes_per_peer, 0, bytes_per_peer, slot);
    if (rc != 0) { kw_set_error("slab exchange: the caller's piece callback failed (status %d)", rc); return KW_ERR_COMM; }
    return KW_OK;
  }
  if (f.exchange_start == nullptr && f.exchange == nullptr) return kw_comm_exchange_start(ctx, slot, send, recv, bytes_per_peer);
  const int rc = (f.exchange_start != nullptr) ? f.exchange_start(f.exchange_user, send, recv, bytes_per_peer, slot)
                                               : f.exchange(f.exchange_user, send, recv, bytes_per_peer);
  if (rc != 0) { kw_set_error("slab exchange: the caller's exchange callback failed (status %d)", rc); return KW_ERR_COMM; }
  return KW_OK;
}
kw_status xwait_one(kw_ctx* ctx, int slot);
kw_status xstart(kw_ctx* ctx, int slot, float2* send, float2* recv)
{ // one spectral scratch array: nz local planes x nyl rows per peer, and the same of the x-Nyquist side array behind them
  const auto& f = ctx->fused;
  const size_t rows = static_cast<size_t>(ctx->c.nz) * f.nyl;
  if (f.side_off == 0) return xstart_bytes(ctx, slot, send, recv, rows * f.PX * sizeof(float2));
  if (f.exchange_start == nullptr && f.exchange == nullptr) // the library's exchange: both pieces in one RCCL group
    return kw_comm_exchange_start2(ctx, slot, send, recv, rows * f.PX * sizeof(float2), send + f.side_off, recv + f.side_off,
                                   rows * sizeof(float2));
  KW_TRY(xstart_bytes(ctx, slot, send, recv, rows * f.PX * sizeof(float2)));
  return xstart_bytes(ctx, slot + KW_COMM_SLOTS, send + f.side_off, recv + f.side_off, rows * sizeof(float2));
}
kw_status xwait(kw_ctx* ctx, int slot)
{
  const auto& f = ctx->fused;
  KW_TRY(xwait_one(ctx, slot));
  if (f.side_off != 0 && f.exchange_start != nullptr) KW_TRY(xwait_one(ctx, slot + KW_COMM_SLOTS)); // callback pair: the side piece
  return KW_OK;
}
kw_status xwait_one(kw_ctx* ctx, int slot)
{
  const auto& f = ctx->fused;
  if (f.exchange_piece == nullptr && f.exchange_start == nullptr && f.exchange == nullptr) return kw_comm_exchange_wait(ctx, slot);
  if (f.exchange_piece != nullptr ? f.exchange_wait != nullptr : f.exchange_start != nullptr)
  {
    const int rc = f.exchange_wait(f.exchange_user, slot);
    if (rc != 0) { kw_set_error("slab exchange: the caller's wait callback failed (status %d)", rc); return KW_ERR_COMM; }
  }
  return KW_OK;
}

// ---- pipelined slab schedule (fused_plan::pipelined) -------------------------------------------------------------
// Buffer roles: s[] plane layout [nz local][ny][P] (x / y passes), t[] per-peer chunks [peer][nz local][nyl][PX] (packed
// y-pass output = forward send; backward receive = packed y-inverse input), r[] transposed [nz global][nyl][PX] (forward
// receive = z-pass in / out = backward send).  An exchange moves the planes of one chunk (or of all chunks) of one
// array: per peer q the bytes at (q * nz local + z0) * nyl * PX of both layouts, plus the same planes of the side array.
enum { X_FWD = 0, X_BACK = 1 };
constexpr int KW_ZSHIFT_SLOT = KW_COMM_SLOTS - 1;
inline int pslot(int dir, int a, int c) { return (dir * 3 + a) * KW_XCHUNKS_MAX + c; }

// planes [z0, z0 + nzp) of the arrays arrs[0..na) in one exchange
kw_status xpieces_start(kw_ctx* ctx, int slot, int dir, const int* arrs, int na, uint32_t z0, uint32_t nzp)
{
  const auto& f = ctx->fused;
  const size_t all = static_cast<size_t>(ctx->c.nz) * f.nyl, first = static_cast<size_t>(z0) * f.nyl, rows = static_cast<size_t>(nzp) * f.nyl;
  const size_t rb = f.PX * sizeof(float2);
  kw_comm_piece pc[6];
  int n = 0;
  for (int i = 0; i < na; i++)
  {
    float2* send = (dir == X_FWD) ? f.t[arrs[i]] : f.r[arrs[i]];
    float2* recv = (dir == X_FWD) ? f.r[arrs[i]] : f.t[arrs[i]];
    pc[n++] = kw_comm_piece{ send, recv, all * rb, first * rb, rows * rb };
    if (f.side_off != 0)
      pc[n++] = kw_comm_piece{ send + f.side_off, recv + f.side_off, all * sizeof(float2), first * sizeof(float2), rows * sizeof(float2) };
  }
  if (f.exchange_piece == nullptr) return kw_comm_exchange_start_pieces(ctx, slot, pc, n);
  for (int i = 0; i < n; i++)
  {
    const int rc = f.exchange_piece(f.exchange_user, const_cast<void*>(pc[i].send), pc[i].recv, pc[i].stride, pc[i].offset,
                                    pc[i].bytes, slot + i * KW_COMM_SLOTS);
    if (rc != 0) { kw_set_error("slab exchange: the caller's piece callback failed (status %d)", rc); return KW_ERR_COMM; }
  }
  return KW_OK;
}
kw_status xpieces_wait(kw_ctx* ctx, int slot, int npieces)
{
  const auto& f = ctx->fused;
  if (f.exchange_piece == nullptr) return kw_comm_exchange_wait(ctx, slot);
  if (f.exchange_wait == nullptr) return KW_OK; // blocking piece callback
  for (int i = 0; i < npieces; i++)
  {
    const int rc = f.exchange_wait(f.exchange_user, slot + i * KW_COMM_SLOTS);
    if (rc != 0) { kw_set_error("slab exchange: the caller's wait callback failed (status %d)", rc); return KW_ERR_COMM; }
  }
  return KW_OK;
}
// chunks [c0, c0 + nc) of the arrays arrs[0..na) in one exchange
kw_status pstart_multi(kw_ctx* ctx, int dir, const int* arrs, int na, int c0, int nc)
{
  auto& f = ctx->fused;
  const uint32_t nzc = ctx->c.nz / f.xchunks;
  const int slot = pslot(dir, arrs[0], c0);
  KW_TRY(xpieces_start(ctx, slot, dir, arrs, na, c0 * nzc, nc * nzc));
  for (int i = 0; i < na; i++)
    for (int c = c0; c < c0 + nc; c++) f.xslot[dir][arrs[i]][c] = static_cast<int8_t>(slot);
  f.slot_waited[slot] = false;
  f.slot_pieces[slot] = static_cast<int8_t>(na * (f.side_off != 0 ? 2 : 1));
  return KW_OK;
}
kw_status pstart(kw_ctx* ctx, int dir, int a, int c0, int nc) { return pstart_multi(ctx, dir, &a, 1, c0, nc); }
// the arrays [a0, a0 + na): one exchange when batching, else one each
kw_status pstart_arrays(kw_ctx* ctx, int dir, int a0, int na, int c0, int nc)
{
  const int arrs[3] = { a0, a0 + 1, a0 + 2 };
  if (ctx->fused.xbatch) return pstart_multi(ctx, dir, arrs, na, c0, nc);
  for (int i = 0; i < na; i++) KW_TRY(pstart(ctx, dir, a0 + i, c0, nc));
  return KW_OK;
}
kw_status pwait(kw_ctx* ctx, int dir, int a, int c)
{
  auto& f = ctx->fused;
  const int slot = f.xslot[dir][a][c];
  if (slot < 0) { kw_set_error("slab pipeline: chunk %d of array %d awaited before its exchange was started", c, a); return KW_ERR_STATE; }
  if (!f.slot_waited[slot])
  {
    KW_TRY(xpieces_wait(ctx, slot, f.slot_pieces[slot]));
    f.slot_waited[slot] = true;
  }
  return KW_OK;
}
kw_status pwait_all(kw_ctx* ctx, int dir, int a)
{
  for (uint32_t c = 0; c < ctx->fused.xchunks; c++) KW_TRY(pwait(ctx, dir, a, static_cast<int>(c)));
  return KW_OK;
}
// forward exchanges a producer started for a consumer that never came (e.g. a run that ended in between): order the
// buffers' reuse after them
kw_status drain_ahead(kw_ctx* ctx)
{
  auto& f = ctx->fused;
  for (int a = 0; a < f.fwd_ahead; a++) KW_TRY(pwait_all(ctx, X_FWD, a));
  f.fwd_ahead = 0;
  return KW_OK;
}
// forward half up to "exchanges started": from real arrays (x-forward first), from chained x-spectra in s[], or nothing
// to do when the producer's tail already sent them chunk by chunk
kw_status pforward_start(kw_ctx* ctx, int narr, const float* const* in)
{
  auto& f = ctx->fused;
  if (in == nullptr && f.fwd_ahead == narr) return KW_OK;
  if (f.fwd_ahead != 0) KW_TRY(drain_ahead(ctx));
  if (f.xbatch)
  { // small messages: multi-array kernels, one exchange for the lot
    if (in != nullptr) KW_TRY(launch_xfwd(ctx, narr, in, f.s));
    KW_TRY(launch_ypass(ctx, -1, narr, f.s, f.t, false, true));
    return pstart_arrays(ctx, X_FWD, 0, narr, 0, static_cast<int>(f.xchunks));
  }
  for (int a = 0; a < narr; a++)
  {
    if (in != nullptr) KW_TRY(launch_xfwd(ctx, 1, in + a, f.s + a));
    KW_TRY(launch_ypass(ctx, -1, 1, f.s + a, f.t + a, false, true));
    KW_TRY(pstart(ctx, X_FWD, a, 0, static_cast<int>(f.xchunks)));
  }
  return KW_OK;
}
// forward + z-pass per array + backward exchanges started chunk-major (chunk 0 of every array first: the tail's first
// chunk is complete after narr chunk transfers, the rest travel while it computes)
template<int MODE> kw_status pslab_chain(kw_ctx* ctx, int narr, const float* const* in, ZArgs z)
{
  auto& f = ctx->fused;
  const int C = static_cast<int>(f.xchunks);
  KW_TRY(pforward_start(ctx, narr, in));
  if (f.xbatch)
  {
    for (int a = 0; a < narr; a++) KW_TRY(pwait_all(ctx, X_FWD, a));
    f.fwd_ahead = 0;
    z.arr0 = 0;
    KW_TRY(launch_zfused<MODE>(ctx, narr, z));
    return pstart_arrays(ctx, X_BACK, 0, narr, 0, C);
  }
  for (int a = 0; a < narr; a++)
  {
    KW_TRY(pwait_all(ctx, X_FWD, a));
    z.arr0 = a;
    KW_TRY(launch_zfused<MODE>(ctx, 1, z));
    KW_TRY(pstart(ctx, X_BACK, a, 0, 1));
  }
  f.fwd_ahead = 0;
  for (int c = 1; c < C; c++)
    for (int a = 0; a < narr; a++) KW_TRY(pstart(ctx, X_BACK, a, c, 1));
  return KW_OK;
}
// plane-local tail per chunk: backward receive -> y-inverse -> x-inverse + epilogue -> (chained) y-forward -> forward send
template<int EPI, bool CHAIN, int TERMS = 0>
kw_status pslab_tail(kw_ctx* ctx, int narr, int ncomp, const XinvArgs& x, int nchain)
{
  auto& f = ctx->fused;
  const uint32_t C = f.xchunks, nzc = ctx->c.nz / C;
  for (uint32_t c = 0; c < C; c++)
  {
    if (f.xbatch || narr == 1)
    {
      for (int a = 0; a < narr; a++) KW_TRY(pwait(ctx, X_BACK, a, static_cast<int>(c)));
      KW_TRY(launch_ypass(ctx, +1, narr, f.t, f.s, true, false, c * nzc, nzc));
    }
    else
    { // every array's y-inverse as soon as that array is back: only the last one stays between the wire and the epilogue
      for (int a = 0; a < narr; a++)
      {
        KW_TRY(pwait(ctx, X_BACK, a, static_cast<int>(c)));
        KW_TRY(launch_ypass(ctx, +1, 1, f.t + a, f.s + a, true, false, c * nzc, nzc));
      }
    }
    KW_TRY((launch_xinv<EPI, CHAIN, TERMS>(ctx, ncomp, x, c * nzc, nzc)));
    if (CHAIN)
    {
      KW_TRY(launch_ypass(ctx, -1, nchain, f.s, f.t, false, true, c * nzc, nzc));
      KW_TRY(pstart_arrays(ctx, X_FWD, 0, nchain, static_cast<int>(c), 1));
    }
  }
  if (CHAIN) f.fwd_ahead = nchain;
  return KW_OK;
}

// Slab mode, narr independent arrays (velocity gradient, absorption, source scaling): software-pipelined per array so
// that the all-to-all of one array is in flight while the y / z passes of the others run.
template<int MODE> kw_status slab_chain(kw_ctx* ctx, int narr, const float* const* in, ZArgs z)
{
  auto& f = ctx->fused;
  if (f.pipelined) return pslab_chain<MODE>(ctx, narr, in, z); // (the caller's tail is pslab_tail)
  for (int a = 0; a < narr; a++)
  {
    if (in != nullptr) KW_TRY(launch_xfwd(ctx, 1, in + a, f.s + a));
    KW_TRY(launch_ypass(ctx, -1, 1, f.s + a, f.t + a, false, true));
    KW_TRY(xstart(ctx, a, f.t[a], f.s[a]));
  }
  for (int a = 0; a < narr; a++)
  {
    KW_TRY(xwait(ctx, a));
    z.arr0 = a;
    KW_TRY(launch_zfused<MODE>(ctx, 1, z));
    KW_TRY(xstart(ctx, a, f.s[a], f.t[a]));
  }
  for (int a = 0; a < narr; a++)
  {
    KW_TRY(xwait(ctx, a));
    KW_TRY(launch_ypass(ctx, +1, 1, f.t + a, f.s + a, true, false));
  }
  return KW_OK;
}

// forward half of a 3-D transform for narr real arrays: x-forward, y-forward, transpose.  Spectra end up in S[]
// ([nz][ny][P] with one rank, transposed [nz_global][nyl][P] in slab mode).
kw_status forward_xy(kw_ctx* ctx, int narr, const float* const* in, int s0 = 0)
{
  auto& f = ctx->fused;
  const int y_done = f.y_done;
  f.y_done = 0;
  if (f.slab && f.pipelined)
  { // (s0 == 0 on slabs) — ends with the transposed spectra in r[]
    KW_TRY(pforward_start(ctx, narr, in));
    for (int i = 0; i < narr; i++) KW_TRY(pwait_all(ctx, X_FWD, i));
    f.fwd_ahead = 0;
    return KW_OK;
  }
  if (in != nullptr) KW_TRY(launch_xfwd(ctx, narr, in, f.s + s0)); // nullptr: x-spectra were chained into S[] already
  else if (y_done >= s0 + narr) return KW_OK;                      // ... and so was their y-pass (chunked producer)
  if (f.two_d) return KW_OK;                                        // the y transform is inside the fused pass
  if (!f.slab) return launch_ypass(ctx, -1, narr, f.s + s0, f.s + s0, false, false);
  KW_TRY(launch_ypass(ctx, -1, narr, f.s + s0, f.t + s0, false, true));
  for (int i = 0; i < narr; i++) KW_TRY(xstart(ctx, s0 + i, f.t[s0 + i], f.s[s0 + i]));
  for (int i = 0; i < narr; i++) KW_TRY(xwait(ctx, s0 + i));
  return KW_OK;
}

// Single rank: the plane-local tail of a stage — y-inverse, x-inverse + epilogue and, when the epilogue chains the
// x-spectra of its results into S[0..nchain), their forward y-pass — runs per chunk of planes, so that what one kernel
// writes is still in the Infinity Cache when the next one reads it.
template<int EPI, bool CHAIN, int TERMS = 0>
kw_status plane_local_tail(kw_ctx* ctx, int narr, int ncomp, const XinvArgs& x, int nchain,
                                                         float2* const* yin = nullptr, float2* const* yout = nullptr,
                                                         const float2* const* ymul = nullptr)
{
  auto& f = ctx->fused;
  const kw_constants& c = ctx->c;
  if (f.plane)
  { // small grids: one launch — every block takes a z-plane and does its y transforms around the x kernels' work
    XinvArgs xp = x;
    for (int i = 0; i < narr; i++)
    { // y-pass i reads yin[i] (times ymul[i]) and leaves its result where some x-inverse reads it: that one takes both over
      const float2* from = yin ? yin[i] : f.s[i];
      const float2* to   = yout ? yout[i] : f.s[i];
      for (int k = 0; k < 3; k++)
        if (x.in[k] == to) { xp.in[k] = from; xp.ymul[k] = ymul ? ymul[i] : nullptr; }
    }
    KW_TRY((launch_xinv<EPI, CHAIN, TERMS>(ctx, ncomp, xp, 0, 0, true)));
    if (CHAIN) f.y_done = nchain;
    return KW_OK;
  }
  uint32_t nch = static_cast<uint32_t>(ctx->tuning.tail_chunks > 0 ? ctx->tuning.tail_chunks : 1);
  while (nch > 1 && (c.nz % nch != 0 || (c.nz / nch * c.ny) % (2 * nl_x(c.nx)) != 0)) nch--;
  const uint32_t nzc = c.nz / nch;
  for (uint32_t ch = 0; ch < nch; ch++)
  {
    KW_TRY(launch_ypass(ctx, +1, narr, yin ? yin : f.s, yout ? yout : f.s, false, false, ch * nzc, nzc, ymul, ymul != nullptr));
    KW_TRY((launch_xinv<EPI, CHAIN, TERMS>(ctx, ncomp, x, ch * nzc, nzc)));
    if (CHAIN) KW_TRY(launch_ypass(ctx, -1, nchain, f.s, f.s, false, false, ch * nzc, nzc));
  }
  if (CHAIN) f.y_done = nchain;
  return KW_OK;
}

// Way back of the pressure gradient after launch_zfused<Z_PGRAD>: S[0] = Q = F_z^-1{kappa F{p}}, S[2] = G_z (x and y
// still transformed).  d/dx and d/dy share Q: the y-inverse produces F_y^-1{ddy Q} into S[1] and F_y^-1{Q} into S[0]
// (one read of Q), the x-inverse of component 0 applies ddx(kx) to its rows.  Two transposes instead of three in slab
// mode, one array less written by the z-pass and read by the y-pass everywhere.
template<int EPI, bool CHAIN> kw_status gradient_tail(kw_ctx* ctx, XinvArgs x, const float2* ddx, const float2* ddy)
{
  auto& f = ctx->fused;
  x.mulx[0] = ddx;
  if (f.two_d)
  { // 2-D: the fused pass along y left Q in S[0] (d/dx: x ddx(kx) in the x-inverse) and G_y in S[2]; two components
    x.in[1] = f.s[2];
    return launch_xinv<EPI, CHAIN>(ctx, 2, x);
  }
  const float2* mul[3] = { ddy, nullptr, nullptr };
  if (f.slab && f.pipelined && f.xbatch)
  { // small messages: Q and G_z come back in one exchange, one multi-array launch per pass, one exchange forward
    const int back[2] = { 0, 2 };
    KW_TRY(pstart_multi(ctx, X_BACK, back, 2, 0, static_cast<int>(f.xchunks)));
    KW_TRY(pwait_all(ctx, X_BACK, 0));
    KW_TRY(pwait_all(ctx, X_BACK, 2));
    float2* yin[3]  = { f.t[0], f.t[0], f.t[2] };
    float2* yout[3] = { f.s[1], f.s[0], f.s[2] };
    KW_TRY(launch_ypass(ctx, +1, 3, yin, yout, true, false, 0, 0, mul, false));
    x.comp0 = 0;
    KW_TRY((launch_xinv<EPI, CHAIN>(ctx, 3, x)));
    if (CHAIN)
    {
      KW_TRY(launch_ypass(ctx, -1, 3, f.s, f.t, false, true));
      KW_TRY(pstart_arrays(ctx, X_FWD, 0, 3, 0, static_cast<int>(f.xchunks)));
      f.fwd_ahead = 3;
    }
    return KW_OK;
  }
  if (f.slab && f.pipelined)
  { // Q (array 0) and G_z (array 2) come back chunk by chunk; the three components leave again as their rows are done
    const uint32_t C = f.xchunks, nzc = ctx->c.nz / C;
    for (uint32_t ch = 0; ch < C; ch++)
    {
      KW_TRY(pstart(ctx, X_BACK, 0, static_cast<int>(ch), 1));
      KW_TRY(pstart(ctx, X_BACK, 2, static_cast<int>(ch), 1));
    }
    float2* yin[2]  = { f.t[0], f.t[0] };
    float2* yout[2] = { f.s[1], f.s[0] };
    for (uint32_t ch = 0; ch < C; ch++)
    {
      const int c = static_cast<int>(ch);
      KW_TRY(pwait(ctx, X_BACK, 0, c));
      KW_TRY(launch_ypass(ctx, +1, 2, yin, yout, true, false, ch * nzc, nzc, mul, false));
      x.comp0 = 0;
      KW_TRY((launch_xinv<EPI, CHAIN>(ctx, 2, x, ch * nzc, nzc)));
      if (CHAIN)
      {
        KW_TRY(launch_ypass(ctx, -1, 2, f.s, f.t, false, true, ch * nzc, nzc));
        KW_TRY(pstart(ctx, X_FWD, 0, c, 1));
        KW_TRY(pstart(ctx, X_FWD, 1, c, 1));
      }
      KW_TRY(pwait(ctx, X_BACK, 2, c));
      KW_TRY(launch_ypass(ctx, +1, 1, f.t + 2, f.s + 2, true, false, ch * nzc, nzc));
      x.comp0 = 2;
      KW_TRY((launch_xinv<EPI, CHAIN>(ctx, 1, x, ch * nzc, nzc)));
      if (CHAIN)
      {
        KW_TRY(launch_ypass(ctx, -1, 1, f.s + 2, f.t + 2, false, true, ch * nzc, nzc));
        KW_TRY(pstart(ctx, X_FWD, 2, c, 1));
      }
    }
    if (CHAIN) f.fwd_ahead = 3;
    return KW_OK;
  }
  if (f.slab)
  {
    KW_TRY(xstart(ctx, 0, f.s[0], f.t[0]));
    KW_TRY(xstart(ctx, 2, f.s[2], f.t[2]));
    KW_TRY(xwait(ctx, 0));
    float2* yin[2]  = { f.t[0], f.t[0] };
    float2* yout[2] = { f.s[1], f.s[0] };
    KW_TRY(launch_ypass(ctx, +1, 2, yin, yout, true, false, 0, 0, mul, false)); // out of place: no ordering needed
    x.comp0 = 0;
    KW_TRY((launch_xinv<EPI, CHAIN>(ctx, 2, x)));
    KW_TRY(xwait(ctx, 2));
    KW_TRY(launch_ypass(ctx, +1, 1, f.t + 2, f.s + 2, true, false));
    x.comp0 = 2;
    return launch_xinv<EPI, CHAIN>(ctx, 1, x);
  }
  float2* yin[3]  = { f.s[0], f.s[0], f.s[2] };
  float2* yout[3] = { f.s[1], f.s[0], f.s[2] };
  return plane_local_tail<EPI, CHAIN>(ctx, 3, 3, x, CHAIN ? 3 : 0, yin, yout, mul);
}

// inverse half: transpose back, y-inverse; leaves [nz][ny][P] spectra (x still transformed) in S[]
kw_status inverse_y(kw_ctx* ctx, int narr, int s0 = 0)
{
  auto& f = ctx->fused;
  if (f.two_d) return KW_OK;
  if (!f.slab) return launch_ypass(ctx, +1, narr, f.s + s0, f.s + s0, false, false);
  if (f.pipelined)
  { // whole arrays: r[] -> t[] -> y-inverse into s[]
    KW_TRY(pstart_arrays(ctx, X_BACK, s0, narr, 0, static_cast<int>(f.xchunks)));
    for (int i = 0; i < narr; i++) KW_TRY(pwait_all(ctx, X_BACK, s0 + i));
    return launch_ypass(ctx, +1, narr, f.t + s0, f.s + s0, true, false);
  }
  for (int i = 0; i < narr; i++) KW_TRY(xstart(ctx, s0 + i, f.s[s0 + i], f.t[s0 + i]));
  for (int i = 0; i < narr; i++) KW_TRY(xwait(ctx, s0 + i));
  return launch_ypass(ctx, +1, narr, f.t + s0, f.s + s0, true, false);
}

kw_status alloc_scratch(kw_ctx* ctx, void* const s[3], void* const t[3])
{
  auto& f = ctx->fused;
  const kw_constants& c = ctx->c;
  const size_t elems = static_cast<size_t>(f.Palloc) * c.ny * c.nz;
  f.owns_scratch     = (s == nullptr);
  for (int i = 0; i < 3; i++)
  {
    if (f.owns_scratch)
    {
      KW_HIP(hipMalloc(reinterpret_cast<void**>(&f.s[i]), elems * sizeof(float2)));
      if (f.slab) KW_HIP(hipMalloc(reinterpret_cast<void**>(&f.t[i]), elems * sizeof(float2)));
    }
    else
    {
      KW_REQUIRE(s[i] != nullptr && (!f.slab || (t != nullptr && t[i] != nullptr)));
      f.s[i] = static_cast<float2*>(s[i]);
      f.t[i] = (f.slab) ? static_cast<float2*>(t[i]) : nullptr;
    }
    KW_HIP(hipMemsetAsync(f.s[i], 0, elems * sizeof(float2), ctx->stream));
    if (f.t[i]) KW_HIP(hipMemsetAsync(f.t[i], 0, elems * sizeof(float2), ctx->stream));
    if (f.pipelined)
    { // the transposed set is always the library's own
      KW_HIP(hipMalloc(reinterpret_cast<void**>(&f.r[i]), elems * sizeof(float2)));
      KW_HIP(hipMemsetAsync(f.r[i], 0, elems * sizeof(float2), ctx->stream));
    }
  }
  return KW_OK;
}

kw_status create_impl(kw_ctx* ctx, void* const s[3], void* const t[3])
{
  KW_CHECK_CONSTS(ctx);
  int ok = 0;
  kw_fused_supported(ctx, &ok);
  if (!ok)
  {
    kw_set_error("kw_fused_create: grid %ux%ux%u (x %u ranks) is not supported by the fused pipeline", ctx->c.nx,
                 ctx->c.ny, ctx->c.nz, ctx->fused.nranks);
    return KW_ERR_INVALID;
  }
  // keep the slab description across the reset
  const auto slab = ctx->fused;
  kw_fused_destroy(ctx);
  auto& f = ctx->fused;
  f.slab = slab.slab; f.nranks = slab.nranks; f.rank = slab.rank; f.exchange = slab.exchange; f.exchange_user = slab.exchange_user;
  f.exchange_start = slab.exchange_start; f.exchange_wait = slab.exchange_wait; f.exchange_piece = slab.exchange_piece;
  KW_HIP(hipSetDevice(ctx->device));
  const kw_constants& c = ctx->c;
  f.nz_global = (f.slab) ? slab.nz_global : c.nz;
  f.nyl       = c.ny / f.nranks;
  f.two_d     = (!f.slab && c.nz == 1);
  if (f.two_d)
  { // 2-D: the z-pass kernels run along y — one "row" per plane, lines of Ny elements with stride P
    f.nz_global = c.ny;
    f.nyl       = 1;
  }
  f.Palloc    = (c.nx_complex + NLMAX - 1) / NLMAX * NLMAX;
  {
    // x-Nyquist column apart (see tile_coord) whenever it is the one bin beyond whole tiles.  On slabs the exchange then
    // moves two pieces per peer — the row chunk [nz local][nyl][Nx/2] and the side chunk [nz local][nyl] — i.e. exactly
    // the Nx/2 + 1 bins per row, in aligned rows.  kw_tuning::side_array = 0 keeps the column in the (padded) rows.
    const bool side = (ctx->tuning.side_array != 0) && (c.nx_complex % NLMAX == 1u) && (c.nx_complex > NLMAX) && !f.two_d;
    f.nxm      = side ? c.nx_complex - 1u : c.nx_complex;
    f.P        = side ? f.nxm : f.Palloc;
    f.side_off = side ? f.P * c.ny * c.nz : 0u;
  }
  // Exchange-side row pitch = the pipeline's pitch: every 16-column tile segment of the packed y-passes and of the
  // transposed z-pass is one aligned 128-B line.  (Rows sent without their padding — nx/2+1 complex — would save 5-10 % of
  // the wire bytes of a grid without a side array, but their tile segments straddle two lines: measured on one rank at
  // 256^3 the z-pass then takes 67 us per array instead of 30 and the packed y-passes 35-42 us instead of 23-27.)
  f.PX = f.P;
  memset(f.xslot, -1, sizeof(f.xslot));
  {
    // Pipelined schedule: whenever the exchange can move plane chunks (the library's own transports, or a piece
    // callback).  kw_tuning::slab_pipeline = 0 keeps the whole-array schedule; slab_chunks sets the chunk count (chunks
    // are whole planes and whole x tiles).
    const kw_tuning& tn = ctx->tuning;
    const bool can = f.slab && (f.exchange_piece != nullptr || (f.exchange == nullptr && f.exchange_start == nullptr));
    f.pipelined    = can && tn.slab_pipeline != 0;
    // Plane chunks are off by default (slab_chunks = 1): a step on 8 GPUs is bound by the links, and every RCCL group
    // has a fixed cost on the wire (~30 us measured on one rank) and on the launching thread (~44 us); with the links
    // modelled (tools/emulate_rank.py) two chunks gain 3-4 % at a 10 us fixed cost and lose 6 % at 30 us.
    uint32_t nch = static_cast<uint32_t>(tn.slab_chunks > 0 ? tn.slab_chunks : 1);
    if (nch > KW_XCHUNKS_MAX) nch = KW_XCHUNKS_MAX;
    while (nch > 1 && (c.nz % nch != 0 || (c.nz / nch * c.ny) % (2 * nl_x(c.nx)) != 0)) nch--;
    f.xchunks = f.pipelined ? nch : 1u;
    // Below 4 MB per peer and array the exchanges are latency- and launch-bound: all arrays of a stage then travel in
    // one exchange per direction (6 per step instead of 13) and the passes run as multi-array launches (slab_batch).
    const size_t per_peer = static_cast<size_t>(c.nz) * f.nyl * c.nx_complex * sizeof(float2);
    f.xbatch = f.pipelined && ((tn.slab_batch >= 0) ? (tn.slab_batch == 1) : (per_peer < (4u << 20)));
    if (f.xbatch) f.xchunks = 1u;
  }
  KW_TRY(alloc_scratch(ctx, s, t));
  const uint32_t lens[3] = { c.nx, c.ny, f.nz_global };
  for (int i = 0; i < 3; i++)
  {
    std::vector<float2> tw(lens[i]);
    for (uint32_t m = 0; m < lens[i]; m++)
    {
      const double ph = -2.0 * M_PI * static_cast<double>(m) / static_cast<double>(lens[i]);
      tw[m]           = make_float2(static_cast<float>(std::cos(ph)), static_cast<float>(std::sin(ph)));
    }
    KW_HIP(hipMalloc(reinterpret_cast<void**>(&f.tw[i]), lens[i] * sizeof(float2)));
    KW_HIP(hipMemcpyAsync(f.tw[i], tw.data(), lens[i] * sizeof(float2), hipMemcpyHostToDevice, ctx->stream));
    KW_HIP(hipStreamSynchronize(ctx->stream));
  }
  f.split512 = (ctx->tuning.split512 != 0);
  // whole-plane x kernels (k_xinv PLANE): square planes of 32 / 64, one GPU, 3-D
  f.plane = (ctx->tuning.plane_kernels != 0) && !f.slab && !f.two_d && c.nx == c.ny && plane_len(static_cast<int>(c.nx)) &&
            supported_len(c.nx);
  if (f.plane)
  {
    const size_t elems = static_cast<size_t>(f.Palloc) * c.ny * c.nz;
    KW_HIP(hipMalloc(reinterpret_cast<void**>(&f.s4), elems * sizeof(float2)));
    KW_HIP(hipMemsetAsync(f.s4, 0, elems * sizeof(float2), ctx->stream));
  }
  f.ready = true;
  return KW_OK;
}

#endif // KW_FUSED_TU == 0
} // namespace

#if KW_FUSED_TU == 1 || KW_FUSED_TU == 3 || KW_FUSED_TU == 5 || KW_FUSED_TU == 6 || KW_FUSED_TU == 7 || KW_FUSED_TU == 8
// passes 1 / 7 / 3: the chained density epilogues (1 and 3 hold the entry points); passes 5 / 8 / 6: the ones that store
// their terms
#if KW_FUSED_TU == 1
kw_status kw_fused_xinv_density(int chain, int terms, kw_ctx* ctx, int ncomp, const void* xinv_args, uint32_t tile0, uint32_t ntiles)
#elif KW_FUSED_TU == 3
kw_status kw_fused_xinv_density_tail(int chain, int terms, kw_ctx* ctx, int ncomp, const void* xinv_args, uint32_t tile0, uint32_t ntiles)
#elif KW_FUSED_TU == 5
kw_status kw_fused_xinv_density_plain(int chain, int terms, kw_ctx* ctx, int ncomp, const void* xinv_args, uint32_t tile0, uint32_t ntiles)
#elif KW_FUSED_TU == 6
kw_status kw_fused_xinv_density_tail_plain(int chain, int terms, kw_ctx* ctx, int ncomp, const void* xinv_args, uint32_t tile0, uint32_t ntiles)
#elif KW_FUSED_TU == 7
kw_status kw_fused_xinv_density_long(int chain, int terms, kw_ctx* ctx, int ncomp, const void* xinv_args, uint32_t tile0, uint32_t ntiles)
#else
kw_status kw_fused_xinv_density_plain_long(int chain, int terms, kw_ctx* ctx, int ncomp, const void* xinv_args, uint32_t tile0, uint32_t ntiles)
#endif
{
  constexpr bool T = (KW_FUSED_TU == 3 || KW_FUSED_TU == 6);
  const XinvArgs& a = *static_cast<const XinvArgs*>(xinv_args);
#if KW_FUSED_TU == 1
  if (ctx->c.nx >= KW_LONG_LINES)
    return chain ? kw_fused_xinv_density_long(chain, terms, ctx, ncomp, xinv_args, tile0, ntiles)
                 : kw_fused_xinv_density_plain_long(chain, terms, ctx, ncomp, xinv_args, tile0, ntiles);
  if (!chain) return kw_fused_xinv_density_plain(chain, terms, ctx, ncomp, xinv_args, tile0, ntiles);
#elif KW_FUSED_TU == 3
  if (!chain) return kw_fused_xinv_density_tail_plain(chain, terms, ctx, ncomp, xinv_args, tile0, ntiles);
#endif
  switch (2 * terms + chain)
  {
#if KW_FUSED_TU == 5 || KW_FUSED_TU == 6 || KW_FUSED_TU == 8
    case 0: return launch_xinv_impl<EPI_DENSITY, false, 0, T>(ctx, ncomp, a, tile0, ntiles);
    case 2: return launch_xinv_impl<EPI_DENSITY, false, 1, T>(ctx, ncomp, a, tile0, ntiles);
    case 4: return launch_xinv_impl<EPI_DENSITY, false, 2, T>(ctx, ncomp, a, tile0, ntiles);
    case 6: return launch_xinv_impl<EPI_DENSITY, false, 3, T>(ctx, ncomp, a, tile0, ntiles);
#else
    case 3: return launch_xinv_impl<EPI_DENSITY, true, 1, T>(ctx, ncomp, a, tile0, ntiles);
    case 5: return launch_xinv_impl<EPI_DENSITY, true, 2, T>(ctx, ncomp, a, tile0, ntiles);
    case 7: return launch_xinv_impl<EPI_DENSITY, true, 3, T>(ctx, ncomp, a, tile0, ntiles);
#endif
    default: kw_set_error("fused pipeline: no density epilogue for chain = %d, terms = %d", chain, terms); return KW_ERR_INVALID;
  }
}
#if KW_FUSED_TU == 1
kw_status kw_fused_xinv_density_plane(int chain, int terms, kw_ctx* ctx, int ncomp, const void* xinv_args, uint32_t tile0, uint32_t ntiles)
{
  const XinvArgs& a = *static_cast<const XinvArgs*>(xinv_args);
  switch (2 * terms + chain)
  {
    case 0: return launch_xinv_plane_impl<EPI_DENSITY, false, 0>(ctx, ncomp, a, tile0, ntiles);
    case 2: return launch_xinv_plane_impl<EPI_DENSITY, false, 1>(ctx, ncomp, a, tile0, ntiles);
    case 3: return launch_xinv_plane_impl<EPI_DENSITY, true, 1>(ctx, ncomp, a, tile0, ntiles);
    case 4: return launch_xinv_plane_impl<EPI_DENSITY, false, 2>(ctx, ncomp, a, tile0, ntiles);
    case 5: return launch_xinv_plane_impl<EPI_DENSITY, true, 2>(ctx, ncomp, a, tile0, ntiles);
    case 6: return launch_xinv_plane_impl<EPI_DENSITY, false, 3>(ctx, ncomp, a, tile0, ntiles);
    case 7: return launch_xinv_plane_impl<EPI_DENSITY, true, 3>(ctx, ncomp, a, tile0, ntiles);
    default: kw_set_error("fused pipeline: no density epilogue for chain = %d, terms = %d", chain, terms); return KW_ERR_INVALID;
  }
}
#endif
#elif KW_FUSED_TU == 2 || KW_FUSED_TU == 4
#if KW_FUSED_TU == 2
kw_status kw_fused_xinv_other_plane(int epi, int chain, kw_ctx* ctx, int ncomp, const void* xinv_args, uint32_t tile0, uint32_t ntiles)
{
  const XinvArgs& a = *static_cast<const XinvArgs*>(xinv_args);
  switch (2 * epi + chain)
  {
    case 2 * EPI_STORE: return launch_xinv_plane_impl<EPI_STORE, false, 0>(ctx, ncomp, a, tile0, ntiles);
    case 2 * EPI_VELOCITY: return launch_xinv_plane_impl<EPI_VELOCITY, false, 0>(ctx, ncomp, a, tile0, ntiles);
    case 2 * EPI_VELOCITY + 1: return launch_xinv_plane_impl<EPI_VELOCITY, true, 0>(ctx, ncomp, a, tile0, ntiles);
    case 2 * EPI_INITVEL: return launch_xinv_plane_impl<EPI_INITVEL, false, 0>(ctx, ncomp, a, tile0, ntiles);
    case 2 * EPI_PSUM: return launch_xinv_plane_impl<EPI_PSUM, false, 0>(ctx, ncomp, a, tile0, ntiles);
    case 2 * EPI_PSUM + 1: return launch_xinv_plane_impl<EPI_PSUM, true, 0>(ctx, ncomp, a, tile0, ntiles);
    default: kw_set_error("fused pipeline: no epilogue %d with chain = %d", epi, chain); return KW_ERR_INVALID;
  }
}
#endif
#if KW_FUSED_TU == 2
kw_status kw_fused_xinv_other(int epi, int chain, kw_ctx* ctx, int ncomp, const void* xinv_args, uint32_t tile0, uint32_t ntiles)
#else
kw_status kw_fused_xinv_other_tail(int epi, int chain, kw_ctx* ctx, int ncomp, const void* xinv_args, uint32_t tile0, uint32_t ntiles)
#endif
{
  constexpr bool T = (KW_FUSED_TU == 4);
  const XinvArgs& a = *static_cast<const XinvArgs*>(xinv_args);
  switch (2 * epi + chain)
  {
    case 2 * EPI_STORE: return launch_xinv_impl<EPI_STORE, false, 0, T>(ctx, ncomp, a, tile0, ntiles);
    case 2 * EPI_VELOCITY: return launch_xinv_impl<EPI_VELOCITY, false, 0, T>(ctx, ncomp, a, tile0, ntiles);
    case 2 * EPI_VELOCITY + 1: return launch_xinv_impl<EPI_VELOCITY, true, 0, T>(ctx, ncomp, a, tile0, ntiles);
    case 2 * EPI_INITVEL: return launch_xinv_impl<EPI_INITVEL, false, 0, T>(ctx, ncomp, a, tile0, ntiles);
    case 2 * EPI_PSUM: return launch_xinv_impl<EPI_PSUM, false, 0, T>(ctx, ncomp, a, tile0, ntiles);
    case 2 * EPI_PSUM + 1: return launch_xinv_impl<EPI_PSUM, true, 0, T>(ctx, ncomp, a, tile0, ntiles);
    default: kw_set_error("fused pipeline: no epilogue %d with chain = %d", epi, chain); return KW_ERR_INVALID;
  }
}
#endif

#if KW_FUSED_TU == 0
extern "C" {

kw_status kw_fused_set_slab(kw_ctx* ctx, uint32_t nranks, uint32_t rank, uint32_t nz_global, kw_exchange_fn fn, void* user)
{
  KW_CHECK_CTX(ctx);
  KW_REQUIRE(nranks >= 1 && rank < nranks);
  if (ctx->fused.ready) { kw_set_error("kw_fused_set_slab: must be called before kw_fused_create"); return KW_ERR_STATE; }
  uint32_t comm_ranks = 0, comm_rank = 0;
  KW_TRY(kw_comm_info(ctx, &comm_ranks, &comm_rank, nullptr));
  if (fn == nullptr && nranks > 1 && comm_ranks == 0)
  {
    kw_set_error("kw_fused_set_slab: %u ranks need an exchange: call kw_comm_init first or pass a callback", nranks);
    return KW_ERR_STATE;
  }
  if (fn == nullptr && comm_ranks != 0 && (comm_ranks != nranks || comm_rank != rank))
  {
    kw_set_error("kw_fused_set_slab: rank %u of %u does not match the communicator (rank %u of %u)", rank, nranks, comm_rank, comm_ranks);
    return KW_ERR_INVALID;
  }
  // one rank with an exchange (callback or communicator) = the slab path against itself
  ctx->fused.slab          = (nranks > 1) || (fn != nullptr) || (comm_ranks != 0);
  ctx->fused.nranks        = nranks;
  ctx->fused.rank          = rank;
  ctx->fused.nz_global     = nz_global;
  ctx->fused.exchange      = fn;
  ctx->fused.exchange_user = user;
  return KW_OK;
}

kw_status kw_fused_set_slab_async(kw_ctx* ctx, kw_exchange_start_fn start, kw_exchange_wait_fn wait)
{
  KW_CHECK_CTX(ctx);
  KW_REQUIRE((start == nullptr) == (wait == nullptr));
  if (ctx->fused.ready) { kw_set_error("kw_fused_set_slab_async: must be called before kw_fused_create"); return KW_ERR_STATE; }
  ctx->fused.exchange_start = start;
  ctx->fused.exchange_wait  = wait;
  return KW_OK;
}

kw_status kw_fused_set_slab_pieces(kw_ctx* ctx, kw_exchange_piece_fn start, kw_exchange_wait_fn wait)
{
  KW_CHECK_CTX(ctx);
  KW_REQUIRE(start != nullptr || wait == nullptr);
  if (ctx->fused.ready) { kw_set_error("kw_fused_set_slab_pieces: must be called before kw_fused_create"); return KW_ERR_STATE; }
  ctx->fused.exchange_piece = start;
  if (start != nullptr) { ctx->fused.exchange_start = nullptr; ctx->fused.exchange_wait = wait; }
  return KW_OK;
}

kw_status kw_fused_supported(kw_ctx* ctx, int* out)
{
  KW_CHECK_CONSTS(ctx);
  KW_REQUIRE(out != nullptr);
  const kw_constants& c = ctx->c;
  const auto& f         = ctx->fused;
  const uint32_t nzg    = (f.slab) ? f.nz_global : c.nz;
  // (the x kernels work on tiles of 2 * NL rows; a row count Ny * Nz that is no whole number of tiles ends in one masked tile)
  bool ok = supported_len(c.nx) && supported_len(c.ny) && supported_len(nzg);
  if (!f.slab && c.nz == 1) ok = supported_len(c.nx) && supported_len(c.ny); // 2-D: x-pass, fused y-pass, x-pass
  if (f.slab) ok = ok && (nzg == c.nz * f.nranks) && (c.ny % f.nranks == 0);
  if (ok && !has_partial_x_tiles(static_cast<int>(c.nx))) ok = (c.ny * c.nz) % (2u * static_cast<uint32_t>(nl_x(c.nx))) == 0;
  const uint64_t P64 = (c.nx_complex + NLMAX - 1) / NLMAX * NLMAX;
  ok = ok && (P64 * c.ny * c.nz < (1ull << 32)) && (static_cast<uint64_t>(c.nx) * c.ny * c.nz < (1ull << 32));
  *out = ok ? 1 : 0;
  return KW_OK;
}

kw_status kw_fused_destroy(kw_ctx* ctx)
{
  KW_CHECK_CTX(ctx);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  (void)kw_comm_sync(ctx); // forward exchanges started ahead by the last stage may still be on the communication stream
  kw_comm_buffers_gone(ctx);
  auto& f = ctx->fused;
  for (int i = 0; i < 3; i++)
  {
    if (f.r[i]) (void)hipFree(f.r[i]);
    f.r[i] = nullptr;
    if (i == 0 && f.s4) { (void)hipFree(f.s4); f.s4 = nullptr; }
    if (f.owns_scratch)
    {
      if (f.s[i]) (void)hipFree(f.s[i]);
      if (f.t[i]) (void)hipFree(f.t[i]);
    }
    if (f.tw[i]) (void)hipFree(f.tw[i]);
    f.s[i] = f.t[i] = nullptr;
    f.tw[i] = nullptr;
  }
  f = kw_ctx::fused_plan();
  return KW_OK;
}

kw_status kw_fused_create(kw_ctx* ctx) { return create_impl(ctx, nullptr, nullptr); }

kw_status kw_fused_create_with_scratch(kw_ctx* ctx, void* const s[3], void* const t[3])
{
  KW_CHECK_CTX(ctx);
  KW_REQUIRE(s != nullptr);
  return create_impl(ctx, s, t);
}

kw_status kw_fused_scratch_bytes(kw_ctx* ctx, size_t* out)
{
  KW_CHECK_CONSTS(ctx);
  KW_REQUIRE(out != nullptr);
  const uint32_t P = (ctx->c.nx_complex + NLMAX - 1) / NLMAX * NLMAX;
  *out             = static_cast<size_t>(P) * ctx->c.ny * ctx->c.nz * sizeof(float2);
  return KW_OK;
}

kw_status kw_fused_reduced_elems(kw_ctx* ctx, size_t* out)
{
  KW_FUSED_READY(ctx);
  KW_REQUIRE(out != nullptr);
  *out = static_cast<size_t>(ctx->fused.Palloc) * ctx->c.ny * ctx->c.nz;
  return KW_OK;
}

// src is [rows][nxc] with rows = ny*nz (one rank: [nz][ny]; slab mode: the transposed [nz_global][nyl]) — same count
kw_status kw_fused_import_reduced(kw_ctx* ctx, float* dst_padded, const float* src)
{
  KW_FUSED_READY(ctx);
  KW_REQUIRE(dst_padded && src);
  const kw_constants& c = ctx->c;
  // factorisation of the z lines, as the z-pass kernels are instantiated
  const bool split = (ctx->fused.nz_global == 512 && ctx->fused.split512);
  uint32_t r1 = Fac<256>::R1, r2 = Fac<256>::R2; // the split lines are built on the 256-point transform
  if (!split)
    switch (ctx->fused.nz_global)
    {
#define X(LEN) case LEN: r1 = Fac<LEN>::R1; r2 = Fac<LEN>::R2; break;
      KW_FUSED_LENGTHS(X)
#undef X
      default: kw_set_error("kw_fused_import_reduced: unsupported length %u", ctx->fused.nz_global); return KW_ERR_INVALID;
    }
  const uint32_t vec = static_cast<uint32_t>(op_vec(static_cast<int>(split ? 2 * r2 : r2)));
  const auto& f = ctx->fused;
  const size_t main_total = static_cast<size_t>(c.ny) * c.nz * f.P; // (f.P = row pitch of the main part: nxm rounded up to 16)
  KW_REQUIRE(f.nz_global <= 1024);
  // columns beyond the imported ones and the unused tail of the array read as zero
  KW_HIP(hipMemsetAsync(dst_padded, 0, static_cast<size_t>(f.Palloc) * c.ny * c.nz * sizeof(float), ctx->stream)); // = kw_fused_reduced_elems
  LAUNCH(k_import_reduced, dim3(f.nyl * (f.P / NLMAX)), dim3(256), dst_padded, src, c.nx_complex, f.nxm, f.P, f.nyl, f.nz_global,
         r1, vec, split ? 1u : 0u);
  if (f.side_off != 0)
  { // ceil(nyl / 16) tiles of 16 ky x nz values
    const size_t side_total = static_cast<size_t>((f.nyl + NLMAX - 1) / NLMAX) * NLMAX * f.nz_global;
    LAUNCH(k_import_reduced_side, dim3(ctx->cu_count * 2), dim3(256), dst_padded + main_total, src, c.nx_complex, f.nyl,
           f.nz_global, side_total, r1, vec, split ? 1u : 0u);
  }
  return KW_OK;
}

// A1-A4: u <- pml_sg*(pml_sg*u - dt/rho0_sg * ifftn(ddk_pos * kappa * fftn(p)) / N)
kw_status kw_fused_velocity(kw_ctx* ctx, const float* p, float* ux, float* uy, float* uz, const float* dtx,
                            const float* dty, const float* dtz, const float* pmlx, const float* pmly, const float* pmlz,
                            const float* kappa_padded, const float* ddx, const float* ddy, const float* ddz,
                            int chain_u_spectra)
{
  KW_FUSED_READY(ctx);
  KW_PROF(ctx, "fused_velocity");
  KW_REQUIRE(p && ux && uy && uz && pmlx && pmly && pmlz && kappa_padded && ddx && ddy && ddz);
  KW_REQUIRE((dtx == nullptr) == (dty == nullptr) && (dtx == nullptr) == (dtz == nullptr));
  float2** S = ctx->fused.s;
  const float* in1[1] = { p };
  const bool p_in_scratch = (chain_u_spectra & KW_FUSED_P_IN_SCRATCH) != 0;
  chain_u_spectra &= KW_FUSED_CHAIN_U;
  KW_TRY(forward_xy(ctx, 1, p_in_scratch ? nullptr : in1));
  ZArgs z{};
  z.in[0] = S[0];
  for (int i = 0; i < 3; i++) z.out[i] = S[i];
  z.op[0] = kappa_padded;
  z.dd[0] = (const float2*)ddx; z.dd[1] = (const float2*)ddy; z.dd[2] = (const float2*)ddz;
  KW_TRY(launch_zfused<Z_PGRAD>(ctx, 1, z));
  XinvArgs x{};
  float* u[3] = { ux, uy, uz };
  const float* dt[3] = { dtx, dty, dtz };
  const float* pml[3] = { pmlx, pmly, pmlz };
  for (int i = 0; i < 3; i++) { x.in[i] = S[i]; x.out[i] = u[i]; x.m0[i] = dt[i]; x.m1[i] = pml[i]; x.fout[i] = S[i]; }
  // whole-plane kernels: the block of u_x would overwrite the plane of Q that the block of u_y reads: its chained spectrum
  // goes to a fourth array, where the density stage picks it up
  if (ctx->fused.plane) x.fout[0] = ctx->fused.s4;
  // chained: the updated velocity rows are forward-transformed along x (and y) on the spot (valid as long as nothing
  // else writes u before kw_fused_density(..., KW_FUSED_U_IN_SCRATCH))
  if (chain_u_spectra) return gradient_tail<EPI_VELOCITY, true>(ctx, x, (const float2*)ddx, (const float2*)ddy);
  return gradient_tail<EPI_VELOCITY, false>(ctx, x, (const float2*)ddx, (const float2*)ddy);
}

// A12 second half: u <- +0.5*dt/rho0_sg * ifftn(ddk_pos * kappa * fftn(p)) / N
kw_status kw_fused_initial_velocity(kw_ctx* ctx, const float* p, float* ux, float* uy, float* uz, const float* dtx,
                                    const float* dty, const float* dtz, const float* kappa_padded, const float* ddx,
                                    const float* ddy, const float* ddz)
{
  KW_FUSED_READY(ctx);
  KW_PROF(ctx, "fused_initial_velocity");
  KW_REQUIRE(p && ux && uy && uz && kappa_padded && ddx && ddy && ddz);
  float2** S = ctx->fused.s;
  const float* in1[1] = { p };
  KW_TRY(forward_xy(ctx, 1, in1));
  ZArgs z{};
  z.in[0] = S[0];
  for (int i = 0; i < 3; i++) z.out[i] = S[i];
  z.op[0] = kappa_padded;
  z.dd[0] = (const float2*)ddx; z.dd[1] = (const float2*)ddy; z.dd[2] = (const float2*)ddz;
  KW_TRY(launch_zfused<Z_PGRAD>(ctx, 1, z));
  XinvArgs x{};
  float* u[3] = { ux, uy, uz };
  const float* dt[3] = { dtx, dty, dtz };
  for (int i = 0; i < 3; i++) { x.in[i] = S[i]; x.out[i] = u[i]; x.m0[i] = dt[i]; }
  return gradient_tail<EPI_INITVEL, false>(ctx, x, (const float2*)ddx, (const float2*)ddy);
}

// A6-A9 (+ the term kernels of A11 when terms != 0): du = ifftn(ddk_neg*kappa*fftn(u))/N; rho update; pressure terms
kw_status kw_fused_density(kw_ctx* ctx, int nonlinear, const float* ux, const float* uy, const float* uz, float* rx,
                           float* ry, float* rz, const float* pmlx, const float* pmly, const float* pmlz,
                           const float* rho0, const float* kappa_padded, const float* ddx, const float* ddy,
                           const float* ddz, float* duxdx, float* duydy, float* duzdz, int terms, const float* bona,
                           float* t0, float* t1, float* t2, int flags)
{
  KW_FUSED_READY(ctx);
  KW_PROF(ctx, "fused_density");
  const bool u_in_scratch = (flags & KW_FUSED_U_IN_SCRATCH) != 0;
  const bool chain_terms  = (flags & KW_FUSED_CHAIN_TERMS) != 0;
  KW_REQUIRE(!chain_terms || terms != 0);
  KW_REQUIRE(ux && uy && uz && rx && ry && rz && pmlx && pmly && pmlz && kappa_padded && ddx && ddy && ddz);
  KW_REQUIRE((duxdx == nullptr) == (duydy == nullptr) && (duxdx == nullptr) == (duzdz == nullptr));
  KW_REQUIRE(terms >= 0 && terms <= 3);
  KW_REQUIRE(terms == 0 || terms == 3 || (t0 && t1 && (terms == 1 || t2)));
  KW_REQUIRE(terms != 3 || t0 != nullptr);
  float2** S = ctx->fused.s;
  const float* in3[3] = { ux, uy, uz };
  ZArgs z{};
  for (int i = 0; i < 3; i++) { z.in[i] = S[i]; z.out[i] = S[i]; }
  if (u_in_scratch && ctx->fused.plane) z.in[0] = ctx->fused.s4; // (see kw_fused_velocity)
  z.op[0] = kappa_padded;
  z.dd[0] = (const float2*)ddx; z.dd[1] = (const float2*)ddy; z.dd[2] = (const float2*)ddz;
  // 2-D: u_z is identically zero; its spectrum travels as zeros (chained stages leave G_y of the velocity stage in S[2])
  if (ctx->fused.two_d && u_in_scratch)
    KW_HIP(hipMemsetAsync(S[2], 0, static_cast<size_t>(ctx->fused.Palloc) * ctx->c.ny * sizeof(float2), ctx->stream));
  if (ctx->fused.slab)
  {
    KW_TRY(slab_chain<Z_VGRAD>(ctx, 3, u_in_scratch ? nullptr : in3, z));
  }
  else
  {
    KW_TRY(forward_xy(ctx, 3, u_in_scratch ? nullptr : in3));
    KW_TRY(launch_zfused<Z_VGRAD>(ctx, 3, z));
  }
  const bool tail_chunked = (!ctx->fused.slab && !ctx->fused.two_d);
  XinvArgs x{};
  float* rho[3] = { rx, ry, rz };
  const float* pml[3] = { pmlx, pmly, pmlz };
  float* du[3] = { duxdx, duydy, duzdz };
  float* t[3] = { t0, t1, t2 };
  for (int i = 0; i < 3; i++) { x.in[i] = S[i]; x.out[i] = rho[i]; x.m1[i] = pml[i]; x.aux[i] = du[i]; x.t[i] = t[i]; }
  x.m0[0]     = rho0;
  x.m0[1]     = bona;
  x.m0[2]     = (terms == 3) ? t1 : nullptr; // lossless pressure: t0 = p (out), t1 = c2 array or NULL (in)
  x.nonlinear = nonlinear;
  x.terms     = terms;
  x.fout[0]   = S[0]; // chained: x-spectrum of rho0 * sum(du)
  x.fout[1]   = S[1]; //          x-spectrum of sum(rho)
  // one specialised kernel per pressure-term mode
#define DENSITY_TAIL(T)                                                                                                \
  do {                                                                                                                 \
    if (tail_chunked)                                                                                                  \
    {                                                                                                                  \
      if (chain_terms) KW_TRY((plane_local_tail<EPI_DENSITY, true, (T) == 0 ? 1 : (T)>(ctx, 3, 1, x, (T) == 3 ? 1 : 2))); \
      else KW_TRY((plane_local_tail<EPI_DENSITY, false, (T)>(ctx, 3, 1, x, 0)));                                       \
    }                                                                                                                  \
    else if (ctx->fused.pipelined)                                                                                     \
    {                                                                                                                  \
      if (chain_terms) KW_TRY((pslab_tail<EPI_DENSITY, true, (T) == 0 ? 1 : (T)>(ctx, 3, 1, x, (T) == 3 ? 1 : 2)));    \
      else KW_TRY((pslab_tail<EPI_DENSITY, false, (T)>(ctx, 3, 1, x, 0)));                                             \
    }                                                                                                                  \
    else if (chain_terms) KW_TRY((launch_xinv<EPI_DENSITY, true, (T) == 0 ? 1 : (T)>(ctx, 1, x)));                      \
    else KW_TRY((launch_xinv<EPI_DENSITY, false, (T)>(ctx, 1, x)));                                                    \
  } while (0)
  switch (terms)
  {
    case 0: DENSITY_TAIL(0); break; // (chain_terms requires terms != 0: checked above)
    case 1: DENSITY_TAIL(1); break;
    case 2: DENSITY_TAIL(2); break;
    default: DENSITY_TAIL(3); break;
  }
#undef DENSITY_TAIL
  return KW_OK;
}

// A6-A8 only: du_i/dx_i = ifftn(ddk_i_neg * kappa * fftn(u_i)) / N stored as arrays — for callers that put something
// between the gradient and the density update (non-uniform grids: duxdx *= dxudxn, SolverCudaKernels.cu:1285-1301)
kw_status kw_fused_velocity_gradient(kw_ctx* ctx, const float* ux, const float* uy, const float* uz, float* duxdx,
                                     float* duydy, float* duzdz, const float* kappa_padded, const float* ddx, const float* ddy,
                                     const float* ddz, int flags)
{
  KW_FUSED_READY(ctx);
  KW_PROF(ctx, "fused_velocity_gradient");
  KW_REQUIRE(ux && uy && uz && duxdx && duydy && duzdz && kappa_padded && ddx && ddy && ddz);
  const bool u_in_scratch = (flags & KW_FUSED_U_IN_SCRATCH) != 0;
  float2** S = ctx->fused.s;
  const float* in3[3] = { ux, uy, uz };
  ZArgs z{};
  for (int i = 0; i < 3; i++) { z.in[i] = S[i]; z.out[i] = S[i]; }
  if (u_in_scratch && ctx->fused.plane) z.in[0] = ctx->fused.s4; // (see kw_fused_velocity)
  z.op[0] = kappa_padded;
  z.dd[0] = (const float2*)ddx; z.dd[1] = (const float2*)ddy; z.dd[2] = (const float2*)ddz;
  if (ctx->fused.two_d && u_in_scratch)
    KW_HIP(hipMemsetAsync(S[2], 0, static_cast<size_t>(ctx->fused.Palloc) * ctx->c.ny * sizeof(float2), ctx->stream));
  if (ctx->fused.slab) KW_TRY(slab_chain<Z_VGRAD>(ctx, 3, u_in_scratch ? nullptr : in3, z));
  else
  {
    KW_TRY(forward_xy(ctx, 3, u_in_scratch ? nullptr : in3));
    KW_TRY(launch_zfused<Z_VGRAD>(ctx, 3, z));
    KW_TRY(inverse_y(ctx, 3));
  }
  XinvArgs x{};
  float* du[3] = { duxdx, duydy, duzdz };
  for (int i = 0; i < 3; i++) { x.in[i] = S[i]; x.out[i] = du[i]; }
  if (ctx->fused.slab && ctx->fused.pipelined) return pslab_tail<EPI_STORE, false>(ctx, 3, 3, x, 0);
  return launch_xinv<EPI_STORE>(ctx, 3, x);
}

// A11 absorbing branch after the terms: p = c2*(first + d*(tau*ifftn(nabla1*fftn(vel_grad_term)) - eta*ifftn(nabla2*fftn(density_sum))))
kw_status kw_fused_absorption_pressure(kw_ctx* ctx, float* p, const float* vel_grad_term, const float* density_sum,
                                       const float* first, const float* nabla1_padded, const float* nabla2_padded,
                                       const float* c2, const float* tau, const float* eta, int flags)
{
  const bool terms_in_scratch = (flags & KW_FUSED_TERMS_IN_SCRATCH) != 0;
  const bool chain_p          = (flags & KW_FUSED_CHAIN_P) != 0;
  KW_FUSED_READY(ctx);
  KW_PROF(ctx, "fused_absorption_pressure");
  KW_REQUIRE(p && first && nabla1_padded && nabla2_padded);
  KW_REQUIRE(terms_in_scratch || (vel_grad_term && density_sum));
  KW_REQUIRE((tau == nullptr) == (eta == nullptr));
  float2** S = ctx->fused.s;
  const float* in2[2] = { vel_grad_term, density_sum };
  ZArgs z{};
  for (int i = 0; i < 2; i++) { z.in[i] = S[i]; z.out[i] = S[i]; }
  z.op[0] = nabla1_padded;
  z.op[1] = nabla2_padded;
  if (ctx->fused.slab)
  {
    KW_TRY(slab_chain<Z_ABSORB>(ctx, 2, terms_in_scratch ? nullptr : in2, z));
  }
  else
  {
    KW_TRY(forward_xy(ctx, 2, terms_in_scratch ? nullptr : in2));
    KW_TRY(launch_zfused<Z_ABSORB>(ctx, 2, z));
  }
  XinvArgs x{};
  x.in[0] = S[0]; x.in[1] = S[1];
  x.out[0] = p;
  x.m0[0] = first; x.m0[1] = c2;
  x.m1[0] = tau;   x.m1[1] = eta;
  x.fout[0] = S[0]; // chained: x-spectrum of the new p
  if (!ctx->fused.slab && !ctx->fused.two_d)
  {
    if (chain_p) KW_TRY((plane_local_tail<EPI_PSUM, true>(ctx, 2, 1, x, 1)));
    else KW_TRY((plane_local_tail<EPI_PSUM, false>(ctx, 2, 1, x, 0)));
  }
  else if (ctx->fused.slab && ctx->fused.pipelined)
  {
    if (chain_p) KW_TRY((pslab_tail<EPI_PSUM, true>(ctx, 2, 1, x, 1)));
    else KW_TRY((pslab_tail<EPI_PSUM, false>(ctx, 2, 1, x, 0)));
  }
  else if (chain_p) KW_TRY((launch_xinv<EPI_PSUM, true>(ctx, 1, x)));
  else KW_TRY(launch_xinv<EPI_PSUM>(ctx, 1, x));
  return KW_OK;
}

// computeVelocityShiftInX/Y/Z + the two 1-D transforms around it (KSpaceFirstOrderSolver.cpp:2714-2735,
// SolverCudaKernels.cu:2617-2710) in one kernel per axis: out = F_axis^-1{ H .* F_axis{in} }, H = full-length Hermitian
// filter with the 1/N of the transform pair folded in (see kwave_hip.h)
kw_status kw_fused_shift_velocity(kw_ctx* ctx, int axis, const float* in, float* out, const float* filter)
{
  KW_FUSED_READY(ctx);
  auto& f = ctx->fused;
  const kw_constants& c = ctx->c;
  KW_REQUIRE(axis >= 0 && axis <= 2 && in != nullptr && out != nullptr && filter != nullptr);
  static const char* const names[3] = { "k_xshift", "k_zfused_shift_y", "k_zfused_shift_z" };
  KW_PROF(ctx, names[axis]);
  if (f.slab && axis == 2)
  { // lines along z cross the slabs: the real array travels as [nz local][ny / P rows per peer][nx] chunks to the
    // owner of each row range ([nz global][nyl][nx] there), is shifted along z and travels back — two exchanges of one
    // real array per sampled step (KSpaceFirstOrderSolver.cpp:2731-2733 on the decomposed grid).  x and y lines are
    // slab-local.  Staging: scratch pair 1 (only S[0] carries a chained spectrum between stages).
    const uint32_t nyl = f.nyl, nzl = c.nz, P = f.nranks;
    const size_t   row = static_cast<size_t>(c.nx) * sizeof(float), chunk = static_cast<size_t>(nzl) * nyl * c.nx;
    float* snd = reinterpret_cast<float*>(f.s[1]);
    float* rcv = reinterpret_cast<float*>(f.t[1]);
    for (uint32_t q = 0; q < P; q++)
      KW_HIP(hipMemcpy2DAsync(snd + q * chunk, nyl * row, in + static_cast<size_t>(q) * nyl * c.nx, c.ny * row, nyl * row, nzl,
                              hipMemcpyDeviceToDevice, ctx->stream));
    KW_TRY(xstart_bytes(ctx, KW_ZSHIFT_SLOT, snd, rcv, chunk * sizeof(float)));
    KW_TRY(xwait_one(ctx, KW_ZSHIFT_SLOT));
    ZArgs z{};
    z.in[0]   = reinterpret_cast<const float2*>(rcv);
    z.out[0]  = reinterpret_cast<float2*>(rcv);
    z.dd[2]   = reinterpret_cast<const float2*>(filter);
    z.tw      = f.tw[2];
    z.nxc     = c.nx / 2;
    z.P       = c.nx / 2;
    z.ny      = nyl;
    z.nz      = f.nz_global;
    z.narr    = 1;
    z.lstride = nyl * z.P;
    z.bstride = z.P;
    const uint32_t nl = nl_z(f.nz_global);
    const dim3 grid((z.nxc + nl - 1) / nl, nyl, 1);
#define M(LEN) LAUNCH((k_zfused<LEN, Z_SHIFT>), grid, dim3((Geo<LEN, nl_z(LEN)>::THREADS)), z)
    KW_LEN_SWITCH(f.nz_global, M)
#undef M
    KW_TRY(xstart_bytes(ctx, KW_ZSHIFT_SLOT, rcv, snd, chunk * sizeof(float)));
    KW_TRY(xwait_one(ctx, KW_ZSHIFT_SLOT));
    for (uint32_t q = 0; q < P; q++)
      KW_HIP(hipMemcpy2DAsync(out + static_cast<size_t>(q) * nyl * c.nx, c.ny * row, snd + q * chunk, nyl * row, nyl * row, nzl,
                              hipMemcpyDeviceToDevice, ctx->stream));
    return KW_OK;
  }
  if (axis == 0)
  {
    XshiftArgs a{ in, out, f.tw[0], reinterpret_cast<const float2*>(filter), c.ny * c.nz, 0u };
    const uint32_t rows_per_tile = 2u * static_cast<uint32_t>(nl_x(c.nx)), full = a.nrows / rows_per_tile;
    if (full > 0)
    {
      const dim3 grid(full, 1, 1);
#define M(LEN) LAUNCH((k_xshift<LEN, false>), grid, dim3(GeoX<LEN>::THREADS), a)
      KW_LEN_SWITCH(c.nx, M)
#undef M
    }
    if (a.nrows % rows_per_tile != 0)
    {
      a.tile0 = full;
#define M(LEN) if constexpr (!has_partial_x_tiles(LEN)) KW_NO_TAIL(LEN) else LAUNCH((k_xshift<LEN, true>), dim3(1, 1, 1), dim3(GeoX<LEN>::THREADS), a)
      KW_LEN_SWITCH(c.nx, M)
#undef M
    }
    return KW_OK;
  }
  // y / z: the real array is read as nx/2 complex columns; lines run along the axis with the matching stride
  ZArgs z{};
  z.in[0]  = reinterpret_cast<const float2*>(in);
  z.out[0] = reinterpret_cast<float2*>(out);
  z.dd[2]  = reinterpret_cast<const float2*>(filter);
  z.tw     = f.tw[axis];
  z.nxc    = c.nx / 2;
  z.P      = c.nx / 2;
  z.ny     = c.ny;
  z.nz     = c.nz;
  z.narr   = 1;
  const uint32_t len   = (axis == 1) ? c.ny : c.nz;
  const uint32_t lines = (axis == 1) ? c.nz : c.ny;
  z.lstride = (axis == 1) ? z.P : c.ny * z.P;
  z.bstride = (axis == 1) ? c.ny * z.P : z.P;
  const uint32_t nl = nl_z(len);
  const dim3 grid((z.nxc + nl - 1) / nl, lines, 1);
#define M(LEN) LAUNCH((k_zfused<LEN, Z_SHIFT>), grid, dim3((Geo<LEN, nl_z(LEN)>::THREADS)), z)
  KW_LEN_SWITCH(len, M)
#undef M
  return KW_OK;
}

// scaleSource body (KSpaceFirstOrderSolver.cpp:2346-2351): scaled <- ifftn(sourceKappa*fftn(scaled))/N, in place
kw_status kw_fused_scale_source(kw_ctx* ctx, float* scaled, const float* source_kappa_padded)
{
  KW_FUSED_READY(ctx);
  KW_PROF(ctx, "fused_scale_source");
  KW_REQUIRE(scaled && source_kappa_padded);
  float2** S = ctx->fused.s;
  const float* in1[1] = { scaled };
  KW_TRY(forward_xy(ctx, 1, in1));
  ZArgs z{};
  z.in[0] = S[0]; z.out[0] = S[0];
  z.op[0] = source_kappa_padded;
  KW_TRY(launch_zfused<Z_SOURCE>(ctx, 1, z));
  KW_TRY(inverse_y(ctx, 1));
  XinvArgs x{};
  x.in[0] = S[0];
  x.out[0] = scaled;
  KW_TRY(launch_xinv<EPI_STORE>(ctx, 1, x));
  return KW_OK;
}

// Pass-level probe for tuning (tools/probe_passes.py): launches ONE pass over the scratch arrays, no physics.
//   0: y-pass forward on s[0] (in place)          1: the same line kernel along z (stride ny*P) on s[0]
//   2: z-fused (forward, x sourceKappa-style multiply with op, inverse) on s[0]   3: y-pass on s[0..2] (3 arrays)
// `op` = a padded reduced real array (e.g. kappa).  Single rank only.
kw_status kw_fused_probe(kw_ctx* ctx, int which, const float* op)
{
  KW_FUSED_READY(ctx);
  auto& f = ctx->fused;
  KW_REQUIRE(!f.slab);
  const kw_constants& c = ctx->c;
  if (which == 0) return launch_ypass(ctx, -1, 1, f.s, f.s, false, false);
  if (which == 3) return launch_ypass(ctx, -1, 3, f.s, f.s, false, false);
  if (which == 1)
  {
    KW_REQUIRE(c.nz == c.ny); // same line-length template
    PassArgs a{};
    a.in[0] = f.s[0]; a.out[0] = f.s[0];
    a.tw  = f.tw[2];
    a.nxc = f.nxm;
    a.P   = f.P;
    a.narr = 1;
    a.ain = a.aout = RowAddr{0u, 0u, 0u, 1u, c.ny}; // element k of line (ky = blockIdx.y): row k*ny + ky
    const dim3 grid(f.P / nl_yz(c.nz), c.ny, 1);
#define M(LEN) LAUNCH((k_ypass<LEN, kFwd, false, false>), grid, dim3(Geo<LEN>::THREADS), a)
    KW_LEN_SWITCH(c.nz, M)
#undef M
    return KW_OK;
  }
  if (which == 10)
  {
    const size_t n4 = static_cast<size_t>(f.P) * c.ny * c.nz / 2;
    LAUNCH(k_probe_copy4, dim3(256 * 16), dim3(256), reinterpret_cast<float4*>(f.s[0]), n4);
    return KW_OK;
  }
  if (which >= 11 && which <= 14)
  { // 11/12: y-line tiles, 13/14: z-line tiles; odd: loads + stores only, even: plus the LDS exchange
    KW_REQUIRE(c.nz == c.ny && Fac<256>::R1 == Fac<256>::R2);
    PassArgs a{};
    a.out[0] = f.s[0];
    a.nxc = f.nxm;
    a.P   = f.P;
    a.ain = (which <= 12) ? RowAddr{0u, 0u, 0u, c.ny, 1u} : RowAddr{0u, 0u, 0u, 1u, c.ny};
    const dim3 grid(f.P / nl_yz(c.ny), c.nz, 1);
#define M(LEN)                                                                                                         \
  if (which & 1) LAUNCH((k_probe_tile<LEN, 0>), grid, dim3(Geo<LEN>::THREADS), a);                                    \
  else LAUNCH((k_probe_tile<LEN, 1>), grid, dim3(Geo<LEN>::THREADS), a)
    KW_LEN_SWITCH(c.ny, M)
#undef M
    return KW_OK;
  }
  if (which == 15 || which == 16)
  { // 15: y-line tiles, 16: z-line tiles, 32 columns wide
    KW_REQUIRE(c.nz == c.ny && c.ny == 256);
    PassArgs a{};
    a.out[0] = f.s[0];
    a.nxc = f.nxm;
    a.P   = f.P;
    a.ain = (which == 15) ? RowAddr{0u, 0u, 0u, c.ny, 1u} : RowAddr{0u, 0u, 0u, 1u, c.ny};
    LAUNCH((k_probe_tile_wide<256>), dim3((f.P + 31) / 32, c.nz, 1), dim3(256), a);
    return KW_OK;
  }
  if (which >= 30 && which <= 37)
  { // bit 0: z-lines instead of y-lines; bit 1: 256-B segments; bit 2: XCD-grouped tile order.  Lines of 256 or 512.
    KW_REQUIRE(c.nz == c.ny && (c.ny == 256 || c.ny == 512));
    const int w = which - 30;
    PassArgs a{};
    a.out[0] = f.s[0];
    a.nxc = f.nxm;
    a.P   = f.P;
    a.ain = (w & 1) ? RowAddr{0u, 0u, 0u, 1u, c.ny} : RowAddr{0u, 0u, 0u, c.ny, 1u};
    const int  vec = (w & 2) ? 2 : 1;
    const dim3 grid((f.P + 16 * vec - 1) / (16 * vec), c.nz, 1);
    KW_REQUIRE(!(w & 4) || (grid.x * grid.y) % 8 == 0);
#define PR(R, V, X) LAUNCH((k_probe_tile_rt<R, V, X>), grid, dim3(256), a)
    if (c.ny == 256)
    {
      if (w == 0 || w == 1) PR(16, 1, false); else if (w == 2 || w == 3) PR(16, 2, false);
      else if (w == 4 || w == 5) PR(16, 1, true); else PR(16, 2, true);
    }
    else
    {
      if (w == 0 || w == 1) PR(32, 1, false); else if (w == 2 || w == 3) PR(32, 2, false);
      else if (w == 4 || w == 5) PR(32, 1, true); else PR(32, 2, true);
    }
#undef PR
    return KW_OK;
  }
  if (which >= 20 && which <= 23)
  { // op doubles as the real arrays: needs 6 * (N + pad) floats (u x3, dt/rho0 x3); 21..23 stagger the arrays by a pad
    KW_REQUIRE(op != nullptr && c.nx == 256);
    XinvArgs a{};
    float* base = const_cast<float*>(op);
    static const size_t pads[4] = { 0, 1024, 17408, 263168 };
    const size_t N = static_cast<size_t>(c.nx) * c.ny * c.nz + pads[which - 20];
    for (int i = 0; i < 3; i++) { a.in[i] = f.s[i]; a.fout[i] = f.s[i]; a.out[i] = base + i * N; a.m0[i] = base + (3 + i) * N; }
    a.P = f.P;
    LAUNCH((k_probe_xinv<256>), dim3(c.ny * c.nz / (2 * nl_x(c.nx)), 3, 1), dim3(GeoX<256>::THREADS), a);
    return KW_OK;
  }
  if (which == 2)
  {
    KW_REQUIRE(op != nullptr);
    ZArgs z{};
    z.in[0] = f.s[0]; z.out[0] = f.s[0];
    z.op[0] = op;
    return launch_zfused<Z_SOURCE>(ctx, 1, z);
  }
  kw_set_error("kw_fused_probe: unknown probe %d", which);
  return KW_ERR_INVALID;
}

} // extern "C"
#endif // KW_FUSED_TU == 0
