/*
 * kwave_oracle.h — CPU oracle for the k-space first-order acoustic step.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ may be imported, linked or executed by the
 * product path (k-wave-fluid-cuda_amd/, include/).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, and only as the checker / the reported CPU baseline.
 *
 * Parity status: "parity unpinned" by the reference's own tests — the reference ships no tests,
 * golden vectors or fixtures and cannot be compiled here (needs nvcc/cuFFT).
 * The oracle is pinned instead by (1) an fp64 closed-form solution of the scheme (tests K1),
 * (2) an independent fp64 NumPy restatement (oracle/kwave_np.py, tests K4) and (3) for the
 * compression basis by the reference's own CompressHelper.cpp compiled into oracle/_ref/.
 *
 * Each function cites the reference file:line whose arithmetic it restates
 * (paths relative to /root/reference).
 */
#ifndef KWAVE_ORACLE_H
#define KWAVE_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Source modes: Parameters/Parameters.h:60-94 (SourceMode kDirichlet=0, kAdditiveNoCorrection=1, kAdditive=2). */
enum { KWO_SRC_DIRICHLET = 0, KWO_SRC_ADDITIVE_NO_CORRECTION = 1, KWO_SRC_ADDITIVE = 2 };
/* Reduce operators: OutputStreams/BaseOutputStream.h ReduceOperator. */
enum { KWO_OP_NONE = 0, KWO_OP_RMS = 1, KWO_OP_MAX = 2, KWO_OP_MIN = 3 };

/* Problem description = what the reference reads from the input file.
 * Arrays are row-major, x fastest; complex arrays interleaved (re,im); indices 0-based.
 * A NULL medium pointer means "scalar medium": the *_s member is used. */
typedef struct kwo_problem
{
  uint64_t nx, ny, nz;
  /* medium */
  const float* c0;          /* [N] or NULL */
  const float* rho0;        /* [N] or NULL; rho0_sg* must be arrays iff rho0 is (Parameters.cpp:449-459) */
  const float* rho0_sgx;
  const float* rho0_sgy;
  const float* rho0_sgz;
  const float* bona;        /* [N] or NULL */
  const float* alpha_coeff; /* [N] or NULL */
  /* k-space derivative operators, complex */
  const float* ddx_k_shift_pos; /* [nx/2+1] */
  const float* ddy_k_shift_pos; /* [ny] */
  const float* ddz_k_shift_pos; /* [nz] */
  const float* ddx_k_shift_neg;
  const float* ddy_k_shift_neg;
  const float* ddz_k_shift_neg;
  /* PML, real */
  const float* pml_x;     /* [nx] */
  const float* pml_y;
  const float* pml_z;
  const float* pml_x_sgx;
  const float* pml_y_sgy;
  const float* pml_z_sgz;
  /* sources */
  const float*    p0_source_input;   /* [N] or NULL */
  const uint64_t* p_source_index;    /* [p_source_n] */
  const float*    p_source_input;    /* [p_source_flag] or [p_source_flag][p_source_n] */
  const uint64_t* u_source_index;    /* [u_source_n] */
  const float*    ux_source_input;
  const float*    uy_source_input;
  const float*    uz_source_input;
  const float*    transducer_source_input;
  const uint64_t* delay_mask;        /* [u_source_n] */
  uint64_t p_source_n, u_source_n;
  /* "flags" are signal lengths: source active while flag > t (KSpaceFirstOrderSolver.cpp:2258,2314,894) */
  uint64_t p_source_flag, ux_source_flag, uy_source_flag, uz_source_flag, transducer_source_flag;
  uint64_t p0_source_flag;
  float dt, dx, dy, dz, c_ref, alpha_power;
  float c0_s, rho0_s, rho0_sgx_s, rho0_sgy_s, rho0_sgz_s, bona_s, alpha_coeff_s;
  int32_t nonlinear_flag, absorbing_flag;
  int32_t p_source_mode, p_source_many, u_source_mode, u_source_many;
  /* non-uniform grid (nonuniform_grid_flag; MatrixContainer.cpp:301-329): derivative scalings per axis, on the regular
   * and on the staggered grid; all NULL on a uniform grid */
  const float* dxudxn;      /* [nx] */
  const float* dyudyn;      /* [ny] */
  const float* dzudzn;      /* [nz] */
  const float* dxudxn_sgx;  /* [nx] */
  const float* dyudyn_sgy;  /* [ny] */
  const float* dzudzn_sgz;  /* [nz] */
} kwo_problem;

typedef struct kwo_sim kwo_sim;

/* pre-processing + allocation (KSpaceFirstOrderSolver.cpp:784-857) */
kwo_sim* kwo_create(const kwo_problem* prob);
void     kwo_destroy(kwo_sim* s);
/* one time step, A1..A12 of SURVEY Appendix A (KSpaceFirstOrderSolver.cpp:885-935 minus sampling) */
void     kwo_step(kwo_sim* s);
uint64_t kwo_time_index(const kwo_sim* s);
/* field access: names "p","ux","uy","uz","rhox","rhoy","rhoz","duxdx","duydy","duzdz",
 * "kappa","nabla1","nabla2","source_kappa","tau","eta","c2","dtrho0sgx","dtrho0sgy","dtrho0sgz" */
float*   kwo_field(kwo_sim* s, const char* name);
/* scalar access for homogeneous parameters: "tau","eta","c2","dtrho0sgx",... */
float    kwo_scalar(const kwo_sim* s, const char* name);

/* stand-alone unnormalised 3-D transforms (contract of cuFFT R2C/C2R: CufftComplexMatrix.cpp:82-130,508-534) */
void kwo_fft_r2c_3d(const float* in, float* out, uint64_t nx, uint64_t ny, uint64_t nz);
void kwo_fft_c2r_3d(const float* in, float* out, uint64_t nx, uint64_t ny, uint64_t nz);
/* 1-D transforms along one axis (0=x,1=y,2=z) of a 3-D array (CufftComplexMatrix.cpp:540-692) */
void kwo_fft_r2c_1d(const float* in, float* out, uint64_t nx, uint64_t ny, uint64_t nz, int axis);
void kwo_fft_c2r_1d(const float* in, float* out, uint64_t nx, uint64_t ny, uint64_t nz, int axis);

/* sampling (OutputStreamsCudaKernels.cu:83-107,164-230,297-316,359-365) */
void kwo_sample_index(int op, float* buf, const float* src, const uint64_t* mask, uint64_t n);
void kwo_sample_cuboid(int op, float* buf, const float* src, const uint32_t tl[3], const uint32_t br[3],
                       const uint32_t size[3], uint64_t n);
void kwo_sample_all(int op, float* buf, const float* src, uint64_t n);
void kwo_post_rms(float* buf, float scale, uint64_t n);

/* shifted (non-staggered) velocity (KSpaceFirstOrderSolver.cpp:2714-2735; SolverCudaKernels.cu:2617-2689) */
void kwo_shifted_velocity(const float* u, float* out, const float* shift_neg_r, uint64_t nx, uint64_t ny,
                          uint64_t nz, int axis);

/* compression basis (Compression/CompressHelper.cpp:672-778): bE, bE_1 each [harmonics*bSize] complex.
 * shifted!=0 gives the velocity-stream variants (phase-shifted by half a step, :740-743). */
void kwo_compress_basis(float period, uint64_t mos, uint64_t harmonics, int shifted, float* bE, float* bE_1);
uint64_t kwo_compress_osize(float period, uint64_t mos);
uint64_t kwo_compress_bsize(float period, uint64_t mos);
/* one sampled step of the compression accumulation (IndexOutputStream.cpp:373-470) */
typedef struct kwo_compress_state
{
  uint64_t n_sens, harmonics, o_size, b_size;
  uint64_t sampled_step;     /* steps seen so far */
  uint64_t compressed_step;  /* frames emitted so far */
  int32_t  no_overlap;
  int32_t  pad_;
  float*   c1;               /* [n_sens*harmonics] complex */
  float*   c2;
} kwo_compress_state;
/* returns 1 and writes frame_out ([n_sens*harmonics] complex) when a frame is emitted, else 0 */
int kwo_compress_step(kwo_compress_state* st, const float* bE, const float* bE_1, const float* x,
                      int is_last_step, float* frame_out);

#ifdef __cplusplus
}
#endif
#endif
