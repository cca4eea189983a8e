/*
 * kwave_hip.h — C-ABI of the MI355X (gfx950) device layer for the k-space first-order acoustic step.
 *
 * This is the drop-in boundary for the per-time-step hot path of klepo/k-Wave-Fluid-CUDA.  The reference
 * has no FFI layer; its device-operator boundary is four C++ surfaces:
 *   (1) namespace SolverCudaKernels            KSpaceSolver/SolverCudaKernels.cuh:52-500
 *   (2) class CufftComplexMatrix (plans+exec)   MatrixClasses/CufftComplexMatrix.h:73-238
 *   (3) namespace OutputStreamsCudaKernels     OutputStreams/OutputStreamsCudaKernels.cuh:47-106
 *   (4) CudaParameters / CudaDeviceConstants   Parameters/CudaParameters.h:140-146, CudaDeviceConstants.cuh:44-116
 * plus the memory verbs of the matrix classes   MatrixClasses/BaseFloatMatrix.h:85-131.
 * Every entry point below names the reference interface (file:line, relative to the reference root) it replaces.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; device pointers are raw HIP device addresses.
 *   - every function returns kw_status (0 = OK).  The reference's convention (void + C++ exception,
 *     Logger/Logger.h:194-216) is restored by the C++ host shims in k-wave-fluid-cuda_amd/host/, which turn
 *     a non-zero status into std::runtime_error(kw_last_error()).
 *   - all launches are asynchronous on the context's stream (reference: default stream, in order).
 *   - data layout: fp32, row-major with x fastest; complex = interleaved (re,im); indices = uint64, 0-based
 *     (KSpaceFirstOrderSolver.cpp:2916-2920, ComplexMatrix.cpp:124-135, IndexMatrix.cpp:161-168).
 *   - a NULL pointer for an optional per-voxel medium array selects the scalar from kw_constants, exactly
 *     like the reference's boolean template flags (e.g. SolverCudaKernels.cu:1401-1440).
 */
#ifndef KWAVE_HIP_H
#define KWAVE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KW_API __attribute__((visibility("default")))

typedef struct kw_ctx kw_ctx;

typedef enum kw_status
{
  KW_OK            = 0,
  KW_ERR_INVALID   = 1, /* bad argument (NULL pointer, zero size, unsupported flag) */
  KW_ERR_HIP       = 2, /* HIP runtime error ("GPU error: ..." of Logger/ErrorMessages.h:331) */
  KW_ERR_FFT       = 3, /* rocFFT error (CufftComplexMatrix.cpp:63-72,706-720) */
  KW_ERR_ALLOC     = 4, /* out of memory (std::bad_alloc of BaseFloatMatrix.cpp:140-143) */
  KW_ERR_STATE     = 5, /* call order violated (constants / plans not set) */
  KW_ERR_NO_DEVICE = 6, /* no usable gfx950 device (CudaParameters.cpp:81-177) */
  KW_ERR_COMM      = 7  /* multi-GPU exchange failed: RCCL error, or a caller-supplied exchange callback returned non-zero */
} kw_status;

/* Parameters/Parameters.h:60-94 */
typedef enum kw_source_mode
{
  KW_SRC_DIRICHLET              = 0,
  KW_SRC_ADDITIVE_NO_CORRECTION = 1,
  KW_SRC_ADDITIVE               = 2
} kw_source_mode;

/* OutputStreams/BaseOutputStream.h ReduceOperator (device-side subset used by the samplers) */
typedef enum kw_reduce_op
{
  KW_OP_NONE = 0, /* buf[i]  = src      */
  KW_OP_RMS  = 1, /* buf[i] += src*src  */
  KW_OP_MAX  = 2, /* buf[i]  = max(buf[i], src) */
  KW_OP_MIN  = 3  /* buf[i]  = min(buf[i], src) */
} kw_reduce_op;

/* Mirror of struct CudaDeviceConstants (Parameters/CudaDeviceConstants.cuh:44-116), filled the way
 * CudaParameters::setUpDeviceConstants does (Parameters/CudaParameters.cpp:238-288).  Passed to kernels as a
 * by-value argument (SGPR-resident) instead of a __constant__ symbol. */
typedef struct kw_constants
{
  uint32_t nx, ny, nz, n_elements;
  uint32_t nx_complex, ny_complex, nz_complex, n_elements_complex;
  float    fft_divider, fft_divider_x, fft_divider_y, fft_divider_z;
  float    dt, dt_by_2, c2;
  float    rho0, dt_rho0, dt_rho0_sgx, dt_rho0_sgy, dt_rho0_sgz;
  float    b_on_a, absorb_tau, absorb_eta;
  uint32_t velocity_source_size, velocity_source_mode, velocity_source_many;
  uint32_t pressure_source_size, pressure_source_mode, pressure_source_many;
} kw_constants;

typedef struct kw_device_info
{
  char     name[128];
  char     arch[64];
  int32_t  device_id;
  int32_t  compute_units;
  int32_t  wavefront_size;
  int32_t  clock_mhz;
  uint64_t total_mem;
  uint64_t free_mem;
  uint64_t lds_per_cu;
  uint64_t l2_bytes;
} kw_device_info;

/* ------------------------------------------------------------------------------------------------------------------
 * Context / device  — replaces CudaParameters::selectDevice (CudaParameters.cpp:81-177)
 * ---------------------------------------------------------------------------------------------------------------- */
/* device_id < 0: first free device (reference: -g not given).  Creates the context's stream. */
KW_API kw_status   kw_init(int device_id, kw_ctx** out_ctx);
KW_API kw_status   kw_destroy(kw_ctx* ctx);
/* message of the last failing call on this thread (ctx may be NULL) */
KW_API const char* kw_last_error(void);
KW_API kw_status   kw_device_info_get(kw_ctx* ctx, kw_device_info* out);
/* use an externally owned hipStream_t (e.g. torch's current stream); NULL restores the context's own stream */
KW_API kw_status   kw_set_stream(kw_ctx* ctx, void* hip_stream);
KW_API void*       kw_get_stream(kw_ctx* ctx);
KW_API kw_status   kw_sync(kw_ctx* ctx);
/* Schedule parameters of the fused pipeline and of the multi-GPU exchange — every switch the library has, in one
 * documented place (the library reads no environment variable).  kw_get_tuning returns the current values (the defaults
 * after kw_init); kw_set_tuning takes effect at the next kw_fused_create.  struct_bytes = sizeof(kw_tuning) of the
 * caller, so that the struct can grow: fields beyond it keep their defaults. */
typedef struct kw_tuning
{
  uint32_t struct_bytes;
  int32_t  side_array;          /* 1: x-Nyquist bins of an even-Nx grid live in a compact side array behind rows of
                                   exactly Nx/2 bins; 0: in the rows, padded to whole 16-bin tiles */
  int32_t  tail_chunks;         /* single GPU: the plane-local tail of a stage (y-inverse, x-inverse + epilogue, chained
                                   x / y forward) runs per chunk of planes (a chunk's spectra stay in the Infinity
                                   Cache between those kernels); 0 or 1 = whole grid (the default: measured within
                                   +-2 % at 320^3 ... 512^3, profiles/r03_tail_chunks.txt), n = that many chunks */
  int32_t  split512;            /* 1: 512-point y / z lines as 2 x 256-point transforms; 0: 16 x 32-point kernels */
  int32_t  slab_pipeline;       /* 1: pipelined slab schedule (third buffer set, forward transposes started by the
                                   producing stage); 0: whole-array schedule */
  int32_t  slab_chunks;         /* plane chunks per array of the pipelined slab tail (1..4) */
  int32_t  slab_batch;          /* all arrays of a stage in ONE exchange per direction: -1 automatic (below 4 MB per
                                   peer and array), 0 never, 1 always */
  int32_t  p2p_blocks_per_peer; /* P2P transport: workgroups that store to one peer (1..8) */
  float    p2p_timeout_s;       /* P2P transport: a rank that waits longer than this for a peer gives up (KW_ERR_COMM) */
  int32_t  plane_kernels;       /* 1: grids with square planes of 32 / 64 points run each stage's tail (y-inverse,
                                   x-inverse + epilogue, chained x / y forward) as ONE launch whose blocks take whole
                                   z-planes; 0: the three-launch form of the larger grids */
} kw_tuning;
KW_API kw_status   kw_get_tuning(kw_ctx* ctx, kw_tuning* out);
KW_API kw_status   kw_set_tuning(kw_ctx* ctx, const kw_tuning* tuning);
/* Launch-bound loops (small grids): record the launches of a fixed sequence of kw_* calls once, replay it per step.
 * kw_graph_begin puts the context's stream into capture mode (nothing executes until the graph is launched);
 * every kw_* call in between must be a pure kernel-launch call (no allocation, copy or synchronisation). */
typedef struct kw_graph kw_graph;
KW_API kw_status   kw_graph_begin(kw_ctx* ctx);
KW_API kw_status   kw_graph_end(kw_ctx* ctx, kw_graph** out);
KW_API kw_status   kw_graph_launch(kw_ctx* ctx, kw_graph* g);
KW_API kw_status   kw_graph_destroy(kw_ctx* ctx, kw_graph* g);
/* Measured device-copy bandwidth: a float4 copy kernel over two freshly allocated buffers of `bytes` each (choose
 * them well above the 256 MB Infinity Cache), reps timed passes after one untimed; GB/s counts read + write.  bench.py
 * reports it beside the spec peak. */
KW_API kw_status   kw_measure_copy_bandwidth(kw_ctx* ctx, size_t bytes, int reps, double* out_gbs);
/* HIP events on the context's stream (bench.py times kernels with these) */
KW_API kw_status   kw_event_create(kw_ctx* ctx, void** out_event);
KW_API kw_status   kw_event_record(kw_ctx* ctx, void* event);
KW_API kw_status   kw_event_synchronize(kw_ctx* ctx, void* event);
KW_API kw_status   kw_event_elapsed_ms(kw_ctx* ctx, void* start, void* stop, float* out_ms);
KW_API kw_status   kw_event_destroy(kw_ctx* ctx, void* event);

/* Per-entry-point device timing (HIP events around every kw_* compute call while enabled); used by bench.py to
 * report kernel durations live.  The reference has only host wall-clock timers (Utils/TimeMeasure.h:90-123). */
typedef struct kw_profile_entry
{
  char     name[64];
  uint64_t calls;
  double   total_ms;
} kw_profile_entry;
KW_API kw_status kw_profile_enable(kw_ctx* ctx, int on);
KW_API int       kw_profile_enabled(kw_ctx* ctx); /* 1 while per-call timing is on (recorded graphs would bypass it) */
/* synchronises, aggregates by entry-point name, clears the recorded events; *n_out = number of entries written */
KW_API kw_status kw_profile_collect(kw_ctx* ctx, kw_profile_entry* out, size_t capacity, size_t* n_out);

/* ------------------------------------------------------------------------------------------------------------------
 * Memory verbs — replace BaseFloatMatrix/BaseIndexMatrix allocate/copyToDevice/copyFromDevice/zeroDeviceMatrix
 * (MatrixClasses/BaseFloatMatrix.cpp:77-80,124-168; BaseIndexMatrix.cpp)
 * ---------------------------------------------------------------------------------------------------------------- */
KW_API kw_status kw_malloc(kw_ctx* ctx, size_t bytes, void** out_dptr);
KW_API kw_status kw_free(kw_ctx* ctx, void* dptr);
KW_API kw_status kw_memcpy_h2d(kw_ctx* ctx, void* dst, const void* src, size_t bytes);
KW_API kw_status kw_memcpy_d2h(kw_ctx* ctx, void* dst, const void* src, size_t bytes);
KW_API kw_status kw_memcpy_d2d(kw_ctx* ctx, void* dst, const void* src, size_t bytes);
KW_API kw_status kw_memcpy_d2h_async(kw_ctx* ctx, void* dst, const void* src, size_t bytes);
KW_API kw_status kw_memset(kw_ctx* ctx, void* dptr, int value, size_t bytes);
/* device -> pinned host on the context's COPY stream, ordered after everything enqueued so far on the compute stream;
 * `event` (kw_event_create) is recorded when the copy has landed.  The compute stream is not blocked: this is how raw
 * sensor series leave the device while the next step computes (reference: zero-copy mapped buffer + cudaEvent,
 * BaseOutputStream.cpp:369-388, IndexOutputStream.cpp:263,354). */
KW_API kw_status kw_memcpy_d2h_overlapped(kw_ctx* ctx, void* dst, const void* src, size_t bytes, void* event);
/* pinned host memory (reference: cudaHostRegister / mapped buffers, BaseOutputStream.cpp:369-388) */
KW_API kw_status kw_host_alloc(kw_ctx* ctx, size_t bytes, void** out_hptr);
KW_API kw_status kw_host_free(kw_ctx* ctx, void* hptr);

/* ------------------------------------------------------------------------------------------------------------------
 * Device constants — replaces CudaDeviceConstants::uploadDeviceConstants (CudaDeviceConstants.cu:58-60)
 * and CudaParameters::setKernelConfiguration (CudaParameters.cpp:195-232; geometry is derived internally
 * from the CU count instead of SM count x 8).
 * ---------------------------------------------------------------------------------------------------------------- */
KW_API kw_status kw_set_constants(kw_ctx* ctx, const kw_constants* constants);
KW_API kw_status kw_get_constants(kw_ctx* ctx, kw_constants* out);

/* ------------------------------------------------------------------------------------------------------------------
 * FFT — replaces CufftComplexMatrix static plan factory + compute* members
 * (MatrixClasses/CufftComplexMatrix.cpp:82-130 plans ND, :144-426 plans 1D, :432-502 destroy, :508-534 exec ND,
 *  :540-692 exec 1D).  Unnormalised, forward sign -i, out-of-place, nx/2+1 bins along x.  Dimensions come from
 * kw_set_constants.  C2R may overwrite its input (as cuFFT may).
 * ---------------------------------------------------------------------------------------------------------------- */
KW_API kw_status kw_fft_create_plans_3d(kw_ctx* ctx);                         /* createR2CFftPlanND + createC2RFftPlanND */
KW_API kw_status kw_fft_create_plans_1d(kw_ctx* ctx, int axis);               /* create{R2C,C2R}FftPlan1D{X,Y,Z} */
KW_API kw_status kw_fft_destroy_plans(kw_ctx* ctx);                           /* destroyAllPlansAndStaticData */
KW_API kw_status kw_fft_r2c_3d(kw_ctx* ctx, const float* in, float* out);     /* computeR2CFftND */
KW_API kw_status kw_fft_c2r_3d(kw_ctx* ctx, float* in, float* out);           /* computeC2RFftND */
/* 1-D transforms along axis (0=x,1=y,2=z); the half-spectrum keeps the [z][y][x] order with the transformed
 * axis shortened to n/2+1 (the reference's transposed/padded layout for Y and Z is an implementation detail) */
KW_API kw_status kw_fft_r2c_1d(kw_ctx* ctx, int axis, const float* in, float* out);  /* computeR2CFft1D{X,Y,Z} */
KW_API kw_status kw_fft_c2r_1d(kw_ctx* ctx, int axis, float* in, float* out);        /* computeC2RFft1D{X,Y,Z} */

/* ------------------------------------------------------------------------------------------------------------------
 * Solver kernels — one entry per SolverCudaKernels wrapper (3-D, uniform grid)
 * ---------------------------------------------------------------------------------------------------------------- */
/* computeVelocityHeterogeneous (.cuh:92, .cu:184-243) when dt_rho0_sg* != NULL,
 * computeVelocityHomogeneousUniform (.cuh:107, .cu:278-335) when all three are NULL */
KW_API kw_status kw_compute_velocity(kw_ctx* ctx, float* ux_sgx, float* uy_sgy, float* uz_sgz, const float* ifft_x,
                                     const float* ifft_y, const float* ifft_z, const float* dt_rho0_sgx,
                                     const float* dt_rho0_sgy, const float* dt_rho0_sgz, const float* pml_x_sgx,
                                     const float* pml_y_sgy, const float* pml_z_sgz);
/* addTransducerSource (.cuh:132, .cu:463-497) */
KW_API kw_status kw_add_transducer_source(kw_ctx* ctx, float* ux_sgx, const uint64_t* velocity_source_index,
                                          const float* transducer_source_input, const uint64_t* delay_mask,
                                          uint64_t time_index);
/* addVelocitySource (.cuh:141-143, .cu:504-555) */
KW_API kw_status kw_add_velocity_source(kw_ctx* ctx, float* velocity, const float* velocity_source_input,
                                        const uint64_t* velocity_source_index, uint64_t time_index);
/* addPressureSource (.cuh:152, .cu:570-660) */
KW_API kw_status kw_add_pressure_source(kw_ctx* ctx, float* rho_x, float* rho_y, float* rho_z,
                                        const float* pressure_source_input, const uint64_t* pressure_source_index,
                                        uint64_t time_index);
/* insertSourceIntoScalingMatrix (.cuh:163-166, .cu:679-733) */
KW_API kw_status kw_insert_source_into_scaling_matrix(kw_ctx* ctx, float* scaled_source, const float* source_input,
                                                      const uint64_t* source_index, uint64_t source_size,
                                                      int many_flag, uint64_t time_index);
/* computeSourceGradient (.cuh:173-174, .cu:740-758) */
KW_API kw_status kw_compute_source_gradient(kw_ctx* ctx, float* source_spectrum, const float* source_kappa);
/* addVelocityScaledSource (.cuh:181-182, .cu:765-786) */
KW_API kw_status kw_add_velocity_scaled_source(kw_ctx* ctx, float* velocity, const float* scaled_source);
/* addPressureScaledSource (.cuh:191-192, .cu:795-826) */
KW_API kw_status kw_add_pressure_scaled_source(kw_ctx* ctx, float* rho_x, float* rho_y, float* rho_z,
                                               const float* scaled_source);
/* addInitialPressureSource (.cuh:208, .cu:864-927); c2 == NULL -> scalar */
KW_API kw_status kw_add_initial_pressure_source(kw_ctx* ctx, float* p, float* rho_x, float* rho_y, float* rho_z,
                                                const float* p0_source_input, const float* c2);
/* computeInitialVelocityHeterogeneous / HomogeneousUniform (.cuh:222,236, .cu:949-1040) */
KW_API kw_status kw_compute_initial_velocity(kw_ctx* ctx, float* ux_sgx, float* uy_sgy, float* uz_sgz,
                                             const float* dt_rho0_sgx, const float* dt_rho0_sgy,
                                             const float* dt_rho0_sgz);
/* computePressureGradient (.cuh:267, .cu:1139-1185) */
KW_API kw_status kw_compute_pressure_gradient(kw_ctx* ctx, float* fft_x, float* fft_y, float* fft_z,
                                              const float* kappa, const float* ddx_k_shift_pos,
                                              const float* ddy_k_shift_pos, const float* ddz_k_shift_pos);
/* computeVelocityGradient (.cuh:281, .cu:1210-1268) */
KW_API kw_status kw_compute_velocity_gradient(kw_ctx* ctx, float* fft_x, float* fft_y, float* fft_z,
                                              const float* kappa, const float* ddx_k_shift_neg,
                                              const float* ddy_k_shift_neg, const float* ddz_k_shift_neg);
/* computeVelocityGradientShiftNonuniform (.cuh:293, .cu:1285-1320): du?d? *= d?ud?n[coord] on a non-uniform grid */
KW_API kw_status kw_compute_velocity_gradient_shift_nonuniform(kw_ctx* ctx, float* duxdx, float* duydy, float* duzdz,
                                                               const float* dxudxn, const float* dyudyn,
                                                               const float* dzudzn);
/* computeDensityNonlinear (.cuh:306, .cu:1358-1440) / computeDensityLinear (.cuh:320, .cu:1470-1545); rho0 NULL -> scalar */
KW_API kw_status kw_compute_density_nonlinear(kw_ctx* ctx, float* rho_x, float* rho_y, float* rho_z,
                                              const float* pml_x, const float* pml_y, const float* pml_z,
                                              const float* duxdx, const float* duydy, const float* duzdz,
                                              const float* rho0);
KW_API kw_status kw_compute_density_linear(kw_ctx* ctx, float* rho_x, float* rho_y, float* rho_z, const float* pml_x,
                                           const float* pml_y, const float* pml_z, const float* duxdx,
                                           const float* duydy, const float* duzdz, const float* rho0);
/* computePressureTermsNonlinear (.cuh:335-338, .cu:1577-1695); b_on_a / rho0 NULL -> scalar */
KW_API kw_status kw_compute_pressure_terms_nonlinear(kw_ctx* ctx, float* density_sum, float* nonlinear_term,
                                                     float* velocity_gradient_sum, const float* rho_x,
                                                     const float* rho_y, const float* rho_z, const float* duxdx,
                                                     const float* duydy, const float* duzdz, const float* b_on_a,
                                                     const float* rho0);
/* computePressureTermsLinear (.cuh:348-350, .cu:1724-1790) */
KW_API kw_status kw_compute_pressure_terms_linear(kw_ctx* ctx, float* density_sum, float* velocity_gradient_sum,
                                                  const float* rho_x, const float* rho_y, const float* rho_z,
                                                  const float* duxdx, const float* duydy, const float* duzdz,
                                                  const float* rho0);
/* computeAbsorbtionTerm (.cuh:365-368, .cu:1812-1840) */
KW_API kw_status kw_compute_absorbtion_term(kw_ctx* ctx, float* fft_part1, float* fft_part2,
                                            const float* absorb_nabla1, const float* absorb_nabla2);
/* sumPressureTermsNonlinear (.cuh:388-391, .cu:1865-1945); c2 / absorb_tau+absorb_eta NULL -> scalars */
KW_API kw_status kw_sum_pressure_terms_nonlinear(kw_ctx* ctx, float* p, const float* nonlinear_term,
                                                 const float* absorb_tau_term, const float* absorb_eta_term,
                                                 const float* c2, const float* absorb_tau, const float* absorb_eta);
/* sumPressureTermsLinear (.cuh:411-414, .cu:1966-2045) */
KW_API kw_status kw_sum_pressure_terms_linear(kw_ctx* ctx, float* p, const float* absorb_tau_term,
                                              const float* absorb_eta_term, const float* density_sum, const float* c2,
                                              const float* absorb_tau, const float* absorb_eta);
/* sumPressureNonlinearLossless (.cuh:429, .cu:2067-2200) */
KW_API kw_status kw_sum_pressure_nonlinear_lossless(kw_ctx* ctx, float* p, const float* rho_x, const float* rho_y,
                                                    const float* rho_z, const float* c2, const float* b_on_a,
                                                    const float* rho0);
/* sumPressureLinearLossless (.cuh:444, .cu:2224-2275) */
KW_API kw_status kw_sum_pressure_linear_lossless(kw_ctx* ctx, float* p, const float* rho_x, const float* rho_y,
                                                 const float* rho_z, const float* c2);
/* computeVelocityShiftInX/Y/Z (.cuh:482-499, .cu:2617-2710); spectrum in the layout of kw_fft_r2c_1d(axis) */
KW_API kw_status kw_compute_velocity_shift(kw_ctx* ctx, int axis, float* spectrum, const float* shift_neg_r);

/* ------------------------------------------------------------------------------------------------------------------
 * Fused spectral pipeline (MI355X fast path; csrc/kw_fused.hip).  Each entry computes one whole stage of the step —
 * FFTs, spectral multiply and the real-space update — with hand-written FFT passes, so the gradients and spectra never
 * make an HBM round trip as separate arrays.  Same arithmetic as the kernels cited; supported when each of Nx, Ny, Nz
 * is one of 16 32 48 64 72 80 96 100 108 120 128 144 160 192 200 216 240 256 288 300 320 324 384 400 432 480 500 512
 * 576 600 640 648 768 1024 (kw_fused_supported says;
 * other grids use the rocFFT entry points above).  kappa / nabla / sourceKappa must first be imported into the
 * pipeline's padded row layout.
 * ---------------------------------------------------------------------------------------------------------------- */
/* Multi-GPU (new with this build; the reference is single-GPU, Readme.md:12-13): Z-slab decomposition with one
 * all-to-all transpose per 3-D FFT.  A context in slab mode holds nz = nz_global/nranks planes of every real array
 * (kw_set_constants gets the LOCAL nz; fft_divider stays 1/(nx*ny*nz_global)) and, in k-space, ny/nranks rows with all
 * nz_global planes ("transposed" layout [nz_global][ny/nranks][P]); reduced operators (kappa, nabla, sourceKappa) are
 * supplied in that transposed layout.
 *
 * The exchange is an all-to-all of equal contiguous chunks (chunk q of `send` goes to rank q; chunk q of `recv` comes
 * from rank q).  Default: the library's own RCCL path — kw_comm_init gives the context a communicator and a
 * communication stream; every exchange is ncclGroupStart / ncclSend + ncclRecv per peer / ncclGroupEnd on that stream,
 * ordered against the context's stream by events (split-phase: the transpose of one array is in flight while the
 * passes of the others run); no host code but the enqueueing runs per exchange.
 *   rank 0: kw_comm_unique_id(id) -> distribute the KW_COMM_ID_BYTES bytes to all ranks by any means (file, socket,
 *   MPI, a torch.distributed store) -> every rank: kw_comm_init(ctx, nranks, rank, id)   [collective, blocks]
 *   -> kw_fused_set_slab(ctx, nranks, rank, nz_global, NULL, NULL) -> kw_fused_create(ctx).
 * One rank with a communicator runs the slab path exchanging with itself (rehearsal on a one-GPU machine). */
#define KW_COMM_ID_BYTES 128
KW_API kw_status kw_comm_unique_id(void* out_id, size_t bytes);  /* ncclGetUniqueId; bytes >= KW_COMM_ID_BYTES */
KW_API kw_status kw_comm_init(kw_ctx* ctx, uint32_t nranks, uint32_t rank, const void* unique_id);
KW_API kw_status kw_comm_destroy(kw_ctx* ctx);                    /* also done by kw_destroy */
KW_API kw_status kw_comm_info(kw_ctx* ctx, uint32_t* nranks, uint32_t* rank, uint64_t* exchanges_started);
/* the same two calls with the RCCL library named by the caller (a path or soname for dlopen; NULL = the default search:
 * the process's own librccl.so.1, then ROCm's) — e.g. a site build of RCCL, or a test double */
KW_API kw_status kw_comm_unique_id_from(const char* rccl_library, void* out_id, size_t bytes);
KW_API kw_status kw_comm_init_with(kw_ctx* ctx, const char* rccl_library, uint32_t nranks, uint32_t rank, const void* unique_id);
/* Device-initiated transport ("P2P"): the ranks map each other's exchange buffers — hipIpc handles between the
 * processes of one node, plain pointers between threads of one process — and every exchange is ONE small kernel on the
 * communication stream that stores this rank's chunks straight into the peers' receive buffers over xGMI, with a
 * credit / full flag rendezvous per peer (csrc/kw_comm.hip k_p2p_exchange).  Same split-phase contract and the same
 * schedules as the RCCL path; what it removes is the per-exchange fixed cost of a library group (launching thread and
 * communication kernel), which is what bounds small and chunked exchanges.  Single node.  Call order on every rank:
 *   kw_comm_init_p2p(ctx, nranks, rank)                 (with or without a prior kw_comm_init)
 *   kw_fused_set_slab(ctx, nranks, rank, nz_global, NULL, NULL);  kw_fused_create(ctx)
 *   kw_comm_p2p_export(ctx, blob, KW_COMM_P2P_BLOB_BYTES)          what this rank publishes: plain bytes
 *   < the caller gathers the blobs of all ranks, in rank order, by any means it has: MPI, a torch.distributed store,
 *     files — all_blobs = nranks * KW_COMM_P2P_BLOB_BYTES bytes, the same on every rank >
 *   kw_comm_p2p_connect(ctx, all_blobs)                  maps the peers' buffers; exchanges go P2P from here on
 * A rank that waits longer than kw_tuning::p2p_timeout_s for a peer stops waiting: the next exchange (or kw_sync of the
 * step loop) returns KW_ERR_COMM — a lost peer ends the run, it does not hang the GPU queue. */
#define KW_COMM_P2P_BLOB_BYTES 1024
KW_API kw_status kw_comm_init_p2p(kw_ctx* ctx, uint32_t nranks, uint32_t rank);
KW_API kw_status kw_comm_p2p_export(kw_ctx* ctx, void* blob, size_t bytes);
KW_API kw_status kw_comm_p2p_connect(kw_ctx* ctx, const void* all_blobs);
/* Link model for schedule studies on a one-GPU machine (tools/emulate_rank.py): ONE rank of an nranks run alone on the
 * GPU — the chunks for the absent peers are copied locally and every peer's transfer is held for latency_us +
 * bytes / link_gbs, as a link of that rate would.  In place of export / connect.  Results are not a simulation's. */
KW_API kw_status kw_comm_p2p_emulate(kw_ctx* ctx, float link_gbs, float latency_us);
/* -1 no communicator, 0 RCCL, 1 P2P not yet connected, 2 P2P, 3 P2P link model */
KW_API kw_status kw_comm_transport(kw_ctx* ctx, int* out_transport);
/* Override: a caller that owns its own communicator passes exchange(user, send, recv, bytes_per_peer), which must be
 * ordered after all prior work on the context's stream and complete (or stream-ordered) before it returns — e.g.
 * torch.distributed.all_to_all_single, or a host-staged all-to-all when several ranks share one GPU (tests).  The
 * callbacks return 0 on success; anything else aborts the step with KW_ERR_COMM. */
typedef int (*kw_exchange_fn)(void* user, void* send, void* recv, size_t bytes_per_peer);
KW_API kw_status kw_fused_set_slab(kw_ctx* ctx, uint32_t nranks, uint32_t rank, uint32_t nz_global,
                                   kw_exchange_fn exchange, void* user);   /* before kw_fused_create; NULL = RCCL path */
/* Optional split-phase form of the same all-to-all, so that transposes overlap with compute: start(user, send, recv,
 * bytes_per_peer, slot) begins the exchange (ordered after the work enqueued so far on the context's stream) and
 * returns; wait(user, slot) makes the context's stream wait for that exchange (slot in 0..5, one exchange in flight per
 * slot).  E.g. all_to_all_single(async_op=True) / work.wait().  Without it the blocking callback is used.
 * A spectral array travels as two pieces when its x-Nyquist bins are kept in the side array (even Nx, the default):
 * the rows on slot 0..2 and the side array on slot + 3 — the callbacks see two exchanges per array, the library's own
 * RCCL path puts both pieces into one group. */
typedef int (*kw_exchange_start_fn)(void* user, void* send, void* recv, size_t bytes_per_peer, int slot);
typedef int (*kw_exchange_wait_fn)(void* user, int slot);
KW_API kw_status kw_fused_set_slab_async(kw_ctx* ctx, kw_exchange_start_fn start, kw_exchange_wait_fn wait);
/* Strided form, for a caller-owned transport that can move PIECES of the arrays: for every peer q, `bytes` bytes at
 * send + q * stride_bytes + offset_bytes go to rank q and land at recv + (sender) * stride_bytes + offset_bytes
 * (stride == bytes, offset 0 is the plain all-to-all above).  start returns after beginning the exchange (or after
 * completing it, with wait == NULL); wait(user, slot) orders the context's stream after it; slot < 256.  With this form —
 * as with the library's own RCCL path — the pipeline runs its pipelined schedule: a third buffer set for the transposed
 * spectra, the forward transposes of the next stage started by the producing stage, small messages batched into one
 * exchange per stage and direction, and with KW_SLAB_CHUNKS=2..4 the plane-local tail of every stage (y-inverse,
 * x-inverse + epilogue, chained forward x / y) run per chunk of planes while the other chunks are on the wire
 * (KW_SLAB_PIPELINE=0: whole-array schedule). */
typedef int (*kw_exchange_piece_fn)(void* user, void* send, void* recv, size_t stride_bytes, size_t offset_bytes, size_t bytes,
                                    int slot);
KW_API kw_status kw_fused_set_slab_pieces(kw_ctx* ctx, kw_exchange_piece_fn start, kw_exchange_wait_fn wait);
KW_API kw_status kw_fused_scratch_bytes(kw_ctx* ctx, size_t* out_bytes_per_array);
/* like kw_fused_create but with caller-owned scratch (each of kw_fused_scratch_bytes bytes): s[3], and t[3] when
 * nranks > 1 — lets the caller register the buffers with its communication library */
KW_API kw_status kw_fused_create_with_scratch(kw_ctx* ctx, void* const s[3], void* const t[3]);
KW_API kw_status kw_fused_supported(kw_ctx* ctx, int* out_supported);
KW_API kw_status kw_fused_create(kw_ctx* ctx);   /* scratch + twiddles; needs kw_set_constants */
KW_API kw_status kw_fused_destroy(kw_ctx* ctx);
KW_API kw_status kw_fused_reduced_elems(kw_ctx* ctx, size_t* out_floats);  /* floats of one imported reduced array */
/* reduced real operator (kappa, nabla1/2, sourceKappa: [nz][ny][nx/2+1]; slab mode: the transposed [nz_global][ny/P][..])
 * -> the pipeline's private tile-blocked layout [ky][kx tile][kz][16] ("*_padded" arguments below) */
KW_API kw_status kw_fused_import_reduced(kw_ctx* ctx, float* dst_padded, const float* src_reduced);
/* computeVelocity (KSpaceFirstOrderSolver.cpp:2087-2119): R2C(p), computePressureGradient (.cu:1139-1157), 3x C2R,
 * computeVelocityHeterogeneous/HomogeneousUniform (.cu:184-215,278-308); dt_rho0_sg* NULL -> scalars */
KW_API kw_status kw_fused_velocity(kw_ctx* ctx, const float* p, float* ux_sgx, float* uy_sgy, float* uz_sgz,
                                   const float* dt_rho0_sgx, const float* dt_rho0_sgy, const float* dt_rho0_sgz,
                                   const float* pml_x_sgx, const float* pml_y_sgy, const float* pml_z_sgz,
                                   const float* kappa_padded, const float* ddx_k_shift_pos,
                                   const float* ddy_k_shift_pos, const float* ddz_k_shift_pos, int chain_u_spectra);
/* chain_u_spectra & KW_FUSED_CHAIN_U: the kernel that updates u also forward-transforms the updated rows along x into the
 * pipeline's scratch, so kw_fused_density(flags & KW_FUSED_U_IN_SCRATCH) skips re-reading u.  Only valid when nothing
 * else (velocity / transducer source injection) writes u in between.
 * chain_u_spectra & KW_FUSED_P_IN_SCRATCH: the spectrum of p was left in scratch by the previous
 * kw_fused_absorption_pressure(flags & KW_FUSED_CHAIN_P) and p has not been written since; p is not read. */
#define KW_FUSED_CHAIN_U       1
#define KW_FUSED_P_IN_SCRATCH  2
#define KW_FUSED_U_IN_SCRATCH 1 /* kw_fused_density: x-spectra of ux,uy,uz are already in scratch */
#define KW_FUSED_CHAIN_TERMS  2 /* kw_fused_density: chain the x-spectra of rho0*sum(du) and sum(rho) into scratch for
                                   kw_fused_absorption_pressure(terms_in_scratch = 1); only the term the pressure sum
                                   re-reads (t1 nonlinear / t0 linear) is stored, the other t arrays are left untouched */
/* second half of addInitialPressureSource (KSpaceFirstOrderSolver.cpp:2368-2395; .cu:949-982) */
KW_API kw_status kw_fused_initial_velocity(kw_ctx* ctx, const float* p, float* ux_sgx, float* uy_sgy, float* uz_sgz,
                                           const float* dt_rho0_sgx, const float* dt_rho0_sgy,
                                           const float* dt_rho0_sgz, const float* kappa_padded,
                                           const float* ddx_k_shift_pos, const float* ddy_k_shift_pos,
                                           const float* ddz_k_shift_pos);
/* computeVelocityGradient + computeDensity{Nonlinear,Linear} (KSpaceFirstOrderSolver.cpp:2126-2173; .cu:1210-1239,
 * 1358-1393, 1470-1497) and, when terms != 0, computePressureTerms{Linear(1),Nonlinear(2)} (.cu:1577-1602,1724-1742)
 * on the updated densities.  duxdx..duzdz may be NULL (gradients not stored).  terms==1: t0 = sum rho,
 * t1 = rho0 * sum du; terms==2: t0 = sum rho, t1 = nonlinear term, t2 = rho0 * sum du.
 * terms==3 (lossless media): the equation of state itself, sumPressure{Nonlinear,Linear}Lossless (.cu:2067-2084,
 * 2224-2236): t0 = p (output), t1 = c2 array or NULL for the scalar (INPUT, not written), t2 unused; with
 * KW_FUSED_CHAIN_TERMS the spectrum of the new p is left for kw_fused_velocity(KW_FUSED_P_IN_SCRATCH). */
KW_API kw_status kw_fused_density(kw_ctx* ctx, int nonlinear, const float* ux_sgx, const float* uy_sgy,
                                  const float* uz_sgz, float* rho_x, float* rho_y, float* rho_z, const float* pml_x,
                                  const float* pml_y, const float* pml_z, const float* rho0,
                                  const float* kappa_padded, const float* ddx_k_shift_neg,
                                  const float* ddy_k_shift_neg, const float* ddz_k_shift_neg, float* duxdx,
                                  float* duydy, float* duzdz, int terms, const float* b_on_a, float* t0, float* t1,
                                  float* t2, int flags);
/* absorbing branch of computePressure{Nonlinear,Linear} after the terms (KSpaceFirstOrderSolver.cpp:2196-2204,
 * 2231-2239; .cu:1812-1820,1865-1879,1966-1980): first = nonlinear term or density sum */
/* computeVelocityGradient alone (KSpaceFirstOrderSolver.cpp:2126-2143, SolverCudaKernels.cu:1210-1239): the three
 * gradients as arrays, for loops that put a step between them and the density update (non-uniform grids, :2145-2149) */
KW_API kw_status kw_fused_velocity_gradient(kw_ctx* ctx, const float* ux_sgx, const float* uy_sgy, const float* uz_sgz,
                                            float* duxdx, float* duydy, float* duzdz, const float* kappa_padded,
                                            const float* ddx_k_shift_neg_r, const float* ddy_k_shift_neg,
                                            const float* ddz_k_shift_neg, int flags /* KW_FUSED_U_IN_SCRATCH */);
KW_API kw_status kw_fused_absorption_pressure(kw_ctx* ctx, float* p, const float* velocity_gradient_term,
                                              const float* density_sum, const float* first,
                                              const float* nabla1_padded, const float* nabla2_padded, const float* c2,
                                              const float* absorb_tau, const float* absorb_eta, int flags);
#define KW_FUSED_TERMS_IN_SCRATCH 1 /* kw_fused_absorption_pressure: the two terms' spectra were chained by kw_fused_density */
#define KW_FUSED_CHAIN_P          2 /* ... and the kernel that writes p forward-transforms it for the next kw_fused_velocity */
/* tuning probe: one pass of the pipeline over its scratch (0 y-pass, 1 line pass along z, 2 z-fused, 3 y-pass x3) */
KW_API kw_status kw_fused_probe(kw_ctx* ctx, int which, const float* padded_reduced_operator);
/* FFT part of scaleSource (KSpaceFirstOrderSolver.cpp:2346-2351; .cu:740-745), in place on scaled_source */
KW_API kw_status kw_fused_scale_source(kw_ctx* ctx, float* scaled_source, const float* source_kappa_padded);
/* Non-staggered velocity (computeShiftedVelocity, KSpaceFirstOrderSolver.cpp:2714-2735): one kernel per axis instead of
 * R2C + computeVelocityShiftIn{X,Y,Z} (SolverCudaKernels.cu:2617-2710) + C2R.  out = F^-1{ filter .* F{in} } along
 * `axis` (0 x, 1 y, 2 z) of the real [nz][ny][nx] array; filter = device array of N_axis complex values, the
 * Hermitian extension of the reference's half-length shift vector with the 1/N of the transform pair folded in:
 *   filter[k] = shift[k]/N (0 < k < N/2), filter[N-k] = conj(filter[k]), filter[0] = Re(shift[0])/N,
 *   filter[N/2] = Re(shift[N/2])/N  (what C2R keeps of those two bins).  Single-GPU pipeline only. */
KW_API kw_status kw_fused_shift_velocity(kw_ctx* ctx, int axis, const float* in, float* out, const float* filter);

/* ------------------------------------------------------------------------------------------------------------------
 * Sampling kernels — replace namespace OutputStreamsCudaKernels (OutputStreams/OutputStreamsCudaKernels.cuh:47-106)
 * ---------------------------------------------------------------------------------------------------------------- */
/* sampleIndex<op> (.cuh:58-62, .cu:83-126) */
KW_API kw_status kw_sample_index(kw_ctx* ctx, kw_reduce_op op, float* sampling_buffer, const float* source_data,
                                 const uint64_t* sensor_data, uint64_t n_samples);
/* the same for up to four operators of one field over one mask in one launch (e.g. -p --p_max): index and value are read
 * once; each buffer gets exactly what its own sampleIndex<op> call would give */
KW_API kw_status kw_sample_index_multi(kw_ctx* ctx, int n_ops, const kw_reduce_op* ops, float* const* sampling_buffers,
                                       const float* source_data, const uint64_t* sensor_data, uint64_t n_samples);
/* sampleCuboid<op> (.cuh:75-81, .cu:164-252): corners are 0-based inclusive (x,y,z), matrix_size = (nx,ny,nz) */
KW_API kw_status kw_sample_cuboid(kw_ctx* ctx, kw_reduce_op op, float* sampling_buffer, const float* source_data,
                                  const uint32_t top_left[3], const uint32_t bottom_right[3],
                                  const uint32_t matrix_size[3], uint64_t n_samples);
/* sampleAll<op> (.cuh:91-95, .cu:297-332) */
KW_API kw_status kw_sample_all(kw_ctx* ctx, kw_reduce_op op, float* sampling_buffer, const float* source_data,
                               uint64_t n_samples);
/* Compression streams (config 5).  The reference samples on the GPU (sampleIndex<kNone>) and correlates with the
 * basis on the CPU one step later (OutputStreams/IndexOutputStream.cpp:373-470); here gather + correlation are one
 * kernel and the accumulators c1/c2 ([n_samples][harmonics] complex) stay on the device:
 *   c1[i,h] += bE[h*b_size+step_local] * x[i];  c2[i,h] += bE_1[h*b_size+step_local] * x[i];
 *   mirror_first_half_frame: c2[i,h] += c1[i,h]  (first saved frame, :388,462-466).  c1 may equal c2 (--no_overlap). */
KW_API kw_status kw_sample_index_compress(kw_ctx* ctx, float* c1, float* c2, const float* source_data,
                                          const uint64_t* sensor_data, uint64_t n_samples, uint32_t harmonics,
                                          const float* bE, const float* bE_1, uint32_t b_size, uint32_t step_local,
                                          int mirror_first_half_frame);
/* --40-bit_complex (IndexOutputStream.cpp:410-436; codec Compression/CompressHelper.cpp:224-389): the accumulators are
 * [n_samples][harmonics] 5-byte packed complex numbers, decoded, updated and re-encoded at every sampled step; max_exp =
 * 138 (pressure) / 114 (velocity).  no_overlap: one buffer (c2 unused), c1 += bE*x + bE_1*x. */
KW_API kw_status kw_sample_index_compress_40b(kw_ctx* ctx, void* c1, void* c2, const float* source_data,
                                              const uint64_t* sensor_data, uint64_t n_samples, uint32_t harmonics,
                                              const float* bE, const float* bE_1, uint32_t b_size, uint32_t step_local,
                                              int mirror_first_half_frame, int no_overlap, int max_exp);
KW_API kw_status kw_intensity_avg_c_accumulate_40b(kw_ctx* ctx, float* iavg, const void* frame_p, const void* frame_u,
                                                   uint64_t n_samples, uint32_t harmonics, int max_exp_p, int max_exp_u);
/* IndexOutputStream::postSample for kIAvgC (IndexOutputStream.cpp:299-342): iavg[i] += sum_h Re(P[i,h]*conj(U[i,h]))/2 */
KW_API kw_status kw_intensity_avg_c_accumulate(kw_ctx* ctx, float* iavg, const float* frame_p, const float* frame_u,
                                               uint64_t n_samples, uint32_t harmonics);
/* buf[i] /= divisor — final division of I_avg_c by the frame count (IndexOutputStream.cpp:482-490) */
KW_API kw_status kw_divide(kw_ctx* ctx, float* buf, float divisor, uint64_t n);
/* ---- post-processing of stored series (KSpaceFirstOrderSolver.cpp:1231-1534 computeAverageIntensities, :1783-2080
 * computeQTerm).  The reference moves every spectrum to the host for the multiply; here the series stay on the device.
 * kw_time_shift_series: series[step][i] (device, steps x n, in place) is advanced by half a time step through its
 *   spectrum along the step axis: X[k] *= (1/steps) * shift[k], k = 0..steps/2 (:1437-1449); shift = device array of
 *   steps/2+1 complex values exp(i*pi*s(k)/steps) computed by the caller (:1253-1260).
 * kw_intensity_avg: iavg[i] = (sum over steps, in step order, of u[step][i]*p[step][i]) / steps (:1492-1513).
 * kw_q_term_sum: out[i] = -(a[i] + b[i] + c[i]) over the grid, c may be NULL in 2-D (:2014-2026). */
KW_API kw_status kw_time_shift_series(kw_ctx* ctx, float* series, const float* shift, uint64_t steps, uint64_t n);
KW_API kw_status kw_intensity_avg(kw_ctx* ctx, float* iavg, const float* p, const float* u, uint64_t steps, uint64_t n);
KW_API kw_status kw_q_term_sum(kw_ctx* ctx, float* out, const float* a, const float* b, const float* c, uint64_t n);
/* postProcessingRms (.cuh:103-105, .cu:359-378) */
KW_API kw_status kw_post_processing_rms(kw_ctx* ctx, float* sampling_buffer, float scaling_coeff, uint64_t n_samples);

#ifdef __cplusplus
}
#endif
#endif /* KWAVE_HIP_H */
