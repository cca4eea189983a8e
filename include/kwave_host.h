/*
 * kwave_host.h — C entry points of the C++ host layer (libkwave_host.so).
 *
 * The host layer is the MI355X build's mirror of the reference's caller side of the hot path:
 * Parameters / MatrixContainer / KSpaceFirstOrderSolver / OutputStreamContainer (k-wave-fluid-cuda_amd/host/).
 * These entry points exist so that non-C++ drivers (bench.py, pytest via ctypes) can run the *same* C++ time loop
 * the command-line program runs — they correspond to main()'s sequence in the reference (main.cpp:840-966):
 *   kwh_create   = Parameters::init + selectDevice + allocateMemory + loadInputData   (main.cpp:857-917)
 *   kwh_run      = the body of computeMainLoop for n steps                            (KSpaceFirstOrderSolver.cpp:885-935)
 *   kwh_finish   = last delayed flush + postProcessing                                (:937-942, :950-1053)
 * Input datasets are named exactly like the HDF5 input file's datasets (Utils/MatrixNames.h, main.cpp:446-563).
 */
#ifndef KWAVE_HOST_H
#define KWAVE_HOST_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KWH_API __attribute__((visibility("default")))

typedef struct kwh_solver kwh_solver;

typedef struct kwh_dataset
{
  const char* name;     /* HDF5 dataset name, e.g. "c0", "ddx_k_shift_pos_r", "sensor_mask_index" */
  const void* data;     /* host pointer: float32 or uint64 */
  int32_t     dtype;    /* 0 = float ("float"), 1 = uint64 ("long") */
  int32_t     pad_;
  uint64_t    nx, ny, nz; /* (x,y,z) sizes = HDF5 dims reversed; complex datasets have the doubled x size halved here */
} kwh_dataset;

/* what the reference takes from the command line for the loop (CommandLineParameters.cpp:264-292) */
typedef struct kwh_options
{
  int32_t  device_idx;                 /* -g ; <0 = first free */
  int32_t  fused_kernels;              /* 1: MI355X fused step kernels, 0: one launch per reference kernel */
  uint64_t sampling_start_time_index;  /* -s (0-based) */
  uint64_t benchmark_time_steps;       /* --benchmark (0 = use Nt) */
  int32_t  p_raw, p_rms, p_max, p_min, p_max_all, p_min_all, p_final;
  int32_t  u_raw, u_rms, u_max, u_min, u_max_all, u_min_all, u_final, u_non_staggered_raw;
  int32_t  p_c, u_non_staggered_c, i_avg_c, no_overlap;
  float    period;
  uint64_t mos, harmonics;
  /* Z-slab decomposition (one process per GPU).  With slab_ranks > 1 the datasets describe this rank's slab: "Nz" is
   * the local plane count, 3-D arrays / pml_z / pml_z_sgz are the local slices, source and sensor indices are local
   * (re-based, 1-based) indices; ddz_* stay global.  The all-to-all is the device library's RCCL path (comm_unique_id
   * below) unless exchange_fn (kw_exchange_fn of kwave_hip.h) overrides it. */
  uint64_t slab_ranks, slab_rank, nz_global;
  void*    exchange_fn;
  void*    exchange_user;
  void*    exchange_start_fn; /* optional kw_exchange_start_fn / kw_exchange_wait_fn pair (both or neither) */
  void*    exchange_wait_fn;
  void*    scratch[6]; /* optional caller-owned pipeline scratch (kw_fused_create_with_scratch), else all NULL */
  /* post-processed quantities (--I_avg, --Q_term, --Q_term_c; KSpaceFirstOrderSolver.cpp:977-1024): time-averaged
   * intensity from the stored p / u_non_staggered series, and Q = -div(I_avg) from it or from the compressed one */
  int32_t  i_avg, q_term, q_term_c;
  int32_t  u_c;        /* --u_c: compression coefficients of the staggered velocities (OutputStreamContainer.cpp:133-142) */
  float    frequency;  /* --frequency [Hz]: period = 1 / (frequency * dt) (Parameters.cpp:468-480); not together with period */
  int32_t  only_post_processing; /* --post: no time loop; I_avg / I_avg_c / Q_term / Q_term_c from the series stored in an
                                    existing output file (KSpaceFirstOrderSolver.cpp:373-415; kwh_post_process_output_file) */
  /* slab runs over the library's own RCCL path (the default multi-GPU exchange): KW_COMM_ID_BYTES bytes from
   * kw_comm_unique_id() on rank 0, the same on every rank; exchange_fn / exchange_start_fn must then be NULL.
   * With slab_ranks == 1 the rank exchanges with itself (rehearsal of the multi-GPU path on one GPU). */
  const void* comm_unique_id;
  int32_t  complex_40bit; /* --40-bit_complex: compression coefficients kept and stored as 5-byte complex numbers
                             (CompressHelper.cpp:224-389; BaseOutputStream.cpp:98-101: c_complex_size 1.25) */
  int32_t  reserved_;
  void*    exchange_piece_fn; /* optional kw_exchange_piece_fn (strided pieces; exchange_wait_fn, if set, is its wait):
                                 a caller-owned transport that lets the pipeline run its pipelined slab schedule.
                                 Give exchange_fn as well (used when the pipeline is told not to pipeline). */
  const void* tuning;         /* optional const kw_tuning* (kwave_hip.h): schedule parameters of the device library */
  int32_t  step_graph;        /* 1: replay the steady-state step from a recorded graph (launch-bound small grids; measured
                                 3-5 % slower than eager launches at 64^3 / 128^3, hence off by default) */
  int32_t  comm_p2p;          /* 1: slab exchange over the device library's P2P transport (kw_comm_init_p2p: mapped peer
                                 buffers, one store kernel per exchange) instead of RCCL groups; needs comm_allgather_fn.
                                 With comm_unique_id as well, the RCCL communicator is created first and stays behind it. */
  void*    comm_allgather_fn; /* kwh_allgather_fn: how the ranks trade their kw_comm_p2p_export blobs */
  void*    comm_allgather_user;
  const char* rccl_library;   /* optional library name / path for the RCCL binding (kw_comm_init_with); NULL = default */
  float    p2p_emulate_link_gbs;   /* > 0 with comm_p2p: link model instead of peers (kw_comm_p2p_emulate; schedule studies */
  float    p2p_emulate_latency_us; /*   with ONE rank of slab_ranks on a one-GPU machine, tools/emulate_rank.py) */
} kwh_options;
/* gathers `bytes` bytes of every rank, in rank order, into all (nranks * bytes); the same result on every rank; 0 = ok.
 * Any transport the launcher has will do: MPI_Allgather, torch.distributed.all_gather on a gloo group, files. */
typedef int (*kwh_allgather_fn)(void* user, const void* mine, void* all, size_t bytes);

KWH_API const char* kwh_last_error(void);
KWH_API int      kwh_create(const kwh_dataset* datasets, size_t n_datasets, const kwh_options* options, kwh_solver** out);
KWH_API int      kwh_destroy(kwh_solver* s);
KWH_API int      kwh_run(kwh_solver* s, uint64_t n_steps);
KWH_API int      kwh_finish(kwh_solver* s);
KWH_API int      kwh_sync(kwh_solver* s);
KWH_API uint64_t kwh_time_index(const kwh_solver* s);
/* the underlying kw_ctx* (include/kwave_hip.h) — for HIP-event timing on the solver's stream */
KWH_API void*    kwh_context(kwh_solver* s);
/* copy a matrix (by reference name: "p","ux_sgx","rhox","kappa_r","absorb_tau", "c0" (= c^2 after pre-processing) ...)
 * from the device into dst; n = number of floats expected (checked) */
KWH_API int      kwh_get_matrix(kwh_solver* s, const char* name, float* dst, uint64_t n);
KWH_API int      kwh_matrix_size(kwh_solver* s, const char* name, uint64_t* n_floats);
/* scalar parameters computed by pre-processing: "absorb_tau","absorb_eta","c2","dt_rho0_sgx",...;
 * "fused_pipeline" = 1 when the grid runs on the hand-written FFT pipeline, 0 on the rocFFT path */
KWH_API int      kwh_get_scalar(kwh_solver* s, const char* name, float* out);
/* output streams by dataset name ("p","p_max","ux",...): size = points per step, steps = stored steps (1 for aggregates) */
KWH_API int      kwh_stream_info(kwh_solver* s, const char* name, uint64_t* size, uint64_t* steps);
KWH_API int      kwh_stream_read(kwh_solver* s, const char* name, float* dst, uint64_t n);
/* Checkpoint / restart (KSpaceFirstOrderSolver.cpp:1176-1224 save, :186-228 recover; MatrixContainer.cpp:504-537): the
 * state of a run is the seven arrays p, rhox, rhoy, rhoz, ux_sgx, uy_sgy, uz_sgz, the time index and the state of every
 * output stream.  These calls move that state in and out of a prepared solver; the HDF5 checkpoint file itself is
 * written by kwh_checkpoint_write / read by kwh_checkpoint_read (libkwave_host_h5.so). */
/* Compression helpers of the reference that run on the host (Compression/CompressHelper.cpp:146-216 findPeriod,
 * :298-389 convertFloatCTo40b, :224-289 convert40bToFloatC; exponent bias e = 138 pressure / 114 velocity).  The
 * solver uses findPeriod itself when compression streams are requested without a period (Parameters.cpp:488-512). */
KWH_API int      kwh_find_period(const float* signal, uint64_t length, float* period);
KWH_API int      kwh_pack_complex_40b(const float* re_im_pairs, uint64_t n, uint8_t* packed5n, int32_t e);
KWH_API int      kwh_unpack_complex_40b(const uint8_t* packed5n, uint64_t n, float* re_im_pairs, int32_t e);
KWH_API int      kwh_set_matrix(kwh_solver* s, const char* name, const float* src, uint64_t n);
KWH_API int      kwh_set_time_index(kwh_solver* s, uint64_t t_index);
KWH_API int      kwh_stream_count(kwh_solver* s, uint64_t* n);
KWH_API int      kwh_stream_name(kwh_solver* s, uint64_t i, char* out, uint64_t cap);
/* the same including the streams that only feed others and are not part of the output (the coefficient series behind
 * --I_avg_c / --Q_term_c alone, the intensities behind --Q_term alone): a checkpoint has to carry those too */
KWH_API int      kwh_stream_count_all(kwh_solver* s, uint64_t* n);
KWH_API int      kwh_stream_name_all(kwh_solver* s, uint64_t i, char* out, uint64_t cap);
/* dst == NULL: only the sizes are returned */
KWH_API int      kwh_stream_checkpoint(kwh_solver* s, const char* name, float* dst, uint64_t cap, uint64_t* n_floats,
                                       uint64_t* sampled_steps);
KWH_API int      kwh_stream_restore(kwh_solver* s, const char* name, const float* src, uint64_t n_floats,
                                    uint64_t sampled_steps);

#ifdef __cplusplus
}
#endif
#endif
