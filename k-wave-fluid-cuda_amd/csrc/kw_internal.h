// kw_internal.h — shared internals of libkwave_hip.so (not part of the public C-ABI).
#ifndef KW_INTERNAL_H
#define KW_INTERNAL_H

#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>

#include "kwave_hip.h"

struct kw_fft_plan
{
  rocfft_plan           plan  = nullptr;
  rocfft_execution_info info  = nullptr;
  size_t                work  = 0;
};

struct kw_comm_state; // kw_comm.hip: communicator (RCCL or P2P) + communication stream + per-slot events
// defaults of every schedule parameter (kwave_hip.h kw_tuning)
inline kw_tuning kw_tuning_defaults()
{
  kw_tuning t{};
  t.struct_bytes        = static_cast<uint32_t>(sizeof(kw_tuning));
  t.side_array          = 1;
  t.tail_chunks         = 0;
  t.split512            = 1;
  t.slab_pipeline       = 1;
  t.slab_chunks         = 1;
  t.slab_batch          = -1;
  t.p2p_blocks_per_peer = 4;
  t.p2p_timeout_s       = 20.f;
  t.plane_kernels       = 1;
  return t;
}
#define KW_COMM_SLOTS 32 /* 2 directions x 3 arrays x KW_XCHUNKS_MAX plane chunks, + spare (z-shift staging) */
#define KW_XCHUNKS_MAX 4

struct kw_ctx
{
  int          device      = 0;
  hipStream_t  own_stream  = nullptr;
  hipStream_t  stream      = nullptr; // stream every launch goes to
  hipStream_t  copy_stream = nullptr; // D2H of sampled series, overlapped with compute
  hipEvent_t   copy_fence  = nullptr; // compute -> copy stream ordering
  bool         have_consts = false;
  kw_constants c{};
  int          cu_count    = 256;
  // rocFFT
  bool         fft_setup   = false;
  kw_fft_plan  r2c_3d, c2r_3d;
  kw_fft_plan  r2c_1d[3], c2r_1d[3];
  void*        fft_work       = nullptr; // one shared work buffer, sized for the largest plan
  size_t       fft_work_bytes = 0;
  // fused spectral pipeline (kw_fused.hip): private padded spectral scratch + per-axis twiddle tables
  struct fused_plan
  {
    bool     ready = false;
    bool     plane = false;                        // small square planes: the x-inverse kernels take whole z-planes and do the y transforms too
    bool     split512 = true;                      // 512-point y / z lines as 2 x 256 (A/B knob against the 16 x 32 kernels)
    uint32_t nxm = 0;                              // columns kept in the rows: nx/2+1, or nx/2 when the x-Nyquist column is kept apart
    uint32_t side_off = 0;                         // element offset of that column's compact array N[z][y] in s[] (0: none)
    uint32_t Palloc = 0;                           // pitch the scratch / imported operator arrays are sized by (nx/2+1 rounded up to 16)
    int      y_done  = 0;                          // chained spectra in s[0..y_done) already carry their forward y-pass
    uint32_t P     = 0;                            // row pitch of the spectra (complex), multiple of 16: nxm rounded up
    uint32_t PX    = 0;                            // row pitch of exchange-side buffers (slab mode: nx/2+1, no padding)
    float2*  s[3]  = {nullptr, nullptr, nullptr};  // three padded complex scratch arrays [nz][ny][P]
    float2*  s4    = nullptr;                      // whole-plane kernels: where the chained spectrum of u_x goes (its block
                                                   // must not overwrite s[0], which the u_y block of the same plane reads)
    float2*  tw[3] = {nullptr, nullptr, nullptr};  // exp(-2 pi i m / n) for n = nx, ny, nz
    // Z-slab decomposition (multi-GPU): this context owns nz = nz_global/nranks planes of the real-space arrays and,
    // after the all-to-all transpose, nyl = ny/nranks rows of every spectrum with all nz_global planes.
    uint32_t nranks = 1, rank = 0, nz_global = 0, nyl = 0;
    bool     slab   = false;                       // transposed z-pass + exchange (nranks > 1, or one rank exchanging with itself)
    float2*  t[3]  = {nullptr, nullptr, nullptr};  // exchange partners of s[] (slab mode only)
    bool     owns_scratch = true;
    kw_exchange_fn exchange = nullptr;             // all-to-all over the ranks (RCCL via the caller)
    void*          exchange_user = nullptr;
    kw_exchange_start_fn exchange_start = nullptr; // optional split-phase pair (overlap with compute)
    kw_exchange_wait_fn  exchange_wait  = nullptr;
    kw_exchange_piece_fn exchange_piece = nullptr; // strided pieces (plane chunks of an array): enables the pipelined schedule
    // Pipelined slab schedule (library's RCCL path or a piece callback): a third buffer set r[] holds the transposed
    // spectra (forward receive = z-pass in / out = backward send), so that the plane-local tail of a stage — backward
    // receive t[] -> y-inverse -> s[] -> x-inverse + epilogue -> chained x / y forward -> t[] -> forward send — runs per
    // chunk of planes while the other chunks are on the wire.
    bool     two_d = false;                        // Nz == 1: x-pass, fused pass along y (in the z-pass kernels' role), x-pass; no y-pass
    bool     pipelined = false;
    uint32_t xchunks   = 1;                        // plane chunks per array of the pipelined tail
    float2*  r[3]      = {nullptr, nullptr, nullptr};
    int      fwd_ahead = 0;                        // arrays whose forward exchange into r[] was started by the producer's tail
    bool     xbatch    = false;                    // small messages: all arrays of a stage travel in ONE exchange per direction
    int8_t   xslot[2][3][KW_XCHUNKS_MAX] = {};     // [dir][array][chunk] -> slot of the started exchange that carries it, -1 none
    bool     slot_waited[KW_COMM_SLOTS] = {};      // the compute stream already waits for that exchange
    int8_t   slot_pieces[KW_COMM_SLOTS] = {};      // pieces of the exchange started on that slot (callback transports wait per piece)
  } fused;
  kw_comm_state* comm = nullptr; // multi-GPU exchange (kw_comm_init / kw_comm_init_p2p)
  kw_tuning      tuning = kw_tuning_defaults(); // kw_set_tuning
  // profiling (kw_profile_enable)
  struct prof_rec { const char* name; hipEvent_t e0, e1; };
  bool                  profiling = false;
  std::vector<prof_rec> prof;
};

// RAII: records a HIP event pair around one entry point while profiling is on
struct kw_prof_scope
{
  kw_ctx* ctx;
  size_t  idx;
  bool    on;
  kw_prof_scope(kw_ctx* c, const char* name) : ctx(c), idx(0), on(c && c->profiling)
  {
    if (!on) return;
    kw_ctx::prof_rec r{name, nullptr, nullptr};
    if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) { on = false; return; }
    (void)hipEventRecord(r.e0, ctx->stream);
    idx = ctx->prof.size();
    ctx->prof.push_back(r);
  }
  ~kw_prof_scope()
  {
    if (on) (void)hipEventRecord(ctx->prof[idx].e1, ctx->stream);
  }
};
#define KW_PROF(ctx, name) kw_prof_scope kw_prof_scope_(ctx, name)

// split-phase all-to-all on the context's communicator (kw_comm.hip); slot < KW_COMM_SLOTS
kw_status kw_comm_exchange_start(kw_ctx* ctx, int slot, const void* send, void* recv, size_t bytes_per_peer);
kw_status kw_comm_exchange_start2(kw_ctx* ctx, int slot, const void* send, void* recv, size_t bytes_per_peer, const void* send2,
                                  void* recv2, size_t bytes_per_peer2);
// general form: n pieces, each "bytes at send + q*stride + offset go to rank q and land at recv + sender*stride + offset"
struct kw_comm_piece { const void* send; void* recv; size_t stride, offset, bytes; };
kw_status kw_comm_exchange_start_pieces(kw_ctx* ctx, int slot, const kw_comm_piece* pieces, int n);
kw_status kw_comm_exchange_wait(kw_ctx* ctx, int slot);
kw_status kw_comm_sync(kw_ctx* ctx); // host waits for the communication stream (no-op without a communicator)
kw_status kw_comm_check(kw_ctx* ctx); // KW_ERR_COMM once a P2P exchange has given up waiting for a peer (no-op otherwise)
void      kw_comm_buffers_gone(kw_ctx* ctx); // kw_fused_destroy: the exchange buffers the P2P transport mapped are being freed

// thread-local error text (kw_last_error)
void kw_set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));

#define KW_CHECK_CTX(ctx)                                                                                              \
  do {                                                                                                                 \
    if ((ctx) == nullptr) { kw_set_error("%s: ctx is NULL", __func__); return KW_ERR_INVALID; }                       \
  } while (0)

#define KW_CHECK_CONSTS(ctx)                                                                                           \
  do {                                                                                                                 \
    KW_CHECK_CTX(ctx);                                                                                                 \
    if (!(ctx)->have_consts) { kw_set_error("%s: kw_set_constants has not been called", __func__); return KW_ERR_STATE; } \
  } while (0)

#define KW_REQUIRE(cond)                                                                                               \
  do {                                                                                                                 \
    if (!(cond)) { kw_set_error("%s: invalid argument: %s", __func__, #cond); return KW_ERR_INVALID; }                \
  } while (0)

// "GPU error: %s routine name: %s in file %s, line %d." — message shape of the reference (ErrorMessages.h:331)
#define KW_HIP(call)                                                                                                   \
  do {                                                                                                                 \
    hipError_t e_ = (call);                                                                                            \
    if (e_ != hipSuccess) {                                                                                            \
      kw_set_error("GPU error: %s routine name: %s in file %s, line %d.", hipGetErrorString(e_), __func__, __FILE__,  \
                   __LINE__);                                                                                          \
      return (e_ == hipErrorOutOfMemory) ? KW_ERR_ALLOC : KW_ERR_HIP;                                                  \
    }                                                                                                                  \
  } while (0)

#define KW_TRY_STATUS(call)                                                                                            \
  do {                                                                                                                 \
    const kw_status st_ = (call);                                                                                      \
    if (st_ != KW_OK) return st_;                                                                                      \
  } while (0)

// checked right after each launch, without synchronising (reference: cudaCheckErrors(cudaGetLastError()))
#define KW_LAUNCH_CHECK() KW_HIP(hipGetLastError())

#endif
