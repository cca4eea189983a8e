// kw_comm.hip — the all-to-all of the Z-slab decomposition, inside the device library: a communicator owned by the
// context, a dedicated communication stream, and split-phase exchanges ordered against the compute stream with events
// (SURVEY §8b last row, §8e "Collective").  New with this build: the reference is single-GPU (Readme.md:12-13).
//
//   start(slot, pieces):  ready[slot] <- compute stream;  comm stream waits for it;  the transport's exchange of every
//                         peer's part of every piece;  done[slot] <- comm stream
//   wait(slot):           compute stream waits for done[slot]           (the host never blocks)
//
// Two transports behind that pair:
//
//   RCCL  (kw_comm_init)      ncclGroupStart; per peer q: ncclSend(chunk q -> q), ncclRecv(chunk q <- q); ncclGroupEnd.
//                             RCCL is bound at run time (dlopen): a process that already holds an RCCL (torch) shares that
//                             instance, a plain C++ caller gets the ROCm one, single-GPU users need no RCCL at all.
//   P2P   (kw_comm_init_p2p / kw_comm_p2p_export / kw_comm_p2p_connect)
//                             the ranks map each other's exchange buffers (hipIpc handles between processes, plain
//                             pointers between threads of one process) and ONE small kernel per exchange on the
//                             communication stream stores this rank's chunks straight into the peers' receive buffers
//                             over xGMI — k_p2p_exchange below.  No library kernel, no proxy thread, no staging buffer:
//                             what an exchange costs besides its bytes is one kernel launch and two flag round trips.
//
// xGMI is a point-to-point mesh: with all P - 1 peers addressed at once every link of the GPU carries its own chunk at
// the same time, so both transports walk the peers starting from the right-hand neighbour (rank r opens with r + 1).
#include "kw_internal.h"

#include <dlfcn.h>
#include <rccl/rccl.h>
#include <unistd.h>

#include <cstdlib>

// ---- P2P transport: shared definitions --------------------------------------------------------------------------------
constexpr int      KW_P2P_MAX_RANKS = 16;
constexpr int      KW_P2P_NB_MAX    = 8;   // blocks per peer (kw_tuning::p2p_blocks_per_peer)
constexpr int      KW_P2P_SELF_BLOCKS = 64; // blocks that copy the rank's own chunk (local HBM: no link to wait for)
constexpr int      KW_P2P_NBUF      = 9;   // s[3], t[3], r[3] of the fused pipeline: every buffer an exchange may land in
constexpr uint32_t KW_P2P_MAGIC     = 0x6b775032u; // "kwP2"
// flag words of one rank (uint32 epochs, written by the peers): credit[slot][sender], full[slot][sender][block]
constexpr size_t   KW_P2P_CREDIT_WORDS = static_cast<size_t>(KW_COMM_SLOTS) * KW_P2P_MAX_RANKS;
constexpr size_t   KW_P2P_FLAG_WORDS   = KW_P2P_CREDIT_WORDS * (1 + KW_P2P_NB_MAX);
__host__ __device__ inline size_t p2p_credit_at(uint32_t slot, uint32_t sender) { return static_cast<size_t>(slot) * KW_P2P_MAX_RANKS + sender; }
__host__ __device__ inline size_t p2p_full_at(uint32_t slot, uint32_t sender, uint32_t b)
{
  return KW_P2P_CREDIT_WORDS + (static_cast<size_t>(slot) * KW_P2P_MAX_RANKS + sender) * KW_P2P_NB_MAX + b;
}

// what one rank publishes (kw_comm_p2p_export); fixed size, plain bytes for any transport the caller has
struct p2p_buffer_desc
{
  hipIpcMemHandle_t handle;  // of the allocation that holds the buffer
  uint64_t          alloc;   // base address of that allocation in the exporting process
  uint64_t          offset;  // of the buffer inside it
  uint64_t          bytes;   // 0: slot not used
};
struct p2p_blob
{
  uint32_t        magic, version, nranks, rank;
  int64_t         pid;
  uint64_t        host;      // boot id hash: IPC handles and pointers mean something on one machine only
  int32_t         device, pad_;
  p2p_buffer_desc buf[KW_P2P_NBUF + 1]; // [KW_P2P_NBUF] = the flag words
};
static_assert(sizeof(p2p_blob) <= KW_COMM_P2P_BLOB_BYTES, "KW_COMM_P2P_BLOB_BYTES too small");

struct p2p_piece
{
  const char* send;       // local send base; the part for peer q starts at q * stride + offset
  char*       recv_local; // local receive base (self chunk, emulation)
  uint64_t    stride, offset, bytes;
  uint32_t    recv_buf;   // registered buffer the receive base lies in ...
  uint64_t    recv_off;   // ... and where
};
struct p2p_args
{
  uint32_t        nranks, rank, slot, epoch, nb, npieces;
  uint32_t*       my_flags;                      // this rank's flag words (the peers write them)
  const uint64_t* peer_tab;                      // [KW_P2P_NBUF + 1][KW_P2P_MAX_RANKS]: peer q's mapping of buffer b
  uint32_t*       status;                        // host-visible: first failure of this context's exchanges (0 = none)
  uint64_t        timeout_ticks;                 // of the 100 MHz wall clock
  float           emulate_bytes_per_tick;        // > 0: link model of tools/emulate_rank.py (one rank alone on the GPU)
  uint32_t        emulate_latency_ticks;
  p2p_piece       piece[8];
};

struct kw_comm_state
{
  enum Transport { RCCL = 0, P2P_PENDING = 1, P2P = 2, P2P_EMULATED = 3 };
  Transport   transport = RCCL;
  void*       lib    = nullptr;
  ncclComm_t  comm   = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t  ready[KW_COMM_SLOTS] = {};
  hipEvent_t  done[KW_COMM_SLOTS]  = {};
  uint32_t    nranks = 0, rank = 0;
  uint64_t    exchanges = 0;
  decltype(&ncclGetUniqueId)    getUniqueId    = nullptr;
  decltype(&ncclCommInitRank)   commInitRank   = nullptr;
  decltype(&ncclCommDestroy)    commDestroy    = nullptr;
  decltype(&ncclGroupStart)     groupStart     = nullptr;
  decltype(&ncclGroupEnd)       groupEnd       = nullptr;
  decltype(&ncclSend)           send           = nullptr;
  decltype(&ncclRecv)           recv           = nullptr;
  decltype(&ncclGetErrorString) getErrorString = nullptr;
  // P2P
  uint32_t*   flags      = nullptr;            // this rank's flag words (device memory, uncached where the runtime allows)
  uint64_t*   peer_tab   = nullptr;            // device copy of the table below
  uint32_t*   status     = nullptr;            // pinned host word
  uint32_t    epoch[KW_COMM_SLOTS] = {};
  struct { uint64_t base = 0, bytes = 0; } reg[KW_P2P_NBUF]; // local buffers, by registration index
  std::vector<void*> opened;                   // hipIpcOpenMemHandle results (closed by release)
  float       emulate_gbs = 0.f, emulate_latency_us = 0.f;
};

namespace
{
#define KW_NCCL(st, call)                                                                                              \
  do {                                                                                                                 \
    const ncclResult_t r_ = (call);                                                                                    \
    if (r_ != ncclSuccess)                                                                                             \
    {                                                                                                                  \
      kw_set_error("RCCL error: %s routine name: %s in file %s, line %d.", (st)->getErrorString(r_), __func__, __FILE__, \
                   __LINE__);                                                                                          \
      return KW_ERR_COMM;                                                                                              \
    }                                                                                                                  \
  } while (0)

// binds the RCCL entry points; `st` keeps the handle.  `library`: NULL = the process's / ROCm's librccl
kw_status bind_rccl(kw_comm_state* st, const char* library)
{
  const char* names[] = { library, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
  for (const char* n : names)
  {
    if (n == nullptr || n[0] == '\0') continue;
    st->lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (st->lib != nullptr || n == library) break; // a library named by the caller is not silently replaced
  }
  if (st->lib == nullptr)
  {
    kw_set_error("kw_comm: cannot load RCCL (%s): %s", library != nullptr ? library : "librccl.so.1", dlerror());
    return KW_ERR_COMM;
  }
#define BIND(member, symbol)                                                                                           \
  st->member = reinterpret_cast<decltype(st->member)>(dlsym(st->lib, symbol));                                         \
  if (st->member == nullptr) { kw_set_error("kw_comm: RCCL has no symbol %s", symbol); return KW_ERR_COMM; }
  BIND(getUniqueId, "ncclGetUniqueId")
  BIND(commInitRank, "ncclCommInitRank")
  BIND(commDestroy, "ncclCommDestroy")
  BIND(groupStart, "ncclGroupStart")
  BIND(groupEnd, "ncclGroupEnd")
  BIND(send, "ncclSend")
  BIND(recv, "ncclRecv")
  BIND(getErrorString, "ncclGetErrorString")
#undef BIND
  return KW_OK;
}

void release(kw_comm_state* st)
{
  if (st == nullptr) return;
  if (st->stream) (void)hipStreamSynchronize(st->stream);
  if (st->comm && st->commDestroy) (void)st->commDestroy(st->comm);
  for (void* p : st->opened) (void)hipIpcCloseMemHandle(p);
  if (st->flags) (void)hipFree(st->flags);
  if (st->peer_tab) (void)hipFree(st->peer_tab);
  if (st->status) (void)hipHostFree(st->status);
  for (int i = 0; i < KW_COMM_SLOTS; i++)
  {
    if (st->ready[i]) (void)hipEventDestroy(st->ready[i]);
    if (st->done[i]) (void)hipEventDestroy(st->done[i]);
  }
  if (st->stream) (void)hipStreamDestroy(st->stream);
  // the library handle stays open: RCCL keeps threads and registered memory that outlive a communicator
  delete st;
}

// communication stream + per-slot events: what both transports need
kw_status create_stream_and_events(kw_comm_state* st)
{
  // Highest stream priority: the runtime keeps a pool of hardware queues per priority, so the communication stream does
  // not share a queue with the compute / copy streams of this process (a P2P exchange kernel waits for its peers while
  // it runs: work queued behind it in the same hardware queue would wait with it), and its few workgroups are placed
  // ahead of the compute kernels' many.
  int least = 0, greatest = 0;
  (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
  if (hipStreamCreateWithPriority(&st->stream, hipStreamNonBlocking, greatest) != hipSuccess)
  {
    kw_set_error("kw_comm_init: cannot create the communication stream");
    return KW_ERR_HIP;
  }
  for (int i = 0; i < KW_COMM_SLOTS; i++)
    if (hipEventCreateWithFlags(&st->ready[i], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&st->done[i], hipEventDisableTiming) != hipSuccess)
    {
      kw_set_error("kw_comm_init: cannot create events");
      return KW_ERR_HIP;
    }
  return KW_OK;
}

uint64_t host_id()
{ // FNV-1a of the kernel's boot id: the same for every process of one machine
  uint64_t h = 1469598103934665603ull;
  FILE* f = fopen("/proc/sys/kernel/random/boot_id", "r");
  if (f == nullptr) return h;
  int ch;
  while ((ch = fgetc(f)) != EOF) { h ^= static_cast<uint64_t>(ch & 0xff); h *= 1099511628211ull; }
  fclose(f);
  return h;
}

// ---- the P2P exchange kernel ----------------------------------------------------------------------------------------
// Grid: nranks - 1 peer groups of nb blocks, then the rank's own chunk (a local copy at HBM speed, KW_P2P_SELF_BLOCKS
// blocks).  Group g works for peer (rank + 1 + g) % nranks.  A peer group runs one rendezvous with its peer, symmetric on
// both sides:
//
//   1. credit   block 0 tells the peer "my receive buffers of exchange (slot, epoch) may be written": this kernel starts
//               only after the compute stream's work up to start(slot) — i.e. after the last reader of what the buffer
//               held before;
//   2. wait for the peer's credit, then store this rank's part of every piece into the peer's mapped receive buffer
//               (16-B stores, block b takes the b-th contiguous share);
//   3. full     system-scope release (L2 write-back, stores acknowledged), then flag full[slot][rank][b] = epoch at the peer;
//   4. wait for the peer's full[slot][peer][b]: its share b has landed here.
//
// When every block has left, all sends have been read and all receives have landed: done[slot] on the communication
// stream means for the compute stream what the end of an RCCL group means.  Kernels launched after it start with the
// acquire every kernel start performs (the XCDs' L2s do not keep lines of the receive buffer across it).
// Flags are monotonic epochs per slot (wrap-safe compare); every spin is bounded by `timeout_ticks` and reports through
// `status` instead of hanging the queue.
__device__ __forceinline__ bool p2p_wait_flag(const uint32_t* flag, uint32_t epoch, uint64_t timeout_ticks)
{
  const uint64_t t0 = wall_clock64();
  for (;;)
  {
    const uint32_t v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (static_cast<int32_t>(v - epoch) >= 0) return true;
    __builtin_amdgcn_s_sleep(64); // ~1.7 us between polls: the waiting wave leaves its SIMD to the compute kernels' waves
    if (wall_clock64() - t0 > timeout_ticks) return false;
  }
}

// share `part` of `nparts` of a byte range, by all threads of the block
__device__ __forceinline__ void p2p_copy(char* __restrict__ dst, const char* __restrict__ src, uint64_t bytes, uint32_t part,
                                         uint32_t nparts)
{
  if (((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src) | bytes) & 15u) == 0)
  {
    const uint64_t n = bytes / 16, per = (n + nparts - 1) / nparts, a = static_cast<uint64_t>(part) * per;
    const uint64_t b = (a + per < n) ? a + per : n;
    const float4* __restrict__ s4 = reinterpret_cast<const float4*>(src);
    float4* __restrict__       d4 = reinterpret_cast<float4*>(dst);
    uint64_t i = a + threadIdx.x;
    const uint64_t step = blockDim.x;
    for (; i + 3 * step < b; i += 4 * step)
    { // four loads in flight per lane
      const float4 v0 = s4[i], v1 = s4[i + step], v2 = s4[i + 2 * step], v3 = s4[i + 3 * step];
      d4[i] = v0; d4[i + step] = v1; d4[i + 2 * step] = v2; d4[i + 3 * step] = v3;
    }
    for (; i < b; i += step) d4[i] = s4[i];
  }
  else
  { // (pieces are whole floats: checked by the host)
    const uint64_t n = bytes / 4, per = (n + nparts - 1) / nparts, a = static_cast<uint64_t>(part) * per;
    const uint64_t b = (a + per < n) ? a + per : n;
    const float* __restrict__ s1 = reinterpret_cast<const float*>(src);
    float* __restrict__       d1 = reinterpret_cast<float*>(dst);
    for (uint64_t i = a + threadIdx.x; i < b; i += blockDim.x) d1[i] = s1[i];
  }
}

__device__ __forceinline__ void p2p_fail(uint32_t* status, uint32_t code, uint32_t slot, uint32_t peer)
{ // the first failure is kept (a plain check-then-store: any of several concurrent failures tells the story); host memory:
  // no atomic read-modify-write across PCIe needed.  bits: code | slot << 8 | peer << 16
  if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0u)
    __hip_atomic_store(status, code | (slot << 8) | (peer << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(256) void k_p2p_exchange(p2p_args a)
{
  const uint32_t npeer = (a.nranks - 1u) * a.nb; // blocks of the peer groups; the rest copy the rank's own chunk
  const bool     self  = blockIdx.x >= npeer;
  const uint32_t g = self ? a.nranks - 1u : blockIdx.x / a.nb;
  const uint32_t b = self ? blockIdx.x - npeer : blockIdx.x % a.nb;
  const uint32_t parts = self ? gridDim.x - npeer : a.nb;
  const uint32_t peer = (a.rank + 1u + g) % a.nranks;
  __shared__ uint32_t ok;
  if (self || a.emulate_bytes_per_tick > 0.f)
  { // own chunk — or, emulating, the chunk of a peer that is not there: a local copy paced like a link
    const uint64_t t0 = wall_clock64();
    uint64_t total = 0;
    for (uint32_t i = 0; i < a.npieces; i++)
    {
      const p2p_piece& pc = a.piece[i];
      const uint64_t at = static_cast<uint64_t>(peer) * pc.stride + pc.offset;
      p2p_copy(pc.recv_local + at, pc.send + at, pc.bytes, b, parts);
      total += pc.bytes;
    }
    if (!self)
    { // the transfer takes latency + bytes / rate on the link to that peer, of which the copy above is a part
      const uint64_t need = a.emulate_latency_ticks + static_cast<uint64_t>(static_cast<float>(total) / a.emulate_bytes_per_tick);
      __syncthreads();
      if (threadIdx.x == 0)
        while (wall_clock64() - t0 < need) __builtin_amdgcn_s_sleep(64); // one lane waits, as in the real rendezvous
      __syncthreads();
    }
    return;
  }
  uint32_t* const peer_flags = reinterpret_cast<uint32_t*>(a.peer_tab[KW_P2P_NBUF * KW_P2P_MAX_RANKS + peer]);
  if (threadIdx.x == 0)
  {
    if (b == 0) __hip_atomic_store(peer_flags + p2p_credit_at(a.slot, a.rank), a.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    ok = p2p_wait_flag(a.my_flags + p2p_credit_at(a.slot, peer), a.epoch, a.timeout_ticks) ? 1u : 0u;
    if (!ok) p2p_fail(a.status, 1u, a.slot, peer);
  }
  __syncthreads();
  if (!ok) return;
  for (uint32_t i = 0; i < a.npieces; i++)
  {
    const p2p_piece& pc = a.piece[i];
    char* const dst = reinterpret_cast<char*>(a.peer_tab[pc.recv_buf * KW_P2P_MAX_RANKS + peer]) + pc.recv_off +
                      static_cast<uint64_t>(a.rank) * pc.stride + pc.offset;
    p2p_copy(dst, pc.send + static_cast<uint64_t>(peer) * pc.stride + pc.offset, pc.bytes, b, a.nb);
  }
  // every wave: stores written back beyond this XCD's L2 and acknowledged, before the flag leaves
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0)
  {
    __hip_atomic_store(peer_flags + p2p_full_at(a.slot, a.rank, b), a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (!p2p_wait_flag(a.my_flags + p2p_full_at(a.slot, peer, b), a.epoch, a.timeout_ticks)) p2p_fail(a.status, 2u, a.slot, peer);
  }
}

// which registered buffer holds [p, p + bytes)?
int find_registered(const kw_comm_state* st, const void* p, uint64_t bytes, uint64_t* off)
{
  const uint64_t a = reinterpret_cast<uint64_t>(p);
  for (int i = 0; i < KW_P2P_NBUF; i++)
    if (st->reg[i].bytes != 0 && a >= st->reg[i].base && a + bytes <= st->reg[i].base + st->reg[i].bytes)
    {
      *off = a - st->reg[i].base;
      return i;
    }
  return -1;
}

kw_status p2p_check_status(kw_comm_state* st)
{
  if (st->status == nullptr) return KW_OK;
  const uint32_t s = *const_cast<volatile uint32_t*>(st->status);
  if (s == 0) return KW_OK;
  kw_set_error("kw_comm (P2P): rank %u gave up waiting for rank %u in exchange slot %u (%s): a peer is late by more than the "
               "time-out, or has stopped", st->rank, (s >> 16) & 0xffu, (s >> 8) & 0xffu,
               (s & 0xffu) == 1u ? "no credit: the peer never started this exchange" : "its data never arrived");
  return KW_ERR_COMM;
}

kw_status p2p_start(kw_ctx* ctx, kw_comm_state* st, int slot, const kw_comm_piece* pieces, int n)
{
  if (st->transport == kw_comm_state::P2P_PENDING)
  {
    kw_set_error("kw_comm (P2P): exchange started before kw_comm_p2p_connect");
    return KW_ERR_STATE;
  }
  KW_TRY_STATUS(p2p_check_status(st));
  p2p_args a{};
  a.nranks = st->nranks; a.rank = st->rank; a.slot = static_cast<uint32_t>(slot);
  a.epoch = ++st->epoch[slot];
  a.nb = static_cast<uint32_t>(ctx->tuning.p2p_blocks_per_peer);
  a.my_flags = st->flags; a.peer_tab = st->peer_tab; a.status = st->status;
  a.timeout_ticks = static_cast<uint64_t>(static_cast<double>(ctx->tuning.p2p_timeout_s) * 1.0e8);
  if (st->transport == kw_comm_state::P2P_EMULATED)
  {
    a.emulate_bytes_per_tick = st->emulate_gbs * 10.f; // GB/s = bytes/ns; one tick = 10 ns
    a.emulate_latency_ticks  = static_cast<uint32_t>(st->emulate_latency_us * 100.f);
  }
  for (int i = 0; i < n; i++)
  {
    const kw_comm_piece& pc = pieces[i];
    if (pc.bytes == 0) continue;
    p2p_piece& q = a.piece[a.npieces++];
    q.send = static_cast<const char*>(pc.send);
    q.recv_local = static_cast<char*>(pc.recv);
    q.stride = pc.stride; q.offset = pc.offset; q.bytes = pc.bytes;
    uint64_t off = 0;
    const int rb = find_registered(st, pc.recv, pc.stride * st->nranks, &off);
    if (rb < 0 && st->transport == kw_comm_state::P2P && st->nranks > 1)
    {
      kw_set_error("kw_comm (P2P): receive buffer %p is not one of the pipeline's exchange buffers", pc.recv);
      return KW_ERR_INVALID;
    }
    q.recv_buf = static_cast<uint32_t>(rb < 0 ? 0 : rb);
    q.recv_off = off;
  }
  if (a.npieces == 0) return KW_OK;
  hipLaunchKernelGGL(k_p2p_exchange, dim3((st->nranks - 1) * a.nb + KW_P2P_SELF_BLOCKS), dim3(256), 0, st->stream, a);
  KW_LAUNCH_CHECK();
  return KW_OK;
}
} // namespace

// ---- used by the pipeline (kw_fused.hip) ----------------------------------------------------------------------------
// One all-to-all of up to eight strided pieces (per array the rows of a chunk of planes and, behind them, the same
// planes of the x-Nyquist side array; up to three arrays): every peer's part of every piece goes out in the same RCCL
// group / the same P2P kernel, so the set costs one launch on the communication stream.
kw_status kw_comm_exchange_start_pieces(kw_ctx* ctx, int slot, const kw_comm_piece* pieces, int n)
{
  kw_comm_state* st = ctx->comm;
  if (st == nullptr) { kw_set_error("kw_comm: no communicator (kw_comm_init has not been called)"); return KW_ERR_STATE; }
  KW_REQUIRE(slot >= 0 && slot < KW_COMM_SLOTS && pieces != nullptr && n >= 1 && n <= 8);
  for (int i = 0; i < n; i++)
    KW_REQUIRE(pieces[i].send != nullptr && pieces[i].recv != nullptr && pieces[i].bytes % sizeof(float) == 0 &&
               pieces[i].offset + pieces[i].bytes <= pieces[i].stride);
  KW_HIP(hipEventRecord(st->ready[slot], ctx->stream));
  KW_HIP(hipStreamWaitEvent(st->stream, st->ready[slot], 0));
  if (st->transport != kw_comm_state::RCCL) KW_TRY_STATUS(p2p_start(ctx, st, slot, pieces, n));
  else
  {
    KW_NCCL(st, st->groupStart());
    for (uint32_t q = 0; q < st->nranks; q++)
    {
      const uint32_t peer = (st->rank + 1 + q) % st->nranks;
      for (int i = 0; i < n; i++)
      {
        const kw_comm_piece& pc = pieces[i];
        if (pc.bytes == 0) continue;
        const size_t at = peer * pc.stride + pc.offset;
        KW_NCCL(st, st->send(static_cast<const char*>(pc.send) + at, pc.bytes / sizeof(float), ncclFloat, static_cast<int>(peer),
                             st->comm, st->stream));
        KW_NCCL(st, st->recv(static_cast<char*>(pc.recv) + at, pc.bytes / sizeof(float), ncclFloat, static_cast<int>(peer), st->comm,
                             st->stream));
      }
    }
    KW_NCCL(st, st->groupEnd());
  }
  KW_HIP(hipEventRecord(st->done[slot], st->stream));
  st->exchanges++;
  return KW_OK;
}

kw_status kw_comm_exchange_start2(kw_ctx* ctx, int slot, const void* send, void* recv, size_t bytes_per_peer,
                                  const void* send2, void* recv2, size_t bytes_per_peer2)
{
  const kw_comm_piece pc[2] = { { send, recv, bytes_per_peer, 0, bytes_per_peer }, { send2, recv2, bytes_per_peer2, 0, bytes_per_peer2 } };
  return kw_comm_exchange_start_pieces(ctx, slot, pc, bytes_per_peer2 != 0 ? 2 : 1);
}

kw_status kw_comm_exchange_start(kw_ctx* ctx, int slot, const void* send, void* recv, size_t bytes_per_peer)
{
  return kw_comm_exchange_start2(ctx, slot, send, recv, bytes_per_peer, nullptr, nullptr, 0);
}

kw_status kw_comm_sync(kw_ctx* ctx)
{
  if (ctx->comm == nullptr || ctx->comm->stream == nullptr) return KW_OK;
  KW_HIP(hipStreamSynchronize(ctx->comm->stream));
  return p2p_check_status(ctx->comm);
}

kw_status kw_comm_check(kw_ctx* ctx) { return (ctx->comm != nullptr) ? p2p_check_status(ctx->comm) : KW_OK; }

void kw_comm_buffers_gone(kw_ctx* ctx)
{ // the pipeline frees its scratch: forget the registration; mappings of the peers' buffers stay until kw_comm_destroy
  kw_comm_state* st = ctx->comm;
  if (st == nullptr) return;
  for (auto& r : st->reg) r = {};
  if (st->transport == kw_comm_state::P2P) st->transport = kw_comm_state::P2P_PENDING;
}

kw_status kw_comm_exchange_wait(kw_ctx* ctx, int slot)
{
  kw_comm_state* st = ctx->comm;
  if (st == nullptr) { kw_set_error("kw_comm: no communicator"); return KW_ERR_STATE; }
  KW_REQUIRE(slot >= 0 && slot < KW_COMM_SLOTS);
  KW_HIP(hipStreamWaitEvent(ctx->stream, st->done[slot], 0));
  return KW_OK;
}

extern "C" {

kw_status kw_comm_unique_id(void* out_id, size_t bytes) { return kw_comm_unique_id_from(nullptr, out_id, bytes); }

kw_status kw_comm_unique_id_from(const char* rccl_library, void* out_id, size_t bytes)
{
  KW_REQUIRE(out_id != nullptr && bytes >= KW_COMM_ID_BYTES);
  static_assert(sizeof(ncclUniqueId) == KW_COMM_ID_BYTES, "KW_COMM_ID_BYTES must match ncclUniqueId");
  kw_comm_state tmp;
  const kw_status st = bind_rccl(&tmp, rccl_library);
  if (st != KW_OK) return st;
  ncclUniqueId id;
  KW_NCCL(&tmp, tmp.getUniqueId(&id));
  memcpy(out_id, &id, sizeof(id));
  return KW_OK;
}

kw_status kw_comm_init(kw_ctx* ctx, uint32_t nranks, uint32_t rank, const void* unique_id)
{
  return kw_comm_init_with(ctx, nullptr, nranks, rank, unique_id);
}

kw_status kw_comm_init_with(kw_ctx* ctx, const char* rccl_library, uint32_t nranks, uint32_t rank, const void* unique_id)
{
  KW_CHECK_CTX(ctx);
  KW_REQUIRE(nranks >= 1 && rank < nranks && unique_id != nullptr);
  if (ctx->comm != nullptr) { kw_set_error("kw_comm_init: the context already has a communicator"); return KW_ERR_STATE; }
  if (ctx->fused.ready) { kw_set_error("kw_comm_init: must be called before kw_fused_create"); return KW_ERR_STATE; }
  KW_HIP(hipSetDevice(ctx->device));
  kw_comm_state* st = new kw_comm_state();
  kw_status rc = bind_rccl(st, rccl_library);
  if (rc != KW_OK) { delete st; return rc; }
  st->nranks = nranks;
  st->rank   = rank;
  auto fail = [&](kw_status code) { release(st); return code; };
  rc = create_stream_and_events(st);
  if (rc != KW_OK) return fail(rc);
  ncclUniqueId id;
  memcpy(&id, unique_id, sizeof(id));
  const ncclResult_t r = st->commInitRank(&st->comm, static_cast<int>(nranks), id, static_cast<int>(rank));
  if (r != ncclSuccess)
  {
    kw_set_error("RCCL error: %s routine name: kw_comm_init (ncclCommInitRank, rank %u of %u)", st->getErrorString(r), rank, nranks);
    st->comm = nullptr;
    return fail(KW_ERR_COMM);
  }
  ctx->comm = st;
  return KW_OK;
}

// ---- P2P transport: set-up ------------------------------------------------------------------------------------------
kw_status kw_comm_init_p2p(kw_ctx* ctx, uint32_t nranks, uint32_t rank)
{
  KW_CHECK_CTX(ctx);
  KW_REQUIRE(nranks >= 1 && nranks <= KW_P2P_MAX_RANKS && rank < nranks);
  if (ctx->fused.ready) { kw_set_error("kw_comm_init_p2p: must be called before kw_fused_create"); return KW_ERR_STATE; }
  KW_HIP(hipSetDevice(ctx->device));
  kw_comm_state* st = ctx->comm;
  if (st != nullptr)
  { // on top of an RCCL communicator of the same shape: RCCL stays behind it (kw_comm_p2p_disconnect falls back to it)
    if (st->nranks != nranks || st->rank != rank)
    {
      kw_set_error("kw_comm_init_p2p: rank %u of %u does not match the communicator (rank %u of %u)", rank, nranks, st->rank, st->nranks);
      return KW_ERR_INVALID;
    }
  }
  else
  {
    st = new kw_comm_state();
    st->nranks = nranks;
    st->rank   = rank;
    const kw_status rc = create_stream_and_events(st);
    if (rc != KW_OK) { release(st); return rc; }
    ctx->comm = st;
  }
  auto fail = [&](const char* what, hipError_t e) {
    kw_set_error("kw_comm_init_p2p: %s: %s", what, hipGetErrorString(e));
    return (e == hipErrorOutOfMemory) ? KW_ERR_ALLOC : KW_ERR_HIP;
  };
  // flag words: written by the peers with system-scope stores, polled here with system-scope loads; uncached device
  // memory where the runtime offers it (no line of it may linger in an L2), fine-grained or plain otherwise
  const size_t fbytes = KW_P2P_FLAG_WORDS * sizeof(uint32_t);
  hipError_t e = hipExtMallocWithFlags(reinterpret_cast<void**>(&st->flags), fbytes, hipDeviceMallocUncached);
  if (e != hipSuccess) { (void)hipGetLastError(); e = hipExtMallocWithFlags(reinterpret_cast<void**>(&st->flags), fbytes, hipDeviceMallocFinegrained); }
  if (e != hipSuccess) { (void)hipGetLastError(); e = hipMalloc(reinterpret_cast<void**>(&st->flags), fbytes); }
  if (e != hipSuccess) return fail("flag words", e);
  if ((e = hipMemset(st->flags, 0, fbytes)) != hipSuccess) return fail("flag words", e);
  if ((e = hipMalloc(reinterpret_cast<void**>(&st->peer_tab), (KW_P2P_NBUF + 1) * KW_P2P_MAX_RANKS * sizeof(uint64_t))) != hipSuccess)
    return fail("peer table", e);
  if ((e = hipHostMalloc(reinterpret_cast<void**>(&st->status), sizeof(uint32_t), hipHostMallocMapped)) != hipSuccess)
    return fail("status word", e);
  *st->status = 0;
  st->transport = kw_comm_state::P2P_PENDING;
  return KW_OK;
}

kw_status kw_comm_p2p_export(kw_ctx* ctx, void* blob, size_t bytes)
{
  KW_CHECK_CTX(ctx);
  KW_REQUIRE(blob != nullptr && bytes >= KW_COMM_P2P_BLOB_BYTES);
  kw_comm_state* st = ctx->comm;
  if (st == nullptr || st->flags == nullptr) { kw_set_error("kw_comm_p2p_export: kw_comm_init_p2p has not been called"); return KW_ERR_STATE; }
  if (!ctx->fused.ready || !ctx->fused.slab)
  {
    kw_set_error("kw_comm_p2p_export: needs the slab pipeline's buffers (kw_fused_set_slab + kw_fused_create first)");
    return KW_ERR_STATE;
  }
  KW_HIP(hipSetDevice(ctx->device));
  memset(blob, 0, KW_COMM_P2P_BLOB_BYTES);
  p2p_blob b{};
  b.magic = KW_P2P_MAGIC; b.version = 1; b.nranks = st->nranks; b.rank = st->rank;
  b.pid = static_cast<int64_t>(getpid());
  b.host = host_id();
  b.device = ctx->device;
  const auto& f = ctx->fused;
  const uint64_t elems = static_cast<uint64_t>(f.Palloc) * ctx->c.ny * ctx->c.nz * sizeof(float2);
  void* bufs[KW_P2P_NBUF + 1] = { f.s[0], f.s[1], f.s[2], f.t[0], f.t[1], f.t[2], f.r[0], f.r[1], f.r[2], st->flags };
  for (int i = 0; i <= KW_P2P_NBUF; i++)
  {
    if (bufs[i] == nullptr) continue;
    const uint64_t nbytes = (i == KW_P2P_NBUF) ? KW_P2P_FLAG_WORDS * sizeof(uint32_t) : elems;
    hipDeviceptr_t base = nullptr;
    size_t         size = 0;
    KW_HIP(hipMemGetAddressRange(&base, &size, bufs[i])); // caller-owned scratch may lie inside a larger allocation
    b.buf[i].alloc  = reinterpret_cast<uint64_t>(base);
    b.buf[i].offset = reinterpret_cast<uint64_t>(bufs[i]) - reinterpret_cast<uint64_t>(base);
    b.buf[i].bytes  = nbytes;
    if (st->nranks > 1) KW_HIP(hipIpcGetMemHandle(&b.buf[i].handle, base));
    if (i < KW_P2P_NBUF) { st->reg[i].base = reinterpret_cast<uint64_t>(bufs[i]); st->reg[i].bytes = nbytes; }
  }
  memcpy(blob, &b, sizeof(b));
  return KW_OK;
}

kw_status kw_comm_p2p_connect(kw_ctx* ctx, const void* all_blobs)
{
  KW_CHECK_CTX(ctx);
  kw_comm_state* st = ctx->comm;
  if (st == nullptr || st->flags == nullptr) { kw_set_error("kw_comm_p2p_connect: kw_comm_init_p2p has not been called"); return KW_ERR_STATE; }
  KW_REQUIRE(all_blobs != nullptr);
  if (st->reg[0].bytes == 0) { kw_set_error("kw_comm_p2p_connect: kw_comm_p2p_export has not been called"); return KW_ERR_STATE; }
  KW_HIP(hipSetDevice(ctx->device));
  std::vector<uint64_t> tab((KW_P2P_NBUF + 1) * KW_P2P_MAX_RANKS, 0);
  const int64_t  pid  = static_cast<int64_t>(getpid());
  const uint64_t host = host_id();
  struct seen_t { hipIpcMemHandle_t h; void* p; };
  std::vector<seen_t> seen; // one mapping per remote allocation, however many buffers lie in it
  for (uint32_t q = 0; q < st->nranks; q++)
  {
    p2p_blob b;
    memcpy(&b, static_cast<const char*>(all_blobs) + static_cast<size_t>(q) * KW_COMM_P2P_BLOB_BYTES, sizeof(b));
    if (b.magic != KW_P2P_MAGIC || b.version != 1 || b.nranks != st->nranks || b.rank != q)
    {
      kw_set_error("kw_comm_p2p_connect: entry %u is not the blob of rank %u of %u (kw_comm_p2p_export, gathered in rank order)", q, q, st->nranks);
      return KW_ERR_INVALID;
    }
    if (b.host != host)
    {
      kw_set_error("kw_comm_p2p_connect: rank %u runs on another machine: the P2P transport is single-node", q);
      return KW_ERR_COMM;
    }
    for (int i = 0; i <= KW_P2P_NBUF; i++)
    {
      if (b.buf[i].bytes == 0) continue;
      void* base = nullptr;
      if (b.pid == pid)
      { // a thread of this process (or this rank itself): the pointer is valid here; another device needs peer access
        base = reinterpret_cast<void*>(b.buf[i].alloc);
        if (b.device != ctx->device)
        {
          const hipError_t e = hipDeviceEnablePeerAccess(b.device, 0);
          if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) KW_HIP(e);
          (void)hipGetLastError();
        }
      }
      else
      {
        for (const seen_t& s : seen)
          if (memcmp(&s.h, &b.buf[i].handle, sizeof(hipIpcMemHandle_t)) == 0) { base = s.p; break; }
        if (base == nullptr)
        {
          const hipError_t e = hipIpcOpenMemHandle(&base, b.buf[i].handle, hipIpcMemLazyEnablePeerAccess);
          if (e != hipSuccess)
          {
            kw_set_error("kw_comm_p2p_connect: cannot map buffer %d of rank %u (hipIpcOpenMemHandle: %s)", i, q, hipGetErrorString(e));
            return KW_ERR_COMM;
          }
          seen.push_back(seen_t{ b.buf[i].handle, base });
          st->opened.push_back(base);
        }
      }
      tab[static_cast<size_t>(i) * KW_P2P_MAX_RANKS + q] = reinterpret_cast<uint64_t>(base) + b.buf[i].offset;
    }
  }
  KW_HIP(hipMemcpy(st->peer_tab, tab.data(), tab.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
  st->transport = kw_comm_state::P2P;
  return KW_OK;
}

kw_status kw_comm_p2p_emulate(kw_ctx* ctx, float link_gbs, float latency_us)
{
  KW_CHECK_CTX(ctx);
  kw_comm_state* st = ctx->comm;
  if (st == nullptr || st->flags == nullptr) { kw_set_error("kw_comm_p2p_emulate: kw_comm_init_p2p has not been called"); return KW_ERR_STATE; }
  KW_REQUIRE(link_gbs > 0.f && latency_us >= 0.f);
  st->emulate_gbs = link_gbs;
  st->emulate_latency_us = latency_us;
  st->transport = kw_comm_state::P2P_EMULATED;
  return KW_OK;
}

kw_status kw_comm_destroy(kw_ctx* ctx)
{
  KW_CHECK_CTX(ctx);
  if (ctx->comm == nullptr) return KW_OK;
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  release(ctx->comm);
  ctx->comm = nullptr;
  return KW_OK;
}

kw_status kw_comm_info(kw_ctx* ctx, uint32_t* nranks, uint32_t* rank, uint64_t* exchanges)
{
  KW_CHECK_CTX(ctx);
  if (nranks) *nranks = ctx->comm ? ctx->comm->nranks : 0;
  if (rank) *rank = ctx->comm ? ctx->comm->rank : 0;
  if (exchanges) *exchanges = ctx->comm ? ctx->comm->exchanges : 0;
  return KW_OK;
}

kw_status kw_comm_transport(kw_ctx* ctx, int* out)
{
  KW_CHECK_CTX(ctx);
  KW_REQUIRE(out != nullptr);
  *out = (ctx->comm == nullptr) ? -1 : static_cast<int>(ctx->comm->transport);
  return KW_OK;
}

} // extern "C"
