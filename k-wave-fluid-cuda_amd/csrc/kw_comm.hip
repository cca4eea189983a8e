// kw_comm.hip — the all-to-all of the Z-slab decomposition, inside the device library: an RCCL communicator owned by
// the context, a dedicated communication stream, and split-phase exchanges ordered against the compute stream with
// events (SURVEY §8b last row, §8e "Collective").  New with this build: the reference is single-GPU (Readme.md:12-13).
//
//   start(slot, send, recv):  ready[slot] <- compute stream;  comm stream waits for it;
//                             ncclGroupStart; per peer q: ncclSend(chunk q of send -> q), ncclRecv(chunk q of recv <- q);
//                             ncclGroupEnd;  done[slot] <- comm stream
//   wait(slot):               compute stream waits for done[slot]           (the host never blocks)
//
// xGMI is a point-to-point mesh: with all P - 1 peers addressed in one group every link of the GPU carries its own
// chunk at the same time.  RCCL is bound at run time (dlopen): a process that already holds an RCCL (torch) shares
// that instance, a plain C++ caller gets the ROCm one, and single-GPU users of this library need no RCCL at all.
#include "kw_internal.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdlib>

struct kw_comm_state
{
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

// binds the RCCL entry points; `st` keeps the handle
kw_status bind_rccl(kw_comm_state* st)
{
  const char* names[] = { getenv("KW_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
  const char* off = getenv("KW_RCCL_DISABLE"); // lets a test walk the caller's "no native exchange" path
  if (off != nullptr && off[0] == '1')
  {
    kw_set_error("kw_comm: RCCL binding disabled (KW_RCCL_DISABLE=1)");
    return KW_ERR_COMM;
  }
  for (const char* n : names)
  {
    if (n == nullptr || n[0] == '\0') continue;
    st->lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (st->lib != nullptr) break;
  }
  if (st->lib == nullptr)
  {
    kw_set_error("kw_comm: cannot load RCCL (librccl.so.1): %s", dlerror());
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
  for (int i = 0; i < KW_COMM_SLOTS; i++)
  {
    if (st->ready[i]) (void)hipEventDestroy(st->ready[i]);
    if (st->done[i]) (void)hipEventDestroy(st->done[i]);
  }
  if (st->stream) (void)hipStreamDestroy(st->stream);
  // the library handle stays open: RCCL keeps threads and registered memory that outlive a communicator
  delete st;
}
} // namespace

// ---- used by the pipeline (kw_fused.hip) ----------------------------------------------------------------------------
// One all-to-all of up to eight strided pieces (per array the rows of a chunk of planes and, behind them, the same
// planes of the x-Nyquist side array; up to three arrays): every peer's part of every piece goes out in the same RCCL group, so the set costs one launch
// of the communication kernel.
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
  KW_NCCL(st, st->groupStart());
  for (uint32_t q = 0; q < st->nranks; q++)
  {
    // peers are taken starting from the right-hand neighbour: rank r talks to r+1, r+2, ... — no two ranks open with
    // the same peer, so the first chunks of every rank go out on different links
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
  if (ctx->comm != nullptr && ctx->comm->stream != nullptr) KW_HIP(hipStreamSynchronize(ctx->comm->stream));
  return KW_OK;
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

kw_status kw_comm_unique_id(void* out_id, size_t bytes)
{
  KW_REQUIRE(out_id != nullptr && bytes >= KW_COMM_ID_BYTES);
  static_assert(sizeof(ncclUniqueId) == KW_COMM_ID_BYTES, "KW_COMM_ID_BYTES must match ncclUniqueId");
  kw_comm_state tmp;
  const kw_status st = bind_rccl(&tmp);
  if (st != KW_OK) return st;
  ncclUniqueId id;
  KW_NCCL(&tmp, tmp.getUniqueId(&id));
  memcpy(out_id, &id, sizeof(id));
  return KW_OK;
}

kw_status kw_comm_init(kw_ctx* ctx, uint32_t nranks, uint32_t rank, const void* unique_id)
{
  KW_CHECK_CTX(ctx);
  KW_REQUIRE(nranks >= 1 && rank < nranks && unique_id != nullptr);
  if (ctx->comm != nullptr) { kw_set_error("kw_comm_init: the context already has a communicator"); return KW_ERR_STATE; }
  if (ctx->fused.ready) { kw_set_error("kw_comm_init: must be called before kw_fused_create"); return KW_ERR_STATE; }
  KW_HIP(hipSetDevice(ctx->device));
  kw_comm_state* st = new kw_comm_state();
  kw_status rc = bind_rccl(st);
  if (rc != KW_OK) { delete st; return rc; }
  st->nranks = nranks;
  st->rank   = rank;
  auto fail = [&](kw_status code) { release(st); return code; };
  if (hipStreamCreateWithFlags(&st->stream, hipStreamNonBlocking) != hipSuccess)
  {
    kw_set_error("kw_comm_init: cannot create the communication stream");
    return fail(KW_ERR_HIP);
  }
  for (int i = 0; i < KW_COMM_SLOTS; i++)
    if (hipEventCreateWithFlags(&st->ready[i], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&st->done[i], hipEventDisableTiming) != hipSuccess)
    {
      kw_set_error("kw_comm_init: cannot create events");
      return fail(KW_ERR_HIP);
    }
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

} // extern "C"
