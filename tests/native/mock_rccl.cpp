// mock_rccl.cpp — TEST INFRASTRUCTURE, not part of the product: a stand-in for librccl.so.1 whose "ranks" are threads
// of ONE process sharing ONE GPU, so that the device library's own exchange path (csrc/kw_comm.hip: groups of
// ncclSend / ncclRecv on a communication stream) can be driven with 2 ... 8 ranks on a one-GPU box.  RCCL itself refuses
// two ranks on one device, and the box has one.
//
// Semantics kept from RCCL for what kw_comm.hip uses: point-to-point operations are matched per ordered (sender,
// receiver) pair in program order; everything between ncclGroupStart and ncclGroupEnd is issued together, so a rank may
// send to and receive from every peer in one group without deadlock; operations are stream-ordered on the stream they
// are given.  Transport: a device-to-device copy on the RECEIVER's stream after the sender's "data ready" event; the
// sender's stream waits for the receiver's "copied" event before the group counts as finished on it.
//
//   selected with KW_RCCL_LIB=<this library>  (kw_comm.hip binds RCCL at run time)
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

extern "C" {
typedef struct mockComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;   // 0 = ncclSuccess
typedef int ncclDataType_t; // 7 = ncclFloat
}

namespace
{
struct Post
{
  const void* ptr   = nullptr;
  size_t      bytes = 0;
  hipEvent_t  ready = nullptr;
  hipEvent_t  copied = nullptr;
  bool        has_copied = false;
};
struct World
{
  int nranks = 0, joined = 0;
  std::map<std::tuple<int, int, uint64_t>, Post> mail;       // (src, dst, sequence number of the pair) -> posted send
  std::map<std::pair<int, int>, uint64_t> send_seq, recv_seq; // per ordered pair
};
std::mutex              g_mu;
std::condition_variable g_cv;
std::map<uint64_t, World> g_worlds; // by the number carried in the unique id
uint64_t                g_next_world = 1;

struct Op { bool send; const void* sptr; void* rptr; size_t bytes; int peer; mockComm* comm; hipStream_t stream; };
thread_local std::vector<Op> t_ops;
thread_local int             t_depth = 0;
} // namespace

struct mockComm { World* w; int rank; };

namespace
{
ncclResult_t flush()
{
  std::vector<Op> ops;
  ops.swap(t_ops);
  std::vector<std::tuple<int, int, uint64_t>> my_sends;
  { // 1. post every send with its "data ready" event
    std::unique_lock<std::mutex> lk(g_mu);
    for (const Op& o : ops)
    {
      if (!o.send) continue;
      World* w = o.comm->w;
      Post p;
      p.ptr = o.sptr; p.bytes = o.bytes;
      if (hipEventCreateWithFlags(&p.ready, hipEventDisableTiming) != hipSuccess) return 1;
      if (hipEventRecord(p.ready, o.stream) != hipSuccess) return 1;
      const auto key = std::make_tuple(o.comm->rank, o.peer, w->send_seq[{o.comm->rank, o.peer}]++);
      w->mail[key] = p;
      my_sends.push_back(key);
    }
    g_cv.notify_all();
  }
  for (const Op& o : ops)
  { // 2. every receive: wait for the matching post, copy on this rank's stream, tell the sender
    if (o.send) continue;
    World* w = o.comm->w;
    std::unique_lock<std::mutex> lk(g_mu);
    const auto key = std::make_tuple(o.peer, o.comm->rank, w->recv_seq[{o.peer, o.comm->rank}]++);
    g_cv.wait(lk, [&] { return w->mail.count(key) != 0; });
    Post& p = w->mail[key];
    if (p.bytes != o.bytes) return 2; // size mismatch between the two sides: what RCCL would hang or corrupt on
    if (hipStreamWaitEvent(o.stream, p.ready, 0) != hipSuccess) return 1;
    if (hipMemcpyAsync(o.rptr, p.ptr, o.bytes, hipMemcpyDeviceToDevice, o.stream) != hipSuccess) return 1;
    if (hipEventCreateWithFlags(&p.copied, hipEventDisableTiming) != hipSuccess) return 1;
    if (hipEventRecord(p.copied, o.stream) != hipSuccess) return 1;
    p.has_copied = true;
    g_cv.notify_all();
  }
  size_t k = 0;
  for (const Op& o : ops)
  { // 3. every send: the buffer is free again on this rank's stream once the receiver has copied it
    if (!o.send) continue;
    World* w = o.comm->w;
    std::unique_lock<std::mutex> lk(g_mu);
    const auto key = my_sends[k++];
    g_cv.wait(lk, [&] { return w->mail[key].has_copied; });
    Post p = w->mail[key];
    w->mail.erase(key);
    lk.unlock();
    if (hipStreamWaitEvent(o.stream, p.copied, 0) != hipSuccess) return 1;
    (void)hipEventDestroy(p.ready);  // (released by the runtime once the enqueued waits have passed)
    (void)hipEventDestroy(p.copied);
  }
  return 0;
}
} // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id)
{
  std::lock_guard<std::mutex> lk(g_mu);
  memset(id, 0, sizeof(*id));
  const uint64_t n = g_next_world++;
  memcpy(id->internal, &n, sizeof(n));
  return 0;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank)
{
  uint64_t n = 0;
  memcpy(&n, id.internal, sizeof(n));
  std::unique_lock<std::mutex> lk(g_mu);
  World& w = g_worlds[n];
  if (w.nranks == 0) w.nranks = nranks;
  if (w.nranks != nranks || rank < 0 || rank >= nranks) return 4;
  w.joined++;
  g_cv.notify_all();
  g_cv.wait(lk, [&] { return w.joined >= w.nranks; }); // collective, like the real one
  *comm = new mockComm{&w, rank};
  return 0;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) { delete comm; return 0; }
ncclResult_t ncclGroupStart(void) { t_depth++; return 0; }
ncclResult_t ncclGroupEnd(void) { return (--t_depth == 0) ? flush() : 0; }

ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream)
{
  if (type != 7) return 4;
  t_ops.push_back(Op{true, buf, nullptr, count * sizeof(float), peer, comm, stream});
  return (t_depth == 0) ? flush() : 0;
}
ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream)
{
  if (type != 7) return 4;
  t_ops.push_back(Op{false, nullptr, buf, count * sizeof(float), peer, comm, stream});
  return (t_depth == 0) ? flush() : 0;
}
const char* ncclGetErrorString(ncclResult_t r)
{
  return r == 0 ? "no error" : r == 2 ? "mock RCCL: send / receive sizes of a pair differ" : r == 4 ? "mock RCCL: invalid argument" : "mock RCCL: HIP error";
}

} // extern "C"
