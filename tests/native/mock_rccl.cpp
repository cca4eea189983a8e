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
//   selected with named by the caller of kw_comm_init_with (kw_comm.hip binds RCCL at run time)
//
// Emulation mode (MOCK_RCCL_EMULATE=1; tools/emulate_rank.py): ONE rank of an N-rank run alone on the GPU.  Every peer is
// looped back to the rank itself (values become meaningless, sizes and buffers are the real ones): a group then costs
//   the launching thread   MOCK_GROUP_HOST_US  (default 44: what an RCCL group was measured to cost),
//   the communication stream   a delay of  latency + (bytes to the busiest peer) / MOCK_LINK_GBS  (default 10 us, 60 GB/s:
//                              every peer has its own xGMI link, so the links run in parallel), then
//   local HBM              one device copy of what the rank sends (a real transfer also reads and writes it locally,
//                              while the links are busy: the delay is shortened by the copy's expected duration).
// A timeline of one rank's kernels against modelled wire time — for choosing between the slab schedules, not a result.
#include <hip/hip_runtime.h>

#include <chrono>
#include <condition_variable>
#include <cstdlib>
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

bool   emulate() { static const bool e = getenv("MOCK_RCCL_EMULATE") != nullptr && getenv("MOCK_RCCL_EMULATE")[0] == '1'; return e; }
double env_or(const char* name, double dflt) { const char* v = getenv(name); return (v != nullptr && v[0] != 0) ? atof(v) : dflt; }

__global__ void k_wire_delay(unsigned long long ticks)
{ // holds the communication stream for the modelled transfer time without touching memory
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}
} // namespace

struct mockComm { World* w; int rank; };

namespace
{
ncclResult_t flush_emulated(const std::vector<Op>& ops)
{
  static const double host_us = env_or("MOCK_GROUP_HOST_US", 44.0), link_gbs = env_or("MOCK_LINK_GBS", 60.0),
                      latency_us = env_or("MOCK_LINK_LATENCY_US", 10.0);
  static int clock_khz = 0;
  if (clock_khz == 0)
  {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&clock_khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || clock_khz <= 0)
      clock_khz = 100000; // 100 MHz
  }
  if (ops.empty()) return 0;
  const auto t_enter = std::chrono::steady_clock::now();
  std::map<int, size_t> to_peer;
  for (const Op& o : ops) if (o.send && o.peer != o.comm->rank) to_peer[o.peer] += o.bytes;
  size_t busiest = 0;
  for (const auto& kv : to_peer) busiest = kv.second > busiest ? kv.second : busiest;
  double wire_us = (busiest == 0) ? 0.0 : latency_us + static_cast<double>(busiest) / (link_gbs * 1e3);
  if (wire_us > 20000.0) wire_us = 20000.0;
  // on a node the transfer's local reads and writes happen WHILE the links are busy; here they are a device copy
  // behind the delay on the same stream: the delay is shortened by the copy's expected duration (2 bytes moved per
  // byte sent at ~4 TB/s) so that delay + copy models the transfer (MOCK_COPY_SERIAL=1: plain sum, pessimistic)
  static const bool serial = getenv("MOCK_COPY_SERIAL") != nullptr && getenv("MOCK_COPY_SERIAL")[0] == '1';
  if (!serial && wire_us > 0.0)
  {
    size_t sent = 0;
    for (const Op& o : ops) if (o.send) sent += o.bytes;
    const double copy_us = 2.0 * static_cast<double>(sent) / 4.0e6;
    wire_us = (wire_us - copy_us > latency_us) ? wire_us - copy_us : latency_us;
  }
  hipStream_t stream = ops[0].stream;
  if (wire_us > 0.0)
    hipLaunchKernelGGL(k_wire_delay, dim3(1), dim3(1), 0, stream, static_cast<unsigned long long>(wire_us * clock_khz / 1000.0));
  // the i-th send to a peer pairs with the i-th receive from it: loop it back (the local HBM traffic of a real
  // transfer).  The per-peer parts of one piece are evenly spaced (peer * stride): one strided copy per piece.
  std::map<int, std::vector<const Op*>> sends, recvs;
  for (const Op& o : ops) (o.send ? sends : recvs)[o.peer].push_back(&o);
  const int    npeers  = static_cast<int>(sends.size());
  const size_t npieces = sends.empty() ? 0 : sends.begin()->second.size();
  for (size_t i = 0; i < npieces; i++)
  {
    const Op* s0 = sends.begin()->second[i];
    const Op* r0 = recvs.begin()->second.size() > i ? recvs.begin()->second[i] : nullptr;
    if (r0 == nullptr || r0->bytes != s0->bytes) return 2;
    size_t pitch = s0->bytes;
    if (npeers > 1)
    {
      auto it = sends.begin(); ++it;
      pitch = static_cast<size_t>(static_cast<const char*>(it->second[i]->sptr) - static_cast<const char*>(s0->sptr));
    }
    if (hipMemcpy2DAsync(r0->rptr, pitch, s0->sptr, pitch, s0->bytes, static_cast<size_t>(npeers), hipMemcpyDeviceToDevice,
                         stream) != hipSuccess) return 1;
  }
  // what the real library costs the launching thread, the calls above included
  while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_enter).count() < host_us) {}
  return 0;
}

ncclResult_t flush()
{
  std::vector<Op> ops;
  ops.swap(t_ops);
  if (emulate()) return flush_emulated(ops);
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
  if (!emulate()) g_cv.wait(lk, [&] { return w.joined >= w.nranks; }); // collective, like the real one
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
