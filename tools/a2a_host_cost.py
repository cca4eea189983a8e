import os, time
for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29578"), ("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0")):
    os.environ.setdefault(k, v)
import torch, torch.distributed as dist
dist.init_process_group("nccl")
torch.cuda.set_device(0)
import sys
nfl = int(sys.argv[1]) if len(sys.argv) > 1 else (1 << 20)
a = torch.zeros(nfl, device="cuda"); b = torch.zeros(nfl, device="cuda")
stream = torch.cuda.Stream()
if len(sys.argv) > 2:
    torch.cuda.set_stream(stream)
for _ in range(20):
    w = dist.all_to_all_single(b, a, async_op=True); w.wait()
torch.cuda.synchronize()
N = 300
t0 = time.perf_counter()
for _ in range(N):
    w = dist.all_to_all_single(b, a, async_op=True); w.wait()
t1 = time.perf_counter(); torch.cuda.synchronize()
print(f"{nfl * 4 >> 20} MB, all_to_all_single async + wait: {1e6*(t1-t0)/N:.1f} us host per call")
pg = dist.distributed_c10d._get_default_group()
t0 = time.perf_counter()
for _ in range(N):
    w = pg.alltoall_base(b, a, [], []); w.wait()
t1 = time.perf_counter(); torch.cuda.synchronize()
print(f"pg.alltoall_base + wait: {1e6*(t1-t0)/N:.1f} us host per call")

# the same with other work enqueued on the stream between the collectives (as inside a time step)
dist.init_process_group("nccl") if not dist.is_initialized() else None
c = torch.zeros(nfl, device="cuda")
acc = 0.0
for _ in range(N):
    for _ in range(3):
        c.add_(1.0)
    t0 = time.perf_counter()
    w = dist.all_to_all_single(b, a, async_op=True)
    acc += time.perf_counter() - t0
    for _ in range(3):
        c.add_(1.0)
    t0 = time.perf_counter()
    w.wait()
    acc += time.perf_counter() - t0
torch.cuda.synchronize()
print(f"interleaved with 6 kernels per collective: {1e6 * acc / N:.1f} us host per call")
dist.destroy_process_group()
