import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
x = torch.zeros(1024 * 21, dtype=torch.int32, device=dev); y = torch.zeros_like(x)
a = torch.randn(2048, 2048, device=dev)
def run(n, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for _ in range(20): dist.all_gather_into_tensor(y, x)
print("all_gather sync-op  us/call:", run(500, lambda: dist.all_gather_into_tensor(y, x)))
def f():
    w = dist.all_gather_into_tensor(y, x, async_op=True); w.wait()
print("all_gather async+wait us/call:", run(500, f))
hs = []
def g():
    hs.append(dist.all_gather_into_tensor(y, x, async_op=True))
    if len(hs) > 2: hs.pop(0).wait()
print("all_gather async pipelined us/call:", run(500, g))
print("matmul only us:", run(200, lambda: a @ a))
def h():
    b = a @ a
    hs.append(dist.all_gather_into_tensor(y, x, async_op=True))
    if len(hs) > 2: hs.pop(0).wait()
print("matmul + pipelined gather us:", run(200, h))
dist.destroy_process_group()
