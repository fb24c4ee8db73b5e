import os, time, torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29551", RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
for n in (16_000_000, 500_000, 100_000, 1_000):
    t = torch.randn(n, device="cuda")
    for _ in range(3): dist.all_reduce(t)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): dist.all_reduce(t)
    torch.cuda.synchronize(); sync = (time.perf_counter() - t0) / 20
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        w = dist.all_reduce(t, async_op=True); w.wait()
    torch.cuda.synchronize(); asy = (time.perf_counter() - t0) / 20
    print("n=%d floats: sync %.1f us, async+wait %.1f us" % (n, sync * 1e6, asy * 1e6))
dist.destroy_process_group()
