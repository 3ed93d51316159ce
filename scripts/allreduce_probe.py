import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29533")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
torch.cuda.set_device(0)
small = torch.zeros(3, dtype=torch.float64, device="cuda")
big = torch.zeros(1745191, device="cuda")
x = torch.randn(4096, 4096, device="cuda")
for t, name in ((small, "24 B"), (big, "7 MB")):
    for _ in range(5): dist.all_reduce(t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): dist.all_reduce(t)
    torch.cuda.synchronize()
    print(name, "blocking-sequence:", (time.perf_counter() - t0) / 50 * 1e6, "us each")
    # interleaved with compute, async
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50):
        y = x @ x
        w = dist.all_reduce(t, async_op=True)
        y = x @ x
        w.wait()
    torch.cuda.synchronize()
    t1 = time.perf_counter() - t0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50):
        y = x @ x
        y = x @ x
    torch.cuda.synchronize()
    t2 = time.perf_counter() - t0
    print(name, "overhead when interleaved with two 4096^3 matmuls:", (t1 - t2) / 50 * 1e6, "us per all-reduce")
dist.destroy_process_group()
