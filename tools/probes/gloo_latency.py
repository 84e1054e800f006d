import os, time, torch, torch.distributed as dist
dist.init_process_group("gloo")
r = dist.get_rank(); n = dist.get_world_size()
mine = torch.tensor([float(r)], dtype=torch.float64)
out = [torch.zeros(1, dtype=torch.float64) for _ in range(n)]
for _ in range(50): dist.all_gather(out, mine)
dist.barrier()
t = time.perf_counter()
for _ in range(500): dist.all_gather(out, mine)
e = (time.perf_counter() - t) / 500
t = time.perf_counter()
for _ in range(200): dist.barrier()
b = (time.perf_counter() - t) / 200
if r == 0: print("ranks %d all_gather %.1f us barrier %.1f us" % (n, e * 1e6, b * 1e6))
