import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
for log2n in (26, 27, 28, 29):
    n = 1 << log2n
    x = torch.arange(n, dtype=torch.int64, device="cuda"); y = torch.full((n,), -1, dtype=torch.int64, device="cuda")
    dist.all_to_all_single(y, x, output_split_sizes=[n], input_split_sizes=[n]); torch.cuda.synchronize()
    bad = int((y != x).sum())
    w = dist.all_to_all_single(y.fill_(-1), x, output_split_sizes=[n], input_split_sizes=[n], async_op=True); w.wait(); torch.cuda.synchronize()
    print("2^%d sync bad=%d async bad=%d first_bad=%s" % (log2n, bad, int((y != x).sum()), int((y != x).nonzero()[0]) if int((y != x).sum()) else None))
dist.destroy_process_group()
