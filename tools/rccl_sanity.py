"""One-rank RCCL sanity check of the collectives bench.py and sharded_db.py issue at N>1 (the GPU box has one GPU)."""
import os
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
x = torch.arange(8, dtype=torch.float32, device="cuda").reshape(2, 4)
y = torch.empty_like(x)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    dist.all_gather_into_tensor(y, x)
torch.cuda.current_stream().wait_stream(s)
dist.barrier(); torch.cuda.synchronize()
assert torch.equal(x, y)
t = torch.tensor([1.5], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t.item()) == 1.5
dist.destroy_process_group()
print("rccl ok")
