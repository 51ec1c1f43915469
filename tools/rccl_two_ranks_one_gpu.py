"""Can two RCCL ranks share the one GPU of a box?  (NCCL refuses duplicate devices; this records what RCCL on this image does.)
   python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 tools/rccl_two_ranks_one_gpu.py"""
import os
import torch
import torch.distributed as dist

rank = int(os.environ["RANK"])
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    x = torch.full((4,), float(rank), device="cuda")
    y = torch.empty(8, device="cuda")
    dist.all_gather_into_tensor(y, x)
    torch.cuda.synchronize()
    print(f"rank {rank}: all_gather over RCCL on one shared GPU -> {y.tolist()}", flush=True)
    dist.destroy_process_group()
except Exception as e:
    print(f"rank {rank}: RCCL refused: {type(e).__name__}: {str(e)[:300]}", flush=True)
