"""Dev check: RCCL (backend "nccl") initialises and runs the collectives bench.py uses, on however many ranks were launched."""
import os, torch, torch.distributed as dist
rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
dev = torch.device("cuda", torch.cuda.current_device())
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
x = torch.ones(13 * 500000, device=dev)
dist.all_reduce(x); dist.barrier()
t = torch.tensor([1.5], dtype=torch.float64, device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX)
torch.cuda.synchronize()
print("rank", rank, "ok", float(x[0]), float(t))
dist.destroy_process_group()
