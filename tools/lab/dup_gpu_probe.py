# Can two ranks share ONE GPU in an RCCL communicator on this stack? (probe; decides whether the N = 2 path can be rehearsed on a 1-GPU box)
import os, sys, torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    x = torch.full((4,), float(rank + 1), device="cuda")
    dist.all_reduce(x)
    torch.cuda.synchronize()
    print("rank", rank, "all_reduce on a shared GPU ->", x.tolist(), flush=True)
    dist.destroy_process_group()
except Exception as e:
    print("rank", rank, "FAILED:", repr(e)[:300], flush=True)
    sys.exit(3)
