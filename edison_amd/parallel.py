"""Data-parallel scoring across the GPUs of one node: one process per GPU, utterances sharded contiguously,
and ONE collective -- an all-gather of the per-class int8 logits (10 B per utterance) over RCCL/xGMI.

The reference has nothing to mirror here (its only transport is a 115200-baud UART, hostinterface.c:96-112);
the shape of this module follows SURVEY.md section 8(e): rank r owns utterances [r*N/W, (r+1)*N/W), weights
and tables (~47 KB) are replicated, frames/utterances are independent so the data path needs no exchange, and
the gathered payload (N*10 bytes, 2.6 MB per rank at 2 097 152 utterances on 8 GPUs) is latency-bound on
xGMI's point-to-point links -- so a single un-bucketed all_gather is the right size.

Works with backend "nccl" (= RCCL on ROCm, CUDA tensors) and "gloo" (CPU tensors; used by the CPU tests).
"""
import os

import torch
import torch.distributed as dist


def shard_range(n_items, rank, world_size):
    """Contiguous shard [lo, hi) of rank; the first n_items % world_size ranks take one extra item."""
    if world_size < 1 or not (0 <= rank < world_size) or n_items < 0:
        raise ValueError("bad shard request")
    base, rem = divmod(n_items, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun's contract).
    Returns (rank, world_size, local_rank). Single-process runs need no rendezvous."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local_rank


def dist_available():
    """Can this process bind RCCL (edison_dist_available)? Creates nothing: no bootstrap thread, no socket."""
    from . import _lib
    return _lib.lib().edison_dist_available() == _lib.OK


def dist_unique_id():
    """128 opaque bytes from rank 0's RCCL (edison_dist_unique_id); every rank needs the same ones for dist_init."""
    import ctypes
    from . import _lib
    buf = ctypes.create_string_buffer(_lib.DIST_ID_BYTES)
    r = _lib.lib().edison_dist_unique_id(ctypes.cast(buf, ctypes.c_void_p))
    if r != _lib.OK:
        raise _lib.EdisonError(r, "edison_dist_unique_id failed (is RCCL installed? EDISON_RCCL_LIB)")
    return buf.raw


def init_context_comm(ctx, rank, world, device):
    """Join `ctx` to a communicator of its own behind the C-ABI: rank 0 creates the id, torch.distributed (whatever
    backend is up: gloo or nccl) carries the 128 bytes to the other ranks -- that is all torch does for this path."""
    if world == 1:
        return
    payload = torch.zeros(128, dtype=torch.uint8)
    if rank == 0:
        payload = torch.frombuffer(bytearray(dist_unique_id()), dtype=torch.uint8).clone()
    payload = payload.to(device) if dist.get_backend() == "nccl" else payload
    dist.broadcast(payload, src=0)
    # ncclCommInitRank is itself a collective; a rank that fails in it (or before it) must not leave the others believing
    # in the communicator: every rank reports, the minimum decides, and a partial success is torn down everywhere
    ok, err = 1, None
    try:
        ctx.dist_init(bytes(payload.cpu().numpy().tobytes()), rank, world)
    except Exception as e:  # noqa: BLE001 -- reported below, after the other ranks have been told
        ok, err = 0, e
    flag = torch.tensor([ok], dtype=torch.int32)
    flag = flag.to(device) if dist.get_backend() == "nccl" else flag
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) != 1:
        if ok:
            ctx.dist_shutdown()
        raise RuntimeError("edison_dist_init failed on %s" % ("this rank: %r" % (err,) if err else "another rank"))


def shard_range_c(n_items, rank, world_size):
    """edison_dist_shard_range: the C-ABI's copy of shard_range (the two must agree: tests/test_distributed_cpu.py)."""
    import ctypes
    from . import _lib
    lo, hi = ctypes.c_int64(), ctypes.c_int64()
    r = _lib.lib().edison_dist_shard_range(int(n_items), int(rank), int(world_size), ctypes.byref(lo), ctypes.byref(hi))
    if r != _lib.OK:
        raise ValueError("bad shard request")
    return lo.value, hi.value


class LogitsGatherer:
    """Pre-allocated all-gather of equal-sized logit shards ([n_local, 10] int8 -> [world*n_local, 10])."""

    def __init__(self, n_local, n_classes=10, device="cpu", group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.n_local, self.n_classes = int(n_local), int(n_classes)
        self.out = torch.empty((self.world * self.n_local, self.n_classes), dtype=torch.int8, device=device)

    def __call__(self, local_logits, async_op=False):
        if local_logits.shape != (self.n_local, self.n_classes) or local_logits.dtype != torch.int8:
            raise ValueError("expected int8 [%d, %d] logits" % (self.n_local, self.n_classes))
        if self.world == 1:
            self.out.copy_(local_logits)
            return self.out if not async_op else None
        work = dist.all_gather_into_tensor(self.out, local_logits.contiguous(), group=self.group, async_op=async_op)
        return work if async_op else self.out


def all_gather_logits(local_logits, n_total=None, group=None):
    """Gather possibly unequal shards (shard_range sizes): pads to the largest shard, gathers once, trims."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local_logits
    world = dist.get_world_size(group)
    n_local = local_logits.shape[0]
    if n_total is None:
        sizes = torch.tensor([n_local], dtype=torch.int64, device=local_logits.device)
        all_sizes = [torch.zeros_like(sizes) for _ in range(world)]
        dist.all_gather(all_sizes, sizes, group=group)
        counts = [int(s.item()) for s in all_sizes]
    else:
        counts = [shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0] for r in range(world)]
    m = max(counts)
    padded = local_logits
    if n_local < m:
        padded = torch.zeros((m,) + tuple(local_logits.shape[1:]), dtype=local_logits.dtype, device=local_logits.device)
        padded[:n_local] = local_logits
    out = torch.empty((world * m,) + tuple(local_logits.shape[1:]), dtype=local_logits.dtype, device=local_logits.device)
    dist.all_gather_into_tensor(out, padded.contiguous(), group=group)
    return torch.cat([out[r * m:r * m + counts[r]] for r in range(world)], dim=0)
