"""GPU tests of the multi-GPU entry points behind the C-ABI (edison_dist_*, edison_kws_batch_sharded_dev) at the one
world size a one-GPU box offers: a communicator of ONE rank built through RCCL itself (ncclGetUniqueId,
ncclCommInitRank, ncclAllGather all execute). N > 1 is unmeasured on hardware until the driver's 8-GPU run; its shard
arithmetic is covered on CPU (tests/test_distributed_cpu.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_world_of_one_through_rccl(oracle_mod, oracle_model):
    import torch
    from edison_amd import parallel
    from edison_amd.context import Context
    c = Context(0)
    try:
        assert c.dist_info() == (0, 1)
        dev = torch.device("cuda", 0)
        c.use_torch_stream()
        rng = np.random.default_rng(9)
        n = 37
        audio = torch.from_numpy(np.clip(rng.normal(0, 3000, n * 31744), -32768, 32767).astype(np.int16)).to(dev)
        lo = torch.zeros((n, 10), dtype=torch.int8, device=dev)
        so, am = torch.zeros_like(lo), torch.zeros((n,), dtype=torch.int32, device=dev)
        feat = torch.zeros((n, 403), dtype=torch.int8, device=dev)
        # outside a communicator the gather is a copy
        all0 = torch.full((n, 10), 99, dtype=torch.int8, device=dev)
        c.kws_sharded_t(audio, n, 31744, all0, feat=feat, logits=lo, softmax=so, argmax=am)
        torch.cuda.synchronize()
        assert torch.equal(all0, lo)
        # a real RCCL communicator of one rank
        c.dist_init(parallel.dist_unique_id(), 0, 1)
        assert c.dist_info() == (0, 1)
        all1 = torch.full((n, 10), 77, dtype=torch.int8, device=dev)
        c.kws_sharded_t(audio, n, 31744, all1, feat=feat, logits=lo, softmax=so, argmax=am)
        torch.cuda.synchronize()
        assert torch.equal(all1, lo)
        o = oracle_mod.cnn(oracle_model, feat.cpu().numpy())
        assert np.array_equal(all1.cpu().numpy(), o["logits"]) and np.array_equal(am.cpu().numpy(), o["argmax"])
        x = torch.randint(-128, 128, (1001, 10), dtype=torch.int32, device=dev).to(torch.int8)
        y = torch.zeros_like(x)
        c.allgather_logits_t(x, 1001, y)
        torch.cuda.synchronize()
        assert torch.equal(x, y)
        # the gather for batches the world size does not divide (edison_dist_allgather_logits_total): in a world of one
        # every total divides, so EDISON_DIST_FORCE_PADDED=1 sends it down the padded road all the same -- zeroed send block
        # of the largest shard + 1 row, ONE ncclAllGather into scratch, compaction copies -- through the real communicator
        import os
        os.environ["EDISON_DIST_FORCE_PADDED"] = "1"
        try:
            for n_odd in (1, 7, 1001):
                y2 = torch.full((n_odd, 10), 55, dtype=torch.int8, device=dev)
                c.allgather_logits_total_t(x[:n_odd].contiguous(), n_odd, y2)
                torch.cuda.synchronize()
                assert torch.equal(x[:n_odd], y2)
            all2 = torch.full((n, 10), 66, dtype=torch.int8, device=dev)
            c.kws_sharded_total_t(audio, n, 31744, all2, feat=feat, logits=lo, softmax=so, argmax=am)
            torch.cuda.synchronize()
            assert torch.equal(all2, all1)
        finally:
            del os.environ["EDISON_DIST_FORCE_PADDED"]
        assert parallel.dist_available()
        with pytest.raises(Exception):
            c.dist_init(parallel.dist_unique_id(), 0, 1)   # a context joins one communicator only
        c.dist_shutdown()
        assert c.dist_info() == (0, 1)
    finally:
        c.close()
