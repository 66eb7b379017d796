"""Process-group plumbing: one process per GPU, torch.distributed (backend "nccl" == RCCL on ROCm; "gloo" in CPU tests).

What shards in StreamingT2V:
  * independent videos                      -> replicas, no data-path collective (bench.py --gpus N, "weak" scaling);
  * the two classifier-free-guidance halves of one chunk -> `CfgPairExchange`: ranks 2k / 2k+1 each run ONE half of the
    CFG batch through StreamingWrapper and exchange the raw network outputs with one all-gather per Euler step
    (25 x 72 x 128 x 4 fp32 = 3.7 MB); exact, because nothing in the forward couples the two halves
    (per-sample GroupNorm statistics, per-sample attention, ControlNet batch splits 7 + 7);
  * the enhancement stage's blending windows -> streamingt2v_amd/blending.py.
The AR chunk loop itself is sequential (chunk k+1 needs decoded frames of chunk k).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None, device=None):
    """Initialise the default group from RANK / WORLD_SIZE / MASTER_* (torch.distributed.run sets them)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 or dist.is_initialized():
        return world
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
    kw = {}
    if backend == "nccl" and device is not None:
        kw["device_id"] = torch.device(device)
    dist.init_process_group(backend, **kw)
    return world


def max_over_ranks(seconds, device="cpu"):
    """Job time = slowest rank (bench contract)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t.item()


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def shard_items(n_items, rank, world):
    """Indices of the independent work items (videos) a rank owns: round-robin."""
    return list(range(rank, n_items, world))


class CfgPairExchange:
    """All-gather of the two CFG halves' network outputs inside a pair of ranks.

    rank parity selects the half (even: unconditional, odd: conditional).  `gather(net_half)` returns the concatenated
    [uncond | cond] tensor on both ranks, i.e. exactly what a single process computes for the CFG batch of 2."""

    def __init__(self, group=None):
        self.group = group
        self.rank = dist.get_rank(group)
        assert dist.get_world_size(group) == 2, "a CFG pair has exactly two ranks"

    @property
    def half(self):
        return self.rank            # 0: uncond half, 1: cond half

    def gather(self, net_half):
        parts = [torch.empty_like(net_half), torch.empty_like(net_half)]
        dist.all_gather(parts, net_half.contiguous(), group=self.group)
        return torch.cat(parts, 0)
