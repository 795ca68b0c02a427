"""Multi-GPU sharding of the HomographyNet path over torch.distributed
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The forward has no exchange step: frame pairs (or MC-dropout samples) are independent units.  Ranks take
contiguous shards; the only collective is one all-gather of the per-pair outputs (72 floats = 288 B per pair)
or of the per-sample head outputs (16 floats per sample).  Messages are KB-sized, i.e. latency-bound: one
all_gather_into_tensor per batch, never one per pair.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(n_items: int, world: int, rank: int):
    """contiguous split of n_items over `world` ranks; the first (n_items % world) ranks get one more"""
    base, extra = divmod(n_items, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def pack_outputs(mean: torch.Tensor, cov: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """[B,8] + [B,64] -> [B,72] (one message per rank)"""
    out[:, :8] = mean
    out[:, 8:] = cov.reshape(-1, 64)
    return out


def _all_gather(dst: torch.Tensor, src: torch.Tensor, group=None):
    world = dist.get_world_size(group)
    if dist.get_backend(group) == "gloo":
        if src.is_cuda:      # single-GPU rehearsal of the multi-rank path (bench.py HNET_BENCH_SHARED_GPU=1): stage through the host
            host = [torch.empty(src.shape, dtype=src.dtype) for _ in range(world)]
            dist.all_gather(host, src.detach().cpu().contiguous(), group=group)
            dst.reshape(world, *src.shape).copy_(torch.stack(host, 0))
            return
        chunks = list(dst.reshape(world, *src.shape).unbind(0))
        dist.all_gather(chunks, src.contiguous(), group=group)
    else:
        dist.all_gather_into_tensor(dst, src.contiguous(), group=group)


def gather_outputs(mean, cov, out, gathered, group=None):
    """all ranks end with gathered [world*B, 72] in rank order (= pair order for contiguous shards)"""
    pack_outputs(mean, cov, out)
    _all_gather(gathered, out, group)
    return gathered


def gather_mc_samples(mean_s, logvar_s, h_part1, group=None):
    """MC-dropout sharding: each rank holds per-sample head outputs [B, n_local, 8] for its contiguous sample
    range (equal n_local on every rank).  Returns ([B, N, 8], [B, N, 8]) with samples in global order; the
    ensemble is then finished in the reference's two-pass order (model_to_trace.py:274-280) by
    hnet_mc_finish_device — NOT by all-reducing sums, which would round differently."""
    world = dist.get_world_size(group)
    b, n_local, _ = mean_s.shape
    both = torch.stack([mean_s, logvar_s], 0).contiguous()               # [2, B, n_local, 8]
    g = torch.empty((world,) + tuple(both.shape), dtype=both.dtype, device=both.device)
    _all_gather(g, both, group)
    g = g.permute(1, 2, 0, 3, 4).reshape(2, b, world * n_local, 8)        # rank-major = global sample order
    return g[0].contiguous(), g[1].contiguous(), h_part1
