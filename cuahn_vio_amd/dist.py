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


def gather_mc_block(both, gathered, group=None):
    """Round 5: config 4 without layout launches.  `both` = this rank's [2, B, n_local, 8] array (hnet_infer_mc_partial_device wrote mean_s into both[0]
    and logvar_s into both[1]); ONE all_gather_into_tensor fills `gathered` [world, 2, B, n_local, 8], which hnet_mc_finish_gathered_device reads in place
    (rank-major = global sample order): no stack, no permute, no .contiguous() between the forward and the finish."""
    _all_gather(gathered, both, group)
    return gathered


class OverlappedGather:
    """The gather of the packed [B, 72] outputs OFF the compute stream: two packed output slabs and two gathered slabs; step i
    writes its outputs into its slab (hnet_infer_batch_packed_device: no packing copies), then `submit(i)` hands a full slab to a side stream, which
    waits for the forward's completion event and runs all_gather_into_tensor there - under the forward of the next step.  `acquire(i)` makes the compute
    stream wait for the gather that last read the slab step i writes into; `result(i)` waits for step i's gather and returns its [world, B, 72] rows.
    group_steps = G > 1: "fewer, larger collectives" - a slab holds the outputs of G consecutive steps and is gathered once, after its last step (the
    message is G x B x 288 bytes; at 32 pairs per GPU a step is 0.35 ms and the Python-side cost of one collective call per step, ~40 us, shows: bench.py
    groups small steps).  With CPU tensors (gloo tests) everything is synchronous: same order of operations, same results.
    The timed region of bench.py still ends with barrier + synchronize, which covers the side stream."""

    def __init__(self, batch, device, group=None, group_steps=1):
        self.group, self.world, self.B, self.G = group, dist.get_world_size(group), batch, max(1, int(group_steps))
        self.cuda = torch.device(device).type == "cuda"
        self.out = [torch.zeros(self.G * batch, 72, device=device) for _ in range(2)]
        self.gathered = [torch.zeros(self.world * self.G * batch, 72, device=device) for _ in range(2)]
        self.pending = [False, False]
        self.submitted_upto = [-1, -1]      # last step whose slab submission covers it (result() checks)
        self.waited = [set(), set()]        # streams that already wait for the slab's pending gather
        if self.cuda:
            # its own priority level: the HIP runtime multiplexes the streams of ONE priority over its hardware queues, and a side stream that lands on the
            # compute stream's queue serialises the gather between two forwards instead of running it under the next one (bench.py, stream mode, saw
            # the same with its copy stream); queues of different priority are never shared.  (The process group's internal RCCL stream: TORCH_NCCL_HIGH_PRIORITY=1)
            self.side = torch.cuda.Stream(device, priority=-1)
            self.ev_done = {}                                            # (slab, stream) -> the forwards of the slab's steps on that stream have finished
            self.ev_gathered = [torch.cuda.Event() for _ in range(2)]    # its gather has finished (recorded on the side stream)

    def _slab(self, i):
        return (i // self.G) % 2

    def buffer(self, i):
        j = i % self.G
        return self.out[self._slab(i)][j * self.B:(j + 1) * self.B]

    def acquire(self, i, compute_stream=None):
        """before the forward of step i writes its part of a slab: the stream that runs it waits for the gather that last read that slab (once per stream and
        slab generation - with several contexts / streams the steps of one slab run on different streams, round 6)"""
        k = self._slab(i)
        if self.cuda and self.pending[k]:
            cs = compute_stream or torch.cuda.current_stream()
            if cs.cuda_stream not in self.waited[k]:
                cs.wait_event(self.ev_gathered[k])
                self.waited[k].add(cs.cuda_stream)

    def submit(self, i, compute_stream=None, flush=False, all_streams=None):
        """after the forward of step i has been enqueued on the compute stream; flush = gather a partly filled slab (end of a run).
        all_streams: every stream that ran a step of this slab (several contexts round-robin, group_steps > 1): the gather waits for all of them"""
        if i % self.G != self.G - 1 and not flush:
            return
        k = self._slab(i)
        self.submitted_upto[k] = i
        if not self.cuda:
            _all_gather(self.gathered[k], self.out[k], self.group)
            self.pending[k] = True
            return
        cs = compute_stream or torch.cuda.current_stream()
        producers = [cs] + [s_ for s_ in (all_streams or []) if s_.cuda_stream != cs.cuda_stream]
        if self.G == 1:
            producers = [cs]                       # one step per slab: one producer
        evs = []
        for s_ in producers:
            ev = self.ev_done.setdefault((k, s_.cuda_stream), torch.cuda.Event())
            ev.record(s_)
            evs.append(ev)
        with torch.cuda.stream(self.side):
            for ev in evs:
                self.side.wait_event(ev)
            _all_gather(self.gathered[k], self.out[k], self.group)
            self.ev_gathered[k].record(self.side)
        self.pending[k] = True
        self.waited[k] = set()

    def flush(self, last_step, compute_stream=None, all_streams=None):
        """gather the partly filled slab that holds step `last_step` (a run whose step count is not a multiple of group_steps: call before the closing
        synchronize of a timed region, so that every step's collective lies inside it)"""
        if last_step % self.G != self.G - 1:
            self.submit(last_step, compute_stream, flush=True, all_streams=all_streams)

    def result(self, i):
        """[world, B, 72]: step i's records of every rank.  The slab of step i must have been submitted (submit of its last step, or flush) AFTER
        step i was enqueued - checked"""
        k, j = self._slab(i), i % self.G
        if self.submitted_upto[k] < i:
            raise RuntimeError(f"OverlappedGather.result({i}): the slab of that step has not been submitted since the step was written")
        if self.submitted_upto[k] // self.G != i // self.G:      # the two slabs alternate: a later group of steps has re-used this one
            raise RuntimeError(f"OverlappedGather.result({i}): the slab of that step has been overwritten by steps {self.submitted_upto[k] // self.G * self.G}..{self.submitted_upto[k]}")
        if self.cuda and self.pending[k]:
            self.ev_gathered[k].synchronize()
        return self.gathered[k].view(self.world, self.G, self.B, 72)[:, j]
