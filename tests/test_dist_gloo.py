"""CPU, world_size 2, gloo: the sharding / gather logic used by bench.py at N>1 (cuahn_vio_amd/dist.py).
The per-rank compute is played by the oracle (test infrastructure) so the collective path, the shard
arithmetic and the rank-invariance of the MC-dropout masks are exercised without a GPU."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, tmp):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cuahn_vio_amd import dist as hd
    from cuahn_vio_amd import synth, weights
    from oracle import pyoracle

    blob = weights.pack_state_dict(weights.synthetic_state(0))
    orc = pyoracle.Oracle(blob, threads=2)
    seed, p = 77, 0.05

    # --- (1) pairs sharded over ranks, one all-gather of [B,72]
    n_pairs = 4
    b0, b1 = hd.shard_range(n_pairs, world, rank)
    prev, curr, prior, _ = synth.make_batch(200, n_pairs)
    mean = torch.zeros(b1 - b0, 8)
    cov = torch.zeros(b1 - b0, 64)
    for i, b in enumerate(range(b0, b1)):
        o = orc.forward(prev[b], curr[b], prior[b], 3, 8, p, seed, b)
        mean[i] = torch.from_numpy(o["mean"])
        cov[i] = torch.from_numpy(o["cov"].reshape(64))
    out = torch.zeros(b1 - b0, 72)
    gathered = torch.zeros(n_pairs, 72)
    hd.gather_outputs(mean, cov, out, gathered)

    # --- (2) MC-dropout samples of ONE pair sharded over ranks, gather of per-sample outputs, two-pass finish
    n_mc = 8
    s0, s1 = hd.shard_range(n_mc, world, rank)
    tr = orc.forward(prev[0], curr[0], None, 3, n_mc, p, seed, 5, want_trace=True)     # trunk (replicated on every rank)
    ms, lv = orc.heads(tr["feat"], s0, s1, p, seed, 5)
    g_ms, g_lv, _ = hd.gather_mc_samples(torch.from_numpy(ms)[None], torch.from_numpy(lv)[None], None)
    m2, c2, _ = orc.finish(g_ms[0].numpy(), g_lv[0].numpy(), tr["H_part1"])
    np.savez(os.path.join(tmp, f"rank{rank}.npz"), gathered=gathered.numpy(), m2=m2, c2=c2, full_mean=tr["mean"], full_cov=tr["cov"])
    dist.barrier()
    dist.destroy_process_group()


def test_world2_gloo_gather_and_mc_sharding(tmp_path, blob):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    from cuahn_vio_amd import synth
    from oracle import pyoracle
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    # every rank holds the same gathered matrix, rows in pair order, equal to a single-process run
    assert np.array_equal(r0["gathered"], r1["gathered"])
    orc = pyoracle.Oracle(blob, threads=2)
    prev, curr, prior, _ = synth.make_batch(200, 4)
    for b in range(4):
        o = orc.forward(prev[b], curr[b], prior[b], 3, 8, 0.05, 77, b)
        assert np.array_equal(r0["gathered"][b, :8], o["mean"])
        assert np.array_equal(r0["gathered"][b, 8:], o["cov"].reshape(64))
    # MC sharding: gathered samples + two-pass finish == the unsharded forward, bitwise, on both ranks
    for r in (r0, r1):
        assert np.array_equal(r["m2"], r["full_mean"]) and np.array_equal(r["c2"], r["full_cov"])


def test_shard_range_covers_everything():
    from cuahn_vio_amd.dist import shard_range
    for n in (1, 7, 32, 256):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(e - b for b, e in spans) - min(e - b for b, e in spans) <= 1
