"""GPU: the multi-rank path rehearsed on ONE MI355X — two processes share the GPU, gather over gloo
(RCCL refuses two ranks per device; the 8-GPU RCCL run is the driver's).  Sharded results against the unsharded forward."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_sharded_ranks_match_single_context(tmp_path, world):
    out = tmp_path / "verdict.json"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(29600 + world), os.path.join(ROOT, "tests", "dist_gpu_worker.py"), str(out)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    v = json.loads(out.read_text())
    # pairs sharded: each rank runs a smaller batch than the single context (other split-K partition): not bitwise
    assert v["pairs_mean_maxdiff"] < 6e-5 and v["pairs_cov_reldiff"] < 1e-5   # fp32 summation-order noise, cf. test_batch_size_invariance
    # MC samples sharded: masks are keyed by the global sample index and the ensemble is finished in the reference's
    # two-pass order on the gathered samples.  Bit-identical to one context when the heads GEMM of a shard gets the same
    # split-K partition as the full one (test_mc_sharding_is_rank_invariant); here M differs (6 x 16 vs 6 x 16/world rows),
    # so only the fp32 summation order differs
    assert v["mc_mean_maxdiff"] < 6e-5 and v["mc_cov_reldiff"] < 1e-5


@pytest.mark.gpu
def test_finish_on_the_gathered_layout_is_bitwise_the_finish_on_reordered_samples(blob):
    """Round 5 (config 4 without layout launches): hnet_mc_finish_gathered_device reads sample s of pair b at rank s / n_local of the buffer an
    all-gather of every rank's [2][B][n_local][8] array fills; hnet_mc_finish_packed_device reads [B][N][8] arrays in global sample order.  Same
    two-pass arithmetic -> the packed [B][72] records must agree bit for bit."""
    import torch
    from cuahn_vio_amd.homography_net import HnetEngine
    dev = torch.device("cuda:0")
    world, B, nl = 4, 3, 8
    g = torch.Generator(device="cpu").manual_seed(5)
    gathered = torch.randn(world, 2, B, nl, 8, generator=g)
    gathered[:, 1] *= 0.01                                            # log-variances
    h1 = (torch.eye(3).reshape(1, 9).repeat(B, 1) + 0.001 * torch.randn(B, 9, generator=g)).to(dev)
    ms = gathered[:, 0].permute(1, 0, 2, 3).reshape(B, world * nl, 8).contiguous().to(dev)
    lv = gathered[:, 1].permute(1, 0, 2, 3).reshape(B, world * nl, 8).contiguous().to(dev)
    gd = gathered.contiguous().to(dev)
    o1, o2 = torch.zeros(B, 72, device=dev), torch.zeros(B, 72, device=dev)
    eng = HnetEngine(blob, variant="full", mc_samples=world * nl, dropout_p=0.05, mc_seed=1, max_batch=B)
    eng.mc_finish_packed_device(ms.data_ptr(), lv.data_ptr(), world * nl, h1.data_ptr(), B, o1.data_ptr())
    eng.mc_finish_gathered_device(gd.data_ptr(), world, nl, h1.data_ptr(), B, o2.data_ptr())
    eng.synchronize()
    eng.close()
    assert torch.isfinite(o1).all() and o1.abs().max() > 0
    assert torch.equal(o1, o2)
