"""The collective path on a real RCCL communicator (VERDICT r2 "execute RCCL once").  A gpurun box has ONE MI355X, which can
host a 1-rank nccl process group: init_process_group("nccl"), all_gather_into_tensor, barrier and all_reduce all execute in librccl.  The
ranks are child processes started before anything in them touches the GPU (never an exec from a GPU process)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _env():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.pop("HNET_BENCH_SHARED_GPU", None)
    return env


def test_gathers_run_through_rccl_on_a_one_rank_communicator(tmp_path):
    out = tmp_path / "rccl.json"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "rccl_worker.py"), str(out)]
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.loads(out.read_text())
    print("RCCL:", res)
    assert res["backend"] == "nccl" and res["world"] == 1
    assert res["own_shard_bitwise"] and res["all_shards_bitwise"] and res["mc_sharded_equals_unsharded_bitwise"]
    assert res["all_reduce_max"] == 1.0
    log = os.environ.get("HNET_RCCL_LOG")              # tracked evidence (profiles/rNN_rccl_one_rank.log)
    if log:
        with open(log, "a") as f:
            f.write("tests/rccl_worker.py: " + json.dumps(res) + "\n")


@pytest.mark.parametrize("extra", [[], ["--pairs-total", "64", "--variant", "prior3", "--mc", "16"], ["--config", "5", "--steps", "8"], ["--config", "4"]])
def test_bench_runs_its_collective_through_rccl_with_one_rank(extra):
    """bench.py --gpus 1 --force-collective: a 1-rank RCCL communicator, the per-step all-gather of the packed [B, 72] outputs on a SIDE stream under the
    next step's forward (round 4: cuahn_vio_amd.dist.OverlappedGather; bench.py itself asserts gathered == local bit for bit on the last step), the
    barrier and the max-over-ranks all-reduce; the oracle check of the last step still gates the run.  Also BASELINE's presets: --config 5 (streamed
    UZH-FPV replay, 256 pairs, the streamed step gathers) and --config 4 (MC samples over the ranks)."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-collective", "--steps", "5", "--warmup", "2",
           "--no-cpu-baseline", "--no-latency", "--no-extras"] + extra
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert res["backend"] == "nccl (RCCL)" and res["rccl_ranks"] == 1 and res["n_gpus"] == 1
    assert res["verify"]["passed"] and res["value"] > 0
    if extra and extra[0] == "--pairs-total":
        assert res["config"]["batch_per_gpu"] == 64 and res["scaling"] == "strong"
    if extra and extra[:2] == ["--config", "5"]:
        assert res["config"]["batch_per_gpu"] == 256 and "STREAMED" in res["config"]["workload"]
    log = os.environ.get("HNET_RCCL_LOG")
    if log:
        with open(log, "a") as f:
            f.write("bench.py --force-collective " + " ".join(extra) + ": " + json.dumps({k: res[k] for k in ("value", "ms_per_step", "backend", "rccl_ranks", "max_px_err")}) + "\n")


def test_cpp_rccl_gather_example_runs(blob, tmp_path):
    """the C++ side of the split (north_star: host code stays C++): tests/cpp/rccl_gather_example.cpp shards the pairs over the visible
    GPUs (one here), runs hnet_infer_batch_device per shard and all-gathers the [B, 72] outputs with ncclAllGather"""
    import test_rccl_cpp_build as tb
    exe = tb.build()
    w = tmp_path / "w.hnw"
    w.write_bytes(blob)
    r = subprocess.run([exe, str(w), "8"], env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "RCCL_GATHER_OK" in r.stdout
    print(r.stdout[-400:])
