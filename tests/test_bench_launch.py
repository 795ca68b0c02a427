"""bench.py's multi-rank launch path, on CPU: `python bench.py --gpus 2` (no launcher, WORLD_SIZE unset) must start two
ranks itself — as child processes, before anything touches a GPU — create the process group and have rank 0 print ONE
JSON line that says n_gpus = 2.  --dry-run stops after the process-group plumbing (gloo; the real run uses RCCL)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*argv, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout          # exactly one JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_gpus_2_starts_two_ranks_itself():
    r = _run("--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run")
    assert r["n_gpus"] == 2 and r["rccl_ranks"] == 2 and r["backend"] == "gloo" and r["dry_run"] is True
    assert r["steps"] == 3 and r["warmup"] == 1
    assert r["max_over_ranks_s"] >= 0.02       # rank 1 sleeps 20 ms, rank 0 10 ms: the MAX over ranks is reported


def test_bench_single_rank_dry_run():
    r = _run("--dry-run")
    assert r["n_gpus"] == 1 and r["rccl_ranks"] == 1


def test_bench_refuses_a_mismatched_world():
    env = {k: v for k, v in os.environ.items()}
    env.update(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry-run"], capture_output=True, text=True,
                       timeout=120, env=env)
    assert p.returncode != 0 and "WORLD_SIZE=2" in (p.stderr + p.stdout)


def test_config_4_preset_at_world_2():
    """BASELINE config 4 (`--config 4` = --mode mc --mc 32 --batch 1): two ranks, the 32 MC-dropout samples split 16 / 16, one gather of the per-sample
    head outputs in global sample order (cuahn_vio_amd.dist.gather_mc_samples, executed here on gloo with CPU tensors)"""
    r = _run("--gpus", "2", "--config", "4", "--dry-run")
    z = r["resolved"]
    assert r["n_gpus"] == 2 and r["gather_checked"] is True
    assert z["mode"] == "mc" and z["mc"] == 32 and z["mc_per_gpu"] == 16 and z["batch_per_gpu"] == 1 and z["gathers"] is True


def test_config_5_preset_at_world_2():
    """BASELINE config 5 (`--config 5` = --pairs-total 256 --mode stream --replay indoor_forward_7 --variant prior3 --mc 16): two ranks, 128 streamed pairs
    each, and the streamed step GATHERS (VERDICT r3 missing #1): three steps through the double-buffered gather of the packed [B, 72] outputs
    (cuahn_vio_amd.dist.OverlappedGather; gloo, CPU tensors), every step's result in rank order"""
    r = _run("--gpus", "2", "--config", "5", "--dry-run")
    z = r["resolved"]
    assert r["n_gpus"] == 2 and r["gather_checked"] is True
    assert z["mode"] == "stream" and z["pairs_total"] == 256 and z["batch_per_gpu"] == 128 and z["replay"] == "indoor_forward_7"
    assert z["variant"] == "prior3" and z["mc"] == 16 and z["gathers"] is True


def test_config_4_refuses_an_indivisible_sample_count():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--config", "4", "--dry-run"], capture_output=True, text=True,
                       timeout=300, env=env)
    assert p.returncode != 0


def test_config_4_preset_at_world_8():
    """BASELINE config 4 at its literal shape: EIGHT ranks, 4 of the 32 MC-dropout samples each, one gather in global sample order"""
    r = _run("--gpus", "8", "--config", "4", "--dry-run")
    z = r["resolved"]
    assert r["n_gpus"] == 8 and r["rccl_ranks"] == 8 and r["gather_checked"] is True
    assert z["mode"] == "mc" and z["mc"] == 32 and z["mc_per_gpu"] == 4 and z["batch_per_gpu"] == 1 and z["gathers"] is True


def test_config_5_preset_at_world_8():
    """BASELINE config 5 at its literal shape: eight ranks, 32 streamed pairs each per step, FOUR steps per all-gather of the packed outputs
    (OverlappedGather(group_steps=4): three slabs = twelve steps through the double-buffered gather, every step's rows of every rank checked); one compute
    context per GPU in the streamed mode (stated in config.parallelism of the real run)"""
    r = _run("--gpus", "8", "--config", "5", "--dry-run")
    z = r["resolved"]
    assert r["n_gpus"] == 8 and r["gather_checked"] is True
    assert z["mode"] == "stream" and z["pairs_total"] == 256 and z["batch_per_gpu"] == 32 and z["gather_group_steps"] == 4 and z["contexts"] == 1
    assert z["variant"] == "prior3" and z["mc"] == 16 and z["gathers"] is True


def test_pairs_mode_at_world_8_keeps_its_contexts_with_grouped_gathers():
    """the resident-input form of config 5's per-GPU shape (--pairs-total 256 over eight ranks = 32 pairs per GPU and step): grouped gathers AND the contexts of the hnet_group"""
    r = _run("--gpus", "8", "--pairs-total", "256", "--variant", "prior3", "--mc", "16", "--dry-run")
    z = r["resolved"]
    assert r["n_gpus"] == 8 and r["gather_checked"] is True and z["batch_per_gpu"] == 32 and z["gather_group_steps"] == 4 and z["contexts"] == 4
