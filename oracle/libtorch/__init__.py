"""Driver of the libtorch C++ harness (TEST / BASELINE INFRASTRUCTURE; BASELINE.json config 1, bench.py `cpu_baseline`).

build()          compiles oracle/libtorch/libtorch_harness.cpp against the installed torch package (oracle/Makefile `libtorch`)
model_path(...)  traces oracle/torch_cpu.py into TorchScript on first use (oracle/_build/restatement_<variant>_n<N>[_<tag>].pt)
run(...)         one harness process: torch::jit::load + forward on one frame pair for a wall-clock budget
Only tests/ and bench.py's cpu_baseline leg import this package.
"""
from __future__ import annotations

import json
import os
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE = os.path.dirname(HERE)
BUILD = os.path.join(ORACLE, "_build")
HARNESS = os.path.join(BUILD, "libtorch_harness")


def build(force: bool = False) -> str:
    src = os.path.join(HERE, "libtorch_harness.cpp")
    if force or not os.path.exists(HARNESS) or os.path.getmtime(HARNESS) < os.path.getmtime(src):
        subprocess.run(["make", "-C", ORACLE, "libtorch"], check=True, stdout=subprocess.DEVNULL)
    return HARNESS


def model_path(state, variant: str, n_mc: int, tag: str = "synth0", out_dir: str = BUILD) -> str:
    """TorchScript file of the restatement with `state`'s weights; `tag` names the weight set in the file name.  Only the
    benchmark's default model is kept under oracle/_build (27 MB, it travels with the snapshot); tests pass a temporary directory"""
    path = os.path.join(out_dir, f"restatement_{variant}_n{n_mc}_{tag}.pt")
    if not os.path.exists(path):
        from . import trace_restatement
        trace_restatement.trace(state, variant, n_mc, path)
    return path


def keep_masks(n_mc: int, p: float, mc_seed: int, pair_seq: int):
    """the four keep masks of include/hnet_rng.h as float32 arrays holding 0 or 1/(1-p), in the module's argument order"""
    from cuahn_vio_amd import mcdrop
    out = []
    for stream, n_el in ((mcdrop.STREAM_MEAN_IN, 5120), (mcdrop.STREAM_MEAN_HID, 256), (mcdrop.STREAM_UNC_IN, 5120), (mcdrop.STREAM_UNC_HID, 256)):
        if p <= 0.0:
            out.append(np.ones((n_mc, n_el), np.float32))
        else:
            out.append(mcdrop.keep_mask(mc_seed, pair_seq, stream, n_mc, n_el, p).astype(np.float32) * np.float32(mcdrop.scale(p)))
    return out


def run(model: str, img1, img2, prior, masks, threads: int = 1, seconds: float = 0.0) -> dict:
    """img1 / img2: [224,320] uint8 (scaled by 1/255 like HomographyNet.cpp:160-163) or float32 in [0,1]; prior: 8 floats or None;
    masks: keep_masks(...).  Returns the harness's JSON plus 'mean' [8], 'cov' [8,8], 'H_part1' [3,3] of the last forward"""
    from ..torch_cpu import _as_f32
    build()
    img1, img2 = _as_f32(img1), _as_f32(img2)
    n_mc = masks[0].shape[0]
    pr = np.zeros(8, np.float32) if prior is None else np.asarray(prior, np.float32).reshape(8)
    with tempfile.TemporaryDirectory() as td:
        fin, fout = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
        with open(fin, "wb") as f:
            for a in (img1, img2, pr, *masks):
                f.write(np.ascontiguousarray(a, np.float32).tobytes())
        r = subprocess.run([HARNESS, model, fin, str(n_mc), str(threads), str(seconds), fout], check=True, capture_output=True, text=True)
        res = json.loads(r.stdout.strip().splitlines()[-1])
        o = np.fromfile(fout, np.float32)
    res["mean"], res["cov"], res["H_part1"] = o[:8].copy(), o[8:72].reshape(8, 8).copy(), o[72:81].reshape(3, 3).copy()
    return res
