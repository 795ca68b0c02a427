import glob
import os
import sys

import numpy as np
import pytest

# torch bundles its own libamdhip64; when libhnet_hip.so (linked against /opt/rocm's) is loaded into the process first,
# a later `import torch` finds "No HIP GPUs".  Tests that hand torch tensors to the C ABI therefore load torch first.
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")

# Tolerances.  The reference's own fp32 run differs from its fp64 evaluation by up to
# 1.4e-4 px on these weights (torch.inverse inside DLT_solve + fp32 activations), so 1e-4 px against the fp32
# golden is below the reference's own noise; the gates are:
TOL_PX_VS_REF32 = 3e-4      # |offset - reference fp32 golden|, px
TOL_PX_VS_REF64 = 1e-4      # |offset - reference fp64 golden|, px  (north_star's 1e-4 px, against the reference evaluated in double)
TOL_PX_VS_ORACLE = 1e-4     # |HIP - oracle (double accumulation)|, px, DEFAULT arithmetic (two fp16 planes, fast sampler): north_star's figure.
                            # Measured: <= 3.8e-5 on the 28 goldens, worst slot of a 256-pair batch 7.5e-5 (profiles/r03_v10_full_batch_check.log)


def tol_px_vs_oracle(precision, worst_slot=False):
    """gate of |HIP - oracle| per arithmetic mode.  north_star's 1e-4 px in every gated mode on the golden, smoke and replay cases (they measure
    <= 4e-5: ADVICE r4 - a 50 % regression of a reference mode must not pass).  worst_slot = True is for the checks that look at EVERY slot of a large
    batch (tests/test_gpu_bench_shapes.py, tools/full_batch_check.py): the oracle accumulates in double, and the worst of 256 slots of the two
    reference modes - exact fp32 MFMA (0) and split-bf16 (2) - sits ON 1e-4 (1.0e-4 measured: the network's own fp32 accumulation noise,
    r03_v10_full_batch_check.log), so those two are gated at 1.5e-4 there; the default mode (3) stays at 1e-4 (7.5e-5 measured).  Plain bf16 (1) is
    reported, never gated."""
    if worst_slot:
        return {3: TOL_PX_VS_ORACLE, 2: 1.5e-4, 0: 1.5e-4}[int(precision)]
    return {3: TOL_PX_VS_ORACLE, 2: TOL_PX_VS_ORACLE, 0: TOL_PX_VS_ORACLE}[int(precision)]
TOL_COV_REL = 2e-5          # max |cov - ref| / max |ref|


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_cases():
    out = []
    for fn in sorted(glob.glob(os.path.join(GOLDEN_DIR, "*.npz"))):
        name = os.path.basename(fn)[:-4]
        if name in ("dlt", "warp_s11") or name.startswith("replay_"):     # operator vectors / trajectory fixtures, not forward cases
            continue
        out.append(name)
    return out


def load_case(name):
    """returns (golden dict, img1, img2, prior, blocks_to_run) with inputs regenerated from seeds"""
    from cuahn_vio_amd import synth
    g = dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))
    kind = str(g["kind"])
    if kind == "pair":
        i1, i2, _ = synth.make_pair(int(g["seed"]), float(g["max_offset"]) if "max_offset" in g else 12.0)
    elif kind == "replay":       # pair (seed, seed + 1) of the committed UZH-FPV trajectory fixture
        from cuahn_vio_amd import replay
        pv, cu, _pr = replay.render_pairs(replay.load_fixture("indoor_forward_7"), int(g["seed"]), 1)
        i1, i2 = pv[0], cu[0]
    elif kind == "noise":
        i1, i2 = synth.make_noise_pair(int(g["seed"]))
    else:
        i1 = np.full((224, 320), 0.2, np.float32)
        i2 = np.full((224, 320), 0.5, np.float32)
    assert synth.crc(i1, i2) == int(g["in_crc"]), "synthetic inputs are not reproduced bit-exactly on this machine"
    prior = g["prior"] if "prior" in g else None
    btr = {"full": 3, "prior3": 3, "prior2": 2, "prior1": 1}[str(g["variant"])]
    return g, i1, i2, prior, btr


_BLOBS = {}


def case_weights(g):
    """(state, blob) of the weight set a golden case was generated on (`weights_seed`, `conv_gain`; tools/gen_golden.py)"""
    from cuahn_vio_amd import weights
    key = (int(g["weights_seed"]) if "weights_seed" in g else 0, float(g["conv_gain"]) if "conv_gain" in g else 1.0)
    if key not in _BLOBS:
        st = weights.variant_state(*key)
        _BLOBS[key] = (st, weights.pack_state_dict(st))
    return _BLOBS[key]


_ORACLES = {}


def case_oracle(g, f32=False):
    from oracle import pyoracle
    key = (int(g["weights_seed"]) if "weights_seed" in g else 0, float(g["conv_gain"]) if "conv_gain" in g else 1.0, f32)
    if key not in _ORACLES:
        _ORACLES[key] = pyoracle.Oracle(case_weights(g)[1], f32=f32)
    return _ORACLES[key]


@pytest.fixture(scope="session")
def blob():
    from cuahn_vio_amd import weights
    return weights.pack_state_dict(weights.synthetic_state(0))


@pytest.fixture(scope="session")
def state():
    from cuahn_vio_amd import weights
    return weights.synthetic_state(0)


@pytest.fixture(scope="session")
def oracle(blob):
    from oracle import pyoracle
    return pyoracle.Oracle(blob)


@pytest.fixture(scope="session")
def oracle_f32(blob):
    from oracle import pyoracle
    return pyoracle.Oracle(blob, f32=True)
