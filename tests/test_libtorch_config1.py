"""BASELINE.json config 1: "single 320x224 grayscale frame pair, 1-block net, libtorch CPU TorchScript path (plumbing, no GPU)".

The reference's CPU path is torch::jit::load + module.forward in C++ (HomographyNet.cpp:89,183-186).  oracle/libtorch/ holds the
same two calls as a stand-alone C++ program on a TorchScript trace of our restatement; here its outputs are checked against the
vectors the REFERENCE model produced (tests/golden/*.npz, tools/gen_golden.py) for the 1-block variant and for the full model with
the MC-dropout masks of include/hnet_rng.h passed in as inputs.  CPU only; builds the harness on first use (~25 s)."""
import numpy as np
import pytest

from conftest import TOL_COV_REL, TOL_PX_VS_REF32, load_case


@pytest.fixture(scope="module")
def state():
    from cuahn_vio_amd import weights
    return weights.synthetic_state(0)


@pytest.mark.parametrize("name", ["prior1_p0_s6", "full_mask16_s8"])
def test_cpp_torchscript_forward_matches_reference_golden(state, name, tmp_path):
    from oracle import libtorch as lt
    g, i1, i2, prior, _btr = load_case(name)
    variant, n_mc, p = str(g["variant"]), int(g["n_mc"]), float(g["p"])
    model = lt.model_path(state, variant, n_mc, out_dir=str(tmp_path))
    masks = lt.keep_masks(n_mc, p, int(g["mc_seed"]) if "mc_seed" in g else 0, int(g["pair_seq"]) if "pair_seq" in g else 0)
    r = lt.run(model, i1, i2, prior, masks, threads=4, seconds=0.0)
    assert r["forwards"] >= 2 and r["ms_per_forward"] > 0
    assert np.abs(r["mean"] - g["mean"]).max() < TOL_PX_VS_REF32
    assert np.abs(r["cov"] - g["cov"]).max() / np.abs(g["cov"]).max() < TOL_COV_REL
    assert np.abs(r["H_part1"] - g["H_part1"]).max() < 1e-4 * max(1.0, float(np.abs(g["H_part1"]).max()))
    # and the Python functional form of the same restatement (what the trace was taken from): same operators, same order
    from oracle.torch_cpu import TorchCpuNet
    o = TorchCpuNet(state).forward(i1, i2, prior, {"full": 3, "prior3": 3, "prior2": 2, "prior1": 1}[variant], n_mc=n_mc, p=p,
                                   mc_seed=int(g["mc_seed"]) if "mc_seed" in g else 0, pair_seq=int(g["pair_seq"]) if "pair_seq" in g else 0)
    assert np.abs(r["mean"] - o["mean"]).max() < 2e-5
