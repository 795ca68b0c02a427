"""HNET_PREC_F16X2 (the default arithmetic: two fp16 planes per activation, three fp16 MFMAs per product, csrc/s3_format.h) has a
range: |weight| < 16, |activation| < 32768 (guaranteed; fp16 itself ends at 65504).  Outside it the context must give the answer of HNET_PREC_BF16X3 — never a silently
wrong or a non-finite one — and say so through hnet_precision."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

PREC_BF16X3, PREC_F16X2 = 2, 3


def _pair(seed):
    from cuahn_vio_amd import synth
    i1, i2, _ = synth.make_pair(seed)
    return i1, i2


def test_default_is_f16x2_and_stays_there_on_ordinary_input(blob, monkeypatch):
    from cuahn_vio_amd.homography_net import HnetEngine
    monkeypatch.delenv("HNET_PRECISION", raising=False)       # the library default, not the environment's choice
    i1, i2 = _pair(3)
    e = HnetEngine(blob, variant="full", mc_samples=4, dropout_p=0.05, mc_seed=1, max_batch=2)
    assert e.precision() == PREC_F16X2
    mean, cov = e.infer_batch(np.stack([i1, i2]), np.stack([i2, i1]), None)
    assert np.isfinite(mean).all() and np.isfinite(cov).all()
    assert e.precision() == PREC_F16X2
    e.close()


def test_activation_overflow_demotes_to_bf16x3_and_returns_its_result(blob):
    """float images scaled far beyond [0, 1]: block_1_1's outputs exceed 65504, the fp16 planes overflow, the outputs turn non-finite;
    hnet_infer_batch re-packs the weights for split-bf16, repeats the call and returns exactly what a BF16X3 context returns"""
    from cuahn_vio_amd.homography_net import HnetEngine
    i1, i2 = _pair(5)
    big1 = (i1.astype(np.float32) / 255.0 * 3.0e6)[None]
    big2 = (i2.astype(np.float32) / 255.0 * 3.0e6)[None]
    ref = HnetEngine(blob, variant="full", mc_samples=4, dropout_p=0.05, mc_seed=1, max_batch=1, precision=PREC_BF16X3)
    m_ref, c_ref = ref.infer_batch(big1, big2, None)
    ref.close()
    assert np.isfinite(m_ref).all(), "the test input must be finite in fp32-range arithmetic"
    e = HnetEngine(blob, variant="full", mc_samples=4, dropout_p=0.05, mc_seed=1, max_batch=1, precision=PREC_F16X2)
    assert e.precision() == PREC_F16X2
    m, c = e.infer_batch(big1, big2, None)
    assert e.precision() == PREC_BF16X3
    assert np.array_equal(m, m_ref) and np.array_equal(c, c_ref)
    # and it stays usable: an ordinary pair afterwards equals the BF16X3 context's answer too
    m2, _ = e.infer_batch(i1[None], i2[None], None)
    ref = HnetEngine(blob, variant="full", mc_samples=4, dropout_p=0.05, mc_seed=1, max_batch=1, precision=PREC_BF16X3)
    m2_ref, _ = ref.infer_batch(i1[None], i2[None], None)
    ref.close()
    e.close()
    assert np.array_equal(m2, m2_ref)


def test_weights_beyond_the_fp16_plane_range_select_bf16x3_at_create(state):
    from cuahn_vio_amd import weights
    from cuahn_vio_amd.homography_net import HnetEngine
    st = {k: v.copy() for k, v in state.items()}
    k = "model_part1.block_2_2.0.weight"
    st[k].reshape(-1)[7] = 20.0                     # one weight >= 16
    e = HnetEngine(weights.pack_state_dict(st), variant="full", mc_samples=1, dropout_p=0.0, max_batch=1, precision=PREC_F16X2)
    assert e.precision() == PREC_BF16X3
    i1, i2 = _pair(2)
    mean, _ = e.infer_batch(i1[None], i2[None], None)
    assert np.isfinite(mean).all()
    e.close()


@pytest.mark.parametrize("seed,conv_gain", [(1, 1.0), (2, 4.0), (3, 0.5)])
def test_other_weight_sets_stay_at_fp32_level(seed, conv_gain):
    """the golden cases all use one weight set; here other seeds, and convolution weights scaled up (activations of O(10^3)) and down
    (features of O(0.1), first-plane residuals in the fp16 subnormal range unless scaled) - both matrix-core fp32-grade modes
    against the oracle's double accumulation, and against each other"""
    from conftest import TOL_PX_VS_ORACLE
    from cuahn_vio_amd import weights
    from cuahn_vio_amd.homography_net import HnetEngine
    from oracle import pyoracle
    st = weights.variant_state(seed, conv_gain)
    blob = weights.pack_state_dict(st)
    i1, i2 = _pair(10 + seed)
    ref = pyoracle.Oracle(blob).forward(i1, i2, n_mc=4, p=0.05, mc_seed=3, pair_seq=0)
    got = {}
    for prec in (PREC_F16X2, PREC_BF16X3):
        e = HnetEngine(blob, variant="full", mc_samples=4, dropout_p=0.05, mc_seed=3, max_batch=1, precision=prec)
        got[prec], _ = e.infer_batch(i1[None], i2[None], None)
        assert e.precision() == prec
        e.close()
        d = float(np.abs(got[prec][0] - ref["mean"]).max())
        print(f"seed {seed} gain {conv_gain} precision {prec}: |hip - oracle| = {d:.2e} px, offsets up to {np.abs(ref['mean']).max():.1f} px")
        assert d < TOL_PX_VS_ORACLE
    assert float(np.abs(got[PREC_F16X2] - got[PREC_BF16X3]).max()) < 1e-4


def test_activations_up_to_32768_are_carried_exactly_and_beyond_that_are_detected(blob, state):
    """the proven range of the two-plane format (tests/cpp/s3_format_check.cpp: exhaustive on the host): a conv fed with |a| ~ 30 000 gives the
    oracle's answer at the fp32 level; fed with |a| ~ 40 000 - inside fp16's 65504 but beyond the guaranteed 32768 - every output is EITHER
    at the fp32 level OR non-finite (an infinite second plane of a near-tie value), never finite and wrong"""
    from cuahn_vio_amd.homography_net import HnetEngine
    from oracle import pyoracle
    layer = 4                                                  # block_2_2: 64 -> 128, 5x5 stride 2 (implicit GEMM on the matrix cores)
    rng = np.random.default_rng(5)
    e = HnetEngine(blob, variant="full", mc_samples=1, dropout_p=0.0, max_batch=1, precision=PREC_F16X2)
    for amp, must_be_finite in ((30000.0, True), (40000.0, False)):
        # ordinary activations with 2 % of the elements at +-(0.6 .. 1.0) x amp: the INPUT planes are what is probed, the outputs (sums over
        # 1600 products, ~32 of them large) stay far inside the range
        x = rng.standard_normal((1, 64, 28, 40)).astype(np.float32)
        big = rng.random(x.shape) < 0.02
        x[big] = (rng.uniform(0.6, 1.0, int(big.sum())) * amp * rng.choice([-1.0, 1.0], int(big.sum()))).astype(np.float32)
        got = e.op_conv(layer, x)
        ref = pyoracle.conv_lrelu(x[0], state["model_part1.block_2_2.0.weight"], state["model_part1.block_2_2.0.bias"], 2)[None]
        fin = np.isfinite(got)
        if must_be_finite:
            assert fin.all()
        scale = np.abs(ref).max()
        assert np.abs(got[fin] - ref[fin]).max() / scale < 2e-5, amp
        print(f"|a| ~ {amp:.0f}: {fin.mean() * 100:.2f} % of the outputs finite, max rel err of those {np.abs(got[fin] - ref[fin]).max() / scale:.1e}")
    e.close()


def test_device_entry_points_raise_the_overflow_flag(blob):
    """the device-resident entry points cannot demote; they OR a device word that hnet_overflow_flag returns and clears"""
    import torch
    from cuahn_vio_amd.homography_net import PIX_F32, PIX_U8, HnetEngine
    i1, i2 = _pair(6)
    dev = torch.device("cuda", 0)
    mean, cov = torch.zeros(2, 8, device=dev), torch.zeros(2, 64, device=dev)
    e = HnetEngine(blob, variant="full", mc_samples=4, dropout_p=0.05, mc_seed=1, max_batch=2, precision=PREC_F16X2)
    assert e.overflow_flag() == 0                              # the warm-up forward of hnet_create left it clear
    p_ok, c_ok = torch.from_numpy(np.stack([i1, i2])).to(dev), torch.from_numpy(np.stack([i2, i1])).to(dev)
    e.infer_batch_device(p_ok.data_ptr(), c_ok.data_ptr(), PIX_U8, None, 2, 0, mean.data_ptr(), cov.data_ptr())
    assert e.overflow_flag() == 0 and torch.isfinite(mean).all()
    big = lambda a: torch.from_numpy((a.astype(np.float32) / 255.0 * 3.0e6)[None]).to(dev)
    pb, cb = big(i1), big(i2)
    e.infer_batch_device(pb.data_ptr(), cb.data_ptr(), PIX_F32, None, 1, 0, mean.data_ptr(), cov.data_ptr())
    assert e.overflow_flag() == 1                              # raised ...
    assert e.overflow_flag() == 0                              # ... and cleared by the poll
    assert e.precision() == PREC_F16X2                         # the device path never switches the mode by itself
    # the sharded path ends in heads_fc2: same word
    ms, lv, h1 = torch.zeros(1, 4, 8, device=dev), torch.zeros(1, 4, 8, device=dev), torch.zeros(1, 9, device=dev)
    e.infer_mc_partial_device(pb.data_ptr(), cb.data_ptr(), PIX_F32, None, 1, 0, ms.data_ptr(), lv.data_ptr(), h1.data_ptr())
    assert e.overflow_flag() == 1
    e.close()


def test_a_nan_prior_is_not_an_overflow(blob):
    """non-finite INPUTS (a diverged filter's NaN prior) give NaN outputs in any arithmetic; the context must keep its mode"""
    from cuahn_vio_amd.homography_net import HnetEngine
    i1, i2 = _pair(7)
    e = HnetEngine(blob, variant="prior3", mc_samples=4, dropout_p=0.05, mc_seed=1, max_batch=1, precision=PREC_F16X2)
    prior = np.full((1, 8), np.nan, np.float32)
    mean, _cov = e.infer_batch(i1[None], i2[None], prior)
    assert not np.isfinite(mean).all()
    assert e.precision() == PREC_F16X2
    m2, _ = e.infer_batch(i1[None], i2[None], np.zeros((1, 8), np.float32))
    assert np.isfinite(m2).all() and e.precision() == PREC_F16X2
    e.close()
