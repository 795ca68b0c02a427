"""The latency path (batch <= 8): the block-tail launch (FC 5120 -> 8 + DLT + composition) and the prior's DLT are recomputed inside the next
block's prep kernel, heads_fc2 + mc_finish run as one launch (csrc/hnet_capi.hip forward_chunk, kernels.h FcArgs); round 5: split-K layers are
finished without reduce launches (last-arriver tiles, reduce-on-load chains: kernels.h LatIO) - all of this is the same arithmetic in the same
order, so the homography of blocks 1 - 3 must be BITWISE that of the multi-launch path (HNET_FUSE_SMALL=0) in every arithmetic mode and variant.
The heads' first FC is a different kernel on the latency path of the default mode (csrc/heads_lat.h: one launch, K summed in another order): there
the outputs agree to fp32 rounding - gated at 5e-5 px / 1e-5 relative in the covariance, the other modes stay bitwise.
(Measured over the cases below: 0 ... 3.0e-5 px between the two orders, each of them 2.2e-5 ... 3.8e-5 px from the double-accumulating oracle: the
mean head's last FC multiplies the hidden units by O(1) weights and cancels to O(5 px), so a 1e-7 relative change of the hidden units is a few 1e-5 px.)"""
TOL_PX_PATHS = 5e-5       # |offset(latency path) - offset(multi-launch path)|, px: different summation order in the heads' first FC only
TOL_COV_PATHS = 1e-5


def _same_outputs(lat, ref, precision):
    """(mean, cov, err) of the latency path against the reference path: bitwise except in the default mode, whose heads FC1 sums K in another order"""
    (m1, c1, e1), (m0, c0, e0) = lat, ref
    if precision != 3:
        return np.array_equal(m1, m0) and np.array_equal(c1, c0) and np.array_equal(e1, e0)
    return (np.abs(m1 - m0).max() < TOL_PX_PATHS and np.abs(c1 - c0).max() / np.abs(c0).max() < TOL_COV_PATHS
            and np.abs(e1.astype(np.float64) - e0.astype(np.float64)).max() < 0.05)
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _engine(blob, fuse, **kw):
    from cuahn_vio_amd.homography_net import HnetEngine
    old = os.environ.get("HNET_FUSE_SMALL")
    os.environ["HNET_FUSE_SMALL"] = "1" if fuse else "0"
    try:
        return HnetEngine(blob, **kw)
    finally:
        if old is None:
            os.environ.pop("HNET_FUSE_SMALL", None)
        else:
            os.environ["HNET_FUSE_SMALL"] = old


@pytest.mark.parametrize("precision", [pytest.param(3, id="f16x2"), pytest.param(2, id="bf16x3"), pytest.param(0, id="fp32")])
@pytest.mark.parametrize("variant,n_mc,batch", [("full", 32, 1), ("prior3", 16, 1), ("prior2", 16, 3), ("prior1", 64, 2), ("full", 16, 8), ("prior3", 5, 7)])
def test_latency_path_is_bitwise_the_multi_launch_path(blob, oracle, variant, n_mc, batch, precision):
    from conftest import TOL_PX_VS_ORACLE
    from cuahn_vio_amd import synth
    prev, curr, prior, _ = synth.make_batch(500 + batch, batch)
    pr = None if variant == "full" else prior
    kw = dict(variant=variant, mc_samples=n_mc, dropout_p=0.05, mc_seed=11, max_batch=8, emit_error_map=True, precision=precision)
    outs = []
    for fuse in (True, False):
        e = _engine(blob, fuse, **kw)
        mean, cov, err = e.infer_batch(prev, curr, pr, pair_seq0=77, want_err=True)
        h1 = np.stack([e.debug_h_part1(b) for b in range(batch)])
        names = [n for n, _f in e.stages()]
        e.close()
        outs.append((mean, cov, err, h1, names))
    (m1, c1, e1, h1, n1), (m0, c0, e0, h0, n0) = outs
    assert np.array_equal(h1, h0) and _same_outputs((m1, c1, e1), (m0, c0, e0), precision)
    # and it is the right answer
    btr = {"full": 3, "prior3": 3, "prior2": 2, "prior1": 1}[variant]
    o = oracle.forward(prev[0], curr[0], None if pr is None else pr[0], btr, n_mc, 0.05, 11, 77)
    assert np.abs(m1[0] - o["mean"]).max() < TOL_PX_VS_ORACLE
    # fewer launches: no block-tail / prior-DLT stage of its own, one heads tail
    # (exact-fp32 mode: block 4's prep does not write the padded planes the fused form needs, so its block-tail launch stays)
    own = [n for n in n1 if n.startswith("fc_dlt_b") or n == "prior_dlt"]
    assert (own == [] or (precision == 0 and own in (["fc_dlt_b3"], ["prior_dlt"]))) and "heads_fc2+mc_finish" in n1
    assert len(n1) < len(n0)


def test_streaming_class_uses_the_latency_path_and_matches(blob):
    """hnet_infer (graph replay of the batch-1 forward) through the class surface: fused and unfused contexts give the same bits"""
    from cuahn_vio_amd import synth
    from cuahn_vio_amd.homography_net import HomographyNet
    import contextlib
    import io
    frames = [synth.make_pair(900 + i)[0] for i in range(4)]
    res = []
    for fuse in ("1", "0"):
        old = os.environ.get("HNET_FUSE_SMALL")
        os.environ["HNET_FUSE_SMALL"] = fuse
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                net = HomographyNet("x_showError.hnw", use_prior=True, blocks_to_run=3, mc_samples=16, dropout_p=0.05, mc_seed=3, weights_blob=blob)
                got = []
                for i, f in enumerate(frames):
                    net.load_current_img(f, float(i))
                    net.network_inference(np.array([1.0, -2.0, 0.5, 3.0, -1.5, 0.25, 2.0, -0.75]), 0)
                    if i:
                        got.append((net._pred_mean.copy(), net._pred_Cov.copy(), net.last_error_map.copy()))
        finally:
            if old is None:
                os.environ.pop("HNET_FUSE_SMALL", None)
            else:
                os.environ["HNET_FUSE_SMALL"] = old
        res.append(got)
    for lat, ref in zip(*res):
        assert _same_outputs(lat, ref, 3)


def _engine_env(blob, env, **kw):
    from cuahn_vio_amd.homography_net import HnetEngine
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return HnetEngine(blob, **kw)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("precision", [pytest.param(3, id="f16x2"), pytest.param(2, id="bf16x3")])
@pytest.mark.parametrize("variant,n_mc,batch", [("full", 32, 1), ("prior3", 16, 1), ("prior1", 64, 2), ("full", 16, 8), ("prior3", 5, 7), ("full", 16, 20)])
def test_split_k_without_reduce_launches_is_bitwise_the_reduce_launches(blob, variant, n_mc, batch, precision):
    """Round 5: the split-K layers of small batches with at most 40 GEMM rows are finished by the last workgroup of a tile to arrive (igemm_s3.h
    s3_splitk_last_arriver) instead of a splitk_reduce* launch (hnet_config.variant 30 keeps those; a reduce-on-load form was measured and removed,
    DESIGN_HISTORY / profiles/r05_experiments_not_shipped.log): same sums in the same order -> the same homography bits, also when the forward is repeated on one context (the tile counters return
    to zero, the partials of an earlier forward are never read)."""
    from cuahn_vio_amd import synth
    prev, curr, prior, _ = synth.make_batch(700 + batch, batch)
    pr = None if variant == "full" else prior
    kw = dict(variant=variant, mc_samples=n_mc, dropout_p=0.05, mc_seed=5, max_batch=batch, emit_error_map=True, precision=precision)
    outs = []
    for env in ({}, {"HNET_S3_TILE": "30"}):
        e = _engine_env(blob, env, **kw)
        reps = [e.infer_batch(prev, curr, pr, pair_seq0=9, want_err=True) for _ in range(3)]
        for r in reps[1:]:
            assert all(np.array_equal(x, y) for x, y in zip(r, reps[0]))
        outs.append((reps[0], np.stack([e.debug_h_part1(b) for b in range(batch)])))
        e.close()
    assert np.array_equal(outs[0][1], outs[1][1])
    assert _same_outputs(outs[0][0], outs[1][0], precision if batch <= 8 else 2)      # (beyond 8 pairs the heads run the same kernels on both sides)


@pytest.mark.parametrize("variant,n_mc,batch", [("full", 32, 1), ("prior3", 16, 1), ("prior1", 8, 2)])
def test_last_arriver_split_k_many_repetitions_against_the_reduce_launches(blob, variant, n_mc, batch):
    """the fence-free last arriver exchanges split-K partials between workgroups outside the HIP memory model (write-through stores, an
    asm vmcnt(0) wait, a workgroup barrier, a relaxed agent-scope ticket) - a failure would be silent wrong sums.  400 back-to-back forwards on resident
    buffers (no host synchronisation in between: the launches of consecutive forwards run tail to head, the tile counters are re-used every 10 us) must ALL
    reproduce the homography of the supported fallback, variant 30 (splitk_reduce* launches), bit for bit, and the packed outputs of the first repetition."""
    import torch
    from cuahn_vio_amd import synth
    from cuahn_vio_amd.homography_net import PIX_U8
    dev = torch.device("cuda:0")
    reps = 400
    ph, ch, prh, _ = synth.make_batch(640 + batch, batch)
    prev, curr, prior = torch.from_numpy(ph).to(dev), torch.from_numpy(ch).to(dev), torch.from_numpy(prh).to(dev)
    d_prior = prior.data_ptr() if variant != "full" else None
    kw = dict(variant=variant, mc_samples=n_mc, dropout_p=0.05, mc_seed=3, max_batch=batch)
    got = {}
    for name, env in (("last_arriver", {}), ("reduce_launches", {"HNET_S3_TILE": "30"})):
        e = _engine_env(blob, env, **kw)
        n = reps if name == "last_arriver" else 3
        out = torch.zeros(n, batch, 72, device=dev)
        for i in range(n):
            e.infer_batch_packed_device(prev.data_ptr(), curr.data_ptr(), PIX_U8, d_prior, batch, 5, out[i].data_ptr())
        e.synchronize()
        torch.cuda.synchronize()
        got[name] = (out.cpu().numpy(), np.stack([e.debug_h_part1(b) for b in range(batch)]))
        e.close()
    o, h = got["last_arriver"]
    differ = [i for i in range(1, reps) if not np.array_equal(o[i], o[0])]
    assert differ == [], f"{len(differ)} of {reps} forwards differ from the first (first at {differ[0]})"
    o30, h30 = got["reduce_launches"]
    assert np.array_equal(h, h30)                                  # blocks 1 - 3: the same sums in the same order
    assert np.abs(o[0][:, :8] - o30[0][:, :8]).max() < TOL_PX_PATHS      # (the heads' first FC of the latency path sums K in another order)
