"""The latency path (batch <= 8): the block-tail launch (FC 5120 -> 8 + DLT + composition) and the prior's DLT are recomputed inside the next
block's prep kernel, heads_fc2 + mc_finish run as one launch (csrc/hnet_capi.hip forward_chunk, kernels.h FcArgs); round 5: split-K layers are
finished without reduce launches (last-arriver tiles, reduce-on-load chains: kernels.h LatIO) - all of this is the same arithmetic in the same
order, so the homography of blocks 1 - 3 must be BITWISE that of the multi-launch path (HNET_FUSE_SMALL=0) in every arithmetic mode and variant.
The heads' first FC is a different kernel on the latency path of the default mode (csrc/heads_lat.h: one launch, K summed in another order): there
the outputs agree to fp32 rounding - gated at 5e-5 px / 1e-5 relative in the covariance, the other modes stay bitwise.
(Measured over the cases below: 0 ... 3.0e-5 px between the two orders, each of them 2.2e-5 ... 3.8e-5 px from the double-accumulating oracle: the
mean head's last FC multiplies the hidden units by O(1) weights and cancels to O(5 px), so a 1e-7 relative change of the hidden units is a few 1e-5 px.)
Round 6: in the default mode the tail layers of every block run as ONE launch on one XCD (csrc/chain_lat.h: an item sums the whole K inside a workgroup instead of
split-K partials in memory - another summation order), so the homography of blocks 1 - 3 agrees with the multi-launch path to fp32 rounding there too (gated
through the offsets: 5e-5 px; HNET_CHAIN=0 keeps the per-layer launches and the bitwise comparison); the other modes stay bitwise."""
TOL_PX_PATHS = 5e-5       # |offset(latency path) - offset(multi-launch path)|, px: different summation orders (heads' first FC; round 6: the tail chains)
TOL_COV_PATHS = 1e-5
TOL_H_PATHS = 2e-6        # relative, on the 3 x 3 homography of blocks 1 - 3 (default mode with the tail chains)


def _same_h(h1, h0, precision, chain=True):
    if precision != 3 or not chain:
        return np.array_equal(h1, h0)
    return float(np.abs(h1 - h0).max() / np.abs(h0).max()) < TOL_H_PATHS


def _same_outputs(lat, ref, precision):
    """(mean, cov, err) of the latency path against the reference path: bitwise except in the default mode, whose heads FC1 sums K in another order"""
    (m1, c1, e1), (m0, c0, e0) = lat, ref
    if precision != 3:
        return np.array_equal(m1, m0) and np.array_equal(c1, c0) and np.array_equal(e1, e0)
    de = np.abs(e1.astype(np.float64) - e0.astype(np.float64))
    # (the u8 error map of the class surface is rounded: a homography that differs in its last bits flips single pixels by one grey level)
    err_ok = (de.max() <= 1.0 and float((de > 0).mean()) < 1e-3) if e1.dtype == np.uint8 else de.max() < 0.05
    return np.abs(m1 - m0).max() < TOL_PX_PATHS and np.abs(c1 - c0).max() / np.abs(c0).max() < TOL_COV_PATHS and err_ok
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
    assert _same_h(h1, h0, precision) and _same_outputs((m1, c1, e1), (m0, c0, e0), precision)
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
    for env in ({"HNET_CHAIN": "0"}, {"HNET_S3_TILE": "30"}):      # (the last-arriver split-K layers are what the chains replace: HNET_CHAIN=0 keeps them)
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
    for name, env in (("last_arriver", {"HNET_CHAIN": "0"}), ("reduce_launches", {"HNET_S3_TILE": "30"})):
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
    assert np.array_equal(h, h30)                                  # blocks 1 - 3: the same sums in the same order (HNET_CHAIN=0 on both sides: _engine_env below)
    assert np.abs(o[0][:, :8] - o30[0][:, :8]).max() < TOL_PX_PATHS      # (the heads' first FC of the latency path sums K in another order)


# ---- round 6: the one-XCD tail chains (csrc/chain_lat.h)
def _chain_outputs(blob, env, variant, n_mc, batch, prev, curr, pr, reps=1):
    e = _engine_env(blob, env, variant=variant, mc_samples=n_mc, dropout_p=0.05, mc_seed=7, max_batch=8, emit_error_map=True, precision=3)
    outs = [e.infer_batch(prev, curr, pr, pair_seq0=21, want_err=True) for _ in range(reps)]
    first = {"full": 0, "prior3": 3, "prior2": 7, "prior1": 13}[variant]
    layers = {l: np.stack([e.debug_layer_output(l, b) for b in range(batch)]) for l in range(first, 20) if l != 13}
    h1 = np.stack([e.debug_h_part1(b) for b in range(batch)])
    names = [n for n, _f in e.stages()]
    flag = e.overflow_flag()
    e.close()
    return outs, layers, h1, names, flag


@pytest.mark.parametrize("variant,n_mc,batch", [("full", 32, 1), ("prior3", 16, 1), ("full", 16, 2), ("prior2", 8, 5), ("full", 16, 8), ("prior1", 8, 3), ("prior3", 16, 7)])
def test_tail_chains_match_the_per_layer_launches(blob, oracle, variant, n_mc, batch):
    """every conv layer's map, the homography and the outputs of the chain launches (one per block, block_x_4+x_5+x_6 etc.) against the per-layer launches
    (HNET_CHAIN=0) of the same context configuration: fp32 rounding apart (another summation order), every repetition bit-identical, no flag raised, fewer launches;
    and against the CPU oracle"""
    from conftest import TOL_PX_VS_ORACLE
    from cuahn_vio_amd import synth
    prev, curr, prior, _ = synth.make_batch(800 + batch, batch)
    pr = None if variant == "full" else prior
    (o1, l1, h1, n1, f1) = _chain_outputs(blob, {}, variant, n_mc, batch, prev, curr, pr, reps=4)
    (o0, l0, h0, n0, f0) = _chain_outputs(blob, {"HNET_CHAIN": "0"}, variant, n_mc, batch, prev, curr, pr)
    for r in o1[1:]:
        assert all(np.array_equal(x, y) for x, y in zip(r, o1[0]))
    assert f1 == 0 and f0 == 0
    assert any("+" in n and n.count("+") >= 1 and n.startswith("block_") and n.split("+")[0][-1] in "234" for n in n1) and len(n1) < len(n0)
    for l in l1:
        d = float(np.abs(l1[l] - l0[l]).max() / max(float(np.abs(l0[l]).max()), 1e-30))
        assert d < 5e-5, (l, d)
    assert _same_h(h1, h0, 3) and _same_outputs(o1[0], o0[0], 3)
    btr = {"full": 3, "prior3": 3, "prior2": 2, "prior1": 1}[variant]
    for b in (0, batch - 1):
        o = oracle.forward(prev[b], curr[b], None if pr is None else pr[b], btr, n_mc, 0.05, 7, 21 + b)
        assert np.abs(o1[0][0][b] - o["mean"]).max() < TOL_PX_VS_ORACLE


@pytest.mark.parametrize("grid", ["8", "3"])
@pytest.mark.parametrize("variant,n_mc,batch", [("full", 16, 1), ("prior3", 16, 8), ("prior2", 8, 5)])
def test_tail_chains_do_not_depend_on_how_many_workgroups_run_where(blob, variant, n_mc, batch, grid):
    """Placement independence, exercised: the chain launched with 8 workgroups (one per XCD under the usual placement: every workgroup works through ALL 32 items of
    every layer - the path taken when fewer workgroups are resident than a layer has items) and with 3 (five XCDs without any workgroup: the pairs they would
    have claimed are picked up by workgroups that are done) gives the bits of the 256-workgroup launch."""
    from cuahn_vio_amd import synth
    prev, curr, prior, _ = synth.make_batch(830 + batch, batch)
    pr = None if variant == "full" else prior
    (o1, l1, h1, _n1, f1) = _chain_outputs(blob, {}, variant, n_mc, batch, prev, curr, pr)
    (o2, l2, h2, _n2, f2) = _chain_outputs(blob, {"HNET_CHAIN_GRID": grid}, variant, n_mc, batch, prev, curr, pr, reps=2)
    assert f1 == 0 and f2 == 0
    assert np.array_equal(h1, h2) and all(np.array_equal(x, y) for x, y in zip(o1[0], o2[0])) and all(np.array_equal(x, y) for x, y in zip(o2[0], o2[1]))
    for l in l1:
        assert np.array_equal(l1[l], l2[l]), l


def test_tail_chains_many_forwards_back_to_back(blob):
    """500 forwards on resident buffers without a host synchronisation in between, alternating batch sizes on ONE context (the counter areas are zeroed by the
    previous chain launch of the stream; a later launch of a larger batch must find the pairs it did not use zeroed too): every forward of a batch size
    reproduces the first one bit for bit, and the flag word stays clear"""
    import torch
    from cuahn_vio_amd import synth
    from cuahn_vio_amd.homography_net import PIX_U8, HnetEngine
    dev = torch.device("cuda:0")
    ph, ch, _prh, _ = synth.make_batch(870, 8)
    prev, curr = torch.from_numpy(ph).to(dev), torch.from_numpy(ch).to(dev)
    e = HnetEngine(blob, variant="full", mc_samples=16, dropout_p=0.05, mc_seed=3, max_batch=8, precision=3)
    reps, sizes = 500, (1, 8, 3, 1, 5)
    out = torch.zeros(reps, 8, 72, device=dev)
    for i in range(reps):
        e.infer_batch_packed_device(prev.data_ptr(), curr.data_ptr(), PIX_U8, None, sizes[i % len(sizes)], 5, out[i].data_ptr())
    e.synchronize()
    torch.cuda.synchronize()
    assert e.overflow_flag() == 0
    o = out.cpu().numpy()
    e.close()
    for i in range(len(sizes), reps):
        b = sizes[i % len(sizes)]
        assert np.array_equal(o[i][:b], o[i % len(sizes)][:b]), i


@pytest.mark.parametrize("use_prior", [True, False])
def test_infer_graph_without_copy_nodes_is_the_graph_with_them(blob, tmp_path, use_prior):
    """hnet_infer's graph lets the kernels read {sequence number, prior} from and write {mean, covariance, error map, flag} to the pinned host block (round 6:
    no memcpy / memset nodes in the dependent chain); HNET_VARIANT_GRAPH_COPIES (HNET_GRAPH_COPIES=1 in the mirror) keeps the copy nodes: the same bits, frame by frame"""
    from cuahn_vio_amd import synth
    from cuahn_vio_amd.homography_net import HomographyNet
    path = str(tmp_path / "w_showError.hnw")
    with open(path, "wb") as f:
        f.write(blob)
    frames = [synth.make_pair(60 + i)[0] for i in range(5)]
    prior = np.array([1.0, -2.0, 0.5, 3.0, -1.5, 0.25, 2.0, -0.75])
    outs = []
    for copies in ("0", "1"):
        old = os.environ.get("HNET_GRAPH_COPIES")
        os.environ["HNET_GRAPH_COPIES"] = copies
        try:
            net = HomographyNet(path, "", use_prior=use_prior, num_of_iteration=1, show_imgs=False, dropout_p=0.05, mc_seed=5)
        finally:
            if old is None:
                os.environ.pop("HNET_GRAPH_COPIES", None)
            else:
                os.environ["HNET_GRAPH_COPIES"] = old
        res = []
        for i, fr in enumerate(frames):
            net.load_current_img(fr, float(i))
            if i:
                net.network_inference(prior + i, 0)
                res.append((net.get_pred_mean().copy(), net.get_pred_Cov().copy(), None if net.last_error_map is None else net.last_error_map.copy()))
        net.close()
        outs.append(res)
    for (m1, c1, e1), (m0, c0, e0) in zip(*outs):
        assert np.isfinite(m1).all() and m1.any() and np.array_equal(m1, m0) and np.array_equal(c1, c0)
        assert (e1 is None and e0 is None) or (e1.any() and np.array_equal(e1, e0))


@pytest.mark.parametrize("variant,n_mc,batch", [("full", 32, 1), ("prior3", 16, 2), ("prior2", 8, 5), ("full", 16, 8)])
def test_block_tail_fc_from_the_chain_partials(blob, oracle, variant, n_mc, batch):
    """Round 6: the last layer of a block's tail chain leaves the block-tail Linear(5120, 8) as 32 partial sums per pair (an item's 8 channels x 20 pixels times their
    slice of the FC weights, csrc/chain_lat.h) and the next warp + pool launch adds them in item order instead of re-reading features and weights in every workgroup
    (HNET_CHAIN_FC=0 / HNET_VARIANT_CHAIN_NO_FC keeps that form): another order of the 5 120 products - the homographies agree to fp32 rounding, the outputs within
    the gate of the other cross-order comparisons of this file, both within 1e-4 px of the oracle"""
    from conftest import TOL_PX_VS_ORACLE
    from cuahn_vio_amd import synth
    from cuahn_vio_amd.homography_net import HnetEngine
    prev, curr, prior, _ = synth.make_batch(800 + batch, batch)
    pr = None if variant == "full" else prior
    res = []
    for fc in ("1", "0"):
        old = os.environ.get("HNET_CHAIN_FC")
        os.environ["HNET_CHAIN_FC"] = fc
        try:
            e = HnetEngine(blob, variant=variant, mc_samples=n_mc, dropout_p=0.05, mc_seed=9, max_batch=8)
        finally:
            if old is None:
                os.environ.pop("HNET_CHAIN_FC", None)
            else:
                os.environ["HNET_CHAIN_FC"] = old
        outs = [e.infer_batch(prev, curr, pr, pair_seq0=31) for _ in range(3)]
        assert all(np.array_equal(o[0], outs[0][0]) and np.array_equal(o[1], outs[0][1]) for o in outs[1:])      # reproducible forward to forward
        h1 = np.stack([e.debug_h_part1(b) for b in range(batch)])
        e.close()
        res.append((outs[0][0], outs[0][1], h1))
    (m1, c1, h1), (m0, c0, h0) = res
    assert float(np.abs(h1 - h0).max() / np.abs(h0).max()) < TOL_H_PATHS
    assert np.abs(m1 - m0).max() < TOL_PX_PATHS and np.abs(c1 - c0).max() / np.abs(c0).max() < TOL_COV_PATHS
    btr = {"full": 3, "prior3": 3, "prior2": 2, "prior1": 1}[variant]
    for b in (0, batch - 1):
        o = oracle.forward(prev[b], curr[b], None if pr is None else pr[b], btr, n_mc, 0.05, 9, 31 + b)
        assert np.abs(m1[b] - o["mean"]).max() < TOL_PX_VS_ORACLE
