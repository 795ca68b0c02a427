"""Round 6, opt-in (HNET_WARP_FUSE=1 / include/hnet.h HNET_VARIANT_WARP_FUSE; it measured slower than the two launches and is not the default): at batches > 8
(default arithmetic mode, u8 images) block 4's input - cat(img1, warp(img2, H)), model_to_trace.py:261-263 - is sampled INSIDE the
block_4_0 + block_4_1 kernel (csrc/conv_b4_fused.h WARPIN) instead of being written as padded fp16 planes by a prep launch of its own.  The sampler is the prep
kernel's fast sampler instruction for instruction (csrc/warp_dev.h, kernels.hip warp_sample_box_fast), so wherever both forms take it the two paths must agree
BIT FOR BIT: on block_4_1's output map and on everything downstream.  Where a patch / tile falls back to the exact sampler (extreme homographies: the two forms
decide per 25 x 80 patch and per 32 x 64 tile respectively) they agree to the sampler's contract instead, gated through the outputs.
Without the switch the prep launch stays."""
import contextlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@contextlib.contextmanager
def _env(**kv):
    old = {k: os.environ.get(k) for k in kv}
    os.environ.update({k: str(v) for k, v in kv.items()})
    try:
        yield
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _engine(blob, fused, **kw):
    from cuahn_vio_amd.homography_net import HnetEngine
    with _env(HNET_WARP_FUSE="1" if fused else "0"):
        return HnetEngine(blob, **kw)


def _run(blob, fused, prev, curr, prior, variant, n_mc, layers=(14,), pairs=(0,)):
    b = prev.shape[0]
    e = _engine(blob, fused, variant=variant, mc_samples=n_mc, dropout_p=0.05, mc_seed=3, max_batch=b, emit_error_map=True)
    mean, cov, err = e.infer_batch(prev, curr, None if variant == "full" else prior, pair_seq0=900, want_err=True)
    maps = [e.debug_layer_output(l, p) for l in layers for p in pairs]
    names = [n for n, _f in e.stages()]
    e.close()
    return mean, cov, err, maps, names


@pytest.mark.parametrize("variant,n_mc,batch", [("full", 16, 16), ("prior3", 16, 9), ("prior2", 8, 40), ("prior1", 4, 33)])
def test_in_kernel_warp_is_bitwise_the_prep_launch(blob, oracle, variant, n_mc, batch):
    from conftest import TOL_PX_VS_ORACLE
    from cuahn_vio_amd import synth
    prev, curr, prior, _ = synth.make_batch(7000 + batch, batch, max_offset=20.0)
    pairs = (0, batch // 2, batch - 1)
    m1, c1, e1, maps1, n1 = _run(blob, True, prev, curr, prior, variant, n_mc, pairs=pairs)
    m0, c0, e0, maps0, n0 = _run(blob, False, prev, curr, prior, variant, n_mc, pairs=pairs)
    assert "prep_b4" in n0 and "prep_b4" not in n1 and len(n1) == len(n0) - 1
    for a, b in zip(maps1, maps0):
        assert a.shape == (16, 112, 160) and np.array_equal(a, b)
    assert np.array_equal(m1, m0) and np.array_equal(c1, c0) and np.array_equal(e1, e0)
    btr = {"full": 3, "prior3": 3, "prior2": 2, "prior1": 1}[variant]
    for b in (0, batch - 1):
        o = oracle.forward(prev[b], curr[b], None if variant == "full" else prior[b], btr, n_mc, 0.05, 3, 900 + b)
        assert np.abs(m1[b] - o["mean"]).max() < TOL_PX_VS_ORACLE


def test_in_kernel_warp_with_wild_priors(blob, oracle):
    """priors far beyond the training range (corner offsets of +- 70 px: rotations, strong perspective, boxes that do not fit the staging area, homographies that
    leave the image): every fallback of the sampler - exact sampler from global memory, zero padding - against the prep launch, through the network's outputs"""
    from conftest import TOL_PX_VS_ORACLE
    from cuahn_vio_amd import synth
    batch = 24
    prev, curr, prior, _ = synth.make_batch(7100, batch)
    rng = np.random.default_rng(5)
    prior = rng.uniform(-70.0, 70.0, size=(batch, 8)).astype(np.float32)
    prior[0] = np.array([150, 100, 150, -100, -150, -100, -150, 100], np.float32)      # the corners pulled to the middle: a 3 x magnification of the centre
    prior[1] = np.array([-300, -200, -300, 200, 300, 200, 300, -200], np.float32)      # pushed far out: most of the warped image is zero padding
    prior[2] = np.array([0, 0, 319, -223, 0, 0, -319, 223], np.float32) * 0.9           # ~ 80 degree twist
    m1, c1, e1, maps1, _ = _run(blob, True, prev, curr, prior, "prior3", 8, pairs=(0, 1, 2, 5))
    m0, c0, e0, maps0, _ = _run(blob, False, prev, curr, prior, "prior3", 8, pairs=(0, 1, 2, 5))
    assert np.isfinite(m1).all() and np.isfinite(c1).all()
    for a, b in zip(maps1, maps0):
        # (a pixel whose sampling position moved by 6e-5 px changes block_4_1's activations in their last bits)
        assert np.abs(a - b).max() <= 2e-3 * max(1.0, float(np.abs(b).max()))
    assert np.abs(m1 - m0).max() < 1e-4
    for b in (0, 1, 2, 7):
        o = oracle.forward(prev[b], curr[b], prior[b], 3, 8, 0.05, 3, 900 + b)
        assert np.abs(m1[b] - o["mean"]).max() < TOL_PX_VS_ORACLE


def test_unaligned_images_keep_the_prep_launch(blob):
    """the in-kernel form fetches the images by 4-byte LDS-DMA: device images that are not 4-byte aligned take the prep launch.  (Not bitwise against the aligned
    run: the pooled prep kernels of blocks 1 - 3 sum their windows through other loads on odd addresses; fp32 rounding.)"""
    import torch
    from cuahn_vio_amd import synth
    from cuahn_vio_amd.homography_net import PIX_U8
    batch = 12
    prev, curr, _prior, _ = synth.make_batch(7200, batch)
    dev = torch.device("cuda:0")
    outs = []
    for shift in (0, 1, 2):
        e = _engine(blob, True, variant="full", mc_samples=8, dropout_p=0.05, mc_seed=1, max_batch=batch)
        raw_p = torch.zeros(prev.size + 16, dtype=torch.uint8, device=dev)
        raw_c = torch.zeros(curr.size + 16, dtype=torch.uint8, device=dev)
        raw_p[shift:shift + prev.size] = torch.from_numpy(prev.reshape(-1)).to(dev)
        raw_c[shift:shift + curr.size] = torch.from_numpy(curr.reshape(-1)).to(dev)
        mean = torch.zeros(batch, 8, device=dev)
        cov = torch.zeros(batch, 64, device=dev)
        torch.cuda.synchronize()
        e.infer_batch_device(raw_p.data_ptr() + shift, raw_c.data_ptr() + shift, PIX_U8, None, batch, 5, mean.data_ptr(), cov.data_ptr())
        e.synchronize()
        outs.append((mean.cpu().numpy(), cov.cpu().numpy()))
        e.close()
    for m, c in outs[1:]:
        assert np.abs(m - outs[0][0]).max() < 5e-6 and np.abs(c - outs[0][1]).max() < 1e-5 * np.abs(outs[0][1]).max()


def test_layer_13_of_a_large_batch_is_refused_not_stale(blob):
    """hnet_debug_layer_output(13) recomputes block_4_0 from the padded planes the prep launch wrote; after a forward that sampled in-kernel they are stale"""
    from cuahn_vio_amd import synth
    from cuahn_vio_amd.homography_net import HnetError
    prev, curr, _prior, _ = synth.make_batch(7300, 10)
    e = _engine(blob, True, variant="full", mc_samples=4, max_batch=10)
    e.infer_batch(prev, curr)
    with pytest.raises(HnetError):
        e.debug_layer_output(13, 0)
    e.infer_batch(prev[:2], curr[:2])                 # a latency-path forward writes the planes again
    assert e.debug_layer_output(13, 1).shape == (8, 224, 320)
    e.close()
