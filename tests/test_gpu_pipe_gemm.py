"""The software-pipelined LDS-DMA implicit GEMM on whole-pair tiles (csrc/igemm_pipe.h, round 4): every layer it serves, element by
element against the oracle's conv (reference op: conv + bias + LeakyReLU(0.1), model_to_trace.py:7-15) and bit for bit against the
four-wave lean kernel it replaces (same K order and MFMA sequence per accumulator)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# layers the kernel serves: (layer index, batch sizes): batches chosen so that the last tile is ragged (valid rows < the tile's)
PIPE_LAYERS = [4, 5, 10, 11, 17, 18]        # block_2_2, 2_3, 3_3, 3_4, 4_4, 4_5


def _geometry(layer):
    from cuahn_vio_amd.weights import CONV_LAYERS
    name, cin, cout, k, s = CONV_LAYERS[layer]
    blk = int(name[6])
    h, w = {1: (28, 40), 2: (56, 80), 3: (112, 160), 4: (224, 320)}[blk]
    for n2, _ci, _co, k2, s2 in CONV_LAYERS:
        if n2 == name:
            break
        if n2[6] == name[6]:
            p2 = (k2 - 1) // 2
            h, w = (h + 2 * p2 - k2) // s2 + 1, (w + 2 * p2 - k2) // s2 + 1
    return name, cin, cout, k, s, h, w


def _engine(blob, tile):
    from cuahn_vio_amd.homography_net import HnetEngine
    old = os.environ.get("HNET_S3_TILE")
    os.environ["HNET_S3_TILE"] = str(tile)
    try:
        return HnetEngine(blob, variant="full", mc_samples=4, dropout_p=0.05, mc_seed=1, max_batch=1, precision=3)
    finally:
        if old is None:
            del os.environ["HNET_S3_TILE"]
        else:
            os.environ["HNET_S3_TILE"] = old


@pytest.fixture(scope="module")
def engines(blob):
    pipe, lean = _engine(blob, 21), _engine(blob, 20)      # 21: the pipelined kernel at any M; 20: never
    yield pipe, lean
    pipe.close()
    lean.close()


@pytest.mark.parametrize("layer", PIPE_LAYERS)
@pytest.mark.parametrize("batch", [1, 3, 5])
def test_pipe_kernel_vs_oracle_and_lean_kernel(engines, state, layer, batch):
    from oracle import pyoracle
    pipe, lean = engines
    name, cin, cout, k, s, h, w = _geometry(layer)
    rng = np.random.default_rng(100 + layer)
    x = rng.standard_normal((batch, cin, h, w)).astype(np.float32)
    prefix = "model_last_block_list.0." if name[6] == "4" else "model_part1."
    wgt, bias = state[prefix + name + ".0.weight"], state[prefix + name + ".0.bias"]
    got = pipe.op_conv(layer, x)
    ref_k = lean.op_conv(layer, x)
    for b in range(batch):
        ref = pyoracle.conv_lrelu(x[b], wgt, bias, s)
        err = np.abs(got[b] - ref).max()
        assert err < 2e-5 * max(1.0, np.abs(ref).max()), f"{name} pair {b}: {err}"
    assert np.array_equal(got, ref_k), f"{name}: pipelined kernel != lean kernel bitwise (max diff {np.abs(got - ref_k).max()})"


REGION_LAYERS = [1, 2, 6, 12, 19]          # block_1_2 (one pair per tile), block_1_3 and block_2_4 / 3_5 / 4_6 (four pairs per tile, K split over the wave halves)


@pytest.mark.parametrize("layer", REGION_LAYERS)
@pytest.mark.parametrize("batch", [1, 3, 5, 8])
def test_region_kernel_vs_oracle(engines, state, layer, batch):
    """igemm_s3_region_kernel (csrc/igemm_region.h: the input region of the tile's pairs resident in LDS, weights straight to registers, 64-channel chunk
    major K order) at the network geometry of every layer it serves, every element against the oracle's conv; against the lean kernel to fp32 rounding
    (another summation order).  Batches 1, 3, 5: ragged last tile of the four-pair form."""
    from oracle import pyoracle
    pipe, lean = engines
    name, cin, cout, k, s, h, w = _geometry(layer)
    rng = np.random.default_rng(300 + 7 * layer + batch)
    x = rng.standard_normal((batch, cin, h, w)).astype(np.float32)
    prefix = "model_last_block_list.0." if name[6] == "4" else "model_part1."
    wgt, bias = state[prefix + name + ".0.weight"], state[prefix + name + ".0.bias"]
    got = pipe.op_conv(layer, x)
    ref_k = lean.op_conv(layer, x)
    for b in range(batch):
        ref = pyoracle.conv_lrelu(x[b], wgt, bias, s)
        err = np.abs(got[b] - ref).max()
        assert err < 2e-5 * max(1.0, np.abs(ref).max()), f"{name} pair {b}: {err}"
    assert np.abs(got - ref_k).max() < 2e-5 * max(1.0, np.abs(ref_k).max())
    assert not np.array_equal(got, ref_k) or batch == 0, f"{name}: identical to the lean kernel bit for bit - did the region kernel run?"
