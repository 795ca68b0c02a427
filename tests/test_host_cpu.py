"""CPU: host logic — weight blob format, mask function, synthetic data determinism, and that the C-ABI library
loads and exports every symbol include/hnet.h declares (no compute calls without a GPU)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_weight_inventory_matches_reference_counts():
    from cuahn_vio_amd import weights
    specs = weights.tensor_specs()
    assert len(specs) == 54
    assert sum(int(np.prod(s)) for _, s in specs) == 6_541_312
    st = weights.synthetic_state(0)
    assert list(st.keys()) == [n for n, _ in specs]
    st2 = weights.synthetic_state(0)
    assert all(np.array_equal(st[k], st2[k]) for k in st)           # deterministic
    assert not np.array_equal(weights.synthetic_state(1)["model_part1.fc_block_1.bias"], st["model_part1.fc_block_1.bias"])


def test_blob_roundtrip_and_errors(tmp_path):
    from cuahn_vio_amd import weights
    st = weights.synthetic_state(3)
    blob = weights.pack_state_dict(st)
    back = weights.unpack_blob(blob)
    assert list(back.keys()) == list(st.keys())
    assert all(np.array_equal(back[k], st[k]) for k in st)
    p = tmp_path / "w.hnw"
    weights.save_blob(str(p), st)
    assert all(np.array_equal(v, st[k]) for k, v in weights.load_blob(str(p)).items())
    bad = dict(st)
    bad.pop("model_part1.fc_block_2.bias")
    with pytest.raises(KeyError):
        weights.pack_state_dict(bad)
    bad = dict(st)
    bad["model_part1.fc_block_2.bias"] = np.zeros(9, np.float32)
    with pytest.raises(ValueError):
        weights.pack_state_dict(bad)
    with pytest.raises(ValueError):
        weights.unpack_blob(b"garbage!" + blob[8:])


def test_oracle_rejects_bad_blob(blob):
    from oracle import pyoracle
    with pytest.raises(ValueError):
        pyoracle.Oracle(b"HNETW001" + b"\0" * 64)
    with pytest.raises(ValueError):
        pyoracle.Oracle(blob[: len(blob) // 2])


def test_mask_function_properties():
    from cuahn_vio_amd import mcdrop
    a = mcdrop.keep_mask(1, 2, 0, 16, 5120, 0.05)
    assert np.array_equal(a, mcdrop.keep_mask(1, 2, 0, 16, 5120, 0.05))
    assert abs(1 - a.mean() - 0.05) < 0.005
    assert mcdrop.keep_mask(1, 2, 0, 4, 256, 0.0).all()            # p = 0 keeps everything
    # streams, pairs and seeds are decorrelated
    for other in (mcdrop.keep_mask(1, 2, 1, 16, 5120, 0.05), mcdrop.keep_mask(1, 3, 0, 16, 5120, 0.05),
                  mcdrop.keep_mask(2, 2, 0, 16, 5120, 0.05)):
        both = (~a & ~other).mean()
        assert abs(both - 0.0025) < 0.001
    # sample_offset = global sample index: a shard sees the same rows
    assert np.array_equal(mcdrop.keep_mask(9, 9, 2, 4, 256, 0.3, sample_offset=8), mcdrop.keep_mask(9, 9, 2, 12, 256, 0.3)[8:])
    assert mcdrop.drop_threshold(0.05) == 838860 and abs(float(mcdrop.scale(0.05)) - 1 / 0.95) < 1e-7


def test_synthetic_pairs_are_consistent_and_deterministic():
    from cuahn_vio_amd import synth
    a1, a2, off = synth.make_pair(5)
    b1, b2, off2 = synth.make_pair(5)
    assert np.array_equal(a1, b1) and np.array_equal(a2, b2) and np.array_equal(off, off2)
    assert a1.dtype == np.uint8 and a1.shape == (224, 320) and a1.max() > 200 and a1.std() > 20
    assert np.abs(off).max() <= 12.0
    # geometric consistency: warping img2 back with the true homography reproduces img1 in the interior
    from oracle import pyoracle
    w = pyoracle.warp(a2, synth.dlt_h(off))
    d = np.abs(w - pyoracle.as_f32_image(a1))[30:-30, 30:-30]
    assert np.median(d) < 0.02
    pr = synth.make_prior(5, off)
    assert np.abs(pr - off).max() < 2 * 1.8 and pr.dtype == np.float32


def test_capi_library_exports_every_declared_symbol():
    """the C-ABI .so loads on a box without a GPU and exports exactly what include/hnet.h declares"""
    from cuahn_vio_amd import _capi
    if not os.path.exists(_capi.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    L = _capi.lib()
    header = open(os.path.join(ROOT, "include", "hnet.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = sorted(set(re.findall(r"\b(hnet_[a-z0-9_]+)\s*\(", header)))
    assert declared == sorted(_capi.SYMBOLS)
    for name in declared:
        assert getattr(L, name) is not None
    assert b"gfx950" in L.hnet_version()
    cfg = _capi.Config()
    L.hnet_default_config(ctypes.byref(cfg))
    assert cfg.struct_size == ctypes.sizeof(_capi.Config) and cfg.mc_samples == 16 and abs(cfg.dropout_p - 0.05) < 1e-7
    assert L.hnet_status_string(4) == b"not ready (need two images)"
    # invalid arguments are rejected before any device work
    assert L.hnet_create_from_memory(ctypes.byref(cfg), None, 0, None) == 1
    assert L.hnet_image_count(None) == 0


def test_product_path_never_touches_the_oracle():
    """the shipped package must not import, link or call anything under oracle/ (no CPU fallback)"""
    pkg = os.path.join(ROOT, "cuahn_vio_amd")
    for dirpath, _d, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "pyoracle" not in txt and "liboracle" not in txt and "hnet_oracle" not in txt, f
    for f in ("hnet.h", "hnet_rng.h"):
        assert "oracle" not in open(os.path.join(ROOT, "include", f)).read().replace("CPU oracle", "")


def test_pth_tar_checkpoint_round_trip(tmp_path, state, blob):
    """the reference's checkpoint container — torch.save({'state_dict': ...}, 'x.pth.tar'), what
    model_to_trace.py:340-344 loads — converts to exactly the blob hnet_create consumes; DataParallel's 'module.' prefix too"""
    import subprocess
    import sys

    import torch
    from cuahn_vio_amd import weights
    sd = {k: torch.from_numpy(v.copy()) for k, v in state.items()}
    p1 = tmp_path / "model_best.pth.tar"
    torch.save({"epoch": 12, "state_dict": sd, "best_EPE": 0.5}, p1)
    out = tmp_path / "w.hnw"
    weights.convert_checkpoint(str(p1), str(out))
    assert out.read_bytes() == blob
    p2 = tmp_path / "dp.pth.tar"
    torch.save({"state_dict": {"module." + k: v for k, v in sd.items()}}, p2)
    out2 = tmp_path / "w2.hnw"
    r = subprocess.run([sys.executable, "-m", "cuahn_vio_amd.weights", str(p2), str(out2)], capture_output=True, text=True,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stderr
    assert out2.read_bytes() == blob
    # strictness: a missing or an extra tensor is an error, like load_state_dict(strict=True)
    bad = dict(sd)
    bad.pop("model_part1.fc_block_1.bias")
    torch.save({"state_dict": bad}, tmp_path / "bad.pth.tar")
    with pytest.raises(KeyError):
        weights.convert_checkpoint(str(tmp_path / "bad.pth.tar"), str(tmp_path / "bad.hnw"))


def test_create_refuses_a_max_batch_beyond_the_kernels_31_bit_plane_offsets():
    """hnet_create validates the configuration before it touches the device: 1 779 pairs is what a 2 GiB buffer descriptor holds of BOTH planes of the
    largest activation (block_4_1's bordered map, kernels.h B42_*: block42_fused_kernel reaches the two planes through one descriptor); a larger
    max_batch must be refused, not silently read as zeros"""
    import ctypes as C
    from cuahn_vio_amd import _capi
    L = _capi.lib()
    cfg = _capi.Config()
    L.hnet_default_config(C.byref(cfg))
    cfg.max_batch = 1780
    h = C.c_void_p()
    junk = (C.c_ubyte * 16)()
    rc = L.hnet_create_from_memory(C.byref(cfg), junk, 16, C.byref(h))
    assert rc == 5 and not h.value          # HNET_ERR_CAPACITY, before the blob is parsed or a device is touched
    cfg.max_batch = 1779
    rc = L.hnet_create_from_memory(C.byref(cfg), junk, 16, C.byref(h))
    assert rc not in (0, 5) and not h.value  # (the junk blob is what is wrong now)


def test_variant_record_round_trip_and_config_validation(state):
    """the `hnet.variant` record of an HNETW001 blob (what the reference bakes into a traced .pt): written / read back by cuahn_vio_amd.weights, skipped by
    readers that do not ask for it; and hnet_create refuses configuration values it does not know before it touches a device"""
    import ctypes as C
    from cuahn_vio_amd import _capi, weights
    b0 = weights.pack_state_dict(state)
    b1 = weights.pack_state_dict(state, dict(variant="prior2", mc_samples=32, dropout_p=0.25, emit_error_map=True))
    assert weights.blob_variant(b0) is None
    assert weights.blob_variant(b1) == {"variant": "prior2", "mc_samples": 32, "dropout_p": 0.25, "emit_error_map": True}
    u = weights.unpack_blob(b1)
    assert all(np.array_equal(u[k], state[k]) for k in state)
    assert len(b1) > len(b0)
    with pytest.raises(KeyError):
        weights.variant_record("prior9")
    with pytest.raises(ValueError):
        weights.variant_record("full", mc_samples=0)
    L = _capi.lib()
    junk = (C.c_ubyte * 16)()
    for field, val in (("graph", 3), ("variant", 99), ("variant", 1 << 20)):
        cfg = _capi.Config()
        L.hnet_default_config(C.byref(cfg))
        setattr(cfg, field, val)
        h = C.c_void_p()
        assert L.hnet_create_from_memory(C.byref(cfg), junk, 16, C.byref(h)) == 1 and not h.value      # HNET_ERR_INVALID_ARG


def test_weights_check_reports_the_fp16_plane_range(state, tmp_path):
    """python -m cuahn_vio_amd.weights FILE --check: max |w| per matrix-core layer against the fp16-plane bound and the arithmetic mode
    hnet_create will pick - on an in-range file and on one with a single weight of 20 (-> HNET_PREC_BF16X3, named)"""
    import subprocess
    import sys
    from cuahn_vio_amd import weights
    rows, mode = weights.weight_range_report(state)
    assert len(rows) == 22 and all(ok for _k, _m, ok in rows) and "F16X2" in mode
    big = {k: v.copy() for k, v in state.items()}
    big["model_part1.block_2_3.0.weight"][3, 5, 1, 1] = -20.0
    rows, mode = weights.weight_range_report(big)
    assert [k for k, _m, ok in rows if not ok] == ["model_part1.block_2_3.0.weight"] and "BF16X3" in mode
    p = tmp_path / "big.hnw"
    weights.save_blob(str(p), big)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "cuahn_vio_amd.weights", str(p), "--check"], capture_output=True, text=True, cwd=root, timeout=120)
    assert r.returncode == 0 and "BEYOND" in r.stdout and "HNET_PREC_BF16X3" in r.stdout.splitlines()[-1]
