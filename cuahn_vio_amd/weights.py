"""HomographyNet weight set: tensor inventory, deterministic synthetic generator, flat blob format.

The reference keeps its parameters in a PyTorch ``state_dict`` of 54 tensors
(reference: trace_pytorch_model/model_to_trace.py:88-115 blocks 1-3, :210-235 block 4;
loaded with ``load_state_dict(strict=True)`` at :344).  The trained checkpoint is not shipped
with the reference (``.MISSING_LARGE_BLOBS:2``), so parity is pinned with seeded synthetic
weights produced here and fed identically to the reference model, the oracle and the HIP path.

Blob format ``HNETW001`` (little endian), consumed by ``hnet_create`` (include/hnet.h):

    char     magic[8]  = "HNETW001"
    uint32   n_tensors
    n_tensors x { uint32 name_len; char name[name_len]; uint32 ndim; uint32 dims[ndim];
                  uint64 data_offset  (bytes, from start of the data section) }
    padding to a multiple of 64 bytes
    data section: float32 tensors in the reference's own layouts
                  (conv [Cout,Cin,kh,kw], linear [out,in], bias [out])

Names are the reference ``state_dict`` keys, so a real checkpoint converts 1:1
(``pack_state_dict``).
"""
from __future__ import annotations

import struct
from collections import OrderedDict

import numpy as np

MAGIC = b"HNETW001"

# (name, cin, cout, k, stride) of every convolution, in execution order per block.
# reference: model_to_trace.py:88-94 (block 1), :99-104 (block 2), :107-113 (block 3), :210-216 (block 4)
CONV_LAYERS = [
    ("block_1_1", 2, 128, 7, 2), ("block_1_2", 128, 128, 5, 2), ("block_1_3", 128, 256, 3, 2),
    ("block_2_1", 2, 64, 7, 2), ("block_2_2", 64, 128, 5, 2), ("block_2_3", 128, 256, 3, 2),
    ("block_2_4", 256, 256, 3, 2),
    ("block_3_0", 2, 16, 7, 1), ("block_3_1", 16, 32, 5, 2), ("block_3_2", 32, 64, 3, 2),
    ("block_3_3", 64, 128, 3, 2), ("block_3_4", 128, 256, 3, 2), ("block_3_5", 256, 256, 3, 2),
    ("block_4_0", 2, 8, 7, 1), ("block_4_1", 8, 16, 5, 2), ("block_4_2", 16, 32, 3, 2),
    ("block_4_3", 32, 64, 3, 2), ("block_4_4", 64, 128, 3, 2), ("block_4_5", 128, 256, 3, 2),
    ("block_4_6", 256, 256, 3, 2),
]
FC_INPUT = 5120      # 256*4*5, reference model_to_trace.py:89
FC_HIDDEN = 256      # reference model_to_trace.py:224


def tensor_specs():
    """Ordered (name, shape) list equal to the reference state_dict (54 tensors, 6 541 312 params)."""
    specs = []

    def conv(prefix, name):
        for n, cin, cout, k, _s in CONV_LAYERS:
            if n == name:
                specs.append((f"{prefix}.{n}.0.weight", (cout, cin, k, k)))
                specs.append((f"{prefix}.{n}.0.bias", (cout,)))

    p1 = "model_part1"
    for n in ("block_1_1", "block_1_2", "block_1_3"):
        conv(p1, n)
    specs += [(f"{p1}.fc_block_1.weight", (8, FC_INPUT)), (f"{p1}.fc_block_1.bias", (8,))]
    for n in ("block_2_1", "block_2_2", "block_2_3", "block_2_4"):
        conv(p1, n)
    specs += [(f"{p1}.fc_block_2.weight", (8, FC_INPUT)), (f"{p1}.fc_block_2.bias", (8,))]
    for n in ("block_3_0", "block_3_1", "block_3_2", "block_3_3", "block_3_4", "block_3_5"):
        conv(p1, n)
    specs += [(f"{p1}.fc_block_3.weight", (8, FC_INPUT)), (f"{p1}.fc_block_3.bias", (8,))]
    lb = "model_last_block_list.0"
    for n in ("block_4_0", "block_4_1", "block_4_2", "block_4_3", "block_4_4", "block_4_5", "block_4_6"):
        conv(lb, n)
    for head in ("fc_block_4_mean", "fc_block_4_uncertainty"):
        specs += [(f"{lb}.{head}.1.weight", (FC_HIDDEN, FC_INPUT)), (f"{lb}.{head}.1.bias", (FC_HIDDEN,)),
                  (f"{lb}.{head}.4.weight", (8, FC_HIDDEN)), (f"{lb}.{head}.4.bias", (8,))]
    return specs


_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    """Vectorised splitmix64 finaliser of counter array ``x`` (uint64, wrapping arithmetic)."""
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform01(seed: int, stream: int, n: int) -> np.ndarray:
    """n reproducible U[0,1) values with 24 random bits each (exact in float32)."""
    with np.errstate(over="ignore"):
        base = _splitmix64(np.array([seed], dtype=np.uint64) ^ (np.uint64(stream) * np.uint64(0xD1B54A32D192ED03)))[0]
        ctr = base + np.arange(n, dtype=np.uint64)
    bits = _splitmix64(ctr) >> np.uint64(40)
    return (bits.astype(np.float64) * (1.0 / 16777216.0)).astype(np.float32)


# the four layers whose outputs are corner offsets in pixels (SURVEY.md §8c "Weights")
_OFFSET_FC = ("model_part1.fc_block_1", "model_part1.fc_block_2", "model_part1.fc_block_3",
              "model_last_block_list.0.fc_block_4_mean.4")


def synthetic_state(seed: int = 0, offset_gain: float = 30.0, offset_bias: float = 3.0) -> "OrderedDict[str, np.ndarray]":
    """Seeded synthetic weights: U(-1/sqrt(fan_in), 1/sqrt(fan_in)) like PyTorch's default init, with
    the corner-offset FC layers scaled so every block moves the corners by O(1-10) px and the warp
    between blocks is really exercised (with default init the offsets are ~0.05 px)."""
    state = OrderedDict()
    for idx, (name, shape) in enumerate(tensor_specs()):
        n = int(np.prod(shape))
        is_bias = name.endswith(".bias")
        if is_bias:
            wshape = dict(tensor_specs())[name[:-4] + "weight"]
            fan_in = int(np.prod(wshape[1:]))
        else:
            fan_in = int(np.prod(shape[1:]))
        bound = 1.0 / np.sqrt(float(fan_in))
        u = uniform01(seed, idx, n).astype(np.float64) * 2.0 - 1.0
        layer = name.rsplit(".", 1)[0]
        if layer in _OFFSET_FC:
            if is_bias:
                vals = u * offset_bias
            else:
                vals = u * bound * offset_gain
        else:
            vals = u * bound
        state[name] = vals.astype(np.float32).reshape(shape)
    return state


def variant_state(seed: int = 0, conv_gain: float = 1.0) -> "OrderedDict[str, np.ndarray]":
    """synthetic_state(seed) with every trunk convolution scaled by `conv_gain` (activations of O(gain^depth)) and the
    5120-input FC layers scaled back so the corner offsets stay at O(1-10) px.  gain 4 drives activations to O(10^3),
    gain 0.5 drives the features to O(0.1): the two ends of the fp16-plane format's range that the golden vectors pin
    (tools/gen_golden.py `weights_seed` / `conv_gain`)."""
    st = synthetic_state(seed)
    if conv_gain != 1.0:
        g = np.float32(conv_gain)
        for k in st:
            if ".block_" in k and k.endswith(".0.weight"):
                st[k] = (st[k] * g).astype(np.float32)
        for k in st:      # undo the gain of the 3 / 4 / 6 / 7 convolutions in front of each 5120-input FC
            if "fc_block_" in k and k.endswith("weight") and st[k].shape[-1] == 5120:
                depth = 7 if "model_last_block_list" in k else (3 if "fc_block_1" in k else (4 if "fc_block_2" in k else 6))
                st[k] = (st[k] / g ** depth).astype(np.float32)
    return st


VARIANT_TENSOR = "hnet.variant"      # optional [8] float32 record of the blob: what the reference bakes into a traced .pt (trace_model.py:16,36-46)
VARIANTS = {"full": (0, 3), "prior3": (1, 3), "prior2": (1, 2), "prior1": (1, 1)}      # name -> (use_prior, blocks_to_run)


def variant_record(variant: str = "prior3", mc_samples: int = 16, dropout_p: float = 0.05, emit_error_map: bool = False) -> np.ndarray:
    """The model variant as the eight floats hnet_create reads when an hnet_config field is HNET_FROM_FILE:
    [record version 1, use_prior, blocks_to_run, mc_samples, dropout_p, emit_error_map, 0, 0].  The reference freezes exactly these into the
    TorchScript file it loads (trace_model.py:16 dropout_rate, :36-46 prior / no prior and the "_showError" twin; model_to_trace.py:72 blocks_to_run,
    :202 MC_dropout_num; HomographyNet.cpp:81-124 only names a file) - one blob per traced variant plays the same role here."""
    use_prior, blocks = VARIANTS[variant]
    if not (1 <= int(mc_samples) <= 256 and 0.0 <= float(dropout_p) < 1.0):
        raise ValueError("mc_samples in 1..256, dropout_p in [0, 1)")
    return np.array([1, use_prior, blocks, int(mc_samples), float(dropout_p), 1 if emit_error_map else 0, 0, 0], dtype="<f4")


def pack_state_dict(state, variant=None) -> bytes:
    """Serialise a name->array mapping (reference state_dict layouts) into the HNETW001 blob.
    Accepts numpy arrays or anything with ``.numpy()`` / ``.detach()`` (torch tensors).
    variant: None (weights only: the caller's hnet_config decides everything) or a ``variant_record(...)`` / a dict of its arguments,
    stored as the extra tensor ``hnet.variant`` that older readers skip."""
    specs = tensor_specs()
    if variant is not None:
        rec = variant_record(**variant) if isinstance(variant, dict) else np.asarray(variant, dtype="<f4")
        if rec.shape != (8,):
            raise ValueError("variant record: eight floats")
        state = OrderedDict(state)
        state[VARIANT_TENSOR] = rec
        specs = specs + [(VARIANT_TENSOR, (8,))]
    arrays = []
    for name, shape in specs:
        if name not in state:
            raise KeyError(f"state dict is missing tensor {name!r}")
        a = state[name]
        if hasattr(a, "detach"):
            a = a.detach().cpu().numpy()
        a = np.ascontiguousarray(np.asarray(a, dtype="<f4"))
        if tuple(a.shape) != tuple(shape):
            raise ValueError(f"{name}: shape {a.shape} != expected {shape}")
        arrays.append(a)
    extra = set(state.keys()) - {n for n, _ in specs} - {VARIANT_TENSOR}
    if extra:
        raise KeyError(f"unexpected tensors in state dict: {sorted(extra)[:4]}")
    head = bytearray(MAGIC + struct.pack("<I", len(specs)))
    off = 0
    for (name, shape), a in zip(specs, arrays):
        nb = name.encode()
        head += struct.pack("<I", len(nb)) + nb + struct.pack("<I", len(shape))
        head += struct.pack(f"<{len(shape)}I", *shape) + struct.pack("<Q", off)
        off += a.nbytes
        off = (off + 63) // 64 * 64
    head += b"\0" * ((-len(head)) % 64)
    data = bytearray(off)
    pos = 0
    for a in arrays:
        data[pos:pos + a.nbytes] = a.tobytes()
        pos = (pos + a.nbytes + 63) // 64 * 64
    return bytes(head) + bytes(data)


def unpack_blob(blob: bytes) -> "OrderedDict[str, np.ndarray]":
    if blob[:8] != MAGIC:
        raise ValueError("not an HNETW001 blob")
    (n,) = struct.unpack_from("<I", blob, 8)
    pos = 12
    entries = []
    for _ in range(n):
        (ln,) = struct.unpack_from("<I", blob, pos); pos += 4
        name = blob[pos:pos + ln].decode(); pos += ln
        (nd,) = struct.unpack_from("<I", blob, pos); pos += 4
        dims = struct.unpack_from(f"<{nd}I", blob, pos); pos += 4 * nd
        (off,) = struct.unpack_from("<Q", blob, pos); pos += 8
        entries.append((name, dims, off))
    data0 = (pos + 63) // 64 * 64
    out = OrderedDict()
    for name, dims, off in entries:
        cnt = int(np.prod(dims))
        out[name] = np.frombuffer(blob, dtype="<f4", count=cnt, offset=data0 + off).reshape(dims).copy()
    return out


def save_blob(path: str, state, variant=None) -> None:
    with open(path, "wb") as f:
        f.write(pack_state_dict(state, variant))


def blob_variant(blob: bytes):
    """the variant record of a blob as a dict, or None"""
    st = unpack_blob(blob)
    if VARIANT_TENSOR not in st:
        return None
    r = st[VARIANT_TENSOR]
    name = {v: k for k, v in VARIANTS.items()}.get((int(r[1]), int(r[2])) if int(r[1]) else (0, 3))
    return {"variant": name, "mc_samples": int(r[3]), "dropout_p": float(r[4]), "emit_error_map": bool(r[5])}


def load_blob(path: str):
    with open(path, "rb") as f:
        return unpack_blob(f.read())


def load_checkpoint(path: str) -> "OrderedDict[str, np.ndarray]":
    """The reference's checkpoint format: ``torch.load(path)['state_dict']`` (trace_pytorch_model/model_to_trace.py:340-344,
    the ``.pth.tar`` the training code writes; the file itself is a missing blob of the reference, .MISSING_LARGE_BLOBS).
    Returns name -> float32 numpy array with the reference's own state_dict keys; a ``module.`` prefix (DataParallel) is
    stripped.  torch is only needed here, never on the inference path."""
    import torch
    ck = torch.load(path, map_location="cpu", weights_only=False)
    sd = ck["state_dict"] if isinstance(ck, dict) and "state_dict" in ck else ck
    out = OrderedDict()
    for k, v in sd.items():
        k = k[len("module."):] if k.startswith("module.") else k
        out[k] = np.ascontiguousarray(v.detach().cpu().numpy(), dtype=np.float32)
    return out


def convert_checkpoint(pth_path: str, blob_path: str, variant=None) -> None:
    """``x.pth.tar`` -> HNETW001 blob for hnet_create (strict: every expected tensor present with the reference's shape,
    nothing else, as ``load_state_dict(strict=True)``); variant: see pack_state_dict"""
    st = load_checkpoint(pth_path)
    st.pop(VARIANT_TENSOR, None)
    save_blob(blob_path, st, variant)


F16X2_WEIGHT_BOUND = 15.99      # hnet_create (csrc/hnet_capi.hip): HNET_PREC_F16X2 carries 4096 w in fp16 planes, every matrix-core weight must stay below 16 (csrc/s3_format.h)


def weight_range_report(state):
    """max |w| of every layer whose contraction runs on the matrix cores (the 20 convs and the heads' Linear(5120, 256) x 2 - exactly the tensors hnet_create
    scans) against the fp16-plane bound: [(tensor name, max |w|, within the bound)], and the arithmetic mode hnet_create selects for this file when asked for
    the default HNET_PREC_F16X2 (a weight beyond the bound -> HNET_PREC_BF16X3: same results, twice the matrix-core work; never an error).  Activations
    (|a| < 32768) cannot be checked from the file: an overflow there shows as a non-finite output, which hnet_infer repairs by demoting the context and
    the device entry points flag (include/hnet.h hnet_overflow_flag)."""
    rows = []
    for name, _cin, _cout, _k, _s in CONV_LAYERS:
        key = ("model_last_block_list.0." if name.startswith("block_4") else "model_part1.") + name + ".0.weight"
        m = float(np.abs(np.asarray(state[key], dtype=np.float32)).max())
        rows.append((key, m, m < F16X2_WEIGHT_BOUND))
    for head in ("fc_block_4_mean", "fc_block_4_uncertainty"):
        key = "model_last_block_list.0." + head + ".1.weight"
        m = float(np.abs(np.asarray(state[key], dtype=np.float32)).max())
        rows.append((key, m, m < F16X2_WEIGHT_BOUND))
    mode = "HNET_PREC_F16X2 (two fp16 planes, three MFMAs per product)" if all(ok for _n, _m, ok in rows) else \
        "HNET_PREC_BF16X3 (a weight is beyond the fp16-plane bound: three bf16 planes, six MFMAs per product; same results)"
    return rows, mode


def _load_any(path: str):
    with open(path, "rb") as f:
        head = f.read(8)
    return load_blob(path) if head == MAGIC else load_checkpoint(path)


if __name__ == "__main__":
    # the counterpart of the reference's trace_model.py: one output file per variant it traces (:36-46)
    #   python -m cuahn_vio_amd.weights ck.pth.tar traced_model_3_blocks_using_prior.hnw --variant prior3 --mc 16 --dropout 0.05
    #   python -m cuahn_vio_amd.weights ck.pth.tar traced_model_3_blocks_using_prior_showError.hnw --variant prior3 --error-map
    import argparse
    ap = argparse.ArgumentParser(prog="python -m cuahn_vio_amd.weights", description="reference checkpoint (.pth.tar) -> HNETW001 blob")
    ap.add_argument("checkpoint")
    ap.add_argument("out", nargs="?", default=None)
    ap.add_argument("--check", action="store_true",
                    help="no conversion: print max |w| of every matrix-core layer of CHECKPOINT (.pth.tar or HNETW001) against the fp16-plane bound and the "
                         "arithmetic mode hnet_create will select - what the first user of the real checkpoint learns before the first frame")
    ap.add_argument("--variant", choices=sorted(VARIANTS), default=None, help="bake the model variant into the blob (default: weights only)")
    ap.add_argument("--mc", type=int, default=16, help="MC-dropout samples N (MC_dropout_num, model_to_trace.py:202)")
    ap.add_argument("--dropout", type=float, default=0.05, help="dropout rate (trace_model.py:16)")
    ap.add_argument("--error-map", action="store_true", help='the "_showError" twin (trace_model.py:40,46)')
    a = ap.parse_args()
    if a.check:
        rows, mode = weight_range_report(_load_any(a.checkpoint))
        for key, m, ok in rows:
            print(f"{key:62s} max |w| = {m:10.6f}  {'ok' if ok else 'BEYOND the fp16-plane bound (< %.2f)' % F16X2_WEIGHT_BOUND}")
        print(f"largest: {max(m for _k, m, _o in rows):.6f}; bound {F16X2_WEIGHT_BOUND}; hnet_create (default precision) will run: {mode}")
        raise SystemExit(0)
    if a.out is None:
        ap.error("OUT is required unless --check")
    var = None if a.variant is None else dict(variant=a.variant, mc_samples=a.mc, dropout_p=a.dropout, emit_error_map=a.error_map)
    convert_checkpoint(a.checkpoint, a.out, var)
    print(f"wrote {a.out}" + ("" if var is None else f" ({var})"))
