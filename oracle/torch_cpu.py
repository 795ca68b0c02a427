"""libtorch-CPU restatement of the HomographyNet forward (TEST / BASELINE INFRASTRUCTURE — not the product).

north_star asks for "the libtorch CPU path timed on the same box's host cores" beside the GPU number.  The
reference's TorchScript file embeds its Python source and cannot travel to the GPU box, and the trained checkpoint
is not shipped; this module is our own functional restatement of the same computation on the same libtorch CPU
operators the traced model executes (aten conv2d / leaky_relu / avg_pool2d / grid_sampler / linalg inverse /
addmm), fed from the HNETW001 weight dict.  It is pinned in tests/test_oracle_golden.py against the golden
vectors the reference model produced (agreement ~1e-5 px: same operators, same order).

Only tests/ and bench.py's `cpu_baseline` leg import this file.  Reference lines followed:
  trunk / block control flow  trace_pytorch_model/model_to_trace.py:124-193
  conv + LeakyReLU(0.1)       model_to_trace.py:7-15
  DLT                         model_to_trace.py:42-61
  warp                        warp.py:60-79
  block 4, heads, ensemble    model_to_trace.py:241-282
  transfer + outputs          model_to_trace.py:18-38, :299-330
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from cuahn_vio_amd import mcdrop
from cuahn_vio_amd.weights import CONV_LAYERS

ROWS, COLS = 224, 320
_BLOCKS = {  # block -> (state_dict prefix, conv layer names, avg-pool factor, fc tensor)
    1: ("model_part1", ["block_1_1", "block_1_2", "block_1_3"], 8, "fc_block_1"),
    2: ("model_part1", ["block_2_1", "block_2_2", "block_2_3", "block_2_4"], 4, "fc_block_2"),
    3: ("model_part1", ["block_3_0", "block_3_1", "block_3_2", "block_3_3", "block_3_4", "block_3_5"], 2, "fc_block_3"),
    4: ("model_last_block_list.0", ["block_4_0", "block_4_1", "block_4_2", "block_4_3", "block_4_4", "block_4_5", "block_4_6"], 1, None),
}
_STRIDE = {n: s for n, _ci, _co, _k, s in CONV_LAYERS}


class TorchCpuNet:
    def __init__(self, state, dtype=torch.float32):
        self.dt = dtype
        self.w = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dtype) for k, v in state.items()}
        self.p4 = torch.tensor([[0.0, 0.0], [0.0, ROWS - 1.0], [COLS - 1.0, ROWS - 1.0], [COLS - 1.0, 0.0]], dtype=dtype)
        v, u = torch.meshgrid(torch.arange(ROWS, dtype=dtype), torch.arange(COLS, dtype=dtype), indexing="ij")
        self.uv1 = torch.stack([u.reshape(-1), v.reshape(-1), torch.ones(ROWS * COLS, dtype=dtype)])     # [3, 71680]
        self.norm = torch.tensor([2.0 / (COLS - 1.0), 2.0 / (ROWS - 1.0)], dtype=dtype)

    # ---- geometry
    def dlt(self, dst):
        """4-point homography p4 -> dst by explicit inverse of the 8x8 system, as the reference does"""
        a = torch.zeros(8, 8, dtype=self.dt)
        x, y = self.p4[:, 0], self.p4[:, 1]
        a[0::2, 0], a[0::2, 1], a[0::2, 2] = x, y, 1.0
        a[1::2, 3], a[1::2, 4], a[1::2, 5] = x, y, 1.0
        a[0::2, 6], a[0::2, 7] = -dst[:, 0] * x, -dst[:, 0] * y
        a[1::2, 6], a[1::2, 7] = -dst[:, 1] * x, -dst[:, 1] * y
        h8 = torch.inverse(a) @ dst.reshape(8, 1)
        return torch.cat([h8.reshape(8), torch.ones(1, dtype=self.dt)]).reshape(3, 3)

    def warp(self, img, h):
        """img [224,320]; bilinear sample at H (u,v,1), zeros outside, align_corners=True"""
        xyz = h @ self.uv1
        g = torch.stack([xyz[0] / xyz[2], xyz[1] / xyz[2]], dim=-1) * self.norm - 1.0
        return F.grid_sample(img.reshape(1, 1, ROWS, COLS), g.reshape(1, ROWS, COLS, 2), mode="bilinear",
                             padding_mode="zeros", align_corners=True).reshape(ROWS, COLS)

    # ---- network pieces
    def trunk(self, block, img1, img2w, hook=None):
        prefix, names, pool, _fc = _BLOCKS[block]
        x = torch.stack([img1, img2w]).unsqueeze(0)
        if pool > 1:
            x = F.avg_pool2d(x, pool)
        for n in names:
            wt = self.w[f"{prefix}.{n}.0.weight"]
            x = F.leaky_relu(F.conv2d(x, wt, self.w[f"{prefix}.{n}.0.bias"], stride=_STRIDE[n], padding=(wt.shape[-1] - 1) // 2), 0.1)
            if hook:
                hook(n, x)
        return x.reshape(1, -1)                                      # NCHW flatten: c*20 + pixel

    def block_h(self, block, img1, img2w, hook=None):
        prefix, _n, _p, fc = _BLOCKS[block]
        off = F.linear(self.trunk(block, img1, img2w, hook), self.w[f"{prefix}.{fc}.weight"], self.w[f"{prefix}.{fc}.bias"])
        if hook:
            hook(fc, off)
        return self.dlt(self.p4 + off.reshape(4, 2))

    def heads(self, feat, n_mc, p, mc_seed, pair_seq):
        """per-sample corner means [N,4,2] and log-variances [N,4,2]; masks from include/hnet_rng.h"""
        lb = "model_last_block_list.0"
        x = feat.repeat(n_mc, 1)
        out = []
        for head, s_in, s_hid in (("fc_block_4_mean", mcdrop.STREAM_MEAN_IN, mcdrop.STREAM_MEAN_HID),
                                  ("fc_block_4_uncertainty", mcdrop.STREAM_UNC_IN, mcdrop.STREAM_UNC_HID)):
            def drop(t, stream):
                if p <= 0.0:
                    return t
                keep = mcdrop.keep_mask(mc_seed, pair_seq, stream, n_mc, t.shape[1], p)
                return t * (torch.from_numpy(keep).to(self.dt) * float(mcdrop.scale(p)))
            h = F.leaky_relu(F.linear(drop(x, s_in), self.w[f"{lb}.{head}.1.weight"], self.w[f"{lb}.{head}.1.bias"]), 0.1)
            out.append(F.linear(drop(h, s_hid), self.w[f"{lb}.{head}.4.weight"], self.w[f"{lb}.{head}.4.bias"]).reshape(n_mc, 4, 2))
        return out[0], out[1] * 1e-3

    @torch.no_grad()
    def forward(self, img1, img2, prior=None, blocks_to_run=3, n_mc=16, p=0.0, mc_seed=0, pair_seq=0, want_err=False, hook=None):
        i1 = torch.from_numpy(_as_f32(img1)).to(self.dt)
        i2 = torch.from_numpy(_as_f32(img2)).to(self.dt)
        if prior is not None:
            h = self.dlt(self.p4 + torch.as_tensor(np.asarray(prior, np.float32).reshape(4, 2)).to(self.dt))
            todo = {3: (2, 3), 2: (3,), 1: ()}[blocks_to_run]
        else:
            h = self.block_h(1, i1, i2, hook)
            todo = (2, 3)
        for b in todo:
            h = h @ self.block_h(b, i1, self.warp(i2, h), hook)
        feat = self.trunk(4, i1, self.warp(i2, h), hook)
        m, lv = self.heads(feat, n_mc, p, mc_seed, pair_seq)
        var = torch.exp(lv)
        mbar = m.mean(dim=0)
        ens = ((mbar - m) ** 2).mean(dim=0) + var.mean(dim=0)         # two-pass population variance + mean aleatoric
        pbar = self.p4 + mbar
        q = h @ torch.cat([pbar, torch.ones(4, 1, dtype=self.dt)], dim=1).T          # [3,4]
        quv = (q[:2] / q[2]).T
        cov = torch.zeros(8, 8, dtype=self.dt)
        for i in range(4):
            g = (h / q[2, i])[:2, :2]
            cov[2 * i:2 * i + 2, 2 * i:2 * i + 2] = g @ torch.diag(ens[i]) @ g.T
        res = {"mean": (quv - self.p4).reshape(8).numpy().astype(np.float32), "cov": cov.numpy().astype(np.float32),
               "H_part1": h.numpy().astype(np.float32)}
        if want_err:
            res["err"] = ((self.warp(i2, h @ self.dlt(pbar)) - i1).abs() * 255.0).numpy().astype(np.float32)
        return res


def _as_f32(img):
    img = np.asarray(img)
    if img.dtype == np.uint8:
        return (img.astype(np.float32) / np.float32(255.0)).reshape(ROWS, COLS)
    return np.ascontiguousarray(img, dtype=np.float32).reshape(ROWS, COLS)
