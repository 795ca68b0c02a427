#!/usr/bin/env python3
"""Traces our restatement of the HomographyNet forward (oracle/torch_cpu.py) into a TorchScript file for the libtorch C++ harness
(TEST / BASELINE INFRASTRUCTURE — not the product; BASELINE.json config 1 and the `cpu_baseline` leg of bench.py).

The reference runs `torch::jit::load(traced_model.pt)` + `module.forward({img1, img2[, prior]})` (cuahn_ros/homography_network/src/
HomographyNet.cpp:89,183-186).  Its .pt file embeds the reference's Python source and its trained weights are not shipped, so neither
can travel to the GPU box; what is traced here is OUR functional restatement (same ATen operators, pinned to the reference's golden
vectors by tests/test_oracle_golden.py) with the weights of an HNETW001 state dict.  Differences from the reference's traced module,
both forced by reproducibility: the four dropout keep-masks are inputs (the reference draws them from PyTorch's global generator,
model_to_trace.py:222-235), and the error map is left out (HomographyNet.cpp:189-201 only displays it).

  python oracle/libtorch/trace_restatement.py --variant full --mc 32 --out oracle/_build/restatement_full_n32.pt
Module signature: forward(img1 f32[224,320], img2 f32[224,320], prior f32[8], keep_mean_in f32[N,5120], keep_mean_hid f32[N,256],
keep_unc_in f32[N,5120], keep_unc_hid f32[N,256]) -> (mean f32[8], cov f32[8,8], H_part1 f32[3,3]); keep masks already hold 0 or 1/(1-p).
"""
from __future__ import annotations

import argparse
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import torch_cpu  # noqa: E402


class Restatement(torch.nn.Module):
    def __init__(self, state, variant: str):
        super().__init__()
        self.net = torch_cpu.TorchCpuNet(state)
        self.variant = variant

    def heads(self, feat, km_in, km_hid, ku_in, ku_hid):
        w = self.net.w
        lb = "model_last_block_list.0"
        out = []
        for head, k_in, k_hid in (("fc_block_4_mean", km_in, km_hid), ("fc_block_4_uncertainty", ku_in, ku_hid)):
            x = feat.repeat(k_in.shape[0], 1) * k_in
            h = F.leaky_relu(F.linear(x, w[f"{lb}.{head}.1.weight"], w[f"{lb}.{head}.1.bias"]), 0.1)
            out.append(F.linear(h * k_hid, w[f"{lb}.{head}.4.weight"], w[f"{lb}.{head}.4.bias"]).reshape(-1, 4, 2))
        return out[0], out[1] * 1e-3

    def forward(self, img1, img2, prior, km_in, km_hid, ku_in, ku_hid):
        n = self.net
        if self.variant == "full":
            h = n.block_h(1, img1, img2)
            todo = (2, 3)
        else:
            h = n.dlt(n.p4 + prior.reshape(4, 2))
            todo = {"prior3": (2, 3), "prior2": (3,), "prior1": ()}[self.variant]
        for b in todo:
            h = h @ n.block_h(b, img1, n.warp(img2, h))
        feat = n.trunk(4, img1, n.warp(img2, h))
        m, lv = self.heads(feat, km_in, km_hid, ku_in, ku_hid)
        var = torch.exp(lv)
        mbar = m.mean(dim=0)
        ens = ((mbar - m) ** 2).mean(dim=0) + var.mean(dim=0)
        pbar = n.p4 + mbar
        q = h @ torch.cat([pbar, torch.ones(4, 1)], dim=1).T
        quv = (q[:2] / q[2]).T
        cov = torch.zeros(8, 8)
        for i in range(4):
            g = (h / q[2, i])[:2, :2]
            cov[2 * i:2 * i + 2, 2 * i:2 * i + 2] = g @ torch.diag(ens[i]) @ g.T
        return (quv - n.p4).reshape(8), cov, h


def example_inputs(n_mc: int):
    return (torch.rand(224, 320), torch.rand(224, 320), torch.zeros(8), torch.ones(n_mc, 5120), torch.ones(n_mc, 256),
            torch.ones(n_mc, 5120), torch.ones(n_mc, 256))


def trace(state, variant: str, n_mc: int, out_path: str):
    mod = Restatement(state, variant).eval()
    with torch.no_grad():
        ts = torch.jit.trace(mod, example_inputs(n_mc), check_trace=False)
    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    ts.save(out_path)
    return out_path


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variant", default="full", choices=["full", "prior3", "prior2", "prior1"])
    ap.add_argument("--mc", type=int, default=32)
    ap.add_argument("--weights", default="", help="HNETW001 blob or .pth.tar; default: the synthetic seed-0 state")
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    from cuahn_vio_amd import weights
    if not a.weights:
        state = weights.synthetic_state(0)
    elif a.weights.endswith((".pth.tar", ".pth", ".pt")):
        state = weights.load_checkpoint(a.weights)
    else:
        state = weights.load_blob(a.weights)
    print(trace(state, a.variant, a.mc, a.out))


if __name__ == "__main__":
    main()
