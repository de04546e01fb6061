#!/usr/bin/env python3
"""Diagnostics for the NGPBRDF kernels: the HIP encoding against the restatement bit for bit (per level), and where the forward pass differs."""
import os, sys
import numpy as np
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import ngp_torch as ng
from iris_amd.model.brdf import NGPBRDF
from iris_amd import _lib as L

def main():
    dev = torch.device("cuda:0")
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
    g = torch.Generator().manual_seed(4)
    params = (torch.rand(ng.n_params(), generator=g) * 2 - 1) * 0.3
    net = NGPBRDF(-2.0, 2.5); net.load_state_dict({"mlp.params": params})
    g = torch.Generator().manual_seed(n)
    pos = torch.rand(n, 3, generator=g) * 4.5 - 2.0
    pd = pos.to(dev)
    feat = torch.zeros(32, n, dtype=torch.int32, device=dev)
    L.check(L.lib().iris_debug_ngp_encode(net._handle(dev), L.ptr(pd), n, L.ptr(feat), L.stream()))
    torch.cuda.synchronize()
    hip = feat.cpu().numpy().view(np.float16).reshape(32, n, 2)
    x = ((pos - np.float32(-2.0)) / np.float32(4.5)) * 2 - 1
    ref = ng.encode(params, x).numpy().reshape(n, 32, 2).transpose(1, 0, 2)
    for l in range(32):
        bad = (hip[l].view(np.uint16) != ref[l].view(np.uint16)).any(1)
        if bad.any():
            i = np.nonzero(bad)[0][:4]
            print("level", l, "mismatching points", int(bad.sum()), "first", i.tolist(), hip[l][i].tolist(), ref[l][i].tolist())
    print("encode compared")
    out = net(pd); torch.cuda.synchronize()
    r = ng.forward(params, pos, -2.0, 2.5)
    for k in ("albedo", "roughness", "metallic"):
        d = (out[k].cpu() - r[k]).abs().reshape(n, -1).max(1).values
        worst = torch.argsort(d, descending=True)[:8]
        print(k, "max %.3e mean %.3e" % (float(d.max()), float(d.mean())), "points > 1e-3:", int((d > 1e-3).sum()), "worst idx", worst.tolist(), "(mod 32:", (worst % 32).tolist(), ")")

if __name__ == "__main__":
    main()
