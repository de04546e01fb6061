import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import ngp_torch as ng
from iris_amd.model.brdf import NGPBRDF
from iris_amd import _lib as L
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(5)
params = (torch.rand(ng.n_params(), generator=g) * 2 - 1) * 0.3
net = NGPBRDF(-2.0, 2.5); net.load_state_dict({"mlp.params": params})
g = torch.Generator().manual_seed(11)
n = 3000
pos = torch.rand(n, 3, generator=g) * 4.5 - 2.0
pos[0] = torch.tensor([-2.0, 2.5, 0.25]); pos[1] = torch.tensor([2.5, -2.0, -2.0])
feat = torch.zeros(32, n, dtype=torch.int32, device=dev)
L.check(L.lib().iris_debug_ngp_encode(net._handle(dev), L.ptr(pos.to(dev)), n, L.ptr(feat), L.stream()))
torch.cuda.synchronize()
hip = feat.cpu().numpy().view(np.uint16).reshape(32, n, 2)
x = ((pos - np.float32(-2.0)) / np.float32(4.5)) * np.float32(2.0) - np.float32(1.0)
ref = ng.encode(params, x).numpy().view(np.uint16).reshape(n, 32, 2).transpose(1, 0, 2)
bad = np.argwhere(hip != ref)
rows, _ = ng.level_tables()
for l, i, f in bad[:40]:
    xs = x[i].numpy()
    p = [float(np.float32(np.float64(np.float32(rows[l][0])) * np.float64(xs[d]) + 0.5)) for d in range(3)]
    print("level", l, "point", i, "feat", f, "hip", hip[l, i, f], "ref", ref[l, i, f], "x", xs.tolist(), "p", p)
