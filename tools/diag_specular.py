"""Diagnostic (GPU box): HIP sample_specular vs golden, by hemisphere of wo."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from iris_amd.model.brdf import BaseBRDF
g = np.load("tests/golden/sample_specular.npz")
dev = torch.device("cuda:0")
T = lambda a: torch.from_numpy(a).to(dev)
nov = (g["wo"] * g["normal"]).sum(-1)
for r_idx in range(6):
    out = BaseBRDF().sample_specular(T(g["u2"]), T(g["wo"]), T(g["normal"]), torch.tensor(g["roughness"][r_idx]))
    wi, pdf, g0, g1 = [t.cpu().numpy() for t in out]
    for nm, a in (("g0", g0), ("g1", g1)):
        b = g[f"{nm}_{r_idx}"]
        for lab, m in (("all", np.ones_like(nov, bool)), ("up", nov > 0.02)):
            e = np.abs(a - b)[m]; bb = b[m]
            worst = np.argmax(e)
            print(r_idx, nm, lab, "relL2 %.2e" % (np.linalg.norm(e) / np.linalg.norm(bb)), "band-ok %.4f" % (e <= 1e-4 * np.abs(bb) + 1e-5).mean(),
                  "worst abs %.3e at val %.3e nov %.3e" % (e[worst], bb[worst], nov[m][worst]))
    print(r_idx, "wi maxabs %.2e" % np.abs(wi - g[f"wi_{r_idx}"]).max())
