#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING THE REFERENCE'S OWN PYTHON.

Runs only in the build container (needs /root/reference); its outputs (small .npz files of
inputs + expected outputs) are committed and are what travels to the GPU box.

How the reference is made importable here: stub modules for the packages
that are absent (mitsuba, tinycudann, cv2, kornia, torchvision, torch_interpolations), CWD =
/root/reference.  Everything below then calls the reference's unmodified functions on torch-CPU.

The one thing the reference cannot do here is intersect rays (utils/path_tracing.py:30-43 is
Mitsuba/OptiX).  For the end-to-end fixture (bake_box.npz) `utils.path_tracing.ray_intersect` is
monkey-patched with the oracle's brute-force closest-hit (oracle/iris_oracle.c, "parity unpinned"
for that one function); every other step of the replayed loop body is reference code.
"""
import math
import os
import sys
import tempfile
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)


def _stub_modules():
    mi = types.ModuleType("mitsuba")
    mi.set_variant = lambda *a, **k: None
    mi.math = types.SimpleNamespace(RayEpsilon=1500 * 2.0 ** -24)
    sys.modules["mitsuba"] = mi
    sys.modules["tinycudann"] = types.ModuleType("tinycudann")
    sys.modules["cv2"] = types.ModuleType("cv2")
    k = types.ModuleType("kornia"); k.create_meshgrid = None
    sys.modules["kornia"] = k
    sys.modules["torch_interpolations"] = types.ModuleType("torch_interpolations")
    tv = types.ModuleType("torchvision"); tvt = types.ModuleType("torchvision.transforms")
    tv.transforms = tvt
    sys.modules["torchvision"] = tv; sys.modules["torchvision.transforms"] = tvt


def box_room():
    """Closed 4 x 3 x 2.6 m box (12 triangles) + a ceiling light quad (2 triangles) = 14 triangles."""
    X, Y, Z = 4.0, 3.0, 2.6
    v = [(0, 0, 0), (X, 0, 0), (X, Y, 0), (0, Y, 0), (0, 0, Z), (X, 0, Z), (X, Y, Z), (0, Y, Z),
         (1.5, 1.0, 2.55), (2.5, 1.0, 2.55), (2.5, 2.0, 2.55), (1.5, 2.0, 2.55)]
    f = [(0, 1, 2), (0, 2, 3), (4, 6, 5), (4, 7, 6), (0, 5, 1), (0, 4, 5), (3, 2, 6), (3, 6, 7),
         (0, 3, 7), (0, 7, 4), (1, 5, 6), (1, 6, 2), (8, 9, 10), (8, 10, 11)]
    return np.asarray(v, np.float32), np.asarray(f, np.int32)


def smooth_radiance(c):
    """Smooth positive radiance field evaluated at voxel centres c (K,3) -> (K,3)."""
    x, y, z = c[:, 0], c[:, 1], c[:, 2]
    r = 0.25 + 0.2 * np.sin(2 * np.pi * x / 1.3) * np.cos(2 * np.pi * y / 1.7)
    g = 0.25 + 0.2 * np.sin(2 * np.pi * y / 1.3 + 1.0) * np.cos(2 * np.pi * z / 1.7)
    b = 0.25 + 0.2 * np.sin(2 * np.pi * z / 1.3 + 2.0) * np.cos(2 * np.pi * x / 1.7)
    return np.stack([r, g, b], -1).astype(np.float32)


def surface_mask(verts, faces, H, vmin, vmax, n_per_edge=96):
    """Occupancy of voxels touched by the mesh surface (dense barycentric point sampling)."""
    mask = np.zeros((H, H, H), bool)
    a = np.linspace(0, 1, n_per_edge, dtype=np.float64)
    u, v = np.meshgrid(a, a, indexing="ij")
    keep = (u + v) <= 1.0
    u, v = u[keep], v[keep]
    for f in faces:
        p0, p1, p2 = verts[f[0]].astype(np.float64), verts[f[1]].astype(np.float64), verts[f[2]].astype(np.float64)
        pts = p0[None] * (1 - u - v)[:, None] + p1[None] * u[:, None] + p2[None] * v[:, None]
        q = np.clip(((pts - vmin) / (vmax - vmin) * H).astype(np.int64), 0, H - 1)
        mask[q[:, 2], q[:, 1], q[:, 0]] = True
    return mask


def main():
    import torch
    import torch.nn.functional as NF
    _stub_modules()
    sys.path.insert(0, REF)
    os.chdir(REF)
    os.makedirs(OUT, exist_ok=True)

    from utils import ops as rops
    from model.brdf import BaseBRDF
    from model.slf import VoxelSLF
    from model.emitter import SLFEmitter
    import utils.path_tracing as rpt
    from utils.dataset import real_ldr, synthetic_ldr
    import oracle

    torch.manual_seed(0)
    rng = np.random.default_rng(0)

    # ------------------------------------------------------------------ a1 ray generation
    def rand_rot():
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        w, x, y, z = q
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                         [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                         [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    H, W = 5, 7
    K = torch.tensor([[6.3, 0, 3.4], [0, 6.1, 2.6], [0, 0, 1]], dtype=torch.float32)
    c2w = torch.tensor(np.concatenate([rand_rot(), rng.normal(size=(3, 1))], 1), dtype=torch.float32)
    d_cam = real_ldr.get_direction(K, (H, W))
    o0, d0 = real_ldr.to_world(d_cam, c2w, False, K)
    o1, d1, dx1, dy1 = real_ldr.to_world(d_cam, c2w, True, K)
    np.savez(os.path.join(OUT, "raygen_real.npz"), K=K.numpy(), c2w=c2w.numpy(), H=H, W=W,
             rays_o=o0.numpy(), rays_d=d0.numpy(), rays_o_diff=o1.numpy(), rays_d_diff=d1.numpy(),
             dxdu=dx1.numpy(), dydv=dy1.numpy())
    focal = 5.7
    dirs = synthetic_ldr.get_ray_directions(H, W, focal)
    so0, sd0 = synthetic_ldr.get_rays(dirs, c2w)
    so1, sd1, sdx, sdy = synthetic_ldr.get_rays(dirs, c2w, focal=focal)
    np.savez(os.path.join(OUT, "raygen_syn.npz"), focal=np.float32(focal), c2w=c2w.numpy(), H=H, W=W,
             rays_o=so0.numpy(), rays_d=sd0.numpy(), rays_o_diff=so1.numpy(), rays_d_diff=sd1.numpy(),
             dxdu=sdx.numpy(), dydv=sdy.numpy())

    # ------------------------------------------------------------------ frame / angle2xyz / double_sided
    def unit(n):
        return (n / np.linalg.norm(n, axis=-1, keepdims=True)).astype(np.float32)
    normals = unit(rng.normal(size=(4096, 3)))
    axes = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], np.float32)
    edge = []
    for s in (+1, -1):
        for e in (-1e-7, 0.0, 1e-7):
            nx = np.float32(s * 0.1 + e)
            r = np.sqrt(1 - float(nx) ** 2)
            edge.append([nx, r * 0.6, r * 0.8])
    normals = np.concatenate([normals, axes, np.asarray(edge, np.float32)], 0)
    frames = rops.get_normal_space(torch.from_numpy(normals)).numpy()
    theta = torch.rand(512) * math.pi / 2
    phi = torch.rand(512) * 2 * math.pi
    xyz = rops.angle2xyz(theta, phi).numpy()
    V = unit(rng.normal(size=(512, 3))); N = unit(rng.normal(size=(512, 3)))
    Nf = rops.double_sided(torch.from_numpy(V), torch.from_numpy(N.copy())).numpy()
    np.savez(os.path.join(OUT, "frame.npz"), normal=normals, frames=frames, theta=theta.numpy(), phi=phi.numpy(), xyz=xyz,
             V=V, N=N, N_flipped=Nf)

    # ------------------------------------------------------------------ a3 sample_diffuse
    brdf = BaseBRDF()
    B = normals.shape[0]
    u2 = torch.rand(B, 2)
    one_m = float(1.0 - 2.0 ** -24)
    u2[:8] = torch.tensor([[0, 0], [0, one_m], [one_m, 0], [one_m, one_m], [0.5, 0.25], [0.25, 0.5], [1e-7, 0.75], [0.999, 0.999]])
    wi, pdf, w = brdf.sample_diffuse(u2, torch.from_numpy(normals))
    np.savez(os.path.join(OUT, "sample_diffuse.npz"), u2=u2.numpy(), normal=normals, wi=wi.numpy(), pdf=pdf.numpy(), weight=w.numpy())

    # ------------------------------------------------------------------ a4 sample_specular (6 roughness levels)
    rough_levels = torch.linspace(0.02, 1.0, 6)
    wo = unit(normals + 0.8 * rng.normal(size=normals.shape))       # mostly above, some below horizon
    graz = unit(np.cross(normals[:64], unit(rng.normal(size=(64, 3)))) + 1e-3 * normals[:64])  # grazing
    wo[:64] = graz
    u2s = torch.rand(B, 2)
    u2s[:8] = u2[:8]
    sp = {"u2": u2s.numpy(), "wo": wo, "normal": normals, "roughness": rough_levels.numpy()}
    for r_idx, r in enumerate(rough_levels):
        wi, pdf, g0, g1 = brdf.sample_specular(u2s, torch.from_numpy(wo), torch.from_numpy(normals), r)
        sp[f"wi_{r_idx}"] = wi.numpy(); sp[f"pdf_{r_idx}"] = pdf.numpy()
        sp[f"g0_{r_idx}"] = g0.numpy(); sp[f"g1_{r_idx}"] = g1.numpy()
    np.savez(os.path.join(OUT, "sample_specular.npz"), **sp)

    # ------------------------------------------------------------------ a10 lerp_specular
    spec = torch.rand(64, 6, 3)
    rr = torch.rand(64, 1) * 0.98 + 0.02
    rr[0] = 0.02; rr[1] = 1.0; rr[2] = 0.216; rr[3] = 0.5
    np.savez(os.path.join(OUT, "lerp_specular.npz"), specular=spec.numpy(), roughness=rr.numpy(),
             out=rops.lerp_specular(spec, rr).numpy())

    # ------------------------------------------------------------------ a5 VoxelSLF
    Hs = 32
    mask = torch.rand(Hs, Hs, Hs) < 0.3
    vmin, vmax = -0.37, 4.21
    slf = VoxelSLF(mask, vmin, vmax)
    slf.radiance[:] = torch.rand(slf.radiance.shape)
    x = torch.rand(8192, 3) * (vmax - vmin) * 1.2 + (vmin - 0.1 * (vmax - vmin))  # some out of box
    # exactly-on-boundary positions
    grid = torch.tensor([vmin + (vmax - vmin) * k / Hs for k in range(Hs + 1)], dtype=torch.float32)
    x[:Hs + 1, 0] = grid; x[Hs + 1:2 * Hs + 2, 1] = grid; x[2 * Hs + 2:3 * Hs + 3, 2] = grid
    idx = slf.spatial_idx(x)
    rgb = slf(x)["rgb"]
    np.savez(os.path.join(OUT, "slf.npz"), mask=mask.numpy(), voxel_min=vmin, voxel_max=vmax, inds=slf.inds.numpy(),
             radiance=slf.radiance.numpy(), x=x.numpy(), idx=idx.numpy(), rgb=rgb.numpy())

    # ------------------------------------------------------------------ a5 SLFEmitter.eval_emitter (reference file formats)
    tmp = tempfile.mkdtemp()
    n_face = 200
    is_emitter = torch.zeros(n_face, dtype=torch.bool)
    is_emitter[torch.randperm(n_face)[:17]] = True
    is_emitter[-1] = True  # idx=-1 wraps onto an emitter row: must still be masked by vis
    K_e = int(is_emitter.sum())
    ev = torch.rand(K_e, 3, 3)
    area = torch.cross(ev[:, 1] - ev[:, 0], ev[:, 2] - ev[:, 0], dim=-1).norm(dim=-1) / 2.0
    area[0] = 0.0  # exercises clamp_min(1e-12)
    erad = torch.zeros(n_face, 3)
    erad[:K_e] = torch.rand(K_e, 3) * 10
    emitter_path = os.path.join(tmp, "emitter.pth"); slf_path = os.path.join(tmp, "vslf.npz")
    torch.save({"is_emitter": is_emitter, "emitter_vertices": ev, "emitter_area": area,
                "emitter_normal": torch.zeros(K_e, 3), "emitter_radiance": erad}, emitter_path)
    torch.save({"mask": mask, "voxel_min": vmin, "voxel_max": vmax, "weight": slf.state_dict()}, slf_path)
    em = SLFEmitter(emitter_path, slf_path)
    Bq = 4096
    pos = torch.rand(Bq, 3) * (vmax - vmin) + vmin
    tri = torch.randint(0, n_face, (Bq,))
    tri[:256] = -1
    tri[256:512] = torch.arange(n_face)[is_emitter][torch.randint(0, K_e, (256,))]
    ldir = torch.zeros(Bq, 3)
    ones_i64 = torch.ones_like(tri)[:, None]
    Le_b, pdf_b, vn_b = em.eval_emitter(pos, ldir, tri, ones_i64, trace_roughness=0.0)      # bake arguments
    Le_n, pdf_n, vn_n = em.eval_emitter(pos, ldir, tri)                                      # roughness=None
    rough_f = torch.rand(Bq, 1)
    Le_r, pdf_r, vn_r = em.eval_emitter(pos, ldir, tri, rough_f, trace_roughness=0.6)       # default threshold
    np.savez(os.path.join(OUT, "eval_emitter.npz"), mask=mask.numpy(), voxel_min=vmin, voxel_max=vmax, inds=slf.inds.numpy(),
             slf_radiance=slf.radiance.numpy(), is_emitter=is_emitter.numpy(), emitter_area=area.numpy(),
             emitter_radiance=erad.numpy(), position=pos.numpy(), triangle_idx=tri.numpy(), roughness=rough_f.numpy(),
             Le_bake=Le_b.numpy(), pdf_bake=pdf_b.numpy(), valid_next_bake=vn_b.numpy(),
             Le_none=Le_n.numpy(), pdf_none=pdf_n.numpy(), valid_next_none=vn_n.numpy(),
             Le_rough=Le_r.numpy(), pdf_rough=pdf_r.numpy(), valid_next_rough=vn_r.numpy())

    # ------------------------------------------------------------------ end-to-end: bake loop body on the box room
    verts, faces = box_room()
    osc = oracle.Scene(verts, faces)

    def ray_intersect_patch(scene, xs, ds):
        p, n, uv, idx, valid = osc.ray_intersect(xs.numpy(), ds.numpy(), brute=True)
        return (torch.from_numpy(p), torch.from_numpy(n), torch.from_numpy(uv), torch.from_numpy(idx), torch.from_numpy(valid))
    rpt.ray_intersect = ray_intersect_patch
    ray_intersect = rpt.ray_intersect

    Hb = 64
    bvmin, bvmax = -0.2, 4.2
    bmask = surface_mask(verts, faces, Hb, bvmin, bvmax)
    bslf = VoxelSLF(torch.from_numpy(bmask), bvmin, bvmax)
    kk, jj, ii = np.where(bmask)
    centres = (np.stack([ii, jj, kk], -1) + 0.5) / Hb * (bvmax - bvmin) + bvmin
    bslf.radiance[:] = torch.from_numpy(smooth_radiance(centres))
    b_is_emitter = torch.zeros(len(faces), dtype=torch.bool); b_is_emitter[12:] = True
    bev = torch.from_numpy(verts[faces[12:]])
    barea = torch.cross(bev[:, 1] - bev[:, 0], bev[:, 2] - bev[:, 0], dim=-1).norm(dim=-1) / 2.0
    brad = torch.zeros(len(faces), 3); brad[:2] = torch.tensor([10.0, 9.0, 8.0])
    torch.save({"is_emitter": b_is_emitter, "emitter_vertices": bev, "emitter_area": barea,
                "emitter_normal": torch.zeros(2, 3), "emitter_radiance": brad}, emitter_path)
    torch.save({"mask": torch.from_numpy(bmask), "voxel_min": bvmin, "voxel_max": bvmax, "weight": bslf.state_dict()}, slf_path)
    emitter = SLFEmitter(emitter_path, slf_path)
    material_net = BaseBRDF()

    Hc = Wc = 32
    Kc = torch.tensor([[0.8 * Wc, 0, Wc / 2], [0, 0.8 * Wc, Hc / 2], [0, 0, 1]], dtype=torch.float32)
    # camera at the room centre, looking along +x, slightly rolled/tilted (OpenCV convention: z forward)
    fwd = np.array([1.0, 0.25, -0.15]); fwd /= np.linalg.norm(fwd)
    right = np.cross(fwd, [0, 0, 1.0]); right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    c2wc = torch.tensor(np.concatenate([np.stack([right, down, fwd], 1), np.array([[2.0], [1.5], [1.3]])], 1), dtype=torch.float32)
    rays_x, rays_d = real_ldr.to_world(real_ldr.get_direction(Kc, (Hc, Wc)), c2wc, False, Kc)
    import mitsuba
    torch.manual_seed(0)

    # ---- replay of bake_shading.py:98-123 (diffuse), spp reduced to 16
    spp = 16
    xs, ds = rays_x, rays_d
    positions, normals_, _, prim_idx, valid = ray_intersect(None, xs, ds)
    position = positions[valid]; normal = normals_[valid]; ds_v = ds[valid]
    Bp = ds_v.shape[0]
    Ld_ = torch.zeros(Bp, 3)
    batch_size = 10240 * 64 // spp
    u2_d = []; tri_d = []; Le_d = []
    for b in range(math.ceil(Bp * 1.0 / batch_size)):
        b0 = b * batch_size; b1 = min(b0 + batch_size, Bp)
        u = torch.rand((b1 - b0) * spp, 2); u2_d.append(u)
        wi, _, _ = material_net.sample_diffuse(u, normal[b0:b1].repeat_interleave(spp, 0))
        p_next, _, _, tri_next, valid_next = ray_intersect(None, position[b0:b1].repeat_interleave(spp, 0) + mitsuba.math.RayEpsilon * wi, wi.reshape(-1, 3))
        roughness_one = torch.ones_like(tri_next)[:, None]
        Le, _, _ = emitter.eval_emitter(p_next, wi, tri_next, roughness_one, trace_roughness=0.0)
        Ld_[b0:b1] = Le.reshape(b1 - b0, spp, 3).mean(1)
        tri_d.append(tri_next); Le_d.append(Le)
    out = {"verts": verts, "faces": faces, "K": Kc.numpy(), "c2w": c2wc.numpy(), "H": Hc, "W": Wc, "spp": spp,
           "slf_mask": bmask, "slf_inds": bslf.inds.numpy(), "slf_radiance": bslf.radiance.numpy(), "voxel_min": bvmin, "voxel_max": bvmax,
           "is_emitter": b_is_emitter.numpy(), "emitter_area": barea.numpy(), "emitter_radiance": brad.numpy(),
           "rays_o": xs.numpy(), "rays_d": ds.numpy(),
           "prim_position": positions.numpy(), "prim_normal": normals_.numpy(), "prim_idx": prim_idx.numpy(), "prim_valid": valid.numpy(),
           "u2_diffuse": torch.cat(u2_d).numpy(), "tri_next_diffuse": torch.cat(tri_d).numpy(), "Le_diffuse": torch.cat(Le_d).numpy(),
           "Ld": Ld_.numpy()}

    # ---- replay of bake_shading.py:149-188 (specular), all spps reduced to 16
    wo_v = -ds[valid]
    roughness_level = torch.linspace(0.02, 1.0, 6)
    out["roughness_level"] = roughness_level.numpy()
    for r_idx, roughness in enumerate(roughness_level):
        Ls0_ = torch.zeros(Bp, 3); Ls1_ = torch.zeros(Bp, 3)
        us = []; tris = []
        for b in range(math.ceil(Bp * 1.0 / batch_size)):
            b0 = b * batch_size; b1 = min(b0 + batch_size, Bp)
            u = torch.rand((b1 - b0) * spp, 2); us.append(u)
            wi, _, g0, g1 = material_net.sample_specular(u, wo_v[b0:b1].repeat_interleave(spp, 0), normal[b0:b1].repeat_interleave(spp, 0), roughness)
            p_next, _, _, tri_next, valid_next = ray_intersect(None, position[b0:b1].repeat_interleave(spp, 0) + mitsuba.math.RayEpsilon * wi, wi.reshape(-1, 3))
            roughness_one = torch.ones_like(tri_next)[:, None]
            Le, _, _ = emitter.eval_emitter(p_next, wi, tri_next, roughness_one, trace_roughness=0.0)
            Ls0_[b0:b1] = (Le * g0).reshape(b1 - b0, spp, 3).mean(1)
            Ls1_[b0:b1] = (Le * g1).reshape(b1 - b0, spp, 3).mean(1)
            tris.append(tri_next)
        out[f"u2_spec_{r_idx}"] = torch.cat(us).numpy(); out[f"tri_next_spec_{r_idx}"] = torch.cat(tris).numpy()
        out[f"Ls0_{r_idx}"] = Ls0_.numpy(); out[f"Ls1_{r_idx}"] = Ls1_.numpy()
    np.savez_compressed(os.path.join(OUT, "bake_box.npz"), **out)

    # ------------------------------------------------------------------ a9 (cfg 5): building blocks
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from stub_material import StubMaterial
    from model.emitter import SLFEmitterLearn
    torch.manual_seed(1)
    bev_real = torch.from_numpy(verts[faces[12:]])                      # true emitter vertices of the box-room light
    torch.save({"is_emitter": b_is_emitter, "emitter_vertices": bev_real, "emitter_area": barea,
                "emitter_normal": torch.zeros(2, 3), "emitter_radiance": brad}, emitter_path)
    em_l = SLFEmitterLearn(emitter_path, slf_path)
    Nq = 2048
    posq = torch.rand(Nq, 3) * torch.tensor([4.0, 3.0, 2.4]) + torch.tensor([0.0, 0.0, 0.05])
    s1q, s2q = torch.rand(Nq), torch.rand(Nq, 2)
    s1q[:4] = torch.tensor([0.0, 1e-13, 0.5, 0.99999994])
    wi_e, pdf_e, tri_e = em_l.sample_emitter(s1q, s2q, posq)
    nq = unit(rng.normal(size=(Nq, 3))); woq = unit(nq + 0.7 * rng.normal(size=(Nq, 3))); wiq = unit(nq + 0.7 * rng.normal(size=(Nq, 3)))
    mat_q = StubMaterial()(posq)
    brdf_q, bpdf_q = material_net.eval_brdf(torch.from_numpy(wiq), torch.from_numpy(woq), torch.from_numpy(nq), mat_q)
    s1b, s2b = torch.rand(Nq), torch.rand(Nq, 2)
    wi_s, pdf_s, w_s = material_net.sample_brdf(s1b, s2b, torch.from_numpy(woq), torch.from_numpy(nq), mat_q)
    np.savez(os.path.join(OUT, "pt_units.npz"), position=posq.numpy(), s1=s1q.numpy(), s2=s2q.numpy(), emitter_vertices=bev_real.numpy(),
             emitter_cdf=em_l.emitter_cdf.numpy(), se_wi=wi_e.numpy(), se_pdf=pdf_e.numpy(), se_tri=tri_e.numpy(),
             normal=nq, wo=woq, wi=wiq, albedo=mat_q["albedo"].numpy(), roughness=mat_q["roughness"].numpy(), metallic=mat_q["metallic"].numpy(),
             brdf=brdf_q.numpy(), brdf_pdf=bpdf_q.numpy(), sb_s1=s1b.numpy(), sb_s2=s2b.numpy(), sb_wi=wi_s.numpy(), sb_pdf=pdf_s.numpy(), sb_weight=w_s.numpy())

    # ------------------------------------------------------------------ a9 (cfg 5): path_tracing_single forward + d/d radiance
    Hp = Wp = 16
    Kp = torch.tensor([[0.8 * Wp, 0, Wp / 2], [0, 0.8 * Wp, Hp / 2], [0, 0, 1]], dtype=torch.float32)
    fwd2 = np.array([0.3, 0.2, 1.0]); fwd2 /= np.linalg.norm(fwd2)      # looking up at the ceiling light
    right2 = np.cross(fwd2, [0, 1.0, 0]); right2 /= np.linalg.norm(right2)
    down2 = np.cross(fwd2, right2)
    c2wp = torch.tensor(np.concatenate([np.stack([right2, down2, fwd2], 1), np.array([[2.0], [1.5], [0.9]])], 1), dtype=torch.float32)
    ro, rd, dxdu, dydv = real_ldr.to_world(real_ldr.get_direction(Kp, (Hp, Wp)), c2wp, True, Kp)
    spp_p = 4
    recorded = []
    real_rand = torch.rand

    def rec_rand(*a, **k):
        k.pop("device", None)
        t = real_rand(*a, **k); recorded.append(t.clone()); return t
    torch.rand = rec_rand
    try:
        torch.manual_seed(2)
        class RefStub(BaseBRDF):            # the reference's NGPBRDF is a BaseBRDF with a forward(position) (model/brdf.py:213-260)
            def forward(self, x):
                return StubMaterial()(x)
        Lp = rpt.path_tracing_single(None, em_l, RefStub(), ro, rd, dxdu, dydv, spp_p)
    finally:
        torch.rand = real_rand
    wgt = torch.rand(Lp.shape[0], 3)
    (gr,) = torch.autograd.grad((Lp * wgt).sum(), em_l.radiance)
    np.savez(os.path.join(OUT, "pt_single.npz"), rays_o=ro.numpy(), rays_d=rd.numpy(), dx_du=dxdu.numpy(), dy_dv=dydv.numpy(), spp=spp_p,
             u0=recorded[0].numpy(), u1=recorded[1].numpy(), u2=recorded[2].numpy(), u3=recorded[3].numpy(), u4=recorded[4].numpy(),
             L=Lp.detach().numpy(), grad_weight=wgt.numpy(), grad_radiance=gr.numpy(), emitter_vertices=bev_real.numpy(),
             emitter_cdf=em_l.emitter_cdf.numpy(), radiance=em_l.radiance.detach().numpy())
    print("pt_single: L mean", float(Lp.mean()), "grad nnz rows", int((gr.abs().sum(-1) > 0).sum()), "draws", [tuple(r.shape) for r in recorded])

    # ------------------------------------------------------------------ 8(f)-1: refine_shading integrators (multi-bounce, no grad)
    def record(fn):
        rec = []

        def rr(*a, **k):
            k.pop("device", None)
            t = real_rand(*a, **k); rec.append(t.clone()); return t
        torch.rand = rr
        try:
            out = fn()
        finally:
            torch.rand = real_rand
        return out, rec
    ref_mat = RefStub()
    with torch.no_grad():
        ppos, pnrm, puv, ptri, pvalid = rpt.ray_intersect(None, ro, NF.normalize(rd, dim=-1))
        rdn = NF.normalize(rd, dim=-1)
        torch.manual_seed(3)
        Li, rec_i = record(lambda: rpt.trace_indirect(None, em_l, ref_mat, ppos[pvalid], -rdn[pvalid], pnrm[pvalid], 3))
        torch.manual_seed(4)
        Ld, rec_d = record(lambda: rpt.path_tracing_det_diff(None, em_l, ref_mat, ppos, rdn, pnrm, puv, ptri, 4, 3))
        torch.manual_seed(5)
        (Ls0, Ls1), rec_s = record(lambda: rpt.path_tracing_det_spec(None, em_l, ref_mat, torch.tensor(0.412), ppos, rdn, pnrm, puv, ptri, 4, 3))
    ref = {"rays_o": ro.numpy(), "rays_d": rdn.numpy(), "position": ppos.numpy(), "normal": pnrm.numpy(), "triangle_idx": ptri.numpy(), "valid": pvalid.numpy(),
           "L_indirect": Li.numpy(), "L_det_diff": Ld.numpy(), "L_det_spec0": Ls0.numpy(), "L_det_spec1": Ls1.numpy(),
           "n_i": len(rec_i), "n_d": len(rec_d), "n_s": len(rec_s)}
    for tag, rec in (("i", rec_i), ("d", rec_d), ("s", rec_s)):
        for k, t in enumerate(rec):
            ref[f"u_{tag}_{k}"] = t.numpy()
    np.savez(os.path.join(OUT, "refine.npz"), **ref)
    print("refine: indirect mean", float(Li.mean()), "det_diff mean", float(Ld.mean()), "det_spec means", float(Ls0.mean()), float(Ls1.mean()),
          "draws", len(rec_i), len(rec_d), len(rec_s))

    # ------------------------------------------------------------------ utils/path_tracing.py:214-318: path_tracing (render.py's integrator): the first
    # bounce of path_tracing_single with trace_roughness 0.6 + trace_indirect for the continuation (no grad), forward + d/d radiance
    # (forward only, as render.py runs it under no_grad: with torch >= 2 the reference's in-place `active_next[active_next.clone()] = valid_next`
    #  (:304) invalidates the index its own `L[active_next] += ...` saved for backward)
    #  Seed: with seed 7 one of the 517 continuing paths draws a direction 3e-4 rad below a wall it sits on; whether origin + RayEpsilon * wi
    #  rounds to the wall plane or one ulp behind it (hit or miss) then depends on the last bit of that direction, which torch's and the
    #  oracle's sin / cos do not share -- and from that bounce on the recorded draws belong to another path set.  Seed 8 has no such ray.)
    torch.manual_seed(int(os.environ.get("IRIS_GOLDEN_PT_FULL_SEED", "8")))
    with torch.no_grad():
        Lf, rec_f = record(lambda: rpt.path_tracing(None, em_l, ref_mat, ro, rd, dxdu, dydv, spp_p, 3))
    def replay(fn, rec):                               # the same draws again
        it = iter(rec)
        torch.rand = lambda *a, **k: next(it).clone()
        try:
            return fn()
        finally:
            torch.rand = real_rand
    with torch.no_grad():
        Lf0 = replay(lambda: rpt.path_tracing(None, em_l, ref_mat, ro, rd, dxdu, dydv, spp_p, 0), rec_f[:5])      # first bounce alone (indir_depth 0)
    full = {"rays_o": ro.numpy(), "rays_d": rd.numpy(), "dx_du": dxdu.numpy(), "dy_dv": dydv.numpy(), "spp": spp_p, "indir_depth": 3, "n_u": len(rec_f),
            "L": Lf.detach().numpy(), "L_first_bounce": Lf0.numpy(), "radiance": em_l.radiance.detach().numpy(), "emitter_vertices": bev_real.numpy(), "emitter_cdf": em_l.emitter_cdf.numpy()}
    for k, t in enumerate(rec_f):
        full[f"u_{k}"] = t.numpy()
    np.savez(os.path.join(OUT, "pt_full.npz"), **full)
    print("pt_full: L mean", float(Lf.mean()), "draws", len(rec_f), [tuple(r.shape) for r in rec_f[:6]])

    # ------------------------------------------------------------------ 8(f)-2: VoxelSLF.scatter_add (mean-pooling builder of slf_bake.py:120-138)
    torch.manual_seed(6)
    Hq = 16
    maskq = torch.rand(Hq, Hq, Hq) < 0.5
    vq = VoxelSLF(maskq, -0.3, 2.9)
    kk, jj, ii = torch.where(maskq)
    pick = torch.randint(0, len(ii), (20000,))
    cen = (torch.stack([ii, jj, kk], -1)[pick].float() + torch.rand(20000, 3) * 0.98 + 0.01) / Hq * (2.9 + 0.3) - 0.3     # inside occupied voxels
    radq = torch.rand(20000, 3) * 3
    vq.scatter_add(cen, radq)
    np.savez(os.path.join(OUT, "slf_scatter.npz"), mask=maskq.numpy(), voxel_min=-0.3, voxel_max=2.9, x=cen.numpy(), rgb=radq.numpy(),
             radiance=vq.radiance.numpy(), count=vq.count.numpy())

    # ------------------------------------------------------------------ 8(f)-3: shading combine of the BRDF trainer
    # train_brdf_crf.py:195-203 replayed through the reference's lerp_specular on rows cut from a packed cache exactly as
    # utils/dataset/scannetpp/dataset.py:359-377 (pack) and :409-414 (slice) do; gradients by torch.autograd.
    torch.manual_seed(7)
    Bc, Rl = 512, 6
    maps = [torch.rand(Bc, 3) * 2 for _ in range(1 + 2 * Rl)]                 # diffuse, spec0[0..5], spec1[0..5] of one "view"
    all_cache = torch.cat([maps[0], torch.cat(maps[1:1 + Rl], -1), torch.cat(maps[1 + Rl:], -1)], 1)   # (B,39)
    idxc = torch.randperm(Bc)[:384]
    cache = all_cache[idxc]
    diffuse_c = cache[..., :3]
    specular0 = cache[..., 3:21].reshape(len(idxc), -1, 3)
    specular1 = cache[..., 21:39].reshape(len(idxc), -1, 3)
    albedo = torch.rand(len(idxc), 3).requires_grad_(True)
    metallic = torch.rand(len(idxc), 1).requires_grad_(True)
    rough_c = (torch.rand(len(idxc), 1) * 0.98 + 0.02)
    rough_c[0] = 0.02; rough_c[1] = 1.0; rough_c[2] = 0.216; rough_c[3] = 0.412; rough_c[4] = 0.5; rough_c[5] = 0.999999
    rough_c.requires_grad_(True)
    kd = albedo * (1 - metallic)
    ks = 0.04 * (1 - metallic) + albedo * metallic
    Ld = kd * diffuse_c
    Ls = ks * rops.lerp_specular(specular0, rough_c) + rops.lerp_specular(specular1, rough_c)
    Lc = Ld + Ls
    gLc = torch.randn(len(idxc), 3)
    ga, gm, gr = torch.autograd.grad(Lc, [albedo, metallic, rough_c], gLc)
    np.savez(os.path.join(OUT, "shade_cached.npz"), **{f"map_{i}": m.numpy() for i, m in enumerate(maps)}, all_cache=all_cache.numpy(),
             idx=idxc.numpy(), diffuse=diffuse_c.numpy(), specular0=specular0.numpy(), specular1=specular1.numpy(),
             albedo=albedo.detach().numpy(), metallic=metallic.detach().numpy(), roughness=rough_c.detach().numpy(), L=Lc.detach().numpy(),
             gL=gLc.numpy(), g_albedo=ga.numpy(), g_metallic=gm.numpy(), g_roughness=gr.numpy())

    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
