"""The refine_shading driver (reference: refine_shading.py:99-177): per view the deterministic first hit, path_tracing_det_diff at spp 128 /
indir_depth 5, path_tracing_det_spec for the six roughness levels at spp 64, every map denoised, the bake's 13 file names.  The integrators
themselves are pinned to the reference's goldens in tests/test_refine.py; here the loop around them:
  test_refine_view_vs_reference_replay  refine_view against the reference's two loops replayed in its own Python (tests/golden/refine_loop.npz)
  test_refine_view_is_the_reference_loop  ragged multi-batch runs against the loop written out over this package's integrators (regression)
  test_refine_cli_*                       the command line on a directory the bake CLI populated"""
import json
import os

import numpy as np
import pytest
import torch

from conftest import golden
from stub_material import StubMaterial
from test_pt_single import _gpu_setup

pytestmark = pytest.mark.gpu


def test_refine_view_vs_reference_replay(tmp_path, oracle_mod):
    """refine_shading.py:109-127 / :144-174 replayed through the reference's path_tracing_det_diff / _spec (tools/make_driver_goldens.py), every
    torch.rand draw recorded per call; refine_view on the same rays with the same draws"""
    from iris_amd import refine_shading as rs
    from conftest import rel_l2
    dev = torch.device("cuda:0")
    f = golden("refine_loop.npz")
    _, _, sc, em = _gpu_setup(tmp_path, dev)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    uniforms = {l: [[T(f[f"u_{l}_{c}_{k}"]) for k in range(int(f[f"n_u_{l}_{c}"]))] for c in range(int(f[f"n_calls_{l}"]))] for l in range(7)}
    out = rs.refine_view(sc, em, StubMaterial(), T(f["rays_x"]), T(f["rays_d"]), int(f["spp_diffuse"]), int(f["spp_specular"]), int(f["indir_depth"]), uniforms=uniforms)
    assert out["n_valid"] == int(f["valid"].sum())
    assert rel_l2(out["diffuse"].cpu().numpy(), f["diffuse"]) <= 1e-4
    for r in range(6):
        assert rel_l2(out["specular0"][r].cpu().numpy(), f[f"specular0_{r}"]) <= 1e-4, r
        assert rel_l2(out["specular1"][r].cpu().numpy(), f[f"specular1_{r}"]) <= 1e-4, r


def test_refine_view_is_the_reference_loop(tmp_path):
    from iris_amd import refine_shading as rs
    from iris_amd.utils.dataset import real_ldr
    from iris_amd.utils.path_tracing import path_tracing_det_diff, path_tracing_det_spec, ray_intersect
    dev = torch.device("cuda:0")
    g, _, sc, em = _gpu_setup(tmp_path, dev)
    H, W = int(g["H"]), int(g["W"])
    xs, ds = real_ldr.to_world(real_ldr.get_direction(g["K"], (H, W)), g["c2w"], False, device=dev)
    mat = StubMaterial()
    spp_d, spp_s, depth = 8, 4, 3
    torch.manual_seed(5); torch.cuda.manual_seed(5)
    out = rs.refine_view(sc, em, mat, xs, ds, spp_d, spp_s, depth, batch_rays=300 * spp_d)       # several ragged batches per lobe
    # the loop of refine_shading.py:109-127 / :144-174 written out, same draw order
    torch.manual_seed(5); torch.cuda.manual_seed(5)
    pos, nrm, uv, tri, valid = ray_intersect(sc, xs, ds)
    B = pos.shape[0]
    Ld = torch.zeros(B, 3, device=dev)
    bs = 300 * spp_d // spp_d
    for b0 in range(0, B, bs):
        b1 = min(b0 + bs, B)
        Ld[b0:b1] = path_tracing_det_diff(sc, em, mat, pos[b0:b1], ds[b0:b1], nrm[b0:b1], uv[b0:b1], tri[b0:b1], spp_d, depth)
    assert torch.equal(out["diffuse"], Ld)
    bs = 300 * spp_d // spp_s
    for r_idx, rough in enumerate(torch.linspace(0.02, 1.0, 6)):
        L0 = torch.zeros(B, 3, device=dev); L1 = torch.zeros(B, 3, device=dev)
        for b0 in range(0, B, bs):
            b1 = min(b0 + bs, B)
            L0[b0:b1], L1[b0:b1] = path_tracing_det_spec(sc, em, mat, rough, pos[b0:b1], ds[b0:b1], nrm[b0:b1], uv[b0:b1], tri[b0:b1], spp_s, depth)
        assert torch.equal(out["specular0"][r_idx], L0) and torch.equal(out["specular1"][r_idx], L1)
    assert out["n_valid"] == int(valid.sum()) and torch.isfinite(Ld).all() and float(Ld.sum()) > 0
    assert float(out["diffuse"][~valid].abs().sum()) == 0.0                                  # misses stay zero


def test_refine_cli_writes_the_bake_file_set(tmp_path):
    from iris_amd import bake_shading as bs, refine_shading as rs
    from iris_amd.utils import exr
    from iris_amd.model.slf import VoxelSLF
    g = golden("bake_box.npz")
    p = golden("pt_single.npz")
    scene_dir = tmp_path / "scene"; scene_dir.mkdir()
    with open(scene_dir / "scene.obj", "w") as fh:
        for v in g["verts"]:
            fh.write("v {} {} {}\n".format(*v))
        for f in g["faces"]:
            fh.write("f {} {} {}\n".format(*(f + 1)))
    H, W = 12, 16
    K = np.array([[0.8 * W, 0, W / 2.0], [0, 0.8 * W, H / 2.0], [0, 0, 1]], np.float32)
    cams = {"img_hw": [H, W], "views": [{"K": K.tolist(), "c2w": g["c2w"].tolist()}, {"K": K.tolist(), "c2w": g["c2w"].tolist()}]}
    json.dump(cams, open(tmp_path / "cams.json", "w"))
    slf = VoxelSLF(torch.from_numpy(g["slf_mask"]), float(g["voxel_min"]), float(g["voxel_max"]))
    slf.radiance[:] = torch.from_numpy(g["slf_radiance"])
    ep, sp = str(tmp_path / "emitter.pth"), str(tmp_path / "vslf.npz")
    torch.save({"is_emitter": torch.from_numpy(g["is_emitter"]), "emitter_vertices": torch.from_numpy(p["emitter_vertices"]), "emitter_area": torch.from_numpy(g["emitter_area"]),
                "emitter_normal": torch.zeros(int(g["is_emitter"].sum()), 3), "emitter_radiance": torch.from_numpy(g["emitter_radiance"])}, ep)
    torch.save({"mask": torch.from_numpy(g["slf_mask"]), "voxel_min": float(g["voxel_min"]), "voxel_max": float(g["voxel_max"]), "weight": slf.state_dict()}, sp)
    out = str(tmp_path / "out")
    argv = ["--scene", str(scene_dir), "--slf_path", sp, "--emitter_path", ep, "--output", out, "--dataset", "generic", "--cameras", str(tmp_path / "cams.json"),
            "--material", "stub_material:material", "--spp_diffuse", "8", "--spp_specular", "4", "--indir_depth", "2", "--seed", "2"]
    # the reference's pipeline runs the bake and the refinement on the SAME --output (README; refine_shading.py:126,172-173 overwrite the bake's
    # files in place): the directory is populated by the bake CLI first, and the refine CLI must replace every file
    bs.main(["--scene", str(scene_dir), "--slf_path", sp, "--emitter_path", ep, "--output", out, "--dataset", "generic", "--cameras", str(tmp_path / "cams.json"),
             "--spp_diffuse", "8", "--spps_specular", "4", "4", "4", "4", "4", "4", "--seed", "2"])
    baked = {f: exr.read_exr(f) for im_id in (0, 1) for f in bs.output_files(out, im_id)}
    assert len(baked) == 26
    rs.main(argv)
    for im_id in (0, 1):
        files = bs.output_files(out, im_id)
        assert len(files) == 13 and all(os.path.exists(f) for f in files)
        img = exr.read_exr(files[0])
        assert img.shape == (H, W, 3) and np.isfinite(img).all() and float(img.sum()) > 0
        for f in (files[0], files[5], files[12]):          # multi-bounce maps, not the single-bounce bake
            assert not np.array_equal(exr.read_exr(f), baked[f]), f
    a, b = exr.read_exr(bs.output_files(out, 0)[0]), exr.read_exr(bs.output_files(out, 1)[0])
    assert not np.array_equal(a, b)                        # same camera, per-view seeds: independent noise
    f0 = bs.output_files(out, 0)[0]
    mt = os.stat(f0).st_mtime_ns
    rs.main(argv + ["--resume"])                           # same settings, completed views: nothing is re-rendered
    assert os.stat(f0).st_mtime_ns == mt
    rs.main(argv + ["--resume", "--spp_diffuse", "4"])     # other settings: the marker does not match, the view is rendered again
    assert os.stat(f0).st_mtime_ns != mt
    mt = os.stat(f0).st_mtime_ns
    rs.main(argv)                                          # default: overwrite, as the reference does
    assert os.stat(f0).st_mtime_ns != mt
    np.testing.assert_array_equal(exr.read_exr(f0), a)     # (and reproducibly: per-view seeds)
    # what a marker must NOT survive (advisor, round 3): a re-bake of the view, a changed input file under an unchanged path
    mt = os.stat(f0).st_mtime_ns
    rs.main(argv + ["--resume"])
    assert os.stat(f0).st_mtime_ns == mt                   # (markers of the run above are valid)
    bs.main(["--scene", str(scene_dir), "--slf_path", sp, "--emitter_path", ep, "--output", out, "--dataset", "generic", "--cameras", str(tmp_path / "cams.json"),
             "--spp_diffuse", "8", "--spps_specular", "4", "4", "4", "4", "4", "4", "--seed", "2", "--overwrite"])
    assert not os.path.exists(os.path.join(out, "diffuse", "000.refined"))       # the bake took the marker with it
    np.testing.assert_array_equal(exr.read_exr(f0), baked[f0])
    rs.main(argv + ["--resume"])                           # ... so the re-baked view is refined again
    np.testing.assert_array_equal(exr.read_exr(f0), a)
    mt = os.stat(f0).st_mtime_ns
    st = os.stat(ep)
    os.utime(ep, ns=(st.st_atime_ns, st.st_mtime_ns + 10_000_000_000))          # the emitter file "changed" (same path, newer)
    rs.main(argv + ["--resume"])
    assert os.stat(f0).st_mtime_ns != mt
    import json as _json
    m = _json.load(open(os.path.join(out, "diffuse", "000.refined")))
    assert len(m["files"]) == 13 and m["run_key"]


def test_refine_cli_runs_from_a_checkpoint(tmp_path):
    """`python -m iris_amd.refine_shading --ckpt last.ckpt` WITHOUT --material: the material network is the reference's NGPBRDF, built from the SLF file's
    (voxel_min, voxel_max) and the checkpoint's 'material.' weights exactly as refine_shading.py:82-92 does.  (Random parameters: no checkpoint of the reference
    exists on this machine.)  The CLI's maps equal refine_view() called with the same network."""
    from iris_amd import refine_shading as rs
    from iris_amd import _lib as L
    from iris_amd.model.brdf import load_ngpbrdf
    from iris_amd.model.emitter import SLFEmitter
    from iris_amd.model.slf import VoxelSLF
    from iris_amd.utils import cameras, exr
    from iris_amd.utils.path_tracing import load_scene
    from iris_amd import bake_shading as bs
    g = golden("bake_box.npz")
    p = golden("pt_single.npz")
    scene_dir = tmp_path / "scene"; scene_dir.mkdir()
    with open(scene_dir / "scene.obj", "w") as fh:
        for v in g["verts"]:
            fh.write("v {} {} {}\n".format(*v))
        for f in g["faces"]:
            fh.write("f {} {} {}\n".format(*(f + 1)))
    H, W = 8, 12
    K = np.array([[0.8 * W, 0, W / 2.0], [0, 0.8 * W, H / 2.0], [0, 0, 1]], np.float32)
    json.dump({"img_hw": [H, W], "views": [{"K": K.tolist(), "c2w": g["c2w"].tolist()}]}, open(tmp_path / "cams.json", "w"))
    slf = VoxelSLF(torch.from_numpy(g["slf_mask"]), float(g["voxel_min"]), float(g["voxel_max"]))
    slf.radiance[:] = torch.from_numpy(g["slf_radiance"])
    ep, sp = str(tmp_path / "emitter.pth"), str(tmp_path / "vslf.npz")
    torch.save({"is_emitter": torch.from_numpy(g["is_emitter"]), "emitter_vertices": torch.from_numpy(p["emitter_vertices"]), "emitter_area": torch.from_numpy(g["emitter_area"]),
                "emitter_normal": torch.zeros(int(g["is_emitter"].sum()), 3), "emitter_radiance": torch.from_numpy(g["emitter_radiance"])}, ep)
    torch.save({"mask": torch.from_numpy(g["slf_mask"]), "voxel_min": float(g["voxel_min"]), "voxel_max": float(g["voxel_max"]), "weight": slf.state_dict()}, sp)
    gen = torch.Generator().manual_seed(9)
    params = (torch.rand(int(L.lib().iris_ngp_n_params()), generator=gen) * 2 - 1) * 0.3
    ckpt = str(tmp_path / "last.ckpt")
    torch.save({"state_dict": {"material.mlp.params": params, "emitter.radiance": torch.zeros(1, 3)}}, ckpt)       # the reference's checkpoint layout
    out = str(tmp_path / "out")
    argv = ["--scene", str(scene_dir), "--slf_path", sp, "--emitter_path", ep, "--output", out, "--dataset", "generic", "--cameras", str(tmp_path / "cams.json"),
            "--ckpt", ckpt, "--spp_diffuse", "8", "--spp_specular", "4", "--indir_depth", "2", "--seed", "2", "--denoise", "none", "--compression", "none"]
    rs.main(argv)
    files = bs.output_files(out, 0)
    assert len(files) == 13 and all(os.path.exists(f) for f in files)
    # the same view through the library call with the same network and the CLI's per-view seed
    dev = torch.device("cuda:0")
    net = load_ngpbrdf(float(g["voxel_min"]), float(g["voxel_max"]), ckpt)
    img_hw, views = cameras.load_generic(str(tmp_path / "cams.json"), 1.0)
    sc = load_scene(str(scene_dir / "scene.obj"), device=dev)
    em = SLFEmitter(ep, sp)
    torch.manual_seed(2 * 1000003 + 0); torch.cuda.manual_seed(2 * 1000003 + 0)
    xs, ds = cameras.view_rays(views[0], img_hw, dev)
    ref = rs.refine_view(sc, em, net, xs, ds, 8, 4, 2)
    got = exr.read_exr(files[0])
    assert got.shape == (H, W, 3) and np.isfinite(got).all() and float(got.sum()) > 0
    np.testing.assert_array_equal(got, ref["diffuse"].reshape(H, W, 3).cpu().numpy())
    with pytest.raises(L.IrisError):
        rs.main([a for a in argv if a not in ("--ckpt", ckpt)])          # neither --ckpt nor --material


def test_refine_cli_scannetpp_runs_the_reference_command_line(tmp_path):
    """scripts/scannetpp/bathroom2/train.sh:97-103's argument list, unchanged: `--dataset_root R --scene S --dataset scannetpp --res_scale s --slf_path ... --emitter_path ...
    --ckpt ... --output ...` -- the mesh from R/data/S/scans/scene.ply, the train views from R/data/S/psdf (refine_shading.py:52-77), the material network from the checkpoint --
    over the 13 files a bake of the same command line wrote (the reference runs both scripts on one --output)."""
    from iris_amd import refine_shading as rs
    from iris_amd import bake_shading as bs
    from iris_amd import _lib as L
    from iris_amd.model.slf import VoxelSLF
    from iris_amd.utils import cameras, exr
    from test_exr_cli import _scannetpp_tree
    g = _scannetpp_tree(tmp_path, "45b0dac5e3")
    b = golden("bake_box.npz"); p = golden("pt_single.npz")
    verts = (b["verts"] - b["verts"].mean(0)) * 4.0
    os.makedirs(str(tmp_path / "data" / "45b0dac5e3" / "scans"))
    with open(str(tmp_path / "data" / "45b0dac5e3" / "scans" / "scene.ply"), "w") as fh:
        fh.write("ply\nformat ascii 1.0\nelement vertex {}\nproperty float x\nproperty float y\nproperty float z\nelement face {}\n"
                 "property list uchar int vertex_indices\nend_header\n".format(len(verts), len(b["faces"])))
        for v in verts:
            fh.write("{!r} {!r} {!r}\n".format(*(float(c) for c in v)))
        for f in b["faces"]:
            fh.write("3 {} {} {}\n".format(*f))
    mask = np.ones((8, 8, 8), bool)
    slf = VoxelSLF(torch.from_numpy(mask), -9.0, 9.0)
    slf.radiance[:] = torch.linspace(0.1, 1.0, slf.radiance.numel()).reshape(slf.radiance.shape)
    K = int(b["is_emitter"].sum())
    ev = torch.from_numpy((p["emitter_vertices"] - b["verts"].mean(0)) * 4.0).float()
    ep, sp = str(tmp_path / "emitter.pth"), str(tmp_path / "vslf.npz")
    torch.save({"is_emitter": torch.from_numpy(b["is_emitter"]), "emitter_vertices": ev, "emitter_area": torch.from_numpy(b["emitter_area"]) * 16,
                "emitter_normal": torch.zeros(K, 3), "emitter_radiance": torch.from_numpy(b["emitter_radiance"])}, ep)
    torch.save({"mask": torch.from_numpy(mask), "voxel_min": -9.0, "voxel_max": 9.0, "weight": slf.state_dict()}, sp)
    gen = torch.Generator().manual_seed(5)
    ckpt = str(tmp_path / "last_0.ckpt")
    torch.save({"state_dict": {"material.mlp.params": (torch.rand(int(L.lib().iris_ngp_n_params()), generator=gen) * 2 - 1) * 0.3, "emitter.radiance": torch.zeros(1, 3)}}, ckpt)
    out = str(tmp_path / "outputs" / "shading")
    s = float(g["rays_res_scale"])
    common = ["--dataset_root", str(tmp_path), "--scene", "45b0dac5e3", "--dataset", "scannetpp", "--res_scale", str(s), "--slf_path", sp, "--emitter_path", ep]
    bs.main(common + ["--output", out, "--spp_diffuse", "4", "--spps_specular", "2", "2", "2", "2", "2", "2", "--denoise", "none", "--compression", "none"])     # train.sh:49-54
    before = exr.read_exr(bs.output_files(out, 1)[0]).copy()
    rs.main(common + ["--ckpt", ckpt, "--output", out, "--spp_diffuse", "4", "--spp_specular", "2", "--indir_depth", "2", "--denoise", "none", "--compression", "none"])   # train.sh:97-103
    img_hw, views = cameras.load_scannetpp(str(tmp_path), "45b0dac5e3", s)
    assert len(views) == 5
    for im_id in range(5):
        files = bs.output_files(out, im_id)
        assert all(os.path.exists(f) for f in files) and os.path.exists(os.path.join(out, "diffuse", "{:03d}.refined".format(im_id)))
        a = exr.read_exr(files[0])
        assert a.shape == (img_hw[0], img_hw[1], 3) and np.isfinite(a).all()
    assert not np.array_equal(exr.read_exr(bs.output_files(out, 1)[0]), before)          # the refined maps replaced the bake's, in place
