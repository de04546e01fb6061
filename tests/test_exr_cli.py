"""EXR shading-cache files and the bake_shading CLI contract (reference: bake_shading.py:29-39,131,202-203)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import golden


def test_exr_roundtrip(tmp_path):
    from iris_amd.utils import exr
    rng = np.random.default_rng(0)
    img = (rng.random((37, 53, 3)) * 7).astype(np.float32)
    img[0, 0] = [0.0, 1e-30, 65504.0]
    for comp in ("none", "zips", "zip"):
        p = str(tmp_path / f"a_{comp}.exr")
        exr.write_exr(p, img, comp)
        h = exr.read_exr_header(p)
        assert [c for c, _ in h["channels"]] == ["B", "G", "R"] and h["height"] == 37 and h["width"] == 53
        np.testing.assert_array_equal(exr.read_exr(p), img)        # float32 EXR is lossless
    raw = open(str(tmp_path / "a_none.exr"), "rb").read()
    assert raw[:4] == bytes([0x76, 0x2F, 0x31, 0x01])


def test_output_file_names():
    from iris_amd import bake_shading as bs
    f = bs.output_files("out", 7)
    assert f[0] == os.path.join("out", "diffuse", "007.exr")
    assert f[1] == os.path.join("out", "specular", "007_0_0.exr") and f[2] == os.path.join("out", "specular", "007_1_0.exr")
    assert f[-1] == os.path.join("out", "specular", "007_1_5.exr") and len(f) == 13


def test_real_camera_loader(tmp_path):
    """cam.txt (origin, lookat, up) -> OpenCV c2w exactly as utils/dataset/real_ldr.py:139-152."""
    from iris_amd.utils import cameras
    n = 12
    rows, ks = [str(n)], [str(n)]
    for i in range(n):
        o = np.array([i * 0.1, 0.2, 0.3]); at = np.array([0.0, 1.0, 0.0]); up = np.array([0.0, 0.0, 1.0])
        rows += [" ".join(map(str, o)), " ".join(map(str, o + at)), " ".join(map(str, up))]
        ks += ["100 0 32", "0 100 24", "0 0 1"]
    (tmp_path / "cam.txt").write_text("\n".join(rows)); (tmp_path / "K_list.txt").write_text("\n".join(ks))
    hw, views = cameras.load_real(str(tmp_path), img_hw=(48, 64))
    assert hw == (48, 64) and len(views) == n - 2            # views 0 and 10 are the validation split
    c2w = views[0]["c2w"]
    np.testing.assert_allclose(c2w[:, 2], [0, 1, 0], atol=1e-6)      # z = viewing direction
    np.testing.assert_allclose(c2w[:, 1], [0, 0, -1], atol=1e-6)     # y = down
    np.testing.assert_allclose(c2w[:, 3], [0.1, 0.2, 0.3], atol=1e-6)


def _scannetpp_tree(root, scene="sc0"):
    """The psdf/ tree of the golden (the two JSON texts the reference's class was given), written as the reference's layout."""
    g = golden("scannetpp_cameras.npz")
    d = os.path.join(str(root), "data", scene, "psdf")
    os.makedirs(d, exist_ok=True)
    open(os.path.join(d, "train_test_lists.json"), "w").write(str(g["lists_json"]))
    open(os.path.join(d, "transforms_all.json"), "w").write(str(g["transforms_json"]))
    return g


def test_scannetpp_camera_loader_equals_the_reference_class(tmp_path):
    """cameras.load_scannetpp == the reference's Scannetpp(root, scene, split='train', pixel=False, res_scale) -- img_hw, Ks, C2Ws bit for bit, view order
    and count included (utils/dataset/scannetpp/dataset.py:78-141; golden from the imported class, tools/make_scannetpp_golden.py)."""
    from iris_amd.utils import cameras
    g = _scannetpp_tree(tmp_path)
    for tag in ("half", "full", "third"):
        hw, views = cameras.load_scannetpp(str(tmp_path), "sc0", float(g["res_scale_" + tag]))
        assert hw == tuple(int(v) for v in g["img_hw_" + tag])
        assert len(views) == len(g["C2Ws_" + tag]) == 5                              # six listed names, one without a frame
        assert [v["name"] for v in views] == ["DSC00012.JPG", "DSC00003.JPG", "DSC00040.JPG", "DSC00007.JPG", "DSC00025.JPG"]   # the LIST's order
        for i, v in enumerate(views):
            assert v["K"].dtype == np.float32 and v["c2w"].dtype == np.float32 and v["c2w"].shape == (3, 4)
            np.testing.assert_array_equal(v["K"], g["Ks_" + tag][i])
            np.testing.assert_array_equal(v["c2w"], g["C2Ws_" + tag][i])
    hw, views = cameras.load_scannetpp(str(tmp_path), "sc0", 1.0, split="test")
    assert [v["name"] for v in views] == ["DSC00005.JPG", "DSC00018.JPG"]
    hw, views = cameras.load_scannetpp(str(tmp_path), "sc0", 1.0, split="all")
    assert len(views) == 7


@pytest.mark.gpu
def test_cli_scannetpp_dataset_runs_the_reference_command_line(tmp_path):
    """scripts/scannetpp/bathroom2/train.sh:49-54's argument list, unchanged (no --cameras, no --img_hw): --dataset_root R --scene S --dataset scannetpp
    --res_scale s --slf_path ... --emitter_path ... --output ...; mesh at R/data/S/scans/scene.ply, cameras from R/data/S/psdf (bake_shading.py:49-69)."""
    from iris_amd import bake_shading as bs
    from iris_amd.utils import exr, cameras
    from iris_amd.model.emitter import SLFEmitter
    from iris_amd.model.slf import VoxelSLF
    from iris_amd.utils.path_tracing import load_scene
    g = _scannetpp_tree(tmp_path, "45b0dac5e3")
    b = golden("bake_box.npz")
    # the golden's cameras sit around the origin at ~2 m: a box room around them (the bake_box room, scaled and centred)
    verts = (b["verts"] - b["verts"].mean(0)) * 4.0
    os.makedirs(str(tmp_path / "data" / "45b0dac5e3" / "scans"))
    ply = str(tmp_path / "data" / "45b0dac5e3" / "scans" / "scene.ply")
    with open(ply, "w") as fh:
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
    ep, sp = str(tmp_path / "emitter.pth"), str(tmp_path / "vslf.npz")
    torch.save({"is_emitter": torch.from_numpy(b["is_emitter"]), "emitter_vertices": torch.zeros(K, 3, 3), "emitter_area": torch.from_numpy(b["emitter_area"]) * 16,
                "emitter_normal": torch.zeros(K, 3), "emitter_radiance": torch.from_numpy(b["emitter_radiance"])}, ep)
    torch.save({"mask": torch.from_numpy(mask), "voxel_min": -9.0, "voxel_max": 9.0, "weight": slf.state_dict()}, sp)
    out = str(tmp_path / "outputs" / "shading")
    s = float(g["rays_res_scale"])
    argv = ["--dataset_root", str(tmp_path), "--scene", "45b0dac5e3", "--dataset", "scannetpp", "--res_scale", str(s),
            "--slf_path", sp, "--emitter_path", ep, "--output", out]
    # the rays of a view == the reference's get_direction / to_world on the class's K and c2w
    dev = torch.device("cuda:0")
    img_hw, views = cameras.load_scannetpp(str(tmp_path), "45b0dac5e3", s)
    assert img_hw == tuple(int(v) for v in g["rays_img_hw"])
    xs, ds = cameras.view_rays(views[int(g["rays_view"])], img_hw, dev)
    np.testing.assert_allclose(torch.cat([xs, ds], -1).cpu().numpy(), g["rays"], rtol=0, atol=2e-6)
    bs.main(argv + ["--spp_diffuse", "8", "--spps_specular", "4", "4", "4", "4", "4", "4", "--denoise", "none"])   # (the reference's 960 spp per pixel are not needed to test the contract)
    for im_id in range(5):
        files = bs.output_files(out, im_id)
        assert all(os.path.exists(f) for f in files)
        assert exr.read_exr(files[0]).shape == (img_hw[0], img_hw[1], 3)
    assert not os.path.exists(bs.output_files(out, 5)[0])
    xs, ds = cameras.view_rays(views[2], img_hw, dev)
    ref = bs.bake_view(load_scene(ply, device=dev), SLFEmitter(ep, sp), xs, ds, 8, [4] * 6, seed=bs.view_seed(0, 2), image_width=img_hw[1])
    f2 = bs.output_files(out, 2)
    np.testing.assert_array_equal(exr.read_exr(f2[0]), ref["diffuse"].reshape(*img_hw, 3).cpu().numpy())
    np.testing.assert_array_equal(exr.read_exr(f2[12]), ref["specular1"][5].reshape(*img_hw, 3).cpu().numpy())
    assert float(ref["diffuse"].sum()) > 0


@pytest.mark.gpu
def test_cli_synthetic_dataset(tmp_path):
    """Tiny FIPT-style scene on disk -> CLI -> 13 EXR files per view == bake_view()."""
    from iris_amd import bake_shading as bs
    from iris_amd.utils import exr, cameras
    from iris_amd.model.emitter import SLFEmitter
    from iris_amd.model.slf import VoxelSLF
    from iris_amd.utils.path_tracing import load_scene
    g = golden("bake_box.npz")
    scene_dir = tmp_path / "scene"; (scene_dir / "train" / "Image").mkdir(parents=True)
    with open(scene_dir / "scene.obj", "w") as fh:
        for v in g["verts"]:
            fh.write("v {} {} {}\n".format(*v))
        for f in g["faces"]:
            fh.write("f {} {} {}\n".format(*(f + 1)))
    H, W = 20, 28
    exr.write_exr(str(scene_dir / "train" / "Image" / "000_0001.exr"), np.zeros((H, W, 3), np.float32))
    c2w = np.eye(4); c2w[:3, :4] = g["c2w"]; c2w[:3, 1:3] *= -1          # OpenCV -> the synthetic (x left, y up) convention is not
    frames = [{"transform_matrix": c2w.tolist()}, {"transform_matrix": (c2w + np.eye(4) * 0).tolist()}]   # needed: any valid pose works
    json.dump({"camera_angle_x": 1.2, "frames": frames}, open(scene_dir / "train" / "transforms.json", "w"))
    slf = VoxelSLF(torch.from_numpy(g["slf_mask"]), float(g["voxel_min"]), float(g["voxel_max"]))
    slf.radiance[:] = torch.from_numpy(g["slf_radiance"])
    K = int(g["is_emitter"].sum())
    ep, sp = str(tmp_path / "emitter.pth"), str(tmp_path / "vslf.npz")
    torch.save({"is_emitter": torch.from_numpy(g["is_emitter"]), "emitter_vertices": torch.zeros(K, 3, 3), "emitter_area": torch.from_numpy(g["emitter_area"]),
                "emitter_normal": torch.zeros(K, 3), "emitter_radiance": torch.from_numpy(g["emitter_radiance"])}, ep)
    torch.save({"mask": torch.from_numpy(g["slf_mask"]), "voxel_min": float(g["voxel_min"]), "voxel_max": float(g["voxel_max"]), "weight": slf.state_dict()}, sp)
    out = str(tmp_path / "out")
    argv = ["--scene", str(scene_dir), "--slf_path", sp, "--emitter_path", ep, "--output", out, "--dataset", "synthetic",
            "--spp_diffuse", "16", "--spps_specular", "8", "8", "8", "8", "8", "8", "--seed", "3"]
    bs.main(argv)
    files = bs.output_files(out, 1)
    assert all(os.path.exists(f) for f in files)
    # same view through the library API
    dev = torch.device("cuda:0")
    img_hw, views = cameras.load_synthetic(str(scene_dir))
    assert img_hw == (H, W)
    xs, ds = cameras.view_rays(views[1], img_hw, dev)
    from iris_amd.utils.denoise import Denoiser
    ref = bs.bake_view(load_scene(str(scene_dir / "scene.obj"), device=dev), SLFEmitter(ep, sp), xs, ds, 16, [8] * 6, seed=bs.view_seed(3, 1), image_width=W,
                       denoiser=Denoiser((W, H), dev))                  # the CLI denoises by default, as the reference does (:129, :198-200)
    np.testing.assert_array_equal(exr.read_exr(files[0]), ref["diffuse"].reshape(H, W, 3).cpu().numpy())
    np.testing.assert_array_equal(exr.read_exr(files[1 + 2 * 4]), ref["specular0"][4].reshape(H, W, 3).cpu().numpy())
    np.testing.assert_array_equal(exr.read_exr(files[2 + 2 * 5]), ref["specular1"][5].reshape(H, W, 3).cpu().numpy())
    assert float(ref["diffuse"].sum()) > 0
    # per-view Philox keys: the noise of two views of one run is independent (view 0 keeps the plain seed)
    assert bs.view_seed(3, 0) == 3 and bs.view_seed(3, 1) != bs.view_seed(3, 2) != 3
    mt = os.path.getmtime(files[0])
    bs.main(argv)                                                        # resume: nothing is re-baked
    assert os.path.getmtime(files[0]) == mt
    # --denoise none writes the raw Monte-Carlo maps; level 0 is never denoised (:198)
    out2 = str(tmp_path / "out_raw")
    bs.main([a if a != out else out2 for a in argv] + ["--denoise", "none"])
    raw = bs.bake_view(load_scene(str(scene_dir / "scene.obj"), device=dev), SLFEmitter(ep, sp), xs, ds, 16, [8] * 6, seed=bs.view_seed(3, 1), image_width=W)
    f2 = bs.output_files(out2, 1)
    np.testing.assert_array_equal(exr.read_exr(f2[0]), raw["diffuse"].reshape(H, W, 3).cpu().numpy())
    np.testing.assert_array_equal(exr.read_exr(f2[1]), exr.read_exr(files[1]))
    assert not np.array_equal(exr.read_exr(f2[0]), exr.read_exr(files[0]))
