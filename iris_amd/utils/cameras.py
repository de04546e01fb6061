"""Camera lists (intrinsics + camera-to-world) of the train views for the bake CLI.

Only the pose / intrinsics part of the reference's dataset classes is needed by bake_shading (images are never read there):
  synthetic : {scene}/train/transforms.json, focal = 0.5 w / tan(0.5 camera_angle_x)          (utils/dataset/synthetic_ldr.py:129-150)
  real      : {scene}/cam.txt (origin, lookat, up per view) + K_list.txt, train split          (utils/dataset/real_ldr.py:25-34,85-166)
  scannetpp : {dataset_root}/data/{scene}/psdf/train_test_lists.json + transforms_all.json (nerfstudio layout: one shared
              fl_x fl_y cx cy h w, per frame an OpenGL camera-to-world), train split                 (utils/dataset/scannetpp/dataset.py:78-141)
  generic   : a JSON file {"img_hw":[H,W], "views":[{"K":3x3, "c2w":3x4}, ...]} (OpenCV convention)
"""
import json
import os

import numpy as np


def _img_hw_from_exr(path, res_scale):
    from .exr import read_exr_header
    h = read_exr_header(path)
    return int(h["height"] * res_scale), int(h["width"] * res_scale)


def load_synthetic(scene_dir, res_scale=1.0, img_hw=None):
    root = os.path.join(scene_dir, "train")
    with open(os.path.join(root, "transforms.json")) as fh:
        meta = json.load(fh)
    if img_hw is None:
        img_hw = _img_hw_from_exr(os.path.join(scene_dir, "train", "Image", "000_0001.exr"), res_scale)
    h, w = img_hw
    focal = float(0.5 * w / np.tan(0.5 * meta["camera_angle_x"]))
    views = [{"kind": "synthetic", "focal": focal, "c2w": np.asarray(f["transform_matrix"], np.float32)[:3, :4]} for f in meta["frames"]]
    return img_hw, views


def _read_cam_params(path):
    with open(path) as fh:
        rows = fh.read().splitlines()
    n = int(rows[0])
    a = np.array([r.split() for r in rows[1:1 + 3 * n]], dtype=np.float32)
    return np.split(a, n, axis=0)


def load_real(scene_dir, res_scale=1.0, img_hw=None, split="train"):
    if img_hw is None:
        img_hw = _img_hw_from_exr(os.path.join(scene_dir, "Image", "000_0001.exr"), res_scale)
    views = []
    cams = _read_cam_params(os.path.join(scene_dir, "cam.txt"))
    Ks = _read_cam_params(os.path.join(scene_dir, "K_list.txt"))
    val_ids = {i * 10 for i in range(16)}                              # get_split_ids, real_ldr.py:85-91
    for i, (cam, K) in enumerate(zip(cams, Ks)):
        if (i in val_ids) == (split == "train"):
            continue
        origin, lookat, up = cam[0], cam[1], cam[2]
        at = (lookat - origin) / np.linalg.norm(lookat - origin)
        R = np.stack((np.cross(-up, at), -up, at), -1).astype(np.float32)   # OpenGL (origin, lookat, up) -> OpenCV c2w
        K = K.copy(); K[:2, :] *= res_scale
        views.append({"kind": "real", "K": K.astype(np.float32), "c2w": np.hstack((R, origin.reshape(3, 1))).astype(np.float32)})
    return img_hw, views


def load_scannetpp(dataset_root, scene_id, res_scale=1.0, split="train"):
    """The pose list of the reference's ``Scannetpp(dataset_root, scene_id, split, pixel=False, res_scale)`` (utils/dataset/scannetpp/dataset.py:78-141;
    built at bake_shading.py:68, refine_shading.py:73, slf_bake.py:62): image size ``(int(h*s), int(w*s))``; ONE intrinsic matrix for every view, its first
    two rows scaled by ``s`` (formed in double, rounded to f32 once, as ``torch.tensor(Ks).float()``); per frame of ``transforms_all.json`` whose file name
    is in the split's list, the 4x4 ``transform_matrix`` with its y and z camera axes negated (OpenGL -> OpenCV) and the top three rows kept; views ordered
    by the position of their name in the split's list (frames not in it are skipped; a listed name without a frame has no view).  The images under
    ``psdf/images`` are never opened: the bake does not use them."""
    root = os.path.join(dataset_root, "data", scene_id, "psdf")
    with open(os.path.join(root, "train_test_lists.json")) as fh:
        lists = json.load(fh)
    names = lists["train"] if split == "train" else lists["test"] if split == "test" else lists["train"] + lists["test"]
    with open(os.path.join(root, "transforms_all.json")) as fh:
        meta = json.load(fh)
    img_hw = (int(meta["h"] * res_scale), int(meta["w"] * res_scale))
    K = np.array([[meta["fl_x"], 0, meta["cx"]], [0, meta["fl_y"], meta["cy"]], [0, 0, 1]], dtype=np.float64)
    K[:2] *= res_scale
    K = K.astype(np.float32)
    first = {}
    for i, n in enumerate(names):                                      # list.index(): the FIRST position of a name
        first.setdefault(n, i)
    found = []
    for frame in meta["frames"]:
        name = frame["file_path"].split("/")[-1]
        if name not in first:
            continue
        c2w = np.array(frame["transform_matrix"], dtype=np.float64)
        c2w[:3, 1:3] *= -1
        found.append((first[name], c2w[:3].astype(np.float32)))
    order = np.argsort(np.array([i for i, _ in found], dtype=np.int64)) if found else []   # the reference's np.argsort(ids)
    views = [{"kind": "real", "K": K.copy(), "c2w": found[j][1], "name": names[found[j][0]]} for j in order]
    return img_hw, views


def load_generic(json_path, res_scale=1.0):
    with open(json_path) as fh:
        meta = json.load(fh)
    h, w = meta["img_hw"]
    views = []
    for v in meta["views"]:
        K = np.asarray(v["K"], np.float32).reshape(3, 3).copy(); K[:2, :] *= res_scale
        views.append({"kind": "real", "K": K, "c2w": np.asarray(v["c2w"], np.float32).reshape(-1)[:12].reshape(3, 4)})
    return (int(h * res_scale), int(w * res_scale)), views


def view_rays(view, img_hw, device):
    """Pixel-centre rays of one view on the GPU: (H*W,3) origins and unit directions."""
    from .dataset import real_ldr, synthetic_ldr
    if view["kind"] == "synthetic":
        return synthetic_ldr.get_rays(synthetic_ldr.get_ray_directions(img_hw[0], img_hw[1], view["focal"]), view["c2w"], device=device)
    return real_ldr.to_world(real_ldr.get_direction(view["K"], img_hw), view["c2w"], False, device=device)
