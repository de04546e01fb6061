#!/usr/bin/env python3
"""tests/golden/scannetpp_cameras.npz: the pose list of the reference's OWN ``Scannetpp`` dataset class on a small synthetic ``psdf/`` tree.

Runs only in the build container (imports /root/reference with the stub modules of tools/make_goldens.py).  The tree is the layout
``utils/dataset/scannetpp/dataset.py:78-141`` reads -- ``data/<scene>/psdf/train_test_lists.json`` and ``transforms_all.json`` -- with the cases that
decide the order and the count of the views: frames stored out of list order, frames of the test split and frames in no list (skipped), a listed name
without a frame (no view), integer entries in a matrix, ``res_scale`` 0.5 on odd image sizes.  Stored: the two JSON texts (the INPUT), and what the class
built from them with ``split='train', pixel=False`` as bake_shading.py:68 does -- ``img_hw``, ``Ks``, ``C2Ws`` -- plus the rays of one view through the
reference's ``get_direction`` / ``to_world`` (what ``__getitem__`` returns as ``batch['rays']`` besides the image it also opens).
"""
import json
import os
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from tools.make_goldens import OUT, REF, _stub_modules   # noqa: E402


def tree(rng):
    def pose(i):
        a, b = 0.7 * i + 0.1, 0.3 * i - 0.2
        Rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
        Rx = np.array([[1, 0, 0], [0, np.cos(b), -np.sin(b)], [0, np.sin(b), np.cos(b)]])
        m = np.eye(4); m[:3, :3] = Rz @ Rx; m[:3, 3] = rng.normal(size=3) * 2
        return m
    train = ["DSC0%04d.JPG" % i for i in (12, 3, 40, 7, 25, 31)]           # the list's order is the view order, not the name order
    test = ["DSC0%04d.JPG" % i for i in (5, 18)]
    frames = []
    for n in ("DSC00025.JPG", "DSC00005.JPG", "DSC00003.JPG", "DSC00099.JPG", "DSC00040.JPG", "DSC00012.JPG", "DSC00018.JPG", "DSC00007.JPG"):
        frames.append({"file_path": "images/" + n, "transform_matrix": pose(len(frames)).tolist()})        # DSC00031 has no frame; DSC00099 is in no list
    ident = [[1, 0, 0, 2], [0, 1, 0, -1], [0, 0, 1, 3], [0, 0, 0, 1]]        # integers in the file
    frames[2]["transform_matrix"] = ident
    meta = {"fl_x": 1163.4453, "fl_y": 1164.6601, "cx": 875.5, "cy": 583.25, "h": 1169, "w": 1753, "camera_model": "PINHOLE", "frames": frames}
    return {"train": train, "test": test}, meta


def main():
    import torch
    _stub_modules()
    sys.path.insert(0, REF)
    os.chdir(REF)
    from utils.dataset.scannetpp.dataset import Scannetpp
    from utils.dataset.real_ldr import get_direction, to_world
    rng = np.random.default_rng(7)
    lists, meta = tree(rng)
    out = {"lists_json": np.array(json.dumps(lists)), "transforms_json": np.array(json.dumps(meta))}
    with tempfile.TemporaryDirectory() as root:
        d = os.path.join(root, "data", "sc0", "psdf")
        os.makedirs(d)
        json.dump(lists, open(os.path.join(d, "train_test_lists.json"), "w"))
        json.dump(meta, open(os.path.join(d, "transforms_all.json"), "w"))
        for tag, s in (("half", 0.5), ("full", 1.0), ("third", 1.0 / 3.0)):
            ds = Scannetpp(root, "sc0", split="train", pixel=False, res_scale=s)
            out["res_scale_" + tag] = np.float64(s)
            out["img_hw_" + tag] = np.array(ds.img_hw, np.int64)
            out["Ks_" + tag] = ds.Ks.numpy()
            out["C2Ws_" + tag] = ds.C2Ws.numpy()
            assert len(ds) == len(ds.C2Ws) == 5
        ds = Scannetpp(root, "sc0", split="train", pixel=False, res_scale=1.0 / 16)
        k, c2w = ds.Ks[3], ds.C2Ws[3]
        xs, dirs = to_world(get_direction(k, ds.img_hw), c2w, False, k)
        out["rays_res_scale"] = np.float64(1.0 / 16); out["rays_view"] = np.int64(3); out["rays_img_hw"] = np.array(ds.img_hw, np.int64)
        out["rays"] = torch.cat([xs, dirs], -1).numpy()
    np.savez_compressed(os.path.join(OUT, "scannetpp_cameras.npz"), **out)
    print("scannetpp_cameras.npz", os.path.getsize(os.path.join(OUT, "scannetpp_cameras.npz")), out["img_hw_half"], out["rays"].shape)


if __name__ == "__main__":
    main()
