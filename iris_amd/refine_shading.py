"""The refine_shading stage (reference: refine_shading.py:99-177) on MI355X: the shading cache re-baked with multi-bounce indirect light
through the current material estimate -- per view a deterministic first hit, then `path_tracing_det_diff` (diffuse, spp 128, indir_depth 5)
and `path_tracing_det_spec` for the six roughness levels (spp 64), denoised and written to the same 13 EXR files the bake wrote.

The integrators are the HIP stages of `iris_amd.utils.path_tracing` (parity: tests/test_refine.py against the reference's goldens).  The
material network is the reference's `NGPBRDF` (model/brdf.py:213-260, a tiny-cuda-nn hash grid + MLP): `iris_amd.model.brdf.NGPBRDF` runs its
inference as HIP kernels (hash-grid gathers + the perceptron on the matrix cores) and is loaded from `--ckpt` exactly as the reference loads it
(refine_shading.py:83-92).  The driver also takes ANY `material_net(position) -> {'albedo','roughness','metallic'}` module: `refine_view(...)` as
a library call, `--material pkg.module:factory` on the command line (the factory is called with the (voxel_min, voxel_max) pair `NGPBRDF` is
constructed from, and `--ckpt` is handed to it when given).
"""
import importlib
import math
import os
import time

import torch

from . import _lib as L
from .bake_shading import N_ROUGHNESS, MapWriter, output_files, roughness_levels
from .utils.path_tracing import path_tracing_det_diff, path_tracing_det_spec, ray_intersect

SPP_DIFFUSE = 128          # refine_shading.py:103
SPP_SPECULAR = 64          # refine_shading.py:137
INDIR_DEPTH = 5            # refine_shading.py:104
REFERENCE_BATCH_RAYS = 10240 * 128   # the reference's rays per integrator call: batch_size = 10240*128//spp pixels (refine_shading.py:107, :139), sized for its GPU memory
BATCH_RAYS = 16 * REFERENCE_BATCH_RAYS   # default here: 21 M paths per call (a few GB of path state next to 288 GB) -- 2 x the paths per second of the reference's batches


@torch.no_grad()
def refine_view(scene, emitter, material_net, xs, ds, spp_diffuse=SPP_DIFFUSE, spp_specular=SPP_SPECULAR, indir_depth=INDIR_DEPTH, batch_rays=BATCH_RAYS,
                denoiser=None, lobes=None, uniforms=None):
    """One iteration of the reference's two per-view loops (refine_shading.py:109-127 and :144-174).
    xs, ds: (N,3) pixel-centre rays of the view on the GPU.  Returns {'diffuse': (N,3), 'specular0': [6 x (N,3)], 'specular1': [...]}
    (zeros at pixels whose primary ray misses).  denoiser: a utils.denoise.Denoiser for the full image; as in the reference EVERY map is
    denoised here (the bake skips roughness level 0, refine_shading does not).  lobes: subset of {0, 1..6}.  uniforms: optional dict
    lobe -> list of draw lists (one per batch) for parity runs."""
    xs = L.require_gpu(xs, torch.float32, "xs").reshape(-1, 3)
    ds = L.require_gpu(ds, torch.float32, "ds").reshape(-1, 3)
    positions, normals, uvs, triangle_idxs, valid = ray_intersect(scene, xs, ds)
    wi = ds
    B, dev = positions.shape[0], positions.device
    want = [l for l in range(N_ROUGHNESS + 1) if lobes is None or l in lobes]
    levels = roughness_levels()
    out = {"specular0": [], "specular1": [], "n_valid": int(valid.sum())}

    def batches(spp):
        bs = max(1, batch_rays // spp)
        return [(b * bs, min((b + 1) * bs, B)) for b in range(math.ceil(B / bs))]

    if 0 in want:
        Ld = torch.zeros(B, 3, device=dev)
        for k, (b0, b1) in enumerate(batches(spp_diffuse)):
            u = None if uniforms is None else uniforms[0][k]
            Ld[b0:b1] = path_tracing_det_diff(scene, emitter, material_net, positions[b0:b1], wi[b0:b1], normals[b0:b1], uvs[b0:b1], triangle_idxs[b0:b1],
                                              spp_diffuse, indir_depth, uniforms=u)
        if torch.isnan(Ld).any():
            raise L.IrisError("refine_view: NaN in the diffuse shading")           # assert L.isnan().any() == False   (:124)
        out["diffuse"] = Ld
    for l in want:
        if l == 0:
            continue
        L0 = torch.zeros(B, 3, device=dev); L1 = torch.zeros(B, 3, device=dev)
        for k, (b0, b1) in enumerate(batches(spp_specular)):
            u = None if uniforms is None else uniforms[l][k]
            a, b = path_tracing_det_spec(scene, emitter, material_net, levels[l - 1], positions[b0:b1], wi[b0:b1], normals[b0:b1], uvs[b0:b1],
                                         triangle_idxs[b0:b1], spp_specular, indir_depth, uniforms=u)
            L0[b0:b1] = a; L1[b0:b1] = b
        if torch.isnan(L0).any() or torch.isnan(L1).any():
            raise L.IrisError("refine_view: NaN in the specular shading")
        out["specular0"].append(L0); out["specular1"].append(L1)
    if denoiser is not None:
        if B != denoiser.H * denoiser.W:
            raise L.IrisError("refine_view: denoising needs the whole image in image order")
        denoiser.set_guides(normals, positions, valid)
        names = ([("diffuse", None)] if "diffuse" in out else []) + [(k, i) for i in range(len(out["specular0"])) for k in ("specular0", "specular1")]
        den = denoiser.denoise_maps([out[k] if i is None else out[k][i] for k, i in names])
        for (k, i), d in zip(names, den):
            if i is None:
                out[k] = d.reshape(B, 3)
            else:
                out[k][i] = d.reshape(B, 3)
    return out


def _load_material(spec, slf_path, ckpt):
    """The material network: by default the reference's own -- NGPBRDF(mask['voxel_min'], mask['voxel_max']) with the checkpoint's 'material.' weights
    (refine_shading.py:82-92) --, or, with --material pkg.module:factory, any callable position -> {'albedo','roughness','metallic'}."""
    if not spec:
        from .model.brdf import load_ngpbrdf
        if not ckpt:
            raise L.IrisError("refine_shading: --ckpt (the checkpoint holding the NGPBRDF weights, refine_shading.py:36,84) or --material pkg.module:factory is required")
        mask = torch.load(slf_path, map_location="cpu")
        return load_ngpbrdf(mask["voxel_min"], mask["voxel_max"], ckpt)
    mod, _, attr = spec.partition(":")
    factory = getattr(importlib.import_module(mod), attr or "material")
    mask = torch.load(slf_path, map_location="cpu")
    try:
        return factory(mask["voxel_min"], mask["voxel_max"], ckpt) if ckpt else factory(mask["voxel_min"], mask["voxel_max"])
    except TypeError:
        return factory()


def main(argv=None):
    """python -m iris_amd.refine_shading --scene S --slf_path vslf.npz --emitter_path emitter.pth --output OUT --dataset synthetic|real|generic
       --ckpt last.ckpt [--material pkg.module:factory]          (the reference's flags; --material replaces the NGPBRDF the reference hard-wires)"""
    import argparse
    from .model.emitter import SLFEmitter
    from .utils import cameras, exr
    from .utils.path_tracing import load_scene
    parser = argparse.ArgumentParser(description=main.__doc__)
    parser.add_argument("--dataset_root", type=str, help="dataset root")
    parser.add_argument("--scene", type=str, required=True, help="dataset folder")
    parser.add_argument("--slf_path", type=str, required=True)
    parser.add_argument("--emitter_path", type=str, required=True)
    parser.add_argument("--output", type=str, required=True, help="last shading folder")
    parser.add_argument("--ckpt", type=str, default=None, help="checkpoint path (handed to the material factory)")
    parser.add_argument("--dataset", type=str, required=True, help="dataset type: synthetic | real | scannetpp | generic")
    parser.add_argument("--ldr_img_dir", type=str, default=None)
    parser.add_argument("--res_scale", type=float, default=1.0)
    # additions (defaults reproduce the reference)
    parser.add_argument("--material", type=str, default=None, help="pkg.module:factory returning material_net(position) -> {'albedo','roughness','metallic'} "
                        "(default: the reference's NGPBRDF, loaded from --ckpt)")
    parser.add_argument("--cameras", type=str, default=None, help="generic camera JSON instead of the dataset's own camera files")
    parser.add_argument("--img_hw", type=int, nargs=2, default=None)
    parser.add_argument("--spp_diffuse", type=int, default=SPP_DIFFUSE)
    parser.add_argument("--spp_specular", type=int, default=SPP_SPECULAR)
    parser.add_argument("--indir_depth", type=int, default=INDIR_DEPTH)
    parser.add_argument("--batch_rays", type=int, default=BATCH_RAYS, help="paths per integrator call (the reference: 10240*128)")
    parser.add_argument("--seed", type=int, default=0)
    parser.add_argument("--compression", type=str, default="zip", choices=["none", "zips", "zip"])
    parser.add_argument("--overwrite", action="store_true", help="(accepted for symmetry with bake_shading; refining always overwrites, as the reference does)")
    parser.add_argument("--resume", action="store_true", help="skip the views a previous refine run with the same settings has completed (marked by a .refined sidecar)")
    parser.add_argument("--denoise", type=str, default="atrous", choices=["atrous", "none"])
    args = parser.parse_args(argv)

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise L.IrisError("refine_shading needs a HIP device; there is no CPU path")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if args.dataset in ("synthetic", "real"):
        mesh_path = os.path.join(args.scene, "scene.obj")
    elif args.dataset == "scannetpp":
        mesh_path = os.path.join(args.dataset_root, "data", args.scene, "scans", "scene.ply")
    else:
        mesh_path = os.path.join(args.scene, "scene.obj") if os.path.exists(os.path.join(args.scene, "scene.obj")) else os.path.join(args.scene, "scene.ply")
    assert os.path.exists(mesh_path), "mesh not found: " + mesh_path
    scene = load_scene(mesh_path, device=device)
    hw = tuple(args.img_hw) if args.img_hw else None
    if args.cameras:
        img_hw, views = cameras.load_generic(args.cameras, args.res_scale)
    elif args.dataset == "synthetic":
        img_hw, views = cameras.load_synthetic(args.scene, args.res_scale, hw)
    elif args.dataset == "real":
        img_hw, views = cameras.load_real(args.scene, args.res_scale, hw)
    elif args.dataset == "scannetpp":                      # Scannetpp(args.dataset_root, args.scene, split='train', pixel=False, res_scale=args.res_scale)
        img_hw, views = cameras.load_scannetpp(args.dataset_root, args.scene, args.res_scale)
    else:
        raise L.IrisError("--dataset {!r}: synthetic | real | scannetpp, or --cameras cameras.json".format(args.dataset))
    emitter = SLFEmitter(args.emitter_path, args.slf_path)
    material_net = _load_material(args.material, args.slf_path, args.ckpt)
    if isinstance(material_net, torch.nn.Module):
        material_net.to(device)
        for p in material_net.parameters():
            p.requires_grad = False
    os.makedirs(os.path.join(args.output, "diffuse"), exist_ok=True)
    os.makedirs(os.path.join(args.output, "specular"), exist_ok=True)
    denoiser = None
    if args.denoise == "atrous":
        from .utils.denoise import Denoiser
        denoiser = Denoiser(img_hw[::-1], device)
    writer = MapWriter(device, img_hw, args.compression)
    start_time = time.time()
    # The reference runs bake_shading and refine_shading on the SAME --output: the refined maps replace the bake's 13 files of every view in place
    # (refine_shading.py:126,172-173).  So a view's files existing says nothing about whether it has been refined: every view is rendered and
    # overwritten, unless --resume finds the sidecar a completed refine of that view left behind (keyed on what determines the result).
    import hashlib
    import json

    def stamp(path):
        """(absolute path, size, mtime in ns) of an input file or directory, or the path alone when it does not exist"""
        if not path or not os.path.exists(path):
            return (str(path),)
        st = os.stat(path)
        return (os.path.abspath(path), st.st_size, st.st_mtime_ns)
    # run_key: everything that determines the refined maps -- the arguments AND the identity of every input file (mesh, SLF, emitters, cameras, material
    # checkpoint: a changed file under an unchanged path is a different run) AND the build of the kernels (a sampler / traversal change alters bits: a resumed
    # run must not mix maps of two arithmetic versions)
    run_key = hashlib.sha256(repr((L.build_id(), args.material, stamp(args.ckpt), stamp(mesh_path), stamp(args.slf_path), stamp(args.emitter_path), stamp(args.cameras),
                                   args.dataset, os.path.abspath(args.scene), args.spp_diffuse, args.spp_specular, args.indir_depth, args.seed, args.res_scale,
                                   args.denoise, tuple(img_hw))).encode()).hexdigest()[:16]

    def files_state(files):
        return [[os.path.basename(f), os.stat(f).st_size, os.stat(f).st_mtime_ns] for f in files]

    def refined(marker, files):
        """the marker of a completed refine of this view, by this run_key, whose 13 files are still the ones it wrote (a re-bake replaces them)"""
        try:
            m = json.load(open(marker))
            return m.get("run_key") == run_key and m.get("files") == files_state(files)
        except Exception:     # noqa  (no marker, a marker of another format, a missing file)
            return False
    for im_id in range(rank, len(views), world):
        files = output_files(args.output, im_id)
        marker = os.path.join(args.output, "diffuse", "{:03d}.refined".format(im_id))
        if args.resume and refined(marker, files):
            continue
        if os.path.exists(marker):
            os.remove(marker)
        torch.manual_seed(args.seed * 1000003 + im_id); torch.cuda.manual_seed(args.seed * 1000003 + im_id)     # the integrators draw with torch.rand
        xs, ds = cameras.view_rays(views[im_id], img_hw, device)
        out = refine_view(scene, emitter, material_net, xs, ds, args.spp_diffuse, args.spp_specular, args.indir_depth, batch_rays=args.batch_rays, denoiser=denoiser)

        def mark(marker=marker, files=files):              # on a writer thread, as soon as THIS view's 13 files are complete: a crash later loses only the views in flight
            tmp = marker + ".part"
            with open(tmp, "w") as fh:
                json.dump({"run_key": run_key, "files": files_state(files)}, fh)
            os.replace(tmp, marker)
        writer.submit(files, torch.stack([out["diffuse"]] + [out[k][r] for r in range(N_ROUGHNESS) for k in ("specular0", "specular1")]).reshape(13, *img_hw, 3), on_written=mark)
    writer.close()                                             # every file and every marker is on disk from here on
    torch.cuda.synchronize()
    print("[refine_shading] rank {}: {:.2f} s".format(rank, time.time() - start_time))


if __name__ == "__main__":
    main()
