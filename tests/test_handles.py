"""Device-side tables follow the modules: SLFEmitter / VoxelSLF cache their handles keyed on the tensors' version counters,
so in-place edits, load_state_dict through the parent module and optimiser steps are picked up without an explicit refresh(); the modules may
live on the CPU (the reference loads them with map_location='cpu'); a mesh / emitter mismatch is an error, not an out-of-bounds read."""
import numpy as np
import pytest
import torch

from test_hip_parity import dev, room_setup, T  # noqa: F401  (fixtures)

pytestmark = pytest.mark.gpu


def test_in_place_edits_and_load_state_dict_reach_the_device(dev, room_setup):
    from iris_amd import bake_shading as bs
    s = room_setup
    em = s["em"]
    P = 512
    pos, nrm = T(s["pos"][:P], dev), T(s["nrm"][:P], dev)
    base = bs.bake_diffuse(s["sc"], em, pos, nrm, 64, seed=9)
    saved = {k: v.clone() for k, v in em.state_dict().items()}
    try:
        with torch.no_grad():                       # in-place edit of both radiance tables, NO refresh()
            em.radiance.mul_(2.0); em.slf.radiance.mul_(2.0)
        assert torch.equal(bs.bake_diffuse(s["sc"], em, pos, nrm, 64, seed=9), base * 2.0)
        sd = {k: v.clone() for k, v in saved.items()}
        sd["radiance"] = saved["radiance"] * 4.0; sd["slf.radiance"] = saved["slf.radiance"] * 4.0
        em.load_state_dict(sd)                      # goes through nn.Module._load_from_state_dict of the PARENT: copies into the child's buffers
        assert torch.equal(bs.bake_diffuse(s["sc"], em, pos, nrm, 64, seed=9), base * 4.0)
        lookup = em.slf(pos)["rgb"]                 # the unfused lookup sees the same rows
        em.slf.radiance.mul_(0.5)
        assert torch.equal(em.slf(pos)["rgb"], lookup * 0.5)
    finally:
        em.load_state_dict(saved)
    assert torch.equal(bs.bake_diffuse(s["sc"], em, pos, nrm, 64, seed=9), base)


def test_emitter_for_another_mesh_is_rejected(dev, room_setup, tmp_path):
    from iris_amd import _lib as L
    from iris_amd import bake_shading as bs
    from iris_amd.utils.path_tracing import Scene
    s = room_setup
    r = s["room"]
    small = Scene(r["vertices"], r["faces"][:1000], device=dev)          # a different mesh: 1000 triangles
    pos, nrm = T(s["pos"][:8], dev), T(s["nrm"][:8], dev)
    with pytest.raises(L.IrisError, match="different number of triangles"):
        bs.bake_diffuse(small, s["em"], pos, nrm, 16)
    with pytest.raises(L.IrisError, match="different number of triangles"):
        bs.bake_lobes(small, s["em"], pos, nrm, pos, [None], [16])


def test_tiny_far_nodes_do_not_lose_rays(dev, oracle_mod):
    """Unused child slots of a node reference a degenerate leaf, never the idle marker: a scene of micrometre triangles far from the ray
    origins (node planes collapse onto one t in f32) still returns the brute-force closest hits."""
    from iris_amd.utils.path_tracing import Scene, ray_intersect
    rng = np.random.default_rng(5)
    n = 3000
    c = rng.random((n, 3)).astype(np.float32) * 1e-3 + np.float32(900.0)            # a 1 mm cloud at distance 900 from the origin
    v = (c[:, None, :] + (rng.random((n, 3, 3)).astype(np.float32) - 0.5) * np.float32(2e-6)).reshape(-1, 3)
    f = np.arange(n * 3, dtype=np.int32).reshape(n, 3)
    o = np.zeros((4096, 3), np.float32)
    d = (c[rng.integers(0, n, 4096)] + (rng.random((4096, 3)).astype(np.float32) - 0.5) * np.float32(1e-6)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    sc = Scene(v, f, device=dev)
    _, _, _, idx, valid = ray_intersect(sc, T(o, dev), T(d, dev))
    osc = oracle_mod.Scene(v, f)
    _, _, _, oidx, ovalid = osc.ray_intersect(o, d, brute=True)
    np.testing.assert_array_equal(idx.cpu().numpy(), oidx)
    np.testing.assert_array_equal(valid.cpu().numpy().astype(bool), ovalid.astype(bool))


def test_voxel_slf_built_on_the_device_and_rebound_buffers(dev):
    """(i) a VoxelSLF constructed from a mask that lives on the GPU (slf_bake counts the occupancy there) builds its index grid and its device
    tables without visiting the host and answers like the one built on the CPU; (ii) ADVICE round 2: a REBOUND radiance buffer
    (`vslf.radiance = vslf.radiance / n`, fresh tensors start at _version 0 and may reuse a freed address) is picked up."""
    from iris_amd.model.slf import VoxelSLF
    torch.manual_seed(0)
    H = 32
    mask = torch.rand(H, H, H) < 0.3
    a = VoxelSLF(mask, -0.2, 3.1)                   # host buffers (the reference's way)
    b = VoxelSLF(mask.to(dev), -0.2, 3.1)           # device buffers
    assert b.inds.is_cuda and torch.equal(a.inds, b.inds.cpu())
    rad = torch.rand(a.radiance.shape[0], 3)
    a.radiance[:] = rad; b.radiance[:] = rad.to(dev)
    x = (torch.rand(20000, 3, device=dev) * 3.6 - 0.4)
    assert torch.equal(a.spatial_idx(x), b.spatial_idx(x)) and torch.equal(a(x)["rgb"], b(x)["rgb"])
    for k in range(6):                              # rebinding in a loop: freed addresses come back, versions restart at 0
        b.radiance = b.radiance / 2.0
        assert torch.equal(b(x)["rgb"], a(x)["rgb"] / 2.0 ** (k + 1))


def test_index_less_device_is_the_current_device(dev, room_setup):
    """`torch.device('cuda')` names torch's CURRENT device, not device 0 (one process per GPU passes index-less devices around after set_device):
    Scene, SLFEmitter and NGPBRDF built with it work and give the bits the explicit index gives."""
    from iris_amd import bake_shading as bs
    from iris_amd.utils.path_tracing import Scene
    s = room_setup
    cur = torch.device("cuda")
    sc = Scene(s["room"]["vertices"], s["room"]["faces"], device=cur)
    P = 256
    pos, nrm = T(s["pos"][:P], dev), T(s["nrm"][:P], dev)
    base = bs.bake_diffuse(s["sc"], s["em"], pos, nrm, 64, seed=5)
    s["em"].refresh()
    s["em"].handle(cur)                                   # tables built for the index-less device
    assert torch.equal(bs.bake_diffuse(sc, s["em"], pos.to(cur), nrm.to(cur), 64, seed=5), base)


def test_fused_leaf_records_follow_scene_and_emitter(dev, room_setup, tmp_path):
    """The tile / view bake kernels shade from a copy of the scene's leaf records that carries the EMITTER's ordinal per triangle (iris_hip.hip fused_tris), cached in the
    emitter handle per scene (at most two).  One emitter used with three scenes of the same mesh size in turn (the third evicts the first), a second emitter with other
    emitter triangles on the same scene, and the first pair again: every bake must equal the bake of the pixel-per-wave kernel, which gathers the ordinal from its table."""
    from iris_amd import _lib as L
    from iris_amd import bake_shading as bs
    from iris_amd.model.emitter import SLFEmitter
    from iris_amd.utils.path_tracing import Scene
    from test_hip_parity import _emitter_files
    from tools import synth
    s = room_setup
    r = s["room"]
    P = 400
    pos, nrm = T(s["pos"][:P], dev), T(s["nrm"][:P], dev)
    verts = r["vertices"]
    scenes = [s["sc"]] + [Scene(verts + np.float32(1e-3 * k), r["faces"], device=dev) for k in (1, 2)]        # three scenes, the same topology
    # a second emitter: other triangles emit
    is_em2 = np.zeros_like(r["is_emitter"]); is_em2[::997] = True
    e2 = synth.emitters_for(verts, r["faces"], is_em2)
    sl = synth.slf_for(verts, r["faces"], 64)
    ep, sp = _emitter_files(tmp_path, e2["is_emitter"], e2["emitter_area"], e2["emitter_radiance"], sl["mask"], sl["inds"], sl["radiance"], sl["voxel_min"], sl["voxel_max"])
    em2 = SLFEmitter(ep, sp)

    def both(sc, em):
        a = bs.bake_diffuse(sc, em, pos, nrm, 64, seed=3, variant=L.BAKE_TILE_SORTED)
        b = bs.bake_diffuse(sc, em, pos, nrm, 64, seed=3, variant=L.BAKE_PIXEL_PER_WAVE)
        assert torch.equal(a, b)
        return a
    first = both(scenes[0], s["em"])
    for sc in scenes[1:] + scenes[:1]:
        both(sc, s["em"])
    other = both(scenes[0], em2)
    assert not torch.equal(other, first)                       # (other emitters, another SLF: another image)
    assert torch.equal(both(scenes[0], s["em"]), first)


def test_resolved_and_plain_hit_slots_give_the_same_maps(dev, room_setup):
    """The tile / view kernels park a finished ray's hit RESOLVED (position + emitter ordinal, iris_bake.h tile_body) unless the caller asks for the per-sample triangle
    ids, which need the plain (u, v, leaf slot) form: both forms through the same shading pass must give the same bits -- maps and per-sample table rows -- on the diffuse
    lobe and on a specular one."""
    from iris_amd import _lib as L
    from iris_amd import bake_shading as bs
    s = room_setup
    P = 700
    pos, nrm, wo = T(s["pos"][:P], dev), T(s["nrm"][:P], dev), T(s["wo"][:P], dev)
    for variant in (L.BAKE_TILE_SORTED,):
        Ld_r, src_r = bs.bake_diffuse(s["sc"], s["em"], pos, nrm, 128, seed=9, want_src=True, variant=variant)
        Ld_p, tri_p, src_p = bs.bake_diffuse(s["sc"], s["em"], pos, nrm, 128, seed=9, want_tri=True, want_src=True, variant=variant)
        assert torch.equal(Ld_r, Ld_p) and torch.equal(src_r, src_p)
        assert int((tri_p >= 0).sum()) > 0.9 * tri_p.numel() and int((src_p <= -2).sum()) > 0          # hits, some of them on emitters
        a0, a1, asrc = bs.bake_specular(s["sc"], s["em"], pos, nrm, wo, 0.3, 96, seed=9, want_src=True, variant=variant)       # (96: a ragged second round)
        b0, b1, _, bsrc = bs.bake_specular(s["sc"], s["em"], pos, nrm, wo, 0.3, 96, seed=9, want_tri=True, want_src=True, variant=variant)
        assert torch.equal(a0, b0) and torch.equal(a1, b1) and torch.equal(asrc, bsrc)
    # other reduce geometries: one sample, several pixels per wave (spp 3, 16), an odd number of rounds (192 = 3 x 64), a ragged fifth round (300), a tile of one pixel (5000)
    for spp, n_px in ((1, 1), (1, 67), (3, 5), (16, 200), (192, 200), (300, 200), (5000, 3)):
        p2, n2 = pos[:n_px].contiguous(), nrm[:n_px].contiguous()
        Ld_r = bs.bake_diffuse(s["sc"], s["em"], p2, n2, spp, seed=spp, variant=L.BAKE_TILE_SORTED)
        Ld_p, _ = bs.bake_diffuse(s["sc"], s["em"], p2, n2, spp, seed=spp, want_tri=True, variant=L.BAKE_TILE_SORTED)
        assert torch.equal(Ld_r, Ld_p), spp
    # the fused view kernel (all lobes behind one launch) takes the resolved form; the lobes baked one by one with triangle ids the plain one
    rough = bs.roughness_levels().tolist()
    lobes = [None, rough[0], rough[3]]
    view = bs.bake_lobes(s["sc"], s["em"], pos, nrm, wo, lobes, [64, 64, 64], seed=4, stream_ids=[0, 1, 4])
    Ld, _ = bs.bake_diffuse(s["sc"], s["em"], pos, nrm, 64, seed=4, stream_id=0, want_tri=True)
    assert torch.equal(view[0], Ld)
    for l, sid in ((1, 1), (2, 4)):
        c0, c1, _ = bs.bake_specular(s["sc"], s["em"], pos, nrm, wo, lobes[l], 64, seed=4, stream_id=sid, want_tri=True)
        assert torch.equal(view[l][0], c0) and torch.equal(view[l][1], c1)
