"""ray_intersect and the scene handle (reference: utils/path_tracing.py:17-48, bake_shading.py:55-61).

The reference wraps Mitsuba's OptiX closest hit; here the scene is a BVH built on the host by
``iris_scene_create`` and traversed by hand-written gfx950 kernels.
"""
import ctypes as C
import math

import numpy as np
import torch

from .. import _lib as L

RayEpsilon = 1500.0 * 2.0 ** -24  # mitsuba.math.RayEpsilon for float32 (bake_shading.py:117)


class Scene:
    """Triangle mesh + BVH resident in HBM.  Stands in for the object ``mitsuba.load_dict`` returns."""

    def __init__(self, vertices, faces, device=None, layout=L.BVH_DEFAULT):
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = torch.device(device)
        v = L.host_f32(vertices).reshape(-1, 3)
        f = faces.detach().cpu().numpy() if isinstance(faces, torch.Tensor) else np.asarray(faces)
        f = np.ascontiguousarray(f, dtype=np.int32).reshape(-1, 3)
        self.n_vertices, self.n_triangles = int(v.shape[0]), int(f.shape[0])
        h = C.c_void_p()
        if int(layout) == L.BVH_DEFAULT:
            L.check(L.lib().iris_scene_create(v.ctypes.data_as(C.c_void_p), v.shape[0], f.ctypes.data_as(C.c_void_p), f.shape[0],
                                              L.device_index(self.device), C.byref(h)))
        else:   # explicit node layout: diagnostics entry point (A/B baseline)
            L.check(L.lib().iris_debug_scene_create(v.ctypes.data_as(C.c_void_p), v.shape[0], f.ctypes.data_as(C.c_void_p), f.shape[0],
                                                    L.device_index(self.device), int(layout), C.byref(h)))
        self._h = h

    @property
    def handle(self):
        return self._h

    def info(self):
        i = L.SceneInfo()
        L.check(L.lib().iris_scene_get_info(self._h, C.byref(i)))
        return {k: getattr(i, k) for k, _ in L.SceneInfo._fields_}

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                L.lib().iris_scene_destroy(h)
            except Exception:
                pass


def load_mesh(path):
    """Minimal OBJ / PLY (ascii or binary little-endian) triangle-mesh reader -> (vertices f32 (V,3), faces i32 (F,3)).
    Replaces the file loading half of ``mitsuba.load_dict`` (bake_shading.py:46-61)."""
    path = str(path)
    if path.lower().endswith(".obj"):
        vs, fs = [], []
        with open(path, "r") as fh:
            for line in fh:
                if line.startswith("v "):
                    vs.append([float(x) for x in line.split()[1:4]])
                elif line.startswith("f "):
                    idx = [int(tok.split("/")[0]) for tok in line.split()[1:]]
                    idx = [i - 1 if i > 0 else len(vs) + i for i in idx]
                    for k in range(1, len(idx) - 1):
                        fs.append([idx[0], idx[k], idx[k + 1]])
        return np.asarray(vs, np.float32).reshape(-1, 3), np.asarray(fs, np.int32).reshape(-1, 3)
    if path.lower().endswith(".ply"):
        with open(path, "rb") as fh:
            assert fh.readline().strip() == b"ply", "not a PLY file"
            fmt, elems, cur = None, [], None
            while True:
                tok = fh.readline().split()
                if not tok:
                    continue
                if tok[0] == b"format":
                    fmt = tok[1].decode()
                elif tok[0] == b"element":
                    cur = {"name": tok[1].decode(), "count": int(tok[2]), "props": []}
                    elems.append(cur)
                elif tok[0] == b"property":
                    cur["props"].append([t.decode() for t in tok[1:]])
                elif tok[0] == b"end_header":
                    break
            np_t = {"char": "i1", "uchar": "u1", "short": "i2", "ushort": "u2", "int": "i4", "uint": "u4", "float": "f4", "double": "f8",
                    "int8": "i1", "uint8": "u1", "int16": "i2", "uint16": "u2", "int32": "i4", "uint32": "u4", "float32": "f4", "float64": "f8"}
            verts = faces = None
            for el in elems:
                if fmt == "ascii":
                    rows = [fh.readline().split() for _ in range(el["count"])]
                    if el["name"] == "vertex":
                        names = [p[-1] for p in el["props"]]
                        ix = [names.index(c) for c in ("x", "y", "z")]
                        verts = np.asarray([[float(r[i]) for i in ix] for r in rows], np.float32)
                    elif el["name"] == "face":
                        out = []
                        for r in rows:
                            n = int(r[0]); idx = [int(x) for x in r[1:1 + n]]
                            for k in range(1, n - 1):
                                out.append([idx[0], idx[k], idx[k + 1]])
                        faces = np.asarray(out, np.int32)
                else:
                    assert fmt == "binary_little_endian", "unsupported PLY format " + str(fmt)
                    if el["name"] == "vertex":
                        dt = np.dtype([(p[-1], "<" + np_t[p[0]]) for p in el["props"]])
                        a = np.frombuffer(fh.read(dt.itemsize * el["count"]), dtype=dt)
                        verts = np.stack([a["x"], a["y"], a["z"]], -1).astype(np.float32)
                    elif el["name"] == "face":
                        lp = [p for p in el["props"] if p[0] == "list"]
                        assert len(el["props"]) == 1 and lp, "unsupported PLY face layout"
                        ct, it = np_t[lp[0][1]], np_t[lp[0][2]]
                        raw = fh.read()
                        csz, isz = np.dtype(ct).itemsize, np.dtype(it).itemsize
                        n0 = int(np.frombuffer(raw[:csz], "<" + ct)[0])
                        stride = csz + n0 * isz
                        if el["count"] * stride <= len(raw) and n0 == 3:   # fast path: all triangles
                            dt = np.dtype([("n", "<" + ct), ("v", "<" + it, (3,))])
                            a = np.frombuffer(raw[:dt.itemsize * el["count"]], dtype=dt)
                            assert (a["n"] == 3).all(), "non-triangle faces"
                            faces = a["v"].astype(np.int32)
                        else:
                            out, off = [], 0
                            for _ in range(el["count"]):
                                n = int(np.frombuffer(raw[off:off + csz], "<" + ct)[0]); off += csz
                                idx = np.frombuffer(raw[off:off + n * isz], "<" + it); off += n * isz
                                for k in range(1, n - 1):
                                    out.append([idx[0], idx[k], idx[k + 1]])
                            faces = np.asarray(out, np.int32)
                    else:
                        raise AssertionError("unsupported PLY element " + el["name"])
            return verts.reshape(-1, 3), faces.reshape(-1, 3)
    raise ValueError("unsupported mesh type: " + path)


def load_scene(mesh_path, device=None, layout=L.BVH_DEFAULT):
    """mitsuba.load_dict({'type':'scene','shape_id':{'type':ply|obj,'filename':...}}) (bake_shading.py:55-61)."""
    v, f = load_mesh(mesh_path)
    return Scene(v, f, device=device, layout=layout)


def ray_intersect(scene, xs, ds):
    """Closest hit of rays with the scene mesh (utils/path_tracing.py:17-48).

    Args:
        xs, ds: Bx3 float32 tensors on the GPU (origins, directions)
    Return:
        positions Bx3, normals Bx3 (unit, face-forwarded against -ds), uvs Bx2 (barycentric b1,b2),
        idx B int64 (-1 = no intersection), valid B bool
    """
    xs = L.require_gpu(xs, torch.float32, "xs").reshape(-1, 3)
    ds = L.require_gpu(ds, torch.float32, "ds").reshape(-1, 3)
    B = xs.shape[0]
    pos = torch.empty(B, 3, device=xs.device, dtype=torch.float32)
    nrm = torch.empty(B, 3, device=xs.device, dtype=torch.float32)
    uv = torch.empty(B, 2, device=xs.device, dtype=torch.float32)
    idx = torch.empty(B, device=xs.device, dtype=torch.int64)
    valid = torch.empty(B, device=xs.device, dtype=torch.bool)
    with torch.cuda.device(xs.device):
        L.check(L.lib().iris_intersect(scene.handle, L.ptr(xs), L.ptr(ds), B, L.ptr(pos), L.ptr(nrm), L.ptr(uv), L.ptr(idx), L.ptr(valid), L.stream()))
    return pos, nrm, uv, idx, valid


# ----------------------------------------------------------------------------------------------------------------------
# cfg 5: the one-bounce MIS path tracer the reference trains through (utils/path_tracing.py:320-407)
# ----------------------------------------------------------------------------------------------------------------------
_SIDE = {}


def _side_stream(dev):
    """one extra HIP stream per (device, calling stream) for stages that are independent of each other: calls issued on different streams (a training loop that runs
    its independent path_tracing_single calls side by side) do not queue their emitter-sampling stages behind one another"""
    key = (torch.device(dev).index if torch.device(dev).index is not None else torch.cuda.current_device(), torch.cuda.current_stream(dev).cuda_stream)
    if key not in _SIDE:
        if len(_SIDE) >= 16:
            _SIDE.pop(next(iter(_SIDE)))
        _SIDE[key] = torch.cuda.Stream(device=dev)
    return _SIDE[key]


class _PtAccumulate(torch.autograd.Function):
    """L = mean_spp(radiance[e0] + coef1*radiance[e1] + coef2*radiance[e2] + const2): gather forward, scatter-add backward.
    Only emitter.radiance receives gradient; geometry and sampled directions carry none."""

    @staticmethod
    def forward(ctx, radiance, e0, path_of, e1, coef1, e2, coef2, const2, B, spp):
        rad = radiance.detach().to(device=e0.device, dtype=torch.float32).contiguous()   # emitter files are loaded on the CPU
        Lout = torch.empty(B, 3, device=rad.device, dtype=torch.float32)
        with torch.cuda.device(rad.device):
            L.check(L.lib().iris_pt_accumulate_fwd(L.ptr(rad), L.ptr(e0), L.ptr(path_of), L.ptr(e1), L.ptr(coef1), L.ptr(e2), L.ptr(coef2), L.ptr(const2),
                                                   B, spp, L.ptr(Lout), L.stream()))
        ctx.save_for_backward(e0, path_of, e1, coef1, e2, coef2)
        ctx.meta = (B, spp, tuple(radiance.shape), radiance.device)
        return Lout

    @staticmethod
    def backward(ctx, gL):
        e0, path_of, e1, coef1, e2, coef2 = ctx.saved_tensors
        B, spp, shape, rad_dev = ctx.meta
        g = torch.zeros(shape, device=gL.device, dtype=torch.float32)
        gL = L.require_gpu(gL.contiguous().to(torch.float32), torch.float32, "grad of L")
        with torch.cuda.device(gL.device):
            L.check(L.lib().iris_pt_accumulate_bwd(L.ptr(gL), L.ptr(e0), L.ptr(path_of), L.ptr(e1), L.ptr(coef1), L.ptr(e2), L.ptr(coef2), B, spp,
                                                   L.ptr(g), L.stream()))
        return (g.to(rad_dev),) + (None,) * 9


def path_tracing_single(scene, emitter_net, material_net, rays_o, rays_d, dx_du, dy_dv, spp, uniforms=None, compact=None, skip_unused_material=True):
    """Path trace the scene with one bounce and power-2 MIS (utils/path_tracing.py:320-407).

    Args as the reference: rays_o, rays_d, dx_du, dy_dv Bx3; spp samples per pixel.  material_net(position) returns
    {'albedo','roughness','metallic'} (the reference's NGPBRDF; any callable works).  uniforms: optional list of the five
    draws the reference makes, [rand(2,B,spp,1), rand(N), rand(N,2), rand(N), rand(N,2)] (parity mode).
    compact: True = compact the paths that continue after the primary hit, as the reference does (its draws are sized by that count, so this is
    the mode of the parity fixtures; costs two host synchronisations per call); False = keep all B*spp paths in place and mask the ones that do
    not continue (missed / emitter primary hits: a per-cent of the rays) -- the same per-path arithmetic and the same sum order, no host
    synchronisation anywhere in the call, which is what a training loop of 262 144-ray calls is bound by; None = False when the draws are this
    function's own (uniforms is None), True otherwise.
    skip_unused_material: the reference evaluates material_net a SECOND time at the sampled hits (:392) and uses the result only for the test
    roughness > trace_roughness = 0.0 (model/emitter.py:209).  A network that declares a lower bound of its roughness above that (`roughness_min`: NGPBRDF's
    sigmoid * 0.98 + 0.02 >= 0.02) decides the test without being evaluated: the call is skipped, the outputs are the same bit for bit
    (tests/test_pt_single.py); False = evaluate it as the reference does.  Any other callable is always evaluated.
    Returns L Bx3, differentiable with respect to emitter_net.radiance.
    """
    return _path_tracing(scene, emitter_net, material_net, rays_o, rays_d, dx_du, dy_dv, spp, 0, uniforms, compact=compact, skip_unused_material=skip_unused_material)


def path_tracing(scene, emitter_net, material_net, rays_o, rays_d, dx_du, dy_dv, spp, indir_depth, uniforms=None):
    """The full integrator render.py uses (utils/path_tracing.py:214-318): the first bounce of path_tracing_single -- with the radiance
    cache consulted at roughness > 0.6 (eval_emitter's default trace_roughness; path_tracing_single passes 0.0) and the MIS denominator of
    the emitter sample unclamped (:260 against :366) -- and every path whose sampled hit is neither an emitter nor a cached diffuse
    surface continued by trace_indirect for up to indir_depth bounces, added with the BRDF weight (:300-316, under no_grad: gradient flows
    through the first bounce only, as in the reference).  uniforms: the five draws of the first bounce, then trace_indirect's four per bounce.
    Returns L Bx3, differentiable with respect to emitter_net.radiance."""
    return _path_tracing(scene, emitter_net, material_net, rays_o, rays_d, dx_du, dy_dv, spp, int(indir_depth), uniforms, full=True)


def _path_tracing(scene, emitter_net, material_net, rays_o, rays_d, dx_du, dy_dv, spp, indir_depth, uniforms, full=False, compact=None, skip_unused_material=True):
    rays_o = L.require_gpu(rays_o, torch.float32, "rays_o").reshape(-1, 3)
    rays_d = L.require_gpu(rays_d, torch.float32, "rays_d").reshape(-1, 3)
    dx_du = L.require_gpu(dx_du, torch.float32, "dx_du").reshape(-1, 3)
    dy_dv = L.require_gpu(dy_dv, torch.float32, "dy_dv").reshape(-1, 3)
    B, dev = rays_o.shape[0], rays_o.device
    lib = L.lib()
    if compact is None:
        compact = uniforms is not None or full
    if full and not compact:
        raise L.IrisError("path_tracing: the continuation (trace_indirect) compacts its paths; compact=False is for path_tracing_single")
    u = list(uniforms) if uniforms is not None else None
    trace_rough = 0.6 if full else 0.0
    # the bound is taken from a network that is EXACTLY NGPBRDF (a subclass may map its roughness otherwise and would inherit the class attribute silently) or that
    # declares one on the INSTANCE (an explicit opt-in: tools/bench_refine.py's timing wrapper copies its network's).  Caveat: with a non-finite network output the
    # reference's test `NaN > 0` is False where the skipped path takes it for True: the "same bits" claim holds for finite outputs.
    from ..model.brdf import NGPBRDF
    rough_min = NGPBRDF.roughness_min if type(material_net) is NGPBRDF else getattr(material_net, "__dict__", {}).get("roughness_min")
    skip_next = bool(skip_unused_material) and rough_min is not None and float(rough_min) > trace_rough     # (see path_tracing_single's docstring)
    if u is not None:
        nxt = lambda *shape: L.require_gpu(u.pop(0), torch.float32, "uniforms").reshape(*shape)             # noqa: E731
    elif not compact:
        # the function's own draws, all path counts known up front (N = B * spp): ONE generator launch for the five tensors of the call (8 N floats) instead of five
        pool, off = torch.rand(8 * B * spp, device=dev), [0]

        def nxt(*shape):
            n = math.prod(shape)
            t = pool[off[0]:off[0] + n].reshape(*shape)
            off[0] += n
            return t
    else:
        nxt = lambda *shape: torch.rand(*shape, device=dev)                                               # noqa: E731

    with torch.cuda.device(dev):
        L.mark()
        dudv = nxt(2, B, spp)
        wi = torch.empty(B * spp, 3, device=dev)
        e0 = torch.empty(B * spp, device=dev, dtype=torch.int32)
        valid_next = torch.empty(B * spp, device=dev, dtype=torch.bool)
        eh, sh = emitter_net.handle(dev), emitter_net.slf.handle(dev)
        radiance = emitter_net.radiance
        if not compact:
            # the head of the un-compacted mode as ONE launch (iris_pt_primary: jitter, closest hit, emitter ordinal, continuation flags, wo = -wi): the arithmetic
            # of the three calls and the torch glue of the compacted branch below, bit for bit (tests/test_pt_single.py compares the two modes)
            N = B * spp
            position = torch.empty(N, 3, device=dev); normal = torch.empty(N, 3, device=dev); wo = torch.empty(N, 3, device=dev)
            path_of = torch.empty(N, device=dev, dtype=torch.int32)
            L.check(lib.iris_pt_primary(scene.handle, eh, L.ptr(rays_o), L.ptr(rays_d), L.ptr(dx_du), L.ptr(dy_dv), L.ptr(dudv), B, spp, L.ptr(wi), L.ptr(wo), L.ptr(position),
                                        L.ptr(normal), L.ptr(e0), L.ptr(valid_next), L.ptr(path_of), L.stream()))
        else:
            L.check(lib.iris_pt_jitter(L.ptr(rays_d), L.ptr(dx_du), L.ptr(dy_dv), L.ptr(dudv), B, spp, L.ptr(wi), L.stream()))
            position, normal, _, triangle_idx, _ = ray_intersect(scene, rays_o.repeat_interleave(spp, 0), wi)
            L.check(lib.iris_pt_primary_emit(eh, L.ptr(triangle_idx), B * spp, L.ptr(e0), L.ptr(valid_next), L.stream()))

        if compact:
            if not bool(valid_next.any()):          # the reference returns the un-reduced (B*spp,3) tensor here (:347-348)
                rdev = radiance.to(device=dev)
                ext = torch.cat([rdev, rdev.new_zeros(1, 3)])
                return ext[torch.where(e0 >= 0, e0.long(), torch.full_like(e0, radiance.shape[0]).long())]
            sel = torch.nonzero(valid_next, as_tuple=False).reshape(-1)
            N = sel.numel()
            path_of = torch.full((B * spp,), -1, device=dev, dtype=torch.int32)
            path_of[sel] = torch.arange(N, device=dev, dtype=torch.int32)
            position, normal, wo = position[sel].contiguous(), normal[sel].contiguous(), (-wi[sel]).contiguous()
        # (un-compacted: every path stays at its ray's index; the ones that do not continue keep path_of = -1 and are ignored by the accumulation -- their stage
        #  outputs are computed on the zeros a miss returns, or on the emitter hit, and never read)

        L.mark("jitter + primary hit")
        mat = material_net(position)
        albedo = mat["albedo"].detach().to(torch.float32).reshape(-1, 3).contiguous()
        rough = mat["roughness"].detach().to(torch.float32).reshape(-1).contiguous()
        metal = mat["metallic"].detach().to(torch.float32).reshape(-1).contiguous()

        L.mark("material (primary hits)")
        # direct illumination: emitter sampling + MIS (:357-382).  Independent of the BRDF-sampling branch below: launched on a side stream, so that
        # its visibility rays run beside the (longer) BRDF rays -- at 262 144 paths per call either kernel alone leaves most of the chip idle
        s1, s2 = nxt(N), nxt(N, 2)
        coef1 = torch.empty(N, 3, device=dev); e1 = torch.empty(N, device=dev, dtype=torch.int32)
        main, side = torch.cuda.current_stream(dev), _side_stream(dev)
        fork = torch.cuda.Event(); fork.record(main)
        with torch.cuda.stream(side):
            side.wait_event(fork)
            L.mark()
            L.check(lib.iris_pt_nee(scene.handle, eh, L.ptr(position), L.ptr(normal), L.ptr(wo), L.ptr(albedo), L.ptr(rough), L.ptr(metal), L.ptr(s1), L.ptr(s2), N,
                                    L.ptr(coef1), L.ptr(e1), 1e-6, 1e-6, 0.0 if full else 1e-6, L.stream()))
            L.mark("nee (side stream)")
            join = torch.cuda.Event(); join.record(side)
        for t_side in (position, normal, wo, albedo, rough, metal, s1, s2, coef1, e1):
            t_side.record_stream(side)     # an exception in a main-stream stage below must not hand these blocks back to the main-stream pool while the side kernel still reads them
        # BRDF sampling + next intersection (:384-391)
        s1b, s2b = nxt(N), nxt(N, 2)
        wi_b = torch.empty(N, 3, device=dev); pdf_b = torch.empty(N, device=dev); w_b = torch.empty(N, 3, device=dev)
        pos_n = torch.empty(N, 3, device=dev); nrm_n = torch.empty(N, 3, device=dev)
        tri_n = torch.empty(N, device=dev, dtype=torch.int64); hit_n = torch.empty(N, device=dev, dtype=torch.bool)
        L.check(lib.iris_pt_brdf_trace(scene.handle, L.ptr(position), L.ptr(normal), L.ptr(wo), L.ptr(albedo), L.ptr(rough), L.ptr(metal), L.ptr(s1b), L.ptr(s2b), N,
                                       L.ptr(wi_b), L.ptr(pdf_b), L.ptr(w_b), L.ptr(pos_n), L.ptr(nrm_n), L.ptr(tri_n), L.ptr(hit_n), 0, 0.0, L.stream()))
        L.mark("brdf sample + trace")
        if skip_next:
            rough_n = None           # every roughness of this network exceeds trace_roughness: the finish stage's test is decided (iris_hip.h)
        else:
            mat_next = material_net(pos_n)
            rough_n = mat_next["roughness"].detach().to(torch.float32).reshape(-1).contiguous()
        L.mark("material (sampled hits)")
        # eval_emitter at the sampled hit + MIS (:394-404)
        coef2 = torch.empty(N, 3, device=dev); const2 = torch.empty(N, 3, device=dev); e2 = torch.empty(N, device=dev, dtype=torch.int32)
        hit_valid = torch.empty(N, device=dev, dtype=torch.bool) if full else None
        L.check(lib.iris_pt_brdf_finish(eh, sh, L.ptr(position), L.ptr(pos_n), L.ptr(nrm_n), L.ptr(wi_b), L.ptr(tri_n), L.ptr(rough_n) if rough_n is not None else None,
                                        L.ptr(pdf_b), L.ptr(w_b), N, L.ptr(coef2), L.ptr(const2), L.ptr(e2), L.ptr(hit_valid) if full else None, trace_rough, 1e-6, L.stream()))
        L.mark("finish")
        main.wait_event(join)              # (before anything frees or reads the tensors the side stream works on)
        if full and indir_depth > 0:
            # the paths whose sampled hit neither ended them nor left the scene go on (:300-316): moved to the front on the device, with the material rows just
            # evaluated at their hits when there are any (trace_indirect's depth 0 would ask the network for them again, :432-433)
            iota = torch.arange(N, device=dev, dtype=torch.int32)
            if rough_n is not None:
                ma, mr, mm = _mat_tensors(mat_next)
                _, (p_k, n_k, w_k, a_k, wo_k), (r_k, m_k), (keep,) = compact_rows(hit_valid, rows3=(pos_n, nrm_n, w_b, ma), neg3=(wi_b,), rows1=(mr, mm), rowsi=(iota,))
                mat0 = (a_k, r_k, m_k)
            else:
                _, (p_k, n_k, w_k, wo_k), _, (keep,) = compact_rows(hit_valid, rows3=(pos_n, nrm_n, w_b), neg3=(wi_b,), rowsi=(iota,))
                mat0 = None
            L_indir = trace_indirect(scene, emitter_net, material_net, p_k, wo_k, n_k, indir_depth, uniforms=u, mat0=mat0)      # (u: what is left of the recorded draws, or None)
            keep = keep.long()
            const2[keep] = const2[keep] + w_k * L_indir                           # rides on the constant term: no gradient, as in the reference
    out = _PtAccumulate.apply(radiance, e0, path_of, e1, coef1, e2, coef2, const2, B, spp)
    L.mark("accumulate")
    return out


# ----------------------------------------------------------------------------------------------------------------------
# refine_shading's integrators: multi-bounce indirect light with NEE + BRDF sampling + MIS
# (utils/path_tracing.py:409-502 trace_indirect) under a deterministic first hit (:50-124 path_tracing_det_diff,
# :126-212 path_tracing_det_spec).  Each bounce is the same three fused stages as path_tracing_single, with trace_indirect's
# constants; the material network is the caller's, evaluated between the stages; paths are compacted every bounce.
# ----------------------------------------------------------------------------------------------------------------------
def _mat_tensors(mat):
    return (mat["albedo"].detach().to(torch.float32).reshape(-1, 3).contiguous(), mat["roughness"].detach().to(torch.float32).reshape(-1).contiguous(),
            mat["metallic"].detach().to(torch.float32).reshape(-1).contiguous())


def _draws(uniforms, dev):
    u = list(uniforms) if uniforms is not None else None
    if u is None:
        return lambda *shape: torch.rand(*shape, device=dev)
    return lambda *shape: L.require_gpu(u.pop(0), torch.float32, "uniforms").reshape(*shape)


def _lobe_trace(scene, position, normal, wo, mat, s1, s2, lobe, roughness=0.0):
    """sample a lobe and find the next intersection (one fused launch)"""
    N, dev = position.shape[0], position.device
    wi = torch.empty(N, 3, device=dev); pdf = torch.empty(N, device=dev); w = torch.empty(N, 3, device=dev)
    pos_n = torch.empty(N, 3, device=dev); nrm_n = torch.empty(N, 3, device=dev)
    tri_n = torch.empty(N, device=dev, dtype=torch.int64); hit = torch.empty(N, device=dev, dtype=torch.bool)
    a, r, m = mat if mat is not None else (None, None, None)
    L.check(L.lib().iris_pt_brdf_trace(scene.handle, L.ptr(position), L.ptr(normal), L.ptr(wo), L.ptr(a), L.ptr(r), L.ptr(m), L.ptr(s1), L.ptr(s2), N,
                                       L.ptr(wi), L.ptr(pdf), L.ptr(w), L.ptr(pos_n), L.ptr(nrm_n), L.ptr(tri_n), L.ptr(hit), int(lobe), float(roughness), L.stream()))
    return wi, pdf, w, pos_n, nrm_n, tri_n


class _Counts:
    """Pinned host words the device-side path counts are copied into (one per device): the ONE thing a bounce hands back to the host -- the number of paths that
    continue, which sizes the next bounce's launches and the material network's evaluation (a caller-supplied callable) -- as a 4-byte copy and an event wait
    instead of torch.nonzero (a scan, a host synchronisation and an index tensor) plus one ATen gather per state array."""
    _host = {}

    @classmethod
    def read(cls, count, dev):
        key = torch.device(dev).index if torch.device(dev).index is not None else torch.cuda.current_device()
        h = cls._host.get(key)
        if h is None:
            h = cls._host[key] = torch.empty(1, dtype=torch.int32).pin_memory()
        h.copy_(count, non_blocking=True)
        ev = torch.cuda.Event(); ev.record(torch.cuda.current_stream(dev)); ev.synchronize()
        return int(h[0])


class _Pool:
    """Several per-path arrays of a bounce out of ONE allocation (a float32 block cut into views, every piece starting at a multiple of 4 floats; int32 / int64 / bool
    pieces are views of the same block): a bounce of trace_indirect needs ~25 arrays, and at the reference's batch size the host thread that allocates and launches is as
    long as the kernels."""

    def __init__(self, n_floats, device):
        self.buf, self.off = torch.empty(int(n_floats) + 64, device=device, dtype=torch.float32), 0

    def _take(self, n_floats):
        o = self.off
        self.off = (o + int(n_floats) + 3) // 4 * 4
        if o + int(n_floats) > self.buf.numel():
            raise L.IrisError("_Pool: block too small for its pieces (a sizing bug)")
        return self.buf[o:o + int(n_floats)]

    def f(self, *shape):
        return self._take(math.prod(shape)).view(*shape)

    def i32(self, n):
        return self._take(n).view(torch.int32)

    def i64(self, n):
        self.off = (self.off + 1) // 2 * 2
        return self._take(2 * n).view(torch.int64)

    def u8(self, n, dtype=torch.bool):
        return self._take((n + 3) // 4).view(torch.uint8)[:n].view(dtype)


def compact_rows(keep, rows3=(), neg3=(), rows1=(), rowsi=()):
    """Boolean indexing of several per-path arrays by one mask, as ONE order-preserving device-side pass (`iris_pt_compact`): returns (count, [a[keep] for a in rows3] +
    [-a[keep] for a in neg3], [a[keep] for a in rows1], [a[keep] for a in rowsi]) -- the outputs are views of N-row buffers cut to the count (one 4-byte read-back)."""
    import ctypes as C
    keep = L.require_gpu(keep, torch.bool, "keep").reshape(-1)
    N, dev = keep.shape[0], keep.device
    in3 = [L.require_gpu(t, torch.float32, "rows3").reshape(N, 3) for t in list(rows3) + list(neg3)]
    in1 = [L.require_gpu(t, torch.float32, "rows1").reshape(N) for t in rows1]
    ini = [L.require_gpu(t, torch.int32, "rowsi").reshape(N) for t in rowsi]
    if max(len(in3), len(in1), len(ini)) > 6:
        raise L.IrisError("compact_rows: at most 6 arrays of each kind")
    ws_bytes = int(L.lib().iris_pt_compact_workspace_bytes(N))
    pool = _Pool((3 * len(in3) + len(in1) + len(ini)) * N + 4 * (len(in3) + len(in1) + len(ini)) + ws_bytes // 4 + 64, dev)
    out3 = [pool.f(N, 3) for _ in in3]; out1 = [pool.f(N) for _ in in1]; outi = [pool.i32(N) for _ in ini]
    count = pool.i32(1)
    ws = pool.u8(max(ws_bytes, 4), torch.uint8)
    arr = lambda ts: (C.c_void_p * max(len(ts), 1))(*[t.data_ptr() for t in ts])          # noqa: E731
    neg = sum(1 << (len(rows3) + k) for k in range(len(neg3)))
    with torch.cuda.device(dev):
        L.check(L.lib().iris_pt_compact(L.ptr(keep), N, len(in3), arr(in3), arr(out3), neg, len(in1), arr(in1), arr(out1), len(ini), arr(ini), arr(outi),
                                        L.ptr(count), L.ptr(ws), ws_bytes, L.stream()))
        n = _Counts.read(count, dev)
    return n, [t[:n] for t in out3], [t[:n] for t in out1], [t[:n] for t in outi]


def _bounce_draws(nxt, own, N, dev):
    """the four draws of a bounce (rand(N), rand(N,2), rand(N), rand(N,2)): recorded ones in the reference's order, or ONE generator launch cut into four
    (every piece starts at a multiple of four floats)"""
    if not own:
        return nxt(N), nxt(N, 2), nxt(N), nxt(N, 2)
    r4 = lambda n: (n + 3) // 4 * 4          # noqa: E731
    o1, o2, o3 = r4(N), r4(N) + r4(2 * N), 2 * r4(N) + r4(2 * N)
    pool = torch.rand(o3 + 2 * N, device=dev)
    return pool[:N], pool[o1:o1 + 2 * N].reshape(N, 2), pool[o2:o2 + N], pool[o3:o3 + 2 * N].reshape(N, 2)


@torch.no_grad()
def trace_indirect(scene, emitter_net, material_net, position, wo, normal, indir_depth, uniforms=None, mat0=None):
    """indirect illumination: up to indir_depth bounces of emitter sampling + BRDF sampling with power-2 MIS, paths ending at
    emitters / the diffuse radiance cache (utils/path_tracing.py:409-502).  Returns L Bx3.  uniforms: optional list of the
    draws in the reference's order, per bounce rand(N), rand(N,2), rand(N), rand(N,2).
    mat0: optional (albedo (B,3), roughness (B), metallic (B)) of `position` when the caller has just evaluated the material network there (the integrators below
    have, for eval_emitter's roughness test): the reference evaluates it again at depth 0 (:432-433), a deterministic network returns the same rows.
    Between two bounces the surviving paths are moved to the front of fresh arrays on the device (`compact_rows`); after the last bounce nothing is moved."""
    position = L.require_gpu(position, torch.float32, "position").reshape(-1, 3)
    wo = L.require_gpu(wo, torch.float32, "wo").reshape(-1, 3)
    normal = L.require_gpu(normal, torch.float32, "normal").reshape(-1, 3)
    B, dev = position.shape[0], position.device
    lib, nxt = L.lib(), _draws(uniforms, dev)
    Lacc = torch.zeros(B, 3, device=dev)
    rows = torch.arange(B, device=dev, dtype=torch.int32)
    throughput = torch.ones(B, 3, device=dev)
    radiance = emitter_net.radiance_on(dev)
    mat = None if mat0 is None else tuple(L.require_gpu(t, torch.float32, "mat0").reshape(*sh) for t, sh in zip(mat0, ((B, 3), (B,), (B,))))
    with torch.cuda.device(dev):
        eh, sh = emitter_net.handle(dev), emitter_net.slf.handle(dev)
        for depth in range(indir_depth):
            N = position.shape[0]
            if N == 0:
                break
            L.mark()
            if depth == 0 and mat is None:
                mat = _mat_tensors(material_net(position))
                L.mark("indirect: material network")
            a, r, m = mat
            s1, s2, s1b, s2b = _bounce_draws(nxt, uniforms is None, N, dev)
            L.mark("indirect: draws")
            pool = _Pool(28 * N + 256, dev)                      # every per-path array of the bounce out of one allocation (26.5 N floats + padding)
            coef1 = pool.f(N, 3); e1 = pool.i32(N)
            # the emitter-sampling stage (sample_emitter, visibility ray, geometry term, eval_brdf, MIS: :434-456) and the BRDF stage (sample_brdf + closest hit, :459-465) of
            # the bounce as ONE launch: the two rays of a path leave from the same point, and their rays are sorted and traced together (iris_pt_bounce)
            wi = pool.f(N, 3); pdf = pool.f(N); w = pool.f(N, 3)
            pos_n = pool.f(N, 3); nrm_n = pool.f(N, 3)
            tri_n = pool.i64(N); hit = pool.u8(N)
            L.check(lib.iris_pt_bounce(scene.handle, eh, L.ptr(position), L.ptr(normal), L.ptr(wo), L.ptr(a), L.ptr(r), L.ptr(m), L.ptr(s1), L.ptr(s2), L.ptr(s1b), L.ptr(s2b), N,
                                       L.ptr(coef1), L.ptr(e1), 1e-12, 1e-12, 0.0, L.ptr(wi), L.ptr(pdf), L.ptr(w), L.ptr(pos_n), L.ptr(nrm_n), L.ptr(tri_n), L.ptr(hit), L.stream()))
            L.mark("indirect: emitter sample + visibility ray, BRDF sample + closest hit (one launch)")
            L.check(lib.iris_pt_apply(L.ptr(Lacc), L.ptr(rows), L.ptr(throughput), L.ptr(radiance), L.ptr(e1), L.ptr(coef1), None, None, N, 1, L.stream()))
            L.mark("indirect: accumulate")
            mat_next = _mat_tensors(material_net(pos_n))
            L.mark("indirect: material network")
            coef2 = pool.f(N, 3); const2 = pool.f(N, 3); e2 = pool.i32(N)
            valid_next = pool.u8(N)
            L.check(lib.iris_pt_brdf_finish(eh, sh, L.ptr(position), L.ptr(pos_n), L.ptr(nrm_n), L.ptr(wi), L.ptr(tri_n), L.ptr(mat_next[1]), L.ptr(pdf), L.ptr(w), N,
                                            L.ptr(coef2), L.ptr(const2), L.ptr(e2), L.ptr(valid_next), 0.6, 1e-12, L.stream()))
            L.mark("indirect: eval_emitter + MIS")
            L.check(lib.iris_pt_apply(L.ptr(Lacc), L.ptr(rows), L.ptr(throughput), L.ptr(radiance), L.ptr(e2), L.ptr(coef2), L.ptr(const2), L.ptr(w), N, 1, L.stream()))
            L.mark("indirect: accumulate")
            if depth + 1 == indir_depth:
                break                                                       # (the reference masks its arrays once more, :488-501, and returns)
            # continue only the paths that neither ended nor left the scene: position = position_next[valid_next], wo = -wi[valid_next], ... (:488-501)
            _, (position, normal, throughput, alb, wo), (rgh, mtl), (rows,) = compact_rows(valid_next, rows3=(pos_n, nrm_n, throughput, mat_next[0]), neg3=(wi,),
                                                                                          rows1=(mat_next[1], mat_next[2]), rowsi=(rows,))
            mat = (alb, rgh, mtl)
            L.mark("indirect: survivors to the front (+ the count's read-back)")
    return Lacc


@torch.no_grad()
def _det_common(scene, emitter_net, material_net, positions, wis, normals, triangle_idxs, spp, indir_depth, lobe, roughness, uniforms, reuse_material=True):
    positions = L.require_gpu(positions, torch.float32, "positions").reshape(-1, 3)
    wis = L.require_gpu(wis, torch.float32, "wis").reshape(-1, 3)
    normals = L.require_gpu(normals, torch.float32, "normals").reshape(-1, 3)
    dev = positions.device
    sel = torch.nonzero(triangle_idxs != -1, as_tuple=False).reshape(-1)
    P = sel.numel()
    if P == 0:
        return sel, None, None
    u = list(uniforms) if uniforms is not None else None
    position = positions[sel].repeat_interleave(spp, 0).contiguous()
    normal = normals[sel].repeat_interleave(spp, 0).contiguous()
    wo = (-wis[sel]).repeat_interleave(spp, 0).contiguous()
    N = P * spp
    with torch.cuda.device(dev):
        # (the reference also evaluates the first-hit material here, :77/:157, and never uses it)
        L.mark()
        s2 = L.require_gpu(u.pop(0), torch.float32, "uniforms").reshape(N, 2) if u is not None else torch.rand(N, 2, device=dev)
        wi, _, w, pos_n, nrm_n, tri_n = _lobe_trace(scene, position, normal, wo, None, None, s2, lobe, roughness)
        L.mark("first bounce: lobe sample + closest hit")
        mat_next = material_net(pos_n)
        L.mark("first bounce: material network")
        Le, _, valid_next = emitter_net.eval_emitter(pos_n, wi, tri_n, mat_next["roughness"])        # default trace_roughness = 0.6
        # the paths that go on (:96-104 / :176-184): their state -- and the material rows just evaluated at their hits, which trace_indirect's depth 0 would ask the
        # network for a second time (:432-433) -- moved to the front on the device
        ma, mr, mm = _mat_tensors(mat_next)
        iota = torch.arange(N, device=dev, dtype=torch.int32)
        _, (p_k, n_k, a_k, wo_k), (r_k, m_k), (keep,) = compact_rows(valid_next, rows3=(pos_n, nrm_n, ma), neg3=(wi,), rows1=(mr, mm), rowsi=(iota,))
        L.mark("first bounce: eval_emitter, survivors to the front")
        L_indir = trace_indirect(scene, emitter_net, material_net, p_k, wo_k, n_k, indir_depth, uniforms=u, mat0=(a_k, r_k, m_k) if reuse_material else None)
        L.mark()
        total = Le.clone()
        total[keep.long()] += L_indir          # both weights multiply (Le + L_indir) of the same path
        L.mark("sum of the path's terms")
    return sel, w, total.reshape(P, spp, 3)


def path_tracing_det_diff(scene, emitter_net, material_net, positions, wis, normals, uvs, triangle_idxs, spp, indir_depth, uniforms=None, reuse_material=True):
    """diffuse shading with a deterministic first intersection and indir_depth bounces of indirect light
    (utils/path_tracing.py:50-124).  Returns Lout Bx3 (zeros where triangle_idxs == -1).
    reuse_material: the material rows evaluated at the sampled hits (for eval_emitter's roughness test, :90) are handed to trace_indirect, whose depth 0 would evaluate
    the network at the same points again (:432-433) -- the same rows for a deterministic network, one evaluation of ~six per path less; False = evaluate twice, as written."""
    Lout = torch.zeros_like(positions.reshape(-1, 3))
    sel, w, total = _det_common(scene, emitter_net, material_net, positions, wis, normals, triangle_idxs, spp, indir_depth, 1, 0.0, uniforms, reuse_material)
    if total is not None:
        Lout[sel] = total.mean(1)              # sample_diffuse's brdf_weight is 1 (model/brdf.py:86)
    return Lout


def path_tracing_det_spec(scene, emitter_net, material_net, roughness_level, positions, wis, normals, uvs, triangle_idxs, spp, indir_depth, uniforms=None, reuse_material=True):
    """the two Fresnel-split specular shadings at one roughness level (utils/path_tracing.py:126-212).  Returns L0out, L1out."""
    L0 = torch.zeros_like(positions.reshape(-1, 3)); L1 = torch.zeros_like(L0)
    r = float(roughness_level.detach().float().cpu().item()) if isinstance(roughness_level, torch.Tensor) else float(roughness_level)
    sel, w, total = _det_common(scene, emitter_net, material_net, positions, wis, normals, triangle_idxs, spp, indir_depth, 2, r, uniforms, reuse_material)
    if total is not None:
        P, spp_ = total.shape[0], total.shape[1]
        g = w.reshape(P, spp_, 3)
        L0[sel] = (g[..., 0:1] * total).mean(1)
        L1[sel] = (g[..., 1:2] * total).mean(1)
    return L0, L1
