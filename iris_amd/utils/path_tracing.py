"""ray_intersect and the scene handle (reference: utils/path_tracing.py:17-48, bake_shading.py:55-61).

The reference wraps Mitsuba's OptiX closest hit; here the scene is a BVH built on the host by
``iris_scene_create`` and traversed by hand-written gfx950 kernels.
"""
import ctypes as C

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
        L.check(L.lib().iris_scene_create(v.ctypes.data_as(C.c_void_p), v.shape[0], f.ctypes.data_as(C.c_void_p), f.shape[0],
                                          self.device.index or 0, int(layout), C.byref(h)))
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
