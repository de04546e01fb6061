"""ctypes binding of ``libiris_oracle.so`` (numpy in / numpy out).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("IRIS_ORACLE_LIB") or os.path.join(_HERE, "libiris_oracle.so")   # override: the sanitizer build (tests/test_sanitizers.py)

RAY_EPSILON = 1500.0 * 2.0 ** -24  # mitsuba.math.RayEpsilon, float32 variants


def build(force=False):
    """Compile the oracle with gcc (Makefile next to this file)."""
    src = os.path.join(_HERE, "iris_oracle.c")
    if os.environ.get("IRIS_ORACLE_LIB"):
        return _SO
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B"] if force else ["make", "-s", "-C", _HERE])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.orc_scene_create.restype = C.c_void_p
        _lib.orc_slf_create.restype = C.c_void_p
        _lib.orc_emitter_create.restype = C.c_void_p
        _lib.orc_num_threads.restype = C.c_int
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def set_mode(mode):
    """0 = literal restatement with libm (pinned to the goldens); 1 = the HIP kernels' exact arithmetic (see iris_oracle.c)."""
    lib().orc_set_mode(C.c_int(int(mode)))


def get_mode():
    return int(lib().orc_get_mode())


class device_arithmetic:
    """with oracle.device_arithmetic(): ... -> oracle results comparable bit for bit with the GPU."""

    def __enter__(self):
        self.prev = get_mode(); set_mode(1)

    def __exit__(self, *a):
        set_mode(self.prev)


def num_threads():
    return int(lib().orc_num_threads())


def set_num_threads(n):
    lib().orc_set_num_threads(C.c_int(int(n)))


# ---------------------------------------------------------------- a1
def raygen_real(K, c2w, H, W, ray_diff=False):
    K = _f32(K).reshape(9); c2w = _f32(c2w).reshape(12)
    n = H * W
    o = np.empty((n, 3), np.float32); d = np.empty((n, 3), np.float32)
    dx = np.empty((n, 3), np.float32) if ray_diff else None
    dy = np.empty((n, 3), np.float32) if ray_diff else None
    lib().orc_raygen_real(_p(K), _p(c2w), C.c_int(H), C.c_int(W), C.c_int(int(ray_diff)), _p(o), _p(d), _p(dx), _p(dy))
    return (o, d, dx, dy) if ray_diff else (o, d)


def raygen_synthetic(focal, c2w, H, W, ray_diff=False):
    c2w = _f32(c2w).reshape(12)
    n = H * W
    o = np.empty((n, 3), np.float32); d = np.empty((n, 3), np.float32)
    dx = np.empty((n, 3), np.float32) if ray_diff else None
    dy = np.empty((n, 3), np.float32) if ray_diff else None
    lib().orc_raygen_synthetic(C.c_float(focal), _p(c2w), C.c_int(H), C.c_int(W), C.c_int(int(ray_diff)), _p(o), _p(d), _p(dx), _p(dy))
    return (o, d, dx, dy) if ray_diff else (o, d)


# ---------------------------------------------------------------- a3 / a4 / a10
def get_normal_space(normal):
    normal = _f32(normal); B = normal.shape[0]
    out = np.empty((B, 3, 3), np.float32)
    lib().orc_get_normal_space(_p(normal), C.c_int64(B), _p(out))
    return out


def double_sided(V, N):
    V = _f32(V); N = _f32(N).copy()
    lib().orc_double_sided(_p(V), _p(N), C.c_int64(N.shape[0]))
    return N


def sample_diffuse(u2, normal):
    u2 = _f32(u2); normal = _f32(normal); B = u2.shape[0]
    wi = np.empty((B, 3), np.float32); pdf = np.empty((B, 1), np.float32); w = np.empty((B, 3), np.float32)
    lib().orc_sample_diffuse(_p(u2), _p(normal), C.c_int64(B), _p(wi), _p(pdf), _p(w))
    return wi, pdf, w


def sample_specular(u2, wo, normal, roughness):
    u2 = _f32(u2); wo = _f32(wo); normal = _f32(normal); B = u2.shape[0]
    wi = np.empty((B, 3), np.float32); pdf = np.empty((B, 1), np.float32)
    g0 = np.empty((B, 1), np.float32); g1 = np.empty((B, 1), np.float32)
    lib().orc_sample_specular(_p(u2), _p(wo), _p(normal), C.c_float(np.float32(roughness)), C.c_int64(B), _p(wi), _p(pdf), _p(g0), _p(g1))
    return wi, pdf, g0, g1


def lerp_specular(specular, roughness):
    specular = _f32(specular); roughness = _f32(roughness).reshape(-1)
    B, R, _ = specular.shape
    out = np.empty((B, 3), np.float32)
    lib().orc_lerp_specular(_p(specular), _p(roughness), C.c_int64(B), C.c_int(R), _p(out))
    return out


# ---------------------------------------------------------------- 8(f)-3 packed shading cache + shading combine
def cache_pack(diffuse, spec0, spec1):
    """13 maps (n,3) -> rows (n, 3+6R) in the reference's order (utils/dataset/scannetpp/dataset.py:359-377)."""
    diffuse = _f32(diffuse); spec0 = [_f32(m) for m in spec0]; spec1 = [_f32(m) for m in spec1]
    n, R = diffuse.shape[0], len(spec0)
    rows = np.empty((n, 3 + 6 * R), np.float32)
    a0 = (C.c_void_p * R)(*[m.ctypes.data for m in spec0]); a1 = (C.c_void_p * R)(*[m.ctypes.data for m in spec1])
    lib().orc_cache_pack(_p(diffuse), a0, a1, C.c_int64(n), C.c_int(R), _p(rows))
    return rows


def cache_gather(rows, idx):
    rows = _f32(rows); idx = np.ascontiguousarray(idx, dtype=np.int64)
    R = (rows.shape[1] - 3) // 6
    out = np.empty((len(idx), rows.shape[1]), np.float32)
    lib().orc_cache_gather(_p(rows), _p(idx), C.c_int64(len(idx)), C.c_int(R), _p(out))
    return out


def shade_cached(rows, idx, albedo, metallic, roughness, gL=None):
    """train_brdf_crf.py:195-203 on rows of the packed cache; with gL also the gradients (g_albedo, g_metallic, g_roughness)."""
    rows = _f32(rows); albedo = _f32(albedo); metallic = _f32(metallic).reshape(-1); roughness = _f32(roughness).reshape(-1)
    idx = None if idx is None else np.ascontiguousarray(idx, dtype=np.int64)
    B, R = albedo.shape[0], (rows.shape[1] - 3) // 6
    L = np.empty((B, 3), np.float32)
    ip = _p(idx) if idx is not None else None
    lib().orc_shade_cached_fwd(_p(rows), ip, _p(albedo), _p(metallic), _p(roughness), C.c_int64(B), C.c_int(R), _p(L))
    if gL is None:
        return L
    gL = _f32(gL)
    ga = np.empty((B, 3), np.float32); gm = np.empty((B, 1), np.float32); gr = np.empty((B, 1), np.float32)
    lib().orc_shade_cached_bwd(_p(rows), ip, _p(albedo), _p(metallic), _p(roughness), _p(gL), C.c_int64(B), C.c_int(R), _p(ga), _p(gm), _p(gr))
    return L, ga, gm, gr


# ---------------------------------------------------------------- 8(f)-4 denoiser substitute (the build's own filter; no reference counterpart)
def denoise(img, normal=None, position=None, valid=None, iterations=5, sigma_l=16.0, sigma_n=128.0, sigma_p=0.05):
    img = _f32(img); H, W, _ = img.shape
    out = np.empty_like(img)
    normal = None if normal is None else _f32(normal).reshape(-1, 3)
    position = None if position is None else _f32(position).reshape(-1, 3)
    valid = None if valid is None else np.ascontiguousarray(np.asarray(valid).reshape(-1), dtype=np.uint8)
    lib().orc_denoise(_p(normal) if normal is not None else None, _p(position) if position is not None else None,
                      _p(valid) if valid is not None else None, C.c_int(H), C.c_int(W), _p(img), _p(out), C.c_int(iterations),
                      C.c_float(sigma_l), C.c_float(sigma_n), C.c_float(sigma_p))
    return out


# ---------------------------------------------------------------- a5
class VoxelSLF:
    def __init__(self, inds, radiance, voxel_min, voxel_max):
        self.inds = np.ascontiguousarray(inds, dtype=np.int64)
        self.radiance = _f32(radiance).reshape(-1, 3)
        self.H = int(self.inds.shape[0])
        self.h = C.c_void_p(lib().orc_slf_create(_p(self.inds), C.c_int(self.H), _p(self.radiance), C.c_int64(self.radiance.shape[0]),
                                                 C.c_double(float(voxel_min)), C.c_double(float(voxel_max))))

    def spatial_idx(self, x):
        x = _f32(x); idx = np.empty(x.shape[0], np.int64)
        lib().orc_slf_spatial_idx(self.h, _p(x), C.c_int64(x.shape[0]), _p(idx))
        return idx

    def forward(self, x):
        x = _f32(x); rgb = np.empty((x.shape[0], 3), np.float32)
        lib().orc_slf_forward(self.h, _p(x), C.c_int64(x.shape[0]), _p(rgb))
        return rgb

    def __del__(self):
        try:
            lib().orc_slf_destroy(self.h)
        except Exception:
            pass


class SLFEmitter:
    def __init__(self, is_emitter, emitter_radiance, emitter_area, slf, emitter_vertices=None, emitter_cdf=None):
        self.is_emitter = np.ascontiguousarray(is_emitter, dtype=np.uint8)
        self.radiance = _f32(emitter_radiance).reshape(-1, 3)
        self.area = _f32(emitter_area).reshape(-1)
        self.slf = slf
        self.h = C.c_void_p(lib().orc_emitter_create(_p(self.is_emitter), C.c_int64(self.is_emitter.shape[0]), _p(self.radiance),
                                                     _p(self.area), C.c_int64(self.area.shape[0])))
        if emitter_vertices is not None:
            self.verts = _f32(emitter_vertices).reshape(-1)
            self.cdf = _f32(emitter_cdf).reshape(-1)
            lib().orc_emitter_set_sampling(self.h, _p(self.verts), _p(self.cdf))

    def sample_emitter(self, s1, s2, position):
        s1 = _f32(s1).reshape(-1); s2 = _f32(s2).reshape(-1, 2); position = _f32(position); N = position.shape[0]
        wi = np.empty((N, 3), np.float32); pdf = np.empty((N, 1), np.float32); tri = np.empty(N, np.int64)
        lib().orc_sample_emitter(self.h, _p(s1), _p(s2), _p(position), C.c_int64(N), _p(wi), _p(pdf), _p(tri))
        return wi, pdf, tri

    def eval_emitter(self, position, triangle_idx, roughness=None, trace_roughness=0.6):
        position = _f32(position); tri = np.ascontiguousarray(triangle_idx, dtype=np.int64); B = position.shape[0]
        r = None if roughness is None else _f32(roughness).reshape(-1)
        Le = np.empty((B, 3), np.float32); pdf = np.empty((B, 1), np.float32); vn = np.empty(B, np.uint8)
        lib().orc_eval_emitter(self.h, self.slf.h, _p(position), _p(tri), _p(r), C.c_float(trace_roughness), C.c_int64(B), _p(Le), _p(pdf), _p(vn))
        return Le, pdf, vn.astype(bool)

    def __del__(self):
        try:
            lib().orc_emitter_destroy(self.h)
        except Exception:
            pass


# ---------------------------------------------------------------- a2
class Scene:
    def __init__(self, vertices, faces):
        self.vertices = _f32(vertices).reshape(-1, 3)
        self.faces = np.ascontiguousarray(faces, dtype=np.int32).reshape(-1, 3)
        self.h = C.c_void_p(lib().orc_scene_create(_p(self.vertices), C.c_int64(self.vertices.shape[0]), _p(self.faces), C.c_int64(self.faces.shape[0])))

    def ray_intersect(self, xs, ds, brute=False, counters=False):
        xs = _f32(xs); ds = _f32(ds); B = xs.shape[0]
        pos = np.empty((B, 3), np.float32); nrm = np.empty((B, 3), np.float32); uv = np.empty((B, 2), np.float32)
        idx = np.empty(B, np.int64); valid = np.empty(B, np.uint8); t = np.empty(B, np.float32)
        cnt = np.zeros(2, np.int64)
        lib().orc_ray_intersect(self.h, _p(xs), _p(ds), C.c_int64(B), C.c_int(0 if brute else 1), _p(pos), _p(nrm), _p(uv), _p(idx), _p(valid), _p(t), _p(cnt))
        self.last_t = t
        if counters:
            return pos, nrm, uv, idx, valid.astype(bool), cnt
        return pos, nrm, uv, idx, valid.astype(bool)

    def __del__(self):
        try:
            lib().orc_scene_destroy(self.h)
        except Exception:
            pass


def philox_u2(seed, idx0, stream, n):
    u = np.empty((n, 2), np.float32)
    lib().orc_philox_u2(C.c_uint64(seed), C.c_uint64(idx0), C.c_uint32(stream), C.c_int64(n), _p(u))
    return u


def bake(scene, emitter, position, normal, spp, wo=None, roughness=None, u2=None, seed=0, stream=0, pix_id=None,
         want_tri=False, counters=False, want_src=False):
    """bake_shading.py:108-123 (roughness None -> diffuse) / :168-188 (specular).
    want_tri / want_src: also return the per-sample hit triangle / radiance-table row (-2 - emitter ordinal, VoxelSLF row, or -1)."""
    position = _f32(position); normal = _f32(normal); P = position.shape[0]
    specular = roughness is not None
    wo_ = _f32(wo) if specular else None
    u2_ = None if u2 is None else _f32(u2).reshape(P * spp, 2)
    pid = None if pix_id is None else np.ascontiguousarray(pix_id, dtype=np.int32)
    out0 = np.empty((P, 3), np.float32); out1 = np.empty((P, 3), np.float32) if specular else None
    tri = np.empty(P * spp, np.int64) if want_tri else None
    src = np.empty(P * spp, np.int64) if want_src else None
    cnt = np.zeros(2, np.int64)
    lib().orc_bake_src(scene.h, emitter.h, emitter.slf.h, _p(position), _p(normal), _p(wo_), C.c_int64(P), C.c_int(spp), _p(u2_),
                       C.c_uint64(seed), C.c_uint32(stream), _p(pid), C.c_float(np.float32(roughness) if specular else -1.0),
                       _p(out0), _p(out1), _p(tri), _p(src), _p(cnt))
    res = (out0, out1) if specular else (out0,)
    if want_tri:
        res = res + (tri,)
    if want_src:
        res = res + (src,)
    if counters:
        res = res + (cnt,)
    return res


# ---------------------------------------------------------------- a9 (cfg 5)
def _mat(mat):
    return _f32(mat["albedo"]).reshape(-1, 3), _f32(mat["roughness"]).reshape(-1), _f32(mat["metallic"]).reshape(-1)


def eval_brdf(wi, wo, normal, mat):
    wi = _f32(wi); wo = _f32(wo); normal = _f32(normal); a, r, m = _mat(mat); N = wi.shape[0]
    brdf = np.empty((N, 3), np.float32); pdf = np.empty((N, 1), np.float32)
    lib().orc_eval_brdf(_p(wi), _p(wo), _p(normal), _p(a), _p(r), _p(m), C.c_int64(N), _p(brdf), _p(pdf))
    return brdf, pdf


def sample_brdf(s1, s2, wo, normal, mat):
    s1 = _f32(s1).reshape(-1); s2 = _f32(s2).reshape(-1, 2); wo = _f32(wo); normal = _f32(normal); a, r, m = _mat(mat); N = wo.shape[0]
    wi = np.empty((N, 3), np.float32); pdf = np.empty((N, 1), np.float32); w = np.empty((N, 3), np.float32)
    lib().orc_sample_brdf(_p(s1), _p(s2), _p(wo), _p(normal), _p(a), _p(r), _p(m), C.c_int64(N), _p(wi), _p(pdf), _p(w))
    return wi, pdf, w


def path_tracing_single(scene, emitter, material_fn, rays_o, rays_d, dx_du, dy_dv, spp, uniforms, radiance=None, trace_roughness=0.0, indir_depth=0):
    """utils/path_tracing.py:320-407.  material_fn(position ndarray) -> dict of ndarrays.  uniforms: the five draws.
    Returns (L (B,3), terms) where terms lets grad_radiance() form dL/d radiance analytically.
    indir_depth > 0 (with trace_roughness 0.6 and the draws of trace_indirect appended): path_tracing, :214-318 -- the same first bounce, the
    paths whose sampled hit is valid continued by trace_indirect (no gradient) and added with the BRDF weight."""
    rays_o = _f32(rays_o); rays_d = _f32(rays_d); dx_du = _f32(dx_du); dy_dv = _f32(dy_dv); B = rays_o.shape[0]
    rad = emitter.radiance if radiance is None else _f32(radiance).reshape(-1, 3)
    u = [np.ascontiguousarray(np.asarray(x, np.float32)) for x in uniforms]
    dudv = u[0].reshape(2, B, spp)
    wi = np.empty((B * spp, 3), np.float32)
    lib().orc_pt_jitter(_p(rays_d), _p(dx_du), _p(dy_dv), _p(dudv), C.c_int64(B), C.c_int(spp), _p(wi))
    pos, nrm, _, tri, vis = scene.ray_intersect(np.repeat(rays_o, spp, 0), wi)
    is_area = emitter.is_emitter[np.where(tri < 0, 0, tri)].astype(bool) & (tri >= 0)
    ord_of = np.cumsum(emitter.is_emitter.astype(np.int64)) - 1
    e0 = np.where(is_area, ord_of[np.where(tri < 0, 0, tri)], -1).astype(np.int32)
    valid_next = (~is_area) & (tri >= 0)
    sel = np.nonzero(valid_next)[0]; N = len(sel)
    path_of = np.full(B * spp, -1, np.int32); path_of[sel] = np.arange(N, dtype=np.int32)
    pos, nrm, wo = _f32(pos[sel]), _f32(nrm[sel]), _f32(-wi[sel])
    a, r, m = _mat(material_fn(pos))
    s1, s2, s1b, s2b = u[1].reshape(-1), u[2].reshape(-1, 2), u[3].reshape(-1), u[4].reshape(-1, 2)
    coef1 = np.empty((N, 3), np.float32); e1 = np.empty(N, np.int32)
    lib().orc_pt_nee(scene.h, emitter.h, _p(pos), _p(nrm), _p(wo), _p(a), _p(r), _p(m), _p(s1), _p(s2), C.c_int64(N), _p(coef1), _p(e1),
                     C.c_float(1e-6), C.c_float(1e-6), C.c_float(0.0 if indir_depth > 0 else 1e-6))     # path_tracing (:260) does not clamp the MIS denominator, path_tracing_single (:366) does
    wi_b = np.empty((N, 3), np.float32); pdf_b = np.empty(N, np.float32); w_b = np.empty((N, 3), np.float32)
    pos_n = np.empty((N, 3), np.float32); nrm_n = np.empty((N, 3), np.float32); tri_n = np.empty(N, np.int64); hit_n = np.empty(N, np.uint8)
    lib().orc_pt_brdf_trace(scene.h, _p(pos), _p(nrm), _p(wo), _p(a), _p(r), _p(m), _p(s1b), _p(s2b), C.c_int64(N), _p(wi_b), _p(pdf_b), _p(w_b),
                            _p(pos_n), _p(nrm_n), _p(tri_n), _p(hit_n), C.c_int(0), C.c_float(0.0))
    _, r_n, _ = _mat(material_fn(pos_n))
    coef2 = np.empty((N, 3), np.float32); const2 = np.empty((N, 3), np.float32); e2 = np.empty(N, np.int32); vn = np.empty(N, np.uint8)
    lib().orc_pt_brdf_finish(emitter.h, emitter.slf.h, _p(pos), _p(pos_n), _p(nrm_n), _p(wi_b), _p(tri_n), _p(r_n), _p(pdf_b), _p(w_b), C.c_int64(N),
                             _p(coef2), _p(const2), _p(e2), _p(vn), C.c_float(trace_roughness), C.c_float(1e-6))
    if indir_depth > 0:                                     # :300-316
        keep = vn.astype(bool)
        Li = trace_indirect(scene, emitter, material_fn, pos_n[keep], -wi_b[keep], nrm_n[keep], indir_depth, u[5:])
        const2[keep] = const2[keep] + w_b[keep] * Li         # (f32: one product, one sum per component)
    Lout = np.empty((B, 3), np.float32)
    lib().orc_pt_accumulate(_p(rad), _p(e0), _p(path_of), _p(e1), _p(coef1), _p(e2), _p(coef2), _p(const2), C.c_int64(B), C.c_int(spp), _p(Lout))
    terms = {"e0": e0, "path_of": path_of, "e1": e1, "coef1": coef1, "e2": e2, "coef2": coef2, "const2": const2, "B": B, "spp": spp,
             "tri_next": tri_n, "position": pos, "position_next": pos_n, "valid_next": vn.astype(bool), "brdf_weight": w_b}
    return Lout, terms


def grad_radiance(terms, gL, n_rad):
    """dL/d radiance (n_rad,3) for upstream gradient gL (B,3): scatter of gL/spp * coefficient (float64 accumulate)."""
    B, spp = terms["B"], terms["spp"]
    g = np.zeros((n_rad, 3), np.float64)
    gp = np.repeat(np.asarray(gL, np.float64) / spp, spp, 0)
    m = terms["e0"] >= 0
    np.add.at(g, terms["e0"][m], gp[m])
    act = terms["path_of"] >= 0
    j = terms["path_of"][act]
    for e, c in ((terms["e1"], terms["coef1"]), (terms["e2"], terms["coef2"])):
        mm = e[j] >= 0
        np.add.at(g, e[j][mm], gp[act][mm] * c[j][mm].astype(np.float64))
    return g.astype(np.float32)


# ---------------------------------------------------------------- refine_shading integrators
def _lobe_trace(scene, pos, nrm, wo, mat, s1, s2, lobe, rough=0.0):
    N = pos.shape[0]
    wi = np.empty((N, 3), np.float32); pdf = np.empty(N, np.float32); w = np.empty((N, 3), np.float32)
    pos_n = np.empty((N, 3), np.float32); nrm_n = np.empty((N, 3), np.float32); tri_n = np.empty(N, np.int64); hit = np.empty(N, np.uint8)
    a, r, m = mat if mat is not None else (None, None, None)
    lib().orc_pt_brdf_trace(scene.h, _p(pos), _p(nrm), _p(wo), _p(a), _p(r), _p(m), _p(s1), _p(s2), C.c_int64(N), _p(wi), _p(pdf), _p(w),
                            _p(pos_n), _p(nrm_n), _p(tri_n), _p(hit), C.c_int(lobe), C.c_float(np.float32(rough)))
    return wi, pdf, w, pos_n, nrm_n, tri_n


def _apply(Lacc, rows, throughput, radiance, e, coef, cst, weight):
    """the HIP pt_apply kernel in numpy f32, same operation order"""
    v = np.zeros_like(coef) if cst is None else cst.copy()
    m = e >= 0
    v[m] = v[m] + coef[m] * radiance[e[m]]
    v = throughput * v
    v[np.isnan(v)] = 0
    Lacc[rows] += v
    if weight is not None:
        throughput *= weight


def path_tracing(scene, emitter, material_fn, rays_o, rays_d, dx_du, dy_dv, spp, indir_depth, uniforms, radiance=None):
    """utils/path_tracing.py:214-318 (render.py's integrator).  uniforms: the 5 draws of the first bounce, then trace_indirect's 4 per bounce."""
    return path_tracing_single(scene, emitter, material_fn, rays_o, rays_d, dx_du, dy_dv, spp, uniforms, radiance, trace_roughness=0.6, indir_depth=indir_depth)


def trace_indirect(scene, emitter, material_fn, position, wo, normal, indir_depth, uniforms):
    """utils/path_tracing.py:409-502"""
    position = _f32(position); wo = _f32(wo); normal = _f32(normal); B = position.shape[0]
    u = [np.ascontiguousarray(np.asarray(x, np.float32)) for x in uniforms]
    Lacc = np.zeros((B, 3), np.float32); rows = np.arange(B); thr = np.ones((B, 3), np.float32)
    rad = emitter.radiance
    mat = None
    for depth in range(indir_depth):
        N = position.shape[0]
        if N == 0:
            break
        if depth == 0:
            mat = _mat(material_fn(position))
        a, r, m = mat
        s1, s2 = u.pop(0).reshape(-1), u.pop(0).reshape(-1, 2)
        assert s1.shape[0] == N and s2.shape[0] == N, f"trace_indirect depth {depth}: {N} paths but draws for {s1.shape[0]} (the recorded uniforms belong to another path set)"
        coef1 = np.empty((N, 3), np.float32); e1 = np.empty(N, np.int32)
        lib().orc_pt_nee(scene.h, emitter.h, _p(position), _p(normal), _p(wo), _p(a), _p(r), _p(m), _p(s1), _p(s2), C.c_int64(N), _p(coef1), _p(e1),
                         C.c_float(1e-12), C.c_float(1e-12), C.c_float(0.0))
        _apply(Lacc, rows, thr, rad, e1, coef1, None, None)
        s1b, s2b = u.pop(0).reshape(-1), u.pop(0).reshape(-1, 2)
        wi, pdf, w, pos_n, nrm_n, tri_n = _lobe_trace(scene, position, normal, wo, mat, s1b, s2b, 0)
        mat_next = _mat(material_fn(pos_n))
        coef2 = np.empty((N, 3), np.float32); const2 = np.empty((N, 3), np.float32); e2 = np.empty(N, np.int32); vn = np.empty(N, np.uint8)
        lib().orc_pt_brdf_finish(emitter.h, emitter.slf.h, _p(position), _p(pos_n), _p(nrm_n), _p(wi), _p(tri_n), _p(mat_next[1]), _p(pdf), _p(w), C.c_int64(N),
                                 _p(coef2), _p(const2), _p(e2), _p(vn), C.c_float(0.6), C.c_float(1e-12))
        _apply(Lacc, rows, thr, rad, e2, coef2, const2, w)
        keep = vn.astype(bool)
        rows, thr = rows[keep], np.ascontiguousarray(thr[keep])
        position, wo, normal = _f32(pos_n[keep]), _f32(-wi[keep]), _f32(nrm_n[keep])
        mat = tuple(np.ascontiguousarray(t[keep]) for t in mat_next)
    return Lacc


def path_tracing_det(scene, emitter, material_fn, positions, wis, normals, triangle_idxs, spp, indir_depth, uniforms, roughness=None):
    """utils/path_tracing.py:50-124 (roughness None -> diffuse) and :126-212 (specular).  Returns Lout or (L0out, L1out)."""
    positions = _f32(positions); wis = _f32(wis); normals = _f32(normals)
    u = [np.ascontiguousarray(np.asarray(x, np.float32)) for x in uniforms]
    sel = np.nonzero(np.asarray(triangle_idxs) != -1)[0]; P = len(sel)
    outs = [np.zeros_like(positions) for _ in range(1 if roughness is None else 2)]
    if P:
        position = _f32(np.repeat(positions[sel], spp, 0)); normal = _f32(np.repeat(normals[sel], spp, 0)); wo = _f32(np.repeat(-wis[sel], spp, 0))
        s2 = u.pop(0).reshape(-1, 2)
        wi, _, w, pos_n, nrm_n, tri_n = _lobe_trace(scene, position, normal, wo, None, None, s2, 1 if roughness is None else 2, 0.0 if roughness is None else roughness)
        mat_next = material_fn(pos_n)
        Le, _, vn = emitter.eval_emitter(pos_n, tri_n, mat_next["roughness"], 0.6)
        total = Le.copy()
        total[vn] += trace_indirect(scene, emitter, material_fn, pos_n[vn], -wi[vn], nrm_n[vn], indir_depth, u)
        if roughness is None:
            outs[0][sel] = total.reshape(P, spp, 3).mean(1, dtype=np.float64).astype(np.float32)
        else:
            for k in range(2):
                outs[k][sel] = (w[:, k:k + 1] * total).reshape(P, spp, 3).mean(1, dtype=np.float64).astype(np.float32)
    return outs[0] if roughness is None else tuple(outs)
