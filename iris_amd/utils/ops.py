"""Shading helpers of the hot path that are callable on their own (reference: utils/ops.py).

Unlike the reference's torch expressions these are HIP kernels WITHOUT a backward pass: an input that requires grad (with grad mode on) raises
IrisError instead of silently cutting the graph.  The differentiable pieces of this package are the autograd.Functions of
utils/path_tracing.py (path_tracing_single w.r.t. the emitter radiance) and utils/shading_cache.py (the BRDF trainer's shading combine)."""
import torch

from .. import _lib as L


def lerp_specular(specular, roughness):
    """Interpolate the 6 baked specular levels by roughness (utils/ops.py:99-118).
    specular: Bx6x3, roughness: Bx1 in [0.02,1.0] -> Bx3."""
    L.no_autograd("lerp_specular", specular, roughness)      # (the differentiable combine is utils/shading_cache.shade_cached)
    specular = L.require_gpu(specular, torch.float32, "specular")
    roughness = L.require_gpu(roughness, torch.float32, "roughness").reshape(-1)
    B, R, _ = specular.shape
    out = torch.empty(B, 3, device=specular.device, dtype=torch.float32)
    with torch.cuda.device(specular.device):
        L.check(L.lib().iris_lerp_specular(L.ptr(specular), L.ptr(roughness), B, R, L.ptr(out), L.stream()))
    return out


# ----------------------------------------------------------------------------------------------------------------------
# The small helpers of utils/ops.py:12-96.  Inside the bake and path-tracing kernels the same device functions are fused; these
# entry points exist so that code written against the reference's `from utils.ops import *` keeps working on GPU tensors.
# ----------------------------------------------------------------------------------------------------------------------
def _flat(*xs):
    """broadcast GPU tensors / Python scalars against each other -> (contiguous flat f32 tensors, broadcast shape)"""
    dev = next(x.device for x in xs if torch.is_tensor(x))
    ts = [x if torch.is_tensor(x) else torch.tensor(float(x), device=dev) for x in xs]
    ts = torch.broadcast_tensors(*[L.require_gpu(t, torch.float32, "argument") for t in ts])
    return [t.contiguous().reshape(-1) for t in ts], ts[0].shape


def _ggx(op, a, b=None, c=None, two=False):
    args = [x for x in (a, b, c) if x is not None]
    L.no_autograd("utils.ops GGX / Fresnel helper", *args)
    flat, shape = _flat(*args)
    flat += [None] * (3 - len(flat))
    n = flat[0].numel()
    out = torch.empty(n, device=flat[0].device, dtype=torch.float32)
    out2 = torch.empty_like(out) if two else None
    with torch.cuda.device(out.device):
        L.check(L.lib().iris_ggx_terms(op, L.ptr(flat[0]), L.ptr(flat[1]) if flat[1] is not None else None, L.ptr(flat[2]) if flat[2] is not None else None, n,
                                       L.ptr(out), L.ptr(out2) if two else None, L.stream()))
    return (out.reshape(shape), out2.reshape(shape)) if two else out.reshape(shape)


def get_normal_space(normal):
    """normal (...,3) unit -> (...,3,3) with columns tangent, bitangent, normal (utils/ops.py:12-30)"""
    L.no_autograd("get_normal_space", normal)
    n = L.require_gpu(normal, torch.float32, "normal")
    flat = n.reshape(-1, 3).contiguous()
    out = torch.empty(flat.shape[0], 3, 3, device=n.device, dtype=torch.float32)
    with torch.cuda.device(n.device):
        L.check(L.lib().iris_get_normal_space(L.ptr(flat), flat.shape[0], L.ptr(out), L.stream()))
    return out.reshape(*n.shape[:-1], 3, 3)


def angle2xyz(theta, phi):
    """spherical -> unit vector (...,3) (utils/ops.py:32-44)"""
    L.no_autograd("angle2xyz", theta, phi)
    (t, p), shape = _flat(theta, phi)
    out = torch.empty(t.numel(), 3, device=t.device, dtype=torch.float32)
    with torch.cuda.device(t.device):
        L.check(L.lib().iris_angle2xyz(L.ptr(t), L.ptr(p), t.numel(), L.ptr(out), L.stream()))
    return out.reshape(*shape, 3)


def double_sided(V, N):
    """flip N (...,3) towards the viewing direction V, IN PLACE as the reference does, and return it (utils/ops.py:85-96)"""
    L.no_autograd("double_sided", V, N)
    V = L.require_gpu(V, torch.float32, "V")
    L.require_gpu(N, torch.float32, "N")
    v = V.expand_as(N).reshape(-1, 3).contiguous()
    work = N if N.is_contiguous() else N.contiguous()
    with torch.cuda.device(N.device):
        L.check(L.lib().iris_double_sided(L.ptr(v), L.ptr(work), v.shape[0], L.stream()))
    if work is not N:
        N.copy_(work)
    return N


def D_GGX(cos_h, eta):
    """GGX normal distribution, eta = roughness (utils/ops.py:77-82)"""
    return _ggx(0, cos_h, eta)


def G1_GGX_Schlick(NoV, eta):
    """1 / (NoV (1 - k) + k), k = (eta + 1)^2 / 8 (utils/ops.py:46-54)"""
    return _ggx(1, NoV, eta)


def G_Smith(NoV, NoL, eta):
    """Smith shadowing-masking divided by NoV NoL (utils/ops.py:56-63)"""
    return _ggx(2, NoV, NoL, eta)


def fresnelSchlick(VoH, F0):
    """F0 + (1 - F0)(1 - VoH)^5 (utils/ops.py:65-68)"""
    return _ggx(3, VoH, F0)


def fresnelSchlick_sep(VoH):
    """the two terms of Schlick's approximation: (1 - x, x), x = (1 - VoH)^5 (utils/ops.py:70-73)"""
    return _ggx(4, VoH, two=True)
