"""BaseBRDF samplers (reference: model/brdf.py:20-136) as single fused gfx950 kernels."""
import torch
import torch.nn as nn

from .. import _lib as L


def diffuse_sampler(sample2, normal):
    """cosine-weighted direction around `normal` (model/brdf.py:20-34): the direction sample_diffuse draws"""
    return BaseBRDF().sample_diffuse(sample2, normal)[0]


def specular_sampler(sample2, roughness, wo, normal):
    """GGX half-vector sample reflected about it (model/brdf.py:36-59): the direction sample_specular draws"""
    return BaseBRDF().sample_specular(sample2, wo, normal, roughness)[0]


class BaseBRDF(nn.Module):
    """Parameter-free BRDF used by bake_shading (bake_shading.py:79)."""

    def __init__(self):
        super().__init__()

    def forward(self):
        pass

    def eval_diffuse(self, wi, normal):
        """(brdf Bx3, pdf Bx1), both relu(n.wi) / pi (model/brdf.py:70-76)"""
        import math
        wi = L.require_gpu(wi, torch.float32, "wi").reshape(-1, 3)
        normal = L.require_gpu(normal, torch.float32, "normal").reshape(-1, 3)
        pdf = ((normal * wi).sum(-1, keepdim=True)).relu() / math.pi
        return pdf.expand(wi.shape[0], 3), pdf

    def eval_specular(self, wi, wo, normal, roughness):
        """(brdf_spec0 Bx1, brdf_spec1 Bx1, pdf Bx1): the GGX lobe split by the two Schlick terms, F = ks F0 + F1 (model/brdf.py:90-110);
        D, G and the Fresnel terms are the HIP helpers of iris_amd.utils.ops.  No backward pass (the reference's version carries gradient to
        roughness, wi and normal): inputs that require grad raise instead of being cut off silently."""
        from ..utils import ops
        L.no_autograd("BaseBRDF.eval_specular", wi, wo, normal, roughness)
        wi = L.require_gpu(wi, torch.float32, "wi").reshape(-1, 3)
        wo = L.require_gpu(wo, torch.float32, "wo").reshape(-1, 3)
        normal = L.require_gpu(normal, torch.float32, "normal").reshape(-1, 3)
        half = torch.nn.functional.normalize(wi + wo, dim=-1)
        dot = lambda a, b: (a * b).sum(-1, keepdim=True).relu()
        n_l, n_v, v_h, n_h = dot(wi, normal), dot(wo, normal), dot(wo, half), dot(normal, half)
        if not torch.is_tensor(roughness):
            roughness = torch.tensor(float(roughness), device=wi.device)
        rough = roughness.detach().to(torch.float32).reshape(-1, 1) if roughness.numel() > 1 else roughness.detach().to(torch.float32).reshape(1, 1)
        D = ops.D_GGX(n_h, rough)
        pdf = D / (4 * v_h.clamp_min(1e-4)) * n_h
        DG = D * ops.G_Smith(n_v, n_l, rough)
        f0, f1 = ops.fresnelSchlick_sep(v_h)
        return DG * f0 / 4.0 * n_l, DG * f1 / 4.0 * n_l, pdf

    def sample_diffuse(self, sample2, normal):
        """Cosine-weighted direction, pdf = relu(n.wi)/pi, weight = 1 (model/brdf.py:78-88).  No backward pass."""
        L.no_autograd("BaseBRDF.sample_diffuse", sample2, normal)
        sample2 = L.require_gpu(sample2, torch.float32, "sample2").reshape(-1, 2)
        normal = L.require_gpu(normal, torch.float32, "normal").reshape(-1, 3)
        B = sample2.shape[0]
        wi = torch.empty(B, 3, device=sample2.device, dtype=torch.float32)
        pdf = torch.empty(B, 1, device=sample2.device, dtype=torch.float32)
        w = torch.empty(B, 3, device=sample2.device, dtype=torch.float32)
        with torch.cuda.device(sample2.device):
            L.check(L.lib().iris_sample_diffuse(L.ptr(sample2), L.ptr(normal), B, L.ptr(wi), L.ptr(pdf), L.ptr(w), L.stream()))
        return wi, pdf, w

    def sample_specular(self, sample2, wo, normal, roughness):
        """GGX half-vector sampling and the two Fresnel-split weights (model/brdf.py:112-136).
        roughness: python float or 0-d tensor (bake_shading.py:161 iterates a linspace), or one value per sample (Bx1).
        No backward pass (the reference's weights carry gradient to roughness; its sampled direction does not: model/brdf.py:46 uses .data)."""
        L.no_autograd("BaseBRDF.sample_specular", sample2, wo, normal, roughness)
        sample2 = L.require_gpu(sample2, torch.float32, "sample2").reshape(-1, 2)
        wo = L.require_gpu(wo, torch.float32, "wo").reshape(-1, 3)
        normal = L.require_gpu(normal, torch.float32, "normal").reshape(-1, 3)
        B = sample2.shape[0]
        each = None
        if isinstance(roughness, torch.Tensor):
            if roughness.numel() == 1:
                roughness = float(roughness.detach().float().cpu().item())
            else:                                           # one roughness per sample (what sample_brdf hands to specular_sampler)
                each = L.require_gpu(roughness.detach(), torch.float32, "roughness").reshape(-1)
                if each.shape[0] != B:
                    raise L.IrisError(f"sample_specular: {each.shape[0]} roughness values for {B} samples")
        wi = torch.empty(B, 3, device=sample2.device, dtype=torch.float32)
        pdf = torch.empty(B, 1, device=sample2.device, dtype=torch.float32)
        w0 = torch.empty(B, 1, device=sample2.device, dtype=torch.float32)
        w1 = torch.empty(B, 1, device=sample2.device, dtype=torch.float32)
        with torch.cuda.device(sample2.device):
            if each is not None:
                L.check(L.lib().iris_sample_specular_v(L.ptr(sample2), L.ptr(wo), L.ptr(normal), L.ptr(each), B, L.ptr(wi), L.ptr(pdf), L.ptr(w0), L.ptr(w1), L.stream()))
            else:
                L.check(L.lib().iris_sample_specular(L.ptr(sample2), L.ptr(wo), L.ptr(normal), roughness, B, L.ptr(wi), L.ptr(pdf), L.ptr(w0), L.ptr(w1), L.stream()))
        return wi, pdf, w0, w1

    @staticmethod
    def _mat(mat, dev):
        albedo = L.require_gpu(mat["albedo"].detach(), torch.float32, "mat['albedo']").reshape(-1, 3)
        rough = L.require_gpu(mat["roughness"].detach(), torch.float32, "mat['roughness']").reshape(-1)
        metal = L.require_gpu(mat["metallic"].detach(), torch.float32, "mat['metallic']").reshape(-1)
        return albedo, rough, metal

    def eval_brdf(self, wi, wo, normal, mat):
        """BRDF value (already multiplied by NoL) and the 50/50 diffuse + GGX sampling pdf (model/brdf.py:138-175).
        mat: {'albedo' Bx3, 'roughness' Bx1, 'metallic' Bx1}.  Returns brdf Bx3, pdf Bx1.  No backward pass: a material that requires
        grad raises (path_tracing_single differentiates w.r.t. the emitter radiance only, as train_emitter.py does)."""
        L.no_autograd("BaseBRDF.eval_brdf", wi, wo, normal, *[mat[k] for k in ("albedo", "roughness", "metallic")])
        wi = L.require_gpu(wi, torch.float32, "wi").reshape(-1, 3)
        wo = L.require_gpu(wo, torch.float32, "wo").reshape(-1, 3)
        normal = L.require_gpu(normal, torch.float32, "normal").reshape(-1, 3)
        albedo, rough, metal = self._mat(mat, wi.device)
        B = wi.shape[0]
        brdf = torch.empty(B, 3, device=wi.device, dtype=torch.float32)
        pdf = torch.empty(B, 1, device=wi.device, dtype=torch.float32)
        with torch.cuda.device(wi.device):
            L.check(L.lib().iris_eval_brdf(L.ptr(wi), L.ptr(wo), L.ptr(normal), L.ptr(albedo), L.ptr(rough), L.ptr(metal), B, L.ptr(brdf), L.ptr(pdf), L.stream()))
        return brdf, pdf

    def sample_brdf(self, sample1, sample2, wo, normal, mat):
        """importance sampling: diffuse lobe where sample1 > 0.5, GGX otherwise; returns wi Bx3, pdf Bx1, brdf/pdf Bx3
        (model/brdf.py:177-210).  No backward pass."""
        L.no_autograd("BaseBRDF.sample_brdf", wo, normal, *[mat[k] for k in ("albedo", "roughness", "metallic")])
        sample1 = L.require_gpu(sample1, torch.float32, "sample1").reshape(-1)
        sample2 = L.require_gpu(sample2, torch.float32, "sample2").reshape(-1, 2)
        wo = L.require_gpu(wo, torch.float32, "wo").reshape(-1, 3)
        normal = L.require_gpu(normal, torch.float32, "normal").reshape(-1, 3)
        albedo, rough, metal = self._mat(mat, wo.device)
        B = wo.shape[0]
        wi = torch.empty(B, 3, device=wo.device, dtype=torch.float32)
        pdf = torch.empty(B, 1, device=wo.device, dtype=torch.float32)
        w = torch.empty(B, 3, device=wo.device, dtype=torch.float32)
        with torch.cuda.device(wo.device):
            L.check(L.lib().iris_sample_brdf(L.ptr(sample1), L.ptr(sample2), L.ptr(wo), L.ptr(normal), L.ptr(albedo), L.ptr(rough), L.ptr(metal), B,
                                             L.ptr(wi), L.ptr(pdf), L.ptr(w), L.stream()))
        return wi, pdf, w


class _TcnnParams(nn.Module):
    """Stands where the reference has `self.mlp = tcnn.NetworkWithInputEncoding(...)`: it owns the ONE flat float32 parameter tensor tiny-cuda-nn's
    torch module registers as `params`, so the state-dict key is `mlp.params` as in the reference's checkpoints (refine_shading.py:84-89 strips the
    'material.' prefix and calls load_state_dict)."""

    def __init__(self, n):
        super().__init__()
        self.params = nn.Parameter(torch.zeros(n, dtype=torch.float32), requires_grad=False)


class NGPBRDF(BaseBRDF):
    """Hash-grid material network, INFERENCE ONLY (model/brdf.py:213-260; the reference loads it from a checkpoint and freezes it in refine_shading.py:83-92
    and train_emitter.py:67-77): HashGrid{32 levels x 2 features, 2^19 entries, base 16, x1.3} -> FullyFusedMLP{64 x 2, ReLU} -> sigmoid, as two HIP
    kernels (level-major gathers; the perceptron on the matrix cores, iris_amd/csrc/iris_ngp.h).  tiny-cuda-nn is third party and CUDA only: its published
    algorithm is implemented, parity unpinned (oracle/ngp_torch.py).  No backward pass: training the material network is train_brdf_crf's job (out of scope)."""

    # roughness = sigmoid(.) * 0.98 + 0.02 (model/brdf.py:258): never below 0.02 for finite network outputs.  path_tracing_single uses its SECOND evaluation of the
    # network (mat_next, utils/path_tracing.py:392) only for the test roughness > trace_roughness = 0.0 (model/emitter.py:209), whose outcome this bound decides:
    # iris_amd.utils.path_tracing skips that evaluation when the network declares a bound above trace_roughness (same outputs, bit for bit).
    roughness_min = 0.02

    def __init__(self, voxel_min, voxel_max):
        super().__init__()
        self.voxel_min, self.voxel_max = float(voxel_min), float(voxel_max)
        self.mlp = _TcnnParams(int(L.lib().iris_ngp_n_params()))
        self._h, self._h_key = None, None

    def _handle(self, device):
        import ctypes as C
        p = self.mlp.params
        key = (str(device), p.data_ptr(), p._version)
        if self._h is None or self._h_key != key:
            self._free()
            host = p.detach().to("cpu", torch.float32).contiguous()
            idx = L.device_index(device)
            h = C.c_void_p()
            L.check(L.lib().iris_ngp_create(C.c_void_p(host.data_ptr()), host.numel(), self.voxel_min, self.voxel_max, idx, C.byref(h)))
            self._h, self._h_key = h, key
        return self._h

    def _free(self):
        h = self.__dict__.get("_h")
        if h is not None:
            self.__dict__["_h"] = None              # (not through nn.Module.__setattr__: at interpreter shutdown its helpers may be gone)
            try:
                L.lib().iris_ngp_destroy(h)
            except Exception:     # noqa  (interpreter shutdown)
                pass

    def __del__(self):
        self._free()

    def __getstate__(self):
        """copy.deepcopy / pickle: the native handle belongs to THIS object (a raw pointer: a copy would free it twice); the copy builds its own on first use"""
        state = self.__dict__.copy()
        state["_h"], state["_h_key"] = None, None
        return state

    def forward(self, position):
        """position Bx3 (world space) -> {'albedo': Bx3, 'roughness': Bx1 in [0.02, 1], 'metallic': Bx1}  (model/brdf.py:243-260)"""
        L.no_autograd("NGPBRDF.forward", position)
        position = L.require_gpu(position, torch.float32, "position")
        shape = position.shape[:-1]
        pos = position.reshape(-1, 3)
        N, dev = pos.shape[0], pos.device
        albedo = torch.empty(N, 3, device=dev); rough = torch.empty(N, device=dev); metal = torch.empty(N, device=dev)
        with torch.cuda.device(dev):
            L.check(L.lib().iris_ngp_forward(self._handle(dev), L.ptr(pos), N, L.ptr(albedo), L.ptr(rough), L.ptr(metal), L.stream()))
        return {"albedo": albedo.reshape(*shape, 3), "roughness": rough.reshape(*shape, 1), "metallic": metal.reshape(*shape, 1)}


def load_ngpbrdf(voxel_min, voxel_max, ckpt):
    """The reference's loading sequence (refine_shading.py:83-92): NGPBRDF(voxel_min, voxel_max), the checkpoint's 'state_dict' entries under 'material.'
    with the prefix stripped, load_state_dict, frozen."""
    net = NGPBRDF(voxel_min, voxel_max)
    try:
        state = torch.load(ckpt, map_location="cpu")["state_dict"]
    except Exception:     # noqa  (a Lightning checkpoint carries hyper-parameters and callback states that the tensors-only unpickler of recent torch refuses;
        state = torch.load(ckpt, map_location="cpu", weights_only=False)["state_dict"]     #  the reference loads its own checkpoints with the full unpickler)
    weight = {k.replace("material.", ""): v for k, v in state.items() if "material." in k}
    net.load_state_dict(weight)
    for p in net.parameters():
        p.requires_grad = False
    return net
