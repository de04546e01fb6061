"""ctypes binding of libiris_hip.so (include/iris_hip.h).  Fails loudly when the library is missing."""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IRIS_HIP_LIB") or os.path.join(_HERE, "libiris_hip.so")   # IRIS_HIP_LIB: A/B builds of the same ABI

BVH_DEFAULT, BVH4_F32, BVH4_Q8 = 0, 1, 3
BAKE_AUTO, BAKE_PIXEL_PER_WAVE, BAKE_TILE_SORTED = 0, 1, 2


class IrisError(RuntimeError):
    pass


class SceneInfo(C.Structure):
    _fields_ = [("n_vertices", C.c_int64), ("n_triangles", C.c_int64), ("layout", C.c_int32), ("n_nodes", C.c_int32),
                ("node_bytes", C.c_int32), ("tri_bytes", C.c_int32), ("depth", C.c_int32), ("n_leaf_records", C.c_int32),
                ("sah_cost", C.c_float), ("build_seconds", C.c_float)]


_P, _I64, _I32, _F, _D, _U64, _U32 = C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_double, C.c_uint64, C.c_uint32

# name -> argtypes (restype is int unless listed in _RESTYPE)
PROTOTYPES = {
    "iris_scene_create": [_P, _I64, _P, _I64, _I32, C.POINTER(_P)],
    "iris_debug_scene_create": [_P, _I64, _P, _I64, _I32, _I32, C.POINTER(_P)],
    "iris_scene_destroy": [_P],
    "iris_scene_get_info": [_P, C.POINTER(SceneInfo)],
    "iris_slf_create": [_P, _I32, _P, _I64, _D, _D, _I32, C.POINTER(_P)],
    "iris_slf_create_dev": [_P, _I32, _P, _I64, _D, _D, _I32, C.POINTER(_P), _P],
    "iris_slf_set_radiance": [_P, _P, _I64, _P],
    "iris_slf_destroy": [_P],
    "iris_emitter_create": [_P, _I64, _P, _I64, _P, _I64, _P, _P, _I32, C.POINTER(_P)],
    "iris_emitter_set_radiance": [_P, _P, _I64, _P],
    "iris_emitter_destroy": [_P],
    "iris_raygen_real": [_P, _P, _I32, _I32, _I32, _P, _P, _P, _P, _P],
    "iris_raygen_synthetic": [_F, _P, _I32, _I32, _I32, _P, _P, _P, _P, _P],
    "iris_intersect": [_P, _P, _P, _I64, _P, _P, _P, _P, _P, _P],
    "iris_sample_diffuse": [_P, _P, _I64, _P, _P, _P, _P],
    "iris_sample_specular": [_P, _P, _P, _F, _I64, _P, _P, _P, _P, _P],
    "iris_sample_specular_v": [_P, _P, _P, _P, _I64, _P, _P, _P, _P, _P],
    "iris_slf_lookup": [_P, _P, _I64, _P, _P, _P],
    "iris_eval_emitter": [_P, _P, _P, _P, _P, _F, _I64, _P, _P, _P, _P],
    "iris_bake_workspace_bytes": [_I64, _I32, _I32],
    "iris_bake_diffuse": [_P, _P, _P, _P, _P, _I64, _I32, _P, _U64, _U32, _P, _P, _P, _P, _U64, _P],
    "iris_bake_specular": [_P, _P, _P, _P, _P, _P, _F, _I64, _I32, _P, _U64, _U32, _P, _P, _P, _P, _P, _U64, _P],
    "iris_debug_bake_diffuse": [_P, _P, _P, _P, _P, _I64, _I32, _P, _U64, _U32, _P, _P, _P, _P, _P, _I32, _P, _U64, _P],
    "iris_debug_bake_specular": [_P, _P, _P, _P, _P, _P, _F, _I64, _I32, _P, _U64, _U32, _P, _P, _P, _P, _P, _P, _I32, _P, _U64, _P],
    "iris_debug_set": [C.c_char_p, C.c_longlong],
    "iris_debug_build_flags": [],
    "iris_debug_source_hash": [],
    "iris_bake_view": [_P, _P, _P, _P, _P, _P, _P, _I64, _I32, _P, _P, _P, _U64, _P, _P, _P, _U64, _P],
    "iris_lerp_specular": [_P, _P, _I64, _I32, _P, _P],
    "iris_unstripe_maps": [_P, _I32, _I32, _I64, _I32, _I32, _I32, _P, _P],
    "iris_get_normal_space": [_P, _I64, _P, _P],
    "iris_double_sided": [_P, _P, _I64, _P],
    "iris_angle2xyz": [_P, _P, _I64, _P, _P],
    "iris_ggx_terms": [_I32, _P, _P, _P, _I64, _P, _P, _P],
    "iris_philox_u2": [_U64, _U64, _U32, _I64, _P, _P],
    "iris_ngp_n_params": [],
    "iris_ngp_create": [_P, _I64, C.c_double, C.c_double, C.c_int, _P],
    "iris_ngp_forward": [_P, _P, _I64, _P, _P, _P, _P],
    "iris_ngp_destroy": [_P],
    "iris_debug_ngp_encode": [_P, _P, _I64, _P, _P],
    "iris_sample_emitter": [_P, _P, _P, _P, _I64, _P, _P, _P, _P],
    "iris_eval_brdf": [_P, _P, _P, _P, _P, _P, _I64, _P, _P, _P],
    "iris_sample_brdf": [_P, _P, _P, _P, _P, _P, _P, _I64, _P, _P, _P, _P],
    "iris_pt_jitter": [_P, _P, _P, _P, _I64, _I32, _P, _P],
    "iris_pt_primary_emit": [_P, _P, _I64, _P, _P, _P],
    "iris_pt_nee": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _P, _P, _F, _F, _F, _P],
    "iris_pt_brdf_trace": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _P, _P, _P, _P, _P, _P, _P, _I32, _F, _P],
    "iris_pt_brdf_finish": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _P, _P, _P, _P, _F, _F, _P],
    "iris_pt_primary": [_P, _P, _P, _P, _P, _P, _P, _I64, _I32, _P, _P, _P, _P, _P, _P, _P, _P],
    "iris_pt_apply": [_P, _P, _P, _P, _P, _P, _P, _P, _I64, _I32, _P],
    "iris_pt_bounce": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _P, _P, _F, _F, _F, _P, _P, _P, _P, _P, _P, _P, _P],
    "iris_pt_compact_workspace_bytes": [_I64],
    "iris_pt_compact": [_P, _I64, _I32, _P, _P, C.c_uint32, _I32, _P, _P, _I32, _P, _P, _P, _P, C.c_uint64, _P],
    "iris_pt_accumulate_fwd": [_P, _P, _P, _P, _P, _P, _P, _P, _I64, _I32, _P, _P],
    "iris_pt_accumulate_bwd": [_P, _P, _P, _P, _P, _P, _P, _I64, _I32, _P, _P],
    "iris_slf_scatter_add": [_P, _P, _P, _I64, _P, _P, _P],
    "iris_voxel_histogram": [_P, _I64, _D, _D, _I32, _P, _P],
    "iris_scatter_add_rows": [_P, _P, _I64, _I64, _P, _P, _P],
    "iris_cache_row_floats": [_I32],
    "iris_cache_pack": [_P, _P, _P, _I64, _I32, _P, _P],
    "iris_cache_gather": [_P, _P, _I64, _I32, _P, _P],
    "iris_shade_cached_fwd": [_P, _P, _P, _P, _P, _I64, _I32, _P, _P],
    "iris_shade_cached_bwd": [_P, _P, _P, _P, _P, _P, _I64, _I32, _P, _P, _P, _P],
    "iris_denoise_workspace_bytes": [_I32, _I32],
    "iris_denoise": [_P, _P, _P, _I32, _I32, _I32, _P, _P, _I32, _F, _F, _F, _P, _U64, _P],
    "iris_bake_tile_max_spp": [],
    "iris_last_error": [],
    "iris_version": [],
}
_RESTYPE = {"iris_scene_destroy": None, "iris_slf_destroy": None, "iris_emitter_destroy": None,
            "iris_last_error": C.c_char_p, "iris_version": C.c_char_p, "iris_debug_build_flags": C.c_char_p, "iris_debug_source_hash": C.c_char_p, "iris_ngp_n_params": C.c_int64, "iris_ngp_destroy": None, "iris_bake_workspace_bytes": C.c_uint64, "iris_pt_compact_workspace_bytes": C.c_uint64, "iris_denoise_workspace_bytes": C.c_uint64}

_lib = None


def lib():
    """The loaded library.  Raises (never falls back) when libiris_hip.so has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise IrisError(f"{LIB_PATH} is missing: build it with `make -C iris_amd/csrc` or "
                            "`python -c 'import __graft_entry__ as g; g.build()'`. There is no CPU fallback.")
        l = C.CDLL(LIB_PATH)
        for name, argtypes in PROTOTYPES.items():
            fn = getattr(l, name)  # AttributeError if the symbol is not exported
            fn.argtypes = argtypes
            fn.restype = _RESTYPE.get(name, C.c_int)
        _lib = l
    return _lib


def source_hash():
    """sha256 over the sources the BAKE kernels are compiled from (device / traversal / tile / bake headers, the host file that sizes and launches them, the BVH
    builder, the ABI headers) and the compiler flags embedded in the LOADED library (iris_debug_build_flags: -fno-slp-vectorize, the scheduler strategy and the
    -D tuning overrides are worth several per cent): stamps counter profiles (tools/pmc_summary.py) so that bench.py can refuse counters that were taken on other
    kernels or on another build of the same sources.  (The headers of the other stages -- material network, shading cache, denoiser, path-tracing stages -- are
    not part of the stamp: bake_view_kernel does not include them.)"""
    import hashlib
    h = hashlib.sha256()
    h.update(lib().iris_debug_build_flags())
    root = os.path.dirname(_HERE)
    files = [os.path.join("iris_amd", "csrc", f) for f in ("bvh_build.cpp", "bvh_build.h", "iris_bake.h", "iris_device.h", "iris_hip.hip", "iris_tile.h", "iris_trace.h")]
    files += [os.path.join("include", "iris_hip.h"), os.path.join("include", "iris_hip_debug.h")]
    for f in files:
        h.update(f.encode()); h.update(open(os.path.join(root, f), "rb").read())
    return h.hexdigest()[:16]


def device_index(device):
    """The HIP device ordinal a torch device names: `torch.device('cuda')` (no index) is torch's CURRENT device on this rank, not device 0 -- one process per
    GPU sets it once (torch.cuda.set_device(LOCAL_RANK)) and then passes index-less devices around."""
    import torch
    device = torch.device(device)
    if device.type != "cuda":
        raise IrisError(f"{device}: the HIP path needs a GPU device (there is no CPU fallback)")
    return device.index if device.index is not None else torch.cuda.current_device()


def build_id():
    """What identifies the ARITHMETIC of the LOADED library for a resumable run: its version string, the compiler flags and the hash of ALL the sources it was
    compiled from -- path-tracing stages, material network, shading cache and denoiser included --, both embedded in the binary by the Makefile (a sampler,
    traversal, network or filter change alters bits: maps of two builds must not be mixed by --resume).  Nothing is read from csrc/ on disk: a stale library
    keeps its own id, an installation without sources works.  A library built around the Makefile carries no hash and is refused."""
    h = lib().iris_debug_source_hash().decode()
    if h == "unknown" or len(h) != 16:
        raise IrisError("libiris_hip.so carries no source hash (built without iris_amd/csrc/Makefile): a resumable run cannot be keyed on it")
    return lib().iris_version().decode() + "|" + lib().iris_debug_build_flags().decode() + "|" + h


class StageTimer:
    """Diagnostics (bench.py extras, tools/bench_pt_single.py): HIP events around the stages of a multi-kernel call.  `with L.StageTimer() as t:` makes every
    `L.mark(name)` inside record an event on the stream that is current THERE; `t.ms()` -> {stage: milliseconds between its mark and the previous mark on the
    same stream} (stages on different streams overlap: the figures do not add up to the wall time).  Not active (no events, no cost) otherwise."""
    active = None

    def __enter__(self):
        self.events = []          # (stream id, name, event)
        StageTimer.active = self
        return self

    def __exit__(self, *a):
        StageTimer.active = None

    def ms(self):
        import torch
        torch.cuda.synchronize()
        out, last = {}, {}
        for sid, name, ev in self.events:
            if sid in last and name is not None:
                out[name] = out.get(name, 0.0) + last[sid].elapsed_time(ev)
            last[sid] = ev
        return out


def mark(name=None):
    """StageTimer: end of stage `name` (None: start of a sequence) on the current stream"""
    t = StageTimer.active
    if t is not None:
        import torch
        ev = torch.cuda.Event(enable_timing=True)
        st = torch.cuda.current_stream()
        ev.record(st)
        t.events.append((st.cuda_stream, name, ev))


def debug_set(key, value):
    """iris_debug_set (include/iris_hip_debug.h): process-wide tuning option; value < 0 restores the default."""
    check(lib().iris_debug_set(key.encode(), int(value)))


def check(rc):
    if rc != 0:
        raise IrisError(lib().iris_last_error().decode("utf-8", "replace"))


def require_gpu(t, dtype, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise IrisError(f"{name}: expected a tensor on a HIP device (got {getattr(t, 'device', type(t))}); "
                        "iris_amd has no CPU path")
    if t.dtype != dtype:
        raise IrisError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def no_autograd(fn, *tensors):
    """The entry points behind `fn` are raw HIP kernels writing into fresh buffers: they have no backward pass.  The reference's counterparts
    are differentiable torch ops, so handing them a tensor that requires grad (with grad mode on) would silently train with zero / None
    gradients -- raise instead.  Call under torch.no_grad() or pass .detach()ed tensors to state that no gradient is wanted."""
    if torch.is_grad_enabled() and any(torch.is_tensor(t) and t.requires_grad for t in tensors):
        raise IrisError(f"{fn}: an input requires grad, but this is a HIP kernel without a backward pass (the reference's torch op is differentiable); "
                        "wrap the call in torch.no_grad() or detach() the inputs")


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def host_f32(a):
    import numpy as np
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().numpy()
    return np.ascontiguousarray(a, dtype=np.float32)
