"""iris_amd -- MI355X-native implementation of the bake_shading hot path of facebookresearch/iris.

The package mirrors the reference's call surface for that path and nothing else:

    reference                               this package
    ------------------------------------    ---------------------------------------------
    utils.path_tracing.ray_intersect        iris_amd.utils.path_tracing.ray_intersect
    mitsuba.load_dict({... mesh ...})       iris_amd.utils.path_tracing.Scene / load_scene
    model.brdf.BaseBRDF                     iris_amd.model.brdf.BaseBRDF
    model.slf.VoxelSLF                      iris_amd.model.slf.VoxelSLF
    model.emitter.SLFEmitter                iris_amd.model.emitter.SLFEmitter
    utils.ops.lerp_specular                 iris_amd.utils.ops.lerp_specular
    utils.dataset.real_ldr.get_direction/   iris_amd.utils.dataset.real_ldr.*
      to_world, synthetic_ldr.get_rays      iris_amd.utils.dataset.synthetic_ldr.*
    bake_shading.py (CLI + loop)            iris_amd.bake_shading

All compute goes through the C ABI of ``libiris_hip.so`` (include/iris_hip.h); there is no CPU fallback.
"""
__version__ = "0.1.0"
