/*
 * iris_hip_debug.h -- diagnostics entry points of libiris_hip.so.  NOT part of the drop-in boundary (include/iris_hip.h):
 * nothing in the reference's call surface maps to these.  They exist for the parity tests, the roofline accounting of
 * bench.py and kernel-choice experiments, and are exported by the same library.
 */
#ifndef IRIS_HIP_DEBUG_H
#define IRIS_HIP_DEBUG_H

#include "iris_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* iris_scene_create with an explicit node layout (IRIS_BVH4_F32: the 128-B f32-plane nodes the Q8 layout is A/B-tested against). */
IRIS_API int iris_debug_scene_create(const float *verts, int64_t nv, const int32_t *faces, int64_t nf, int device, int layout,
                            iris_scene **out);

/* Kernel variants of the fused bake; AUTO = what iris_bake_* run (TILE_SORTED when a workspace is given). */
enum { IRIS_BAKE_AUTO = 0, IRIS_BAKE_PIXEL_PER_WAVE = 1, IRIS_BAKE_TILE_SORTED = 2 };

/* iris_bake_diffuse / iris_bake_specular plus:
 *   src_next (P*spp) int64, nullable: per sample, the radiance-table row eval_emitter read: -2 - emitter ordinal for an emitter
 *            triangle, the VoxelSLF row for the radiance cache, -1 for empty space or a miss (parity bookkeeping: a sample whose
 *            (tri_next, src_next) differs from the oracle's is a discrete "flip", everything else differs by rounding only);
 *   stats    nullable device uint64[20]: when given, an INSTRUMENTED build of the kernel (4 waves/SIMD) adds
 *            {rays, node visits, triangle tests, wave-level node steps, wave-level triangle steps, rays whose stack exceeded
 *            8 / 12 / 16 entries, tail sum (pixel-per-wave), node visits / wave-level node steps while a tile drains,
 *            node visits with node index < 21 / 85 / 341 / 1365, wave-level node steps in which >= 32 of the lanes at a node sit at
 *            the SAME node of the same octant table / the lanes that share it / the steps in which all of them do, wave-level node steps
 *            that could not take the branch-free pushes (a lane's stack within three entries of the LDS part's end), 0};
 *   variant  IRIS_BAKE_*.  All variants and the instrumented builds return identical bits. */
IRIS_API int iris_debug_bake_diffuse(const iris_scene *, const iris_emitter *, const iris_slf *, const float *pos, const float *nrm,
                            int64_t P, int spp, const float *u2, uint64_t seed, uint32_t stream_id, const int32_t *pix_id,
                            float *Ld, int64_t *tri_next, int64_t *src_next, uint64_t *stats, int variant, void *workspace,
                            uint64_t workspace_bytes, iris_stream_t);
IRIS_API int iris_debug_bake_specular(const iris_scene *, const iris_emitter *, const iris_slf *, const float *pos, const float *nrm,
                             const float *wo, float roughness, int64_t P, int spp, const float *u2, uint64_t seed,
                             uint32_t stream_id, const int32_t *pix_id, float *Ls0, float *Ls1, int64_t *tri_next, int64_t *src_next,
                             uint64_t *stats, int variant, void *workspace, uint64_t workspace_bytes, iris_stream_t);

/* Process-wide tuning options (value < 0 restores the default).  Results never depend on them.
 *   "bvh_max_leaf" 1..7 (4)      "bvh_tri_cost_x100": SAH cost of a triangle test relative to a node visit, in % (70)
 *   "bvh_presplit_x10": triangles whose box is longer than this many tenths of the median triangle's are referenced through several
 *                       clipped boxes (80; 0 = off)
 *   "phase_min" lanes (12)      "tile_target_rays" (4096)      "tiles_per_block" (4)
 *   "pt_tile_min": smallest batch the path-tracing stages route through the tile-sorted kernel
 * NOT thread-safe: plain process-wide globals that iris_scene_create / the launches read.  Set them before creating handles, from one thread. */
IRIS_API int iris_debug_set(const char *key, long long value);
/* NGPBRDF: the hash-grid encoding alone (tests compare it bit for bit with the restatement; the perceptron behind it only to a tolerance):
 * features of N <= 2^20 positions as the kernels hand them over, feat[level * N + i] = the level's two half features of point i (one uint32). */
IRIS_API int iris_debug_ngp_encode(const iris_ngp *, const float *position, int64_t N, uint32_t *feat, iris_stream_t);
/* The compiler flags this library was built with (iris_amd/csrc/Makefile embeds them): part of the stamp that ties a counter profile to a build. */
IRIS_API const char *iris_debug_build_flags(void);
/* sha256 (16 hex digits) over EVERY file this library was compiled from (kernels of all stages, host code, ABI headers), taken by the Makefile at build
 * time: identifies the arithmetic of the loaded binary without looking at the sources on disk ("unknown" for a build that did not go through the Makefile). */
IRIS_API const char *iris_debug_source_hash(void);

#ifdef __cplusplus
}
#endif
#endif /* IRIS_HIP_DEBUG_H */
