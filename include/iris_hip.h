/*
 * iris_hip.h -- C ABI of libiris_hip.so, the MI355X (gfx950) implementation of the bake_shading hot path
 * of facebookresearch/iris.
 *
 * The reference has no FFI of its own: the path is ordinary Python calls into Mitsuba/OptiX and ATen.  The
 * entry points below are therefore exactly the calls a binding of that path needs, one per reference
 * callable (cited per function as reference-file:line).  See INTEGRATION.md for the ctypes stub a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - Every pointer is a DEVICE pointer to contiguous memory owned by the caller (a torch tensor), except
 *     the inputs of the *_create functions, which are HOST pointers copied during the call.
 *   - f32 = IEEE binary32; idx arrays are int64 (the reference's torch.long); masks are uint8 (torch.bool).
 *   - Work is enqueued on the caller's hipStream_t and is asynchronous; nothing here synchronises.
 *   - Handles are immutable after creation (except iris_emitter_set_radiance) and may be shared by streams.
 *   - Return value 0 = OK; non-zero = error, message from iris_last_error() (thread local).
 *   - No C++ exception crosses this boundary.  There is no CPU fallback: without a HIP device every
 *     compute entry point fails with IRIS_ERR_HIP.
 */
#ifndef IRIS_HIP_H
#define IRIS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define IRIS_API __attribute__((visibility("default")))
#else
#define IRIS_API
#endif

typedef void *iris_stream_t; /* hipStream_t */

typedef struct iris_scene iris_scene;     /* triangle mesh + BVH            (mitsuba scene, bake_shading.py:55-61) */
typedef struct iris_slf iris_slf;         /* VoxelSLF                       (model/slf.py:16-39)                   */
typedef struct iris_emitter iris_emitter; /* SLFEmitter's emitter tables    (model/emitter.py:134-173)             */
typedef struct iris_ngp iris_ngp;         /* NGPBRDF's hash grid + MLP      (model/brdf.py:213-260)                */

enum { IRIS_OK = 0, IRIS_ERR_ARG = 1, IRIS_ERR_HIP = 2, IRIS_ERR_BUILD = 3 };

#define IRIS_RAY_EPSILON 8.940696716308594e-05f /* mitsuba.math.RayEpsilon (float32) = 1500 * 2^-24 */

/* BVH node layouts (reported by iris_scene_get_info; iris_scene_create always builds IRIS_BVH4_Q8) */
enum { IRIS_BVH_DEFAULT = 0, IRIS_BVH4_F32 = 1 /* 128-B nodes, f32 planes (A/B baseline, iris_hip_debug.h) */, IRIS_BVH4_Q8 = 3 /* 64-B nodes, 8-bit planes */ };

typedef struct {
    int64_t n_vertices, n_triangles;
    int32_t layout;        /* IRIS_BVH4_F32 | IRIS_BVH4_Q8 */
    int32_t n_nodes;       /* wide nodes */
    int32_t node_bytes;    /* bytes per node record (64: quantised planes); the default layout keeps EIGHT records per node, one per ray octant
                              (children in that octant's front-to-back order, planes pre-swapped near / far): table = 8 * n_nodes * node_bytes */
    int32_t tri_bytes;     /* stride of a leaf-triangle record (64: component-major (p0.x,p1.x,p2.x,id) (y...) (z...), csrc/iris_trace.h) */
    int32_t depth;         /* wide-tree depth */
    int32_t n_leaf_records;/* leaf-triangle records: n_triangles + the extra references of split long triangles */
    float   sah_cost;      /* cost of the WIDE tree as the collapse minimises it: sum over wide nodes of A_node + sum over leaves of A_leaf * records *
                              tri_cost (0.7), areas relative to the root, duplicated references of split triangles included (csrc/bvh_build.cpp) */
    float   build_seconds;
} iris_scene_info;

/* ---- handles ----------------------------------------------------------------------------------------- */

/* mitsuba.load_dict({'type':'scene','shape_id':{...}})  (bake_shading.py:55-61).  Host SAH build + upload. */
IRIS_API int iris_scene_create(const float *verts, int64_t nv, const int32_t *faces, int64_t nf, int device, iris_scene **out);
IRIS_API void iris_scene_destroy(iris_scene *);
IRIS_API int iris_scene_get_info(const iris_scene *, iris_scene_info *out);

/* VoxelSLF(mask, voxel_min, voxel_max) + load_state_dict (model/slf.py:18-39, model/emitter.py:144-147).
 * inds: (H,H,H) int64 [z][y][x], -1 = empty; radiance: (kv,3).  voxel_min/max are the python floats stored in
 * vslf.npz (slf_bake.py:140-145); the denominator is float32(voxel_max - voxel_min) as in torch-CPU. */
IRIS_API int iris_slf_create(const int64_t *inds, int H, const float *radiance, int64_t kv, double voxel_min, double voxel_max,
                    int device, iris_slf **out);
/* the same from DEVICE buffers (inds int64 (H,H,H), radiance (kv,3) on `device`): the pre-bake stages build the grid on the GPU (slf_bake.py:116-118) */
IRIS_API int iris_slf_create_dev(const int64_t *inds_dev, int H, const float *radiance_dev, int64_t kv, double voxel_min, double voxel_max,
                    int device, iris_slf **out, iris_stream_t);
/* refresh the radiance rows from a DEVICE pointer (kv,3): mean pooling rewrites them (slf_bake.py:138, slf_refine.py:106) */
IRIS_API int iris_slf_set_radiance(iris_slf *, const float *radiance_dev, int64_t kv, iris_stream_t);
IRIS_API void iris_slf_destroy(iris_slf *);

/* SLFEmitter.__init__ (model/emitter.py:149-173): is_emitter (nf) bool, radiance (n_rad,3) indexed by EMITTER
 * ORDINAL (model/emitter.py:203), area (k).  emitter_idx / emitter_pdf=1/k are derived here. */
/* verts (k,3,3) = emitter_vertices and cdf (k) = emitter_cdf (model/emitter.py:155,170) are only needed by
 * iris_sample_emitter / iris_pt_nee and may be NULL. */
IRIS_API int iris_emitter_create(const uint8_t *is_emitter, int64_t nf, const float *radiance, int64_t n_rad, const float *area,
                        int64_t k, const float *verts, const float *cdf, int device, iris_emitter **out);
/* SLFEmitterLearn.radiance is a parameter (model/emitter.py:268): refresh the device copy from a DEVICE pointer. */
IRIS_API int iris_emitter_set_radiance(iris_emitter *, const float *radiance_dev, int64_t n_rad, iris_stream_t);
IRIS_API void iris_emitter_destroy(iris_emitter *);

/* ---- a1: ray generation ------------------------------------------------------------------------------ */
/* get_direction + to_world (utils/dataset/real_ldr.py:49-83; ScanNet++ utils/dataset/scannetpp/dataset.py:202-215).
 * K (9) and c2w (12) are passed BY VALUE from the host (row-major).  rays_o, rays_d: (H*W,3).  ray_diff: also
 * dxdu, dydv and un-normalised rays_d. */
IRIS_API int iris_raygen_real(const float K[9], const float c2w[12], int H, int W, int ray_diff, float *rays_o, float *rays_d,
                     float *dxdu, float *dydv, iris_stream_t);
/* get_ray_directions + get_rays (utils/dataset/synthetic_ldr.py:21-57) */
IRIS_API int iris_raygen_synthetic(float focal, const float c2w[12], int H, int W, int ray_diff, float *rays_o, float *rays_d,
                          float *dxdu, float *dydv, iris_stream_t);

/* ---- a2: ray_intersect(scene, xs, ds)  (utils/path_tracing.py:17-48) ---------------------------------- */
/* Any output pointer may be NULL.  Miss: idx=-1, valid=0, pos/nrm/uv=0. */
IRIS_API int iris_intersect(const iris_scene *, const float *xs, const float *ds, int64_t B, float *pos, float *nrm, float *uv,
                   int64_t *idx, uint8_t *valid, iris_stream_t);

/* ---- a3/a4: BaseBRDF.sample_diffuse / sample_specular (model/brdf.py:78-88, :112-136) ------------------- */
IRIS_API int iris_sample_diffuse(const float *u2, const float *normal, int64_t B, float *wi, float *pdf, float *weight,
                        iris_stream_t);
IRIS_API int iris_sample_specular(const float *u2, const float *wo, const float *normal, float roughness, int64_t B, float *wi,
                         float *pdf, float *w0, float *w1, iris_stream_t);
/* the same with one roughness per sample (model/brdf.py:36-59 specular_sampler is called with a Bx1 roughness by sample_brdf) */
IRIS_API int iris_sample_specular_v(const float *u2, const float *wo, const float *normal, const float *roughness, int64_t B,
                           float *wi, float *pdf, float *w0, float *w1, iris_stream_t);

/* ---- a5: VoxelSLF.spatial_idx/forward (model/slf.py:41-70), SLFEmitter.eval_emitter (model/emitter.py:180-221) */
IRIS_API int iris_slf_lookup(const iris_slf *, const float *x, int64_t B, int64_t *idx /*nullable*/, float *rgb /*nullable*/,
                    iris_stream_t);
/* roughness: NULL <=> the reference's roughness=None; otherwise (B) f32. */
IRIS_API int iris_eval_emitter(const iris_emitter *, const iris_slf *, const float *position, const int64_t *triangle_idx,
                      const float *roughness, float trace_roughness, int64_t B, float *Le, float *emit_pdf,
                      uint8_t *valid_next, iris_stream_t);

/* ---- a3..a7 fused: the bake loop body (bake_shading.py:108-123 diffuse, :168-188 specular) -------------- */
/* pos,nrm[,wo]: (P,3) records of the valid pixels.  u2: (P*spp,2) explicit uniforms in the reference's order
 * (row = pixel*spp + sample), or NULL -> in-kernel Philox4x32-10 keyed by (seed, pix_id[p]*spp+s, stream);
 * pix_id (P) int32 nullable (defaults to p) makes the sample set independent of how pixels are sharded.
 * Ld/Ls0/Ls1: (P,3) = mean over spp of Le, Le*g0, Le*g1.  tri_next (P*spp) int64, nullable: the per-sample hit triangle
 * (the reference's `triangle_idx` of ray_intersect, bake_shading.py:117 / :180).
 * workspace: device scratch of iris_bake_workspace_bytes() bytes for the tile-sorted kernel (8 tile-queue counters, the
 * workgroups' per-ray slots -- sampled direction, then hit -- and their traversal-stack overflow slabs; contents are
 * scratch, nothing survives the call); NULL (or spp > iris_bake_tile_max_spp()) selects the simpler pixel-per-wave kernel.  Both give
 * identical bits. */
IRIS_API uint64_t iris_bake_workspace_bytes(int64_t P, int spp, int specular);   /* 0 if spp > iris_bake_tile_max_spp() */
IRIS_API int iris_bake_tile_max_spp(void);   /* largest spp the tile-sorted / view kernels take (5120: the LDS ray list) */
IRIS_API int iris_bake_diffuse(const iris_scene *, const iris_emitter *, const iris_slf *, const float *pos, const float *nrm,
                      int64_t P, int spp, const float *u2, uint64_t seed, uint32_t stream_id, const int32_t *pix_id,
                      float *Ld, int64_t *tri_next, void *workspace, uint64_t workspace_bytes, iris_stream_t);
IRIS_API int iris_bake_specular(const iris_scene *, const iris_emitter *, const iris_slf *, const float *pos, const float *nrm,
                       const float *wo, float roughness, int64_t P, int spp, const float *u2, uint64_t seed,
                       uint32_t stream_id, const int32_t *pix_id, float *Ls0, float *Ls1, int64_t *tri_next,
                       void *workspace, uint64_t workspace_bytes, iris_stream_t);

/* All lobes of one view (bake_shading.py:93-204) in ONE launch with one tile queue: n_lobes <= 8; roughness[l] < 0 selects the
 * diffuse lobe (out1[l] may be NULL), otherwise the specular lobe of that roughness; spp[l] <= iris_bake_tile_max_spp(); Philox uniforms only.
 * roughness / spp / stream_ids / out0 / out1 are HOST arrays.  Outputs are bit-identical to the per-lobe entry points. */
IRIS_API int iris_bake_view(const iris_scene *, const iris_emitter *, const iris_slf *, const float *pos, const float *nrm, const float *wo,
                   const int32_t *pix_id, int64_t P, int n_lobes, const float *roughness, const int32_t *spp,
                   const uint32_t *stream_ids, uint64_t seed, float *const *out0, float *const *out1, void *workspace,
                   uint64_t workspace_bytes, iris_stream_t);

/* ---- (e) multi-GPU: one view sharded over the ranks of a node in interleaved row stripes (no reference counterpart: bake_shading.py:41 pins one
 * device).  After the single collective, `gathered` holds (world, n_maps, n_max, 3): rank r's maps at its local pixels (its rows -- stripe s of
 * stripe_rows rows belongs to rank s % world -- ascending, row-major, padded to n_max); `full` (n_maps, H*W, 3) receives the image order. */
IRIS_API int iris_unstripe_maps(const float *gathered, int world, int n_maps, int64_t n_max, int H, int W, int stripe_rows, float *full, iris_stream_t);

/* ---- a10: lerp_specular (utils/ops.py:99-118): specular (B,R,3), roughness (B) -> (B,3) ------------------ */
IRIS_API int iris_lerp_specular(const float *specular, const float *roughness, int64_t B, int R, float *out, iris_stream_t);

/* ---- a3 / a4 helpers (utils/ops.py:12-96) as calls of their own; the bake and path-tracing kernels use the same device functions fused.
 * get_normal_space: normal (B,3) -> (B,3,3), columns tangent, bitangent, normal.  double_sided: N (B,3) flipped in place towards V.
 * angle2xyz: theta, phi (B) -> unit (B,3).  ggx_terms, element-wise over B (the caller broadcasts):
 *   op 0 D_GGX(a = cos_h, b = roughness)   1 G1_GGX_Schlick(a = NoV, b = roughness)   2 G_Smith(a = NoV, b = NoL, c = roughness)
 *   op 3 fresnelSchlick(a = VoH, b = F0)   4 fresnelSchlick_sep(a = VoH) -> out = 1 - x, out2 = x with x = (1 - VoH)^5 */
IRIS_API int iris_get_normal_space(const float *normal, int64_t B, float *out, iris_stream_t);
IRIS_API int iris_double_sided(const float *V, float *N, int64_t B, iris_stream_t);
IRIS_API int iris_angle2xyz(const float *theta, const float *phi, int64_t B, float *out, iris_stream_t);
IRIS_API int iris_ggx_terms(int op, const float *a, const float *b, const float *c, int64_t B, float *out, float *out2, iris_stream_t);

/* ---- a9 (BASELINE cfg 5): path_tracing_single (utils/path_tracing.py:320-407) ------------------------------ */
/* Building blocks of the call surface: */
/* SLFEmitter.sample_emitter (model/emitter.py:224-255): s1 (N), s2 (N,2), position (N,3) -> wi (N,3), pdf (N), tri (N) */
IRIS_API int iris_sample_emitter(const iris_emitter *, const float *s1, const float *s2, const float *position, int64_t N, float *wi,
                        float *pdf, int64_t *tri, iris_stream_t);
/* BaseBRDF.eval_brdf (model/brdf.py:138-175): albedo (N,3), roughness (N), metallic (N) -> brdf (N,3), pdf (N) */
IRIS_API int iris_eval_brdf(const float *wi, const float *wo, const float *normal, const float *albedo, const float *roughness,
                   const float *metallic, int64_t N, float *brdf, float *pdf, iris_stream_t);
/* BaseBRDF.sample_brdf (model/brdf.py:177-210) -> wi (N,3), pdf (N), brdf/pdf weight (N,3) */
IRIS_API int iris_sample_brdf(const float *s1, const float *s2, const float *wo, const float *normal, const float *albedo,
                     const float *roughness, const float *metallic, int64_t N, float *wi, float *pdf, float *weight,
                     iris_stream_t);
/* Stages.  The material network is third party (tiny-cuda-nn) and is evaluated by the caller between them, exactly where
 * the reference calls material_net (:355, :392).  L is linear in emitter.radiance, the only tensor that receives gradient:
 * stages emit (emitter ordinal, rgb coefficient) pairs, accumulate_fwd gathers, accumulate_bwd scatter-adds. */
/* :338-340  wi = normalize(rays_d + dx_du*(u-0.5) + dy_dv*(v-0.5)); dudv = the reference's torch.rand(2,B,spp,1) */
IRIS_API int iris_pt_jitter(const float *rays_d, const float *dxdu, const float *dydv, const float *dudv, int64_t B, int spp, float *wi,
                   iris_stream_t);
/* :344  eval_emitter(position, wi, triangle_idx) with the radiance gather factored out: e0 = emitter ordinal or -1 */
IRIS_API int iris_pt_primary_emit(const iris_emitter *, const int64_t *tri, int64_t N, int32_t *e0, uint8_t *valid_next, iris_stream_t);
/* :357-382  emitter sampling + visibility ray + geometry term + eval_brdf + MIS -> term1 = coef1 * radiance[e1].
 * g_eps / pdf_eps / mis_eps: the clamp_min constants of the caller: 1e-6,1e-6,1e-6 in path_tracing_single (:370,:373,:379),
 * 1e-12,1e-12,none(<=0) in trace_indirect (:445,:448,:453). */
IRIS_API int iris_pt_nee(const iris_scene *, const iris_emitter *, const float *pos, const float *nrm, const float *wo, const float *albedo,
                const float *roughness, const float *metallic, const float *s1, const float *s2, int64_t N, float *coef1, int32_t *e1,
                float g_eps, float pdf_eps, float mis_eps, iris_stream_t);
/* :338-344 as ONE launch (the un-compacted mode of path_tracing_single): jitter, closest hit of rays_o[b] + t * wi, the primary hit's emitter ordinal, which paths
 * continue.  dudv (2,B,spp) as iris_pt_jitter; wi, wo = -wi, pos, nrm (B*spp,3) as iris_pt_jitter / iris_intersect give them; e0, valid_next as iris_pt_primary_emit;
 * path_of[i] = i where the path continues (a hit that is not an emitter), -1 otherwise. */
IRIS_API int iris_pt_primary(const iris_scene *, const iris_emitter *, const float *rays_o, const float *rays_d, const float *dxdu, const float *dydv,
                    const float *dudv, int64_t B, int spp, float *wi, float *wo, float *pos, float *nrm, int32_t *e0, uint8_t *valid_next, int32_t *path_of,
                    iris_stream_t);
/* :384-391  lobe sampling + next intersection.  lobe 0: sample_brdf(s1,s2,wo,normal,mat); lobe 1: sample_diffuse(s2,normal)
 * (path_tracing_det_diff :93-97, weight 1); lobe 2: sample_specular(s2,wo,normal,lobe_roughness) (path_tracing_det_spec :174-178),
 * weight = (g0,g1,0). */
IRIS_API int iris_pt_brdf_trace(const iris_scene *, const float *pos, const float *nrm, const float *wo, const float *albedo,
                       const float *roughness, const float *metallic, const float *s1, const float *s2, int64_t N, float *wi,
                       float *pdf, float *weight, float *pos_next, float *nrm_next, int64_t *tri_next, uint8_t *valid,
                       int lobe, float lobe_roughness, iris_stream_t);
/* trace_indirect's two tracing stages of a bounce (utils/path_tracing.py:434-471) in ONE launch: iris_pt_nee with the draws (s1, s2) and iris_pt_brdf_trace
 * (lobe 0: sample_brdf) with the draws (s1b, s2b) on the same N paths -- the visibility ray and the BRDF ray of a path leave from the same point; their rays are
 * sorted by direction together and traced by the same persistent lanes.  Outputs as the two calls' (same bits). */
IRIS_API int iris_pt_bounce(const iris_scene *, const iris_emitter *, const float *pos, const float *nrm, const float *wo, const float *albedo,
                   const float *roughness, const float *metallic, const float *s1, const float *s2, const float *s1b, const float *s2b, int64_t N,
                   float *coef1, int32_t *e1, float g_eps, float pdf_eps, float mis_eps, float *wi, float *pdf, float *weight, float *pos_next,
                   float *nrm_next, int64_t *tri_next, uint8_t *valid, iris_stream_t);
/* :394-404  eval_emitter(..., mat_next.roughness, 0.0) + geometry term + MIS -> term2 = coef2 * radiance[e2] + const2.
 * roughness_next may be NULL: "every roughness exceeds trace_roughness" -- the only use of mat_next in the reference's path_tracing_single is the test
 * roughness > trace_roughness = 0.0 (model/emitter.py:209), and NGPBRDF's roughness is sigmoid * 0.98 + 0.02 >= 0.02 (model/brdf.py:258): a caller that knows its
 * material network's lower bound skips the second network evaluation (same outputs, bit for bit). */
IRIS_API int iris_pt_brdf_finish(const iris_emitter *, const iris_slf *, const float *pos, const float *pos_next, const float *nrm_next,
                        const float *wi, const int64_t *tri_next, const float *roughness_next, const float *pdf, const float *weight,
                        int64_t N, float *coef2, float *const2, int32_t *e2, uint8_t *valid_next /*nullable*/, float trace_roughness,
                        float g_eps, iris_stream_t);
/* trace_indirect's accumulation (utils/path_tracing.py:454-456,:462,:484-486): L[rows[i]] += throughput[i] * (coef[i]*radiance[e[i]]
 * + cst[i]) with NaN -> 0, then throughput[i] *= weight[i].  rows, throughput, e/coef, cst, weight are each nullable. */
IRIS_API int iris_pt_apply(float *L, const int32_t *rows, float *throughput, const float *radiance, const int32_t *e, const float *coef,
                  const float *cst, const float *weight, int64_t N, int nan_to_zero, iris_stream_t);
/* trace_indirect's end of a bounce (utils/path_tracing.py:488-501: `position = position[valid_next]` and its siblings): the rows with keep[i] != 0 are moved to the front
 * of the output arrays IN ORDER, as boolean indexing does -- n3 arrays of 3 floats per row (bit k of negate3: dst3[k] = -src3[k], the reference's `wo = -wi`), n1 of one
 * float, ni of one int32 (each at most 6; the pointer arrays are host arrays of device pointers, read during the call); *count (device int32) receives the number of rows
 * kept.  Outputs must hold N rows and must not alias the inputs.  Two launches on the stream, no host round trip, no index tensors.  workspace: device scratch of
 * iris_pt_compact_workspace_bytes(N) bytes. */
IRIS_API uint64_t iris_pt_compact_workspace_bytes(int64_t N);
IRIS_API int iris_pt_compact(const uint8_t *keep, int64_t N, int n3, const float *const *src3, float *const *dst3, uint32_t negate3,
                    int n1, const float *const *src1, float *const *dst1, int ni, const int32_t *const *srci, int32_t *const *dsti,
                    int32_t *count, void *workspace, uint64_t workspace_bytes, iris_stream_t);
/* :406  L (B,3) = mean over spp; path_of (B*spp) maps a path to its row in the compacted stage arrays (or -1).
 * radiance: the (n_rad,3) parameter tensor itself (model/emitter.py:268). */
IRIS_API int iris_pt_accumulate_fwd(const float *radiance, const int32_t *e0, const int32_t *path_of, const int32_t *e1, const float *coef1,
                           const int32_t *e2, const float *coef2, const float *const2, int64_t B, int spp, float *L,
                           iris_stream_t);
/* dL/d radiance: g_radiance (n_rad,3), zero-initialised by the caller, += scatter of gL (B,3) */
IRIS_API int iris_pt_accumulate_bwd(const float *gL, const int32_t *e0, const int32_t *path_of, const int32_t *e1, const float *coef1,
                           const int32_t *e2, const float *coef2, int64_t B, int spp, float *g_radiance, iris_stream_t);

/* ---- 8(f)-2: pooling builders of the stages that write vslf.npz / emitter.pth ------------------------------- */
/* VoxelSLF.scatter_add (model/slf.py:56-61): radiance_acc (kv,3) f32 += rgb, count (kv) int64 += 1 at spatial_idx(x) */
IRIS_API int iris_slf_scatter_add(const iris_slf *, const float *x, const float *rgb, int64_t B, float *radiance_acc, int64_t *count,
                         iris_stream_t);
/* occupancy histogram of slf_bake.py:104-110: hist (H^3) f32, index x + y*H + z*H*H */
IRIS_API int iris_voxel_histogram(const float *x, int64_t B, double voxel_min, double voxel_max, int H, float *hist, iris_stream_t);
/* per-triangle sums of extract_emitter_ldr.py:90-95: out (F,3) += values (B,3) at idx (B); count (F) f32 nullable */
IRIS_API int iris_scatter_add_rows(const float *values, const int64_t *idx, int64_t B, int64_t F, float *out, float *count, iris_stream_t);

/* ---- 8(f)-3: the shading cache resident in HBM + the BRDF trainer's shading combine --------------------------- */
/* Floats per packed row for R roughness levels (R <= 8): [d.rgb 0 | per level: spec0.rgb spec1.rgb], padded to 16 B (40 for R=6).
 * The VALUES are the reference's (pixels, 3+6R) table of utils/dataset/scannetpp/dataset.py:359-377; the order is ours. */
IRIS_API int iris_cache_row_floats(int R);
/* pack the 1+2R maps (n,3) of one view into rows (n, iris_cache_row_floats(R)); spec0 / spec1 are HOST arrays of R device pointers */
IRIS_API int iris_cache_pack(const float *diffuse, const float *const *spec0, const float *const *spec1, int64_t n, int R, float *rows,
                    iris_stream_t);
/* the loader's batch slice (dataset.py:409-414): out (B, 3+6R) = [diffuse | specular0 (R,3) | specular1 (R,3)] of rows[idx] (idx NULL = identity) */
IRIS_API int iris_cache_gather(const float *rows, const int64_t *idx, int64_t B, int R, float *out, iris_stream_t);
/* train_brdf_crf.py:195-203 fused with the slice: L (B,3) = kd*diffuse + ks*lerp_specular(spec0,rough) + lerp_specular(spec1,rough),
 * kd = albedo*(1-metallic), ks = 0.04*(1-metallic) + albedo*metallic; albedo (B,3), metallic (B), roughness (B) */
IRIS_API int iris_shade_cached_fwd(const float *rows, const int64_t *idx, const float *albedo, const float *metallic, const float *roughness,
                          int64_t B, int R, float *L, iris_stream_t);
/* its gradient for an incoming gL (B,3): g_albedo (B,3), g_metallic (B), g_roughness (B); any output may be NULL */
IRIS_API int iris_shade_cached_bwd(const float *rows, const int64_t *idx, const float *albedo, const float *metallic, const float *roughness,
                          const float *gL, int64_t B, int R, float *g_albedo, float *g_metallic, float *g_roughness, iris_stream_t);

/* ---- 8(f)-4: denoiser substitute for mitsuba.OptixDenoiser (bake_shading.py:81,129,198-200) --------------------- */
/* Variance-guided edge-avoiding a-trous filter over (H,W,3) maps, guided by the primary hits of the view: normal / position (H*W,3) f32
 * and valid (H*W) u8 (each nullable: no guide of that kind).  in / out: HOST arrays of n_maps device pointers (in[m] == out[m] allowed);
 * maps are filtered four at a time sharing the geometric weights.  iterations in [1,8] (stride 2^i); defaults used by the Python
 * mirror: 5, sigma_l 16, sigma_n 128, sigma_p 0.05 (tools/tune_denoise.py).  Not bit-comparable with OptiX (closed): judged on PSNR against a high-spp bake. */
IRIS_API uint64_t iris_denoise_workspace_bytes(int H, int W);
IRIS_API int iris_denoise(const float *normal, const float *position, const uint8_t *valid, int H, int W, int n_maps, const float *const *in,
                 float *const *out, int iterations, float sigma_l, float sigma_n, float sigma_p, void *workspace, uint64_t workspace_bytes,
                 iris_stream_t);

/* ---- the material network of the refine / emitter-training stages, inference only ----------------------- */
/* NGPBRDF (model/brdf.py:213-260; loaded and frozen at refine_shading.py:83-92, train_emitter.py:67-77): tiny-cuda-nn
 * NetworkWithInputEncoding(3, 5, HashGrid{n_levels 32, 2 features, log2_hashmap_size 19, base 16, per_level_scale 1.3},
 * FullyFusedMLP{64 neurons, 2 hidden layers, ReLU}) + sigmoid.  params: HOST float32[n_params] = the `mlp.params` tensor of the reference's
 * state dict ([MLP weights 64x64, 64x64, 16x64 row-major | grid tables level by level, 2 features per entry]); n_params must equal
 * iris_ngp_n_params().  tiny-cuda-nn is third party: the published algorithm is implemented, parity unpinned (oracle/ngp_torch.py). */
IRIS_API int64_t iris_ngp_n_params(void);
IRIS_API int iris_ngp_create(const float *params, int64_t n_params, double voxel_min, double voxel_max, int device, iris_ngp **out);
/* forward(position): position (N,3) f32 world space -> albedo (N,3), roughness (N) in [0.02,1], metallic (N), all f32 device pointers.
 * Outputs lie on the HALF grid as the reference's do (model/brdf.py:255: the network's half output -> sigmoid -> half -> .float(); roughness * 0.98 + 0.02
 * in f32 afterwards).  The handle owns the feature buffer the two kernels of a call exchange (2^20 points x 128 B); calls on ONE handle from different
 * streams or host threads are serialised by the library (a device-side event wait, a host mutex); different handles are independent. */
IRIS_API int iris_ngp_forward(const iris_ngp *, const float *position, int64_t N, float *albedo, float *roughness, float *metallic, iris_stream_t);
IRIS_API void iris_ngp_destroy(iris_ngp *);

/* ---- misc --------------------------------------------------------------------------------------------- */
/* Philox uniforms exactly as the bake kernels draw them (for tests): u2[i] = U(seed, idx0+i, stream_id). */
IRIS_API int iris_philox_u2(uint64_t seed, uint64_t idx0, uint32_t stream_id, int64_t n, float *u2, iris_stream_t);
IRIS_API const char *iris_last_error(void);
IRIS_API const char *iris_version(void);

#ifdef __cplusplus
}
#endif
#endif /* IRIS_HIP_H */
