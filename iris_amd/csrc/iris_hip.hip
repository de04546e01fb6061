// libiris_hip.so -- kernels and C ABI (include/iris_hip.h).  gfx950 / wave64 only.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <cstdlib>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/iris_hip.h"
#include "../../include/iris_hip_debug.h"
#include "bvh_build.h"
#include "iris_device.h"
#include "iris_trace.h"
#include "iris_bake.h"
#include "iris_pt.h"
#include "iris_cache.h"
#include "iris_denoise.h"
#include "iris_ngp.h"

using namespace iris;

// ======================================================================================================
// error plumbing
// ======================================================================================================
static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define HIP_TRY(expr)                                                                                     \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess) return fail(IRIS_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)
#define API_BEGIN try {
#define API_END                                                                   \
    }                                                                             \
    catch (const std::exception& ex) { return fail(IRIS_ERR_BUILD, ex.what()); }  \
    catch (...) { return fail(IRIS_ERR_BUILD, "unknown C++ exception"); }

extern "C" IRIS_API const char* iris_last_error(void) { return g_err.c_str(); }

// ---- diagnostics options (iris_hip_debug.h): process-wide, set by tests / experiments only; -1 = the built-in default
static long long g_opt_bvh_tri_cost_x100 = -1, g_opt_bvh_max_leaf = -1, g_opt_phase_min = -1, g_opt_tile_target_rays = -1, g_opt_tiles_per_block = -1, g_opt_pt_tile_min = -1, g_opt_bvh_presplit_x10 = -1, g_opt_joint_max_rays = -1;
extern "C" IRIS_API int iris_debug_set(const char* key, long long value) {
    if (!key) return fail(IRIS_ERR_ARG, "iris_debug_set: null key");
    const std::string k(key);
    if (k == "bvh_max_leaf") g_opt_bvh_max_leaf = value;
    else if (k == "bvh_tri_cost_x100") g_opt_bvh_tri_cost_x100 = value;
    else if (k == "bvh_presplit_x10") g_opt_bvh_presplit_x10 = value;
    else if (k == "phase_min") g_opt_phase_min = value;
    else if (k == "tile_target_rays") g_opt_tile_target_rays = value;
    else if (k == "tiles_per_block") g_opt_tiles_per_block = value;
    else if (k == "pt_tile_min") g_opt_pt_tile_min = value;
    else if (k == "joint_max_rays") g_opt_joint_max_rays = value;
    else return fail(IRIS_ERR_ARG, "iris_debug_set: unknown option " + k);
    return IRIS_OK;
}
extern "C" IRIS_API const char* iris_version(void) { return "iris_hip 0.1 (gfx950)"; }
#ifndef IRIS_NO_FUSED_RECORDS
#define IRIS_NO_FUSED_RECORDS 0      // (A/B: 1 = the bake kernels keep the plain leaf records and gather the emitter ordinal from its own table)
#endif
#ifndef IRIS_BUILD_FLAGS
#define IRIS_BUILD_FLAGS "unknown"
#endif
extern "C" IRIS_API const char* iris_debug_build_flags(void) { return IRIS_BUILD_FLAGS; }
#ifndef IRIS_SOURCE_HASH
#define IRIS_SOURCE_HASH "unknown"
#endif
extern "C" IRIS_API const char* iris_debug_source_hash(void) { return IRIS_SOURCE_HASH; }

// ======================================================================================================
// handles
// ======================================================================================================
static std::atomic<uint64_t> g_scene_uid{1};
struct iris_scene {
    int device = 0;
    uint64_t uid = g_scene_uid.fetch_add(1);    // identity of this scene for caches keyed on it (a pointer can be reused after iris_scene_destroy)
    SceneDev dev{};
    void* d_nodes = nullptr;
    void* d_tris = nullptr;
    iris_scene_info info{};
};
struct iris_slf {
    int device = 0;
    SlfDev dev{};
    void* d_inds = nullptr;
    void* d_rad = nullptr;
    int64_t kv = 0;
};
struct iris_emitter {
    int device = 0;
    EmitDev dev{};
    void* d_ord = nullptr;
    void* d_rad = nullptr;
    void* d_area = nullptr;
    void* d_verts = nullptr;   // (K,3,3) emitter_vertices   (sample_emitter only)
    void* d_cdf = nullptr;     // (K) emitter_cdf
    void* d_ord2tri = nullptr; // (K) triangle index per emitter ordinal
    EmitSampleDev sample{};
    bool can_sample = false;
    int64_t n_rad = 0, k = 0;
    // Fused leaf records (round 5): the scene's leaf-record table with this emitter's ordinal of every triangle in the record's free fourth plane, so that the shading pass of
    // the bake kernels reads it from the line it fetches anyway.  Built on first use per scene (iris_bake_view), kept for the handle's life (at most two scenes).
    struct Fused { uint64_t scene_uid; void* d_tris; hipEvent_t ready; bool done; };
    mutable std::mutex fused_mu;
    mutable std::vector<Fused> fused;
    mutable std::vector<Fused> retired;      // tables pushed out by a third scene: a launch that was handed one may still be on its way -- freed with the handle, never earlier
};

// Launches of the one-ray-per-lane kernels (iris_intersect, the path-tracing stages below their tiling threshold) of at most this many rays run in LATENCY MODE
// (iris_trace.h trace_q8_joint: node and triangle loads of an iteration issued together): they do not fill the chip, and what they wait for is their longest wave's
// dependent round trips.  iris_debug_set("joint_max_rays") overrides (0 = never).  Results do not depend on it.
constexpr long long kJointMaxRays = 1 << 20;
static bool joint_launch(int64_t n_rays) { return n_rays <= (g_opt_joint_max_rays >= 0 ? g_opt_joint_max_rays : kJointMaxRays) && !IRIS_NODE80; }

static int grid_for(int64_t n, int block, int max_blocks) {
    int64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > max_blocks) g = max_blocks;
    return (int)g;
}
static int num_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) cus = p.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

// ------------------------------------------------------------------------------------------------------
extern "C" IRIS_API int iris_scene_create(const float* verts, int64_t nv, const int32_t* faces, int64_t nf, int device, iris_scene** out) {
    return iris_debug_scene_create(verts, nv, faces, nf, device, IRIS_BVH_DEFAULT, out);
}
extern "C" IRIS_API int iris_debug_scene_create(const float* verts, int64_t nv, const int32_t* faces, int64_t nf, int device, int layout,
                                       iris_scene** out) {
    API_BEGIN
    if (!out || nv < 0 || nf < 0 || (nf > 0 && (!verts || !faces))) return fail(IRIS_ERR_ARG, "iris_scene_create: bad arguments");
    if (nf >= (1 << 26) - 1) return fail(IRIS_ERR_ARG, "iris_scene_create: more than 2^26 triangles (32-bit byte offsets into the 64-B records)");
    for (int64_t i = 0; i < nf * 3; ++i)
        if (faces[i] < 0 || faces[i] >= nv) return fail(IRIS_ERR_ARG, "iris_scene_create: face index out of range");
    if (layout == IRIS_BVH_DEFAULT) layout = IRIS_BVH4_Q8;
    if (layout != IRIS_BVH4_F32 && layout != IRIS_BVH4_Q8) return fail(IRIS_ERR_ARG, "iris_scene_create: unknown BVH layout");
    HIP_TRY(hipSetDevice(device));
    auto t0 = std::chrono::steady_clock::now();
    int max_leaf = 4;
    if (g_opt_bvh_max_leaf > 0) max_leaf = (int)std::min<long long>(7, g_opt_bvh_max_leaf);   // iris_debug_set("bvh_max_leaf")
    const float tri_cost = g_opt_bvh_tri_cost_x100 > 0 ? (float)g_opt_bvh_tri_cost_x100 * 0.01f : 0.7f;   // iris_debug_set("bvh_tri_cost_x100")
    const float presplit = g_opt_bvh_presplit_x10 >= 0 ? (float)g_opt_bvh_presplit_x10 * 0.1f : 8.f;       // iris_debug_set("bvh_presplit_x10"); 0 = off
    WideBvh bvh = build_wide_bvh(verts, nv, faces, nf, 4, max_leaf, 2e-5f, tri_cost, presplit);
    if (bvh.tri_order.size() >= (size_t)(1 << 26) - 1) return fail(IRIS_ERR_BUILD, "iris_scene_create: more than 2^26 leaf records");
    if (3 * bvh.depth + 4 > kStackLds + kStackSpill) return fail(IRIS_ERR_BUILD, "iris_scene_create: BVH too deep for the traversal stack");
    if (layout == IRIS_BVH4_Q8 && bvh.nodes.size() * kNodeBytes * 8 >= ((size_t)1 << 32)) return fail(IRIS_ERR_BUILD, "iris_scene_create: the eight octant copies of the node table exceed 4 GiB (32-bit byte offsets)");

    // ---- encode nodes ----
    const size_t nn = bvh.nodes.size();
    // Unused child slots carry an inverted quantised box, which the slab test rejects -- except when the planes of a tiny node far from
    // the ray origin collapse onto one t (b absorbs q * a): their reference is therefore a 1-triangle leaf on a degenerate record appended
    // to the triangle table (all zeros: det = 0, never accepted), never kEmptyRef, which the traversal also uses as "lane idle".
    const uint32_t dummy_leaf = kLeafBit | ((uint32_t)bvh.tri_order.size() << 3) | 1u;
    auto child_ref = [&](const WideNode& w, int s) -> uint32_t {
        if (s >= w.n) return dummy_leaf;
        if (w.child[s] >= 0) return (uint32_t)w.child[s];
        return kLeafBit | ((uint32_t)w.leaf_start[s] << 3) | (uint32_t)w.leaf_count[s];
    };
    const int node_floats = layout == IRIS_BVH4_Q8 ? (int)kNodeBytes / 4 : 32;
    const int n_copies = layout == IRIS_BVH4_Q8 ? 8 : 1;       // Q8: one copy of the node table per ray octant (see below)
    std::vector<float> nodes(nn * node_floats * n_copies);
    for (size_t i = 0; i < nn; ++i) {
        const WideNode& w = bvh.nodes[i];
        float* p = nodes.data() + i * node_floats;
        if (layout == IRIS_BVH4_F32) {   // 128 B: lox[4] hix[4] loy[4] hiy[4] loz[4] hiz[4] ref[4] pad[4]
            for (int s = 0; s < 4; ++s) {
                p[0 + s] = w.lo[s][0]; p[4 + s] = w.hi[s][0];
                p[8 + s] = w.lo[s][1]; p[12 + s] = w.hi[s][1];
                p[16 + s] = w.lo[s][2]; p[20 + s] = w.hi[s][2];
                uint32_t ref = child_ref(w, s);
                std::memcpy(&p[24 + s], &ref, 4);
                p[28 + s] = 0.f;
            }
        } else {                         // 64 B: origin.xyz, scale.x | scale.yz, qlo_x, qlo_y | qlo_z, qhi_x, qhi_y, qhi_z | ref[4]
            float org[3], hi3[3];
            for (int k = 0; k < 3; ++k) {
                org[k] = INFINITY; hi3[k] = -INFINITY;
                for (int s = 0; s < w.n; ++s) { org[k] = std::min(org[k], w.lo[s][k]); hi3[k] = std::max(hi3[k], w.hi[s][k]); }
                if (w.n == 0) { org[k] = 0.f; hi3[k] = 0.f; }
            }
            uint32_t ebytes = 0;
            uint8_t q[6][4];   // planes lo_x lo_y lo_z hi_x hi_y hi_z
            double ext_max = 0.0;
            for (int k = 0; k < 3; ++k) ext_max = std::max(ext_max, (double)hi3[k] - (double)org[k]);
            for (int k = 0; k < 3; ++k) {
                const double ext = IRIS_NODE80 ? ext_max : (double)hi3[k] - (double)org[k];     // (80-B nodes: ONE plane scale per node, that of the longest axis)
                int e = -126;
                if (ext > 0) e = std::max(-126, (int)std::ceil(std::log2(ext / 255.0)));
                while (std::ldexp(255.0, e) < ext) ++e;                       // 255 * 2^e must cover the extent
                const double sc = std::ldexp(1.0, e);
                ebytes |= (uint32_t)(e + 127) << (8 * k);
                for (int s = 0; s < 4; ++s) {
                    if (s >= w.n) { q[k][s] = 255; q[3 + k][s] = 0; continue; }   // inverted box: never hit
                    int lo = (int)std::floor(((double)w.lo[s][k] - (double)org[k]) / sc);
                    int hi = (int)std::ceil(((double)w.hi[s][k] - (double)org[k]) / sc);
                    lo = std::min(255, std::max(0, lo)); hi = std::min(255, std::max(0, hi));
                    while (lo > 0 && (double)org[k] + lo * sc > (double)w.lo[s][k]) --lo;       // decoded box must contain the f32 box
                    while (hi < 255 && (double)org[k] + hi * sc < (double)w.hi[s][k]) ++hi;
                    if ((double)org[k] + hi * sc < (double)w.hi[s][k]) return fail(IRIS_ERR_BUILD, "iris_scene_create: node quantisation failed");
                    q[k][s] = (uint8_t)lo; q[3 + k][s] = (uint8_t)hi;
                }
            }
            // the plane scales 2^e are stored as floats (not as exponent bytes): decoding them on the device -- two ALU operations per axis
            // right behind the load, in front of every slab test -- was measured 10 % slower on the whole bake
            // (times 2^24: the kernel feeds the plane bytes to v_fma_mix_f32 as f16 subnormals q * 2^-24, iris_trace.h node_step)
            //
            // ONE COPY PER RAY OCTANT (round 3): copy o (bit 0: d.x < 0, bit 1: d.y < 0, bit 2: d.z < 0) holds the children in the front-to-back order
            // the node's binary splits give a ray of that octant (WideNode::order), and per axis the plane the ray meets FIRST in the "near" bytes:
            //   {origin.xyz, scale.x} {scale.y, scale.z, near_x[4], near_y[4]} {near_z[4], far_x[4], far_y[4], far_z[4]} {ref[4]}
            // so that a node visit neither selects planes by the ray's signs (6 selects) nor sorts the children (5 compare-exchanges): 8 x 64 B per
            // node -- 105 MB for the bench scene next to 288 GB -- against ~30 vector instructions per visit.  Child references are node indices
            // (the same in every copy); a ray adds its copy's base offset (SceneDev::oct_stride).
            for (int o = 0; o < 8; ++o) {
                float* po = nodes.data() + ((size_t)o * nn + i) * node_floats;
                po[0] = org[0]; po[1] = org[1]; po[2] = org[2];
#if IRIS_NODE80
                // {origin.xyz, scale} {ref[4]} {x planes of children 0..3} {y planes} {z planes}: a plane word = near byte | far byte << 16 (two f16 subnormals)
                po[3] = std::ldexp(1.0f, (int)(ebytes & 0xffu) - 127 + 24);
                for (int j = 0; j < 4; ++j) {
                    const int sl = j < w.n ? (int)w.order[o][j] : j;
                    const uint32_t ref = child_ref(w, sl);
                    std::memcpy(&po[4 + j], &ref, 4);
                    for (int k = 0; k < 3; ++k) {
                        const bool neg = (o >> k) & 1;
                        const uint32_t near_q = neg ? q[3 + k][sl] : q[k][sl], far_q = neg ? q[k][sl] : q[3 + k][sl];
                        const uint32_t word = near_q | (far_q << 16);
                        std::memcpy(&po[8 + 4 * k + j], &word, 4);
                    }
                }
                continue;
#endif
                for (int k = 0; k < 3; ++k) po[3 + k] = std::ldexp(1.0f, (int)((ebytes >> (8 * k)) & 0xffu) - 127 + 24);
                uint8_t qo[6][4];
                for (int j = 0; j < 4; ++j) {
                    const int sl = j < w.n ? (int)w.order[o][j] : j;          // (unused slots: the canonical inverted box (lo 255, hi 0) goes through the same swap, so it is empty for either sign)
                    for (int k = 0; k < 3; ++k) {
                        const bool neg = (o >> k) & 1;
                        qo[k][j] = neg ? q[3 + k][sl] : q[k][sl];             // near
                        qo[3 + k][j] = neg ? q[k][sl] : q[3 + k][sl];         // far
                    }
                    const uint32_t ref = child_ref(w, sl);
                    std::memcpy(&po[12 + j], &ref, 4);
                }
                for (int k = 0; k < 6; ++k) std::memcpy(&po[6 + k], qo[k], 4);
            }
        }
    }
    // ---- encode leaf triangles (64 B, one per 64-B line), component-major so that the watertight test's axis permutation is an address
    // offset: (p0.x, p1.x, p2.x, id) (p0.y, p1.y, p2.y, id) (p0.z, p1.z, p2.z, id) (0, 0, 0, 0) ----
    const size_t nt = bvh.tri_order.size();
    std::vector<float> tris((nt + 1) * 16, 0.f);      // + the degenerate record unused child slots point to (id -1, never accepted: det = 0)
    { const int32_t none = -1; for (int k = 0; k < 3; ++k) std::memcpy(&tris[nt * 16 + 4 * k + 3], &none, 4); }
    for (size_t i = 0; i < nt; ++i) {
        int32_t f = bvh.tri_order[i];
        float* p = tris.data() + i * 16;
        for (int v = 0; v < 3; ++v) {
            const float* pv = verts + (int64_t)faces[(int64_t)f * 3 + v] * 3;
            for (int k = 0; k < 3; ++k) p[4 * k + v] = pv[k];
        }
        for (int k = 0; k < 3; ++k) std::memcpy(&p[4 * k + 3], &f, 4);
    }
    iris_scene* s = new iris_scene();
    s->device = device;
    if (hipMalloc(&s->d_nodes, nodes.size() * 4) != hipSuccess || hipMalloc(&s->d_tris, tris.size() * 4) != hipSuccess ||
        hipMemcpy(s->d_nodes, nodes.data(), nodes.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(s->d_tris, tris.data(), tris.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(s->d_nodes); (void)hipFree(s->d_tris); delete s;
        return fail(IRIS_ERR_HIP, "iris_scene_create: device allocation / upload failed");
    }
    s->dev.nodes = (const float4*)s->d_nodes;
    s->dev.tris = (const float4*)s->d_tris;
    s->dev.n_nodes = (int)nn;
    s->dev.n_tris = (int)nt;
    s->dev.layout = layout == IRIS_BVH4_Q8 ? kLayoutQ8 : kLayoutF32;
    s->dev.oct_stride = layout == IRIS_BVH4_Q8 ? (uint32_t)(nn * kNodeBytes) : 0u;
    s->dev.phase_min = kPhaseMin;
    if (g_opt_phase_min >= 0) s->dev.phase_min = (int)g_opt_phase_min;  // iris_debug_set("phase_min") (results do not depend on it)
    s->info.n_vertices = nv; s->info.n_triangles = nf; s->info.layout = layout; s->info.n_nodes = (int32_t)nn;
    s->info.node_bytes = layout == IRIS_BVH4_Q8 ? (int)kNodeBytes : 128; s->info.tri_bytes = 64; s->info.depth = bvh.depth;
    s->info.sah_cost = bvh.sah_cost; s->info.n_leaf_records = (int32_t)nt;
    s->info.build_seconds = std::chrono::duration<float>(std::chrono::steady_clock::now() - t0).count();
    *out = s;
    return IRIS_OK;
    API_END
}
extern "C" IRIS_API void iris_scene_destroy(iris_scene* s) {
    if (!s) return;
    (void)hipFree(s->d_nodes); (void)hipFree(s->d_tris);
    delete s;
}
extern "C" IRIS_API int iris_scene_get_info(const iris_scene* s, iris_scene_info* out) {
    if (!s || !out) return fail(IRIS_ERR_ARG, "iris_scene_get_info: null");
    *out = s->info;
    return IRIS_OK;
}

extern "C" IRIS_API int iris_slf_create(const int64_t* inds, int H, const float* radiance, int64_t kv, double voxel_min, double voxel_max,
                               int device, iris_slf** out) {
    API_BEGIN
    if (!out || !inds || H <= 0 || H > 1024 || kv < 0 || (kv > 0 && !radiance)) return fail(IRIS_ERR_ARG, "iris_slf_create: bad arguments");
    HIP_TRY(hipSetDevice(device));
    const size_t n = (size_t)H * H * H;
    std::vector<int32_t> i32(n);
    for (size_t i = 0; i < n; ++i) {
        int64_t v = inds[i];
        if (v < -1 || v >= kv) return fail(IRIS_ERR_ARG, "iris_slf_create: inds entry out of range");
        i32[i] = (int32_t)v;
    }
    std::vector<float> rad((size_t)std::max<int64_t>(kv, 1) * 4, 0.f);
    for (int64_t i = 0; i < kv; ++i) { rad[i * 4] = radiance[i * 3]; rad[i * 4 + 1] = radiance[i * 3 + 1]; rad[i * 4 + 2] = radiance[i * 3 + 2]; }
    iris_slf* s = new iris_slf();
    s->device = device; s->kv = kv;
    HIP_TRY(hipMalloc(&s->d_inds, n * 4));
    HIP_TRY(hipMalloc(&s->d_rad, rad.size() * 4));
    HIP_TRY(hipMemcpy(s->d_inds, i32.data(), n * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(s->d_rad, rad.data(), rad.size() * 4, hipMemcpyHostToDevice));
    s->dev.inds = (const int32_t*)s->d_inds; s->dev.radiance = (const float4*)s->d_rad; s->dev.H = H;
    s->dev.vmin = (float)voxel_min;
    s->dev.den = (float)(voxel_max - voxel_min);
    *out = s;
    return IRIS_OK;
    API_END
}
// The same from DEVICE buffers (the pre-bake stages build the grid on the GPU: slf_bake.py:116-118 constructs VoxelSLF from the occupancy mask it has
// just counted there): no 8 H^3-byte round trip through the host.  One 4-byte read-back reports an out-of-range entry.
__global__ void slf_inds_narrow_kernel(const int64_t* __restrict__ src, int32_t* __restrict__ dst, int64_t n, int64_t kv, int* __restrict__ bad) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t v = src[i];
        if (v < -1 || v >= kv) *bad = 1;
        dst[i] = (int32_t)v;
    }
}
__global__ void pad_rows_kernel(const float* __restrict__ src, float4* __restrict__ dst, int64_t n);
extern "C" IRIS_API int iris_slf_create_dev(const int64_t* inds_dev, int H, const float* radiance_dev, int64_t kv, double voxel_min, double voxel_max,
                                   int device, iris_slf** out, iris_stream_t stream) {
    API_BEGIN
    if (!out || !inds_dev || H <= 0 || H > 1024 || kv < 0 || (kv > 0 && !radiance_dev)) return fail(IRIS_ERR_ARG, "iris_slf_create_dev: bad arguments");
    HIP_TRY(hipSetDevice(device));
    const size_t n = (size_t)H * H * H;
    iris_slf* s = new iris_slf();
    s->device = device; s->kv = kv;
    int* d_bad = nullptr;
    auto cleanup = [&](int rc, const char* msg) { (void)hipFree(s->d_inds); (void)hipFree(s->d_rad); (void)hipFree(d_bad); delete s; return fail(rc, msg); };
    if (hipMalloc(&s->d_inds, n * 4) != hipSuccess || hipMalloc(&s->d_rad, (size_t)std::max<int64_t>(kv, 1) * 16) != hipSuccess || hipMalloc(&d_bad, 4) != hipSuccess)
        return cleanup(IRIS_ERR_HIP, "iris_slf_create_dev: out of device memory");
    hipStream_t st = (hipStream_t)stream;
    (void)hipMemsetAsync(d_bad, 0, 4, st);
    (void)hipMemsetAsync(s->d_rad, 0, (size_t)std::max<int64_t>(kv, 1) * 16, st);
    hipLaunchKernelGGL(slf_inds_narrow_kernel, dim3(grid_for((int64_t)n, 256, 8192)), dim3(256), 0, st, inds_dev, (int32_t*)s->d_inds, (int64_t)n, kv, d_bad);
    if (kv > 0) hipLaunchKernelGGL(pad_rows_kernel, dim3(grid_for(kv, 256, 1024)), dim3(256), 0, st, radiance_dev, (float4*)s->d_rad, kv);
    int bad = 0;
    if (hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return cleanup(IRIS_ERR_HIP, "iris_slf_create_dev: HIP error");
    if (bad) return cleanup(IRIS_ERR_ARG, "iris_slf_create_dev: inds entry out of range");
    (void)hipFree(d_bad);
    s->dev.inds = (const int32_t*)s->d_inds; s->dev.radiance = (const float4*)s->d_rad; s->dev.H = H;
    s->dev.vmin = (float)voxel_min;
    s->dev.den = (float)(voxel_max - voxel_min);
    *out = s;
    return IRIS_OK;
    API_END
}
extern "C" IRIS_API void iris_slf_destroy(iris_slf* s) {
    if (!s) return;
    (void)hipFree(s->d_inds); (void)hipFree(s->d_rad);
    delete s;
}

// ======================================================================================================
// NGPBRDF (model/brdf.py:213-260): hash-grid encoding + MLP, inference
// ======================================================================================================
struct iris_ngp {
    int device = 0;
    NgpLevels lv{};
    void* d_grid = nullptr;     // half2 entries
    void* d_w = nullptr;        // kNgpMlpParams halves
    void* d_feat = nullptr;     // [32 levels][kChunk] half2: the encoded features of one chunk of points
    float vmin = 0.f, den = 1.f;
    uint64_t n_entries = 0;
    // d_feat is the handle's ONE scratch buffer: forwards of one handle are serialised on the device -- a call on another stream than the previous call's first
    // waits (on the device, not the host) for the event that call recorded behind its last kernel.  Host threads are serialised by the mutex.
    mutable std::mutex mu;
    mutable hipEvent_t last_use = nullptr;
    mutable hipStream_t last_stream = nullptr;
    mutable bool used = false;
};
constexpr int kNgpChunk = 1 << 20;       // points per encode / MLP launch pair: 128 MiB of features
static uint64_t ngp_levels(NgpLevels& lv) {
    // tiny-cuda-nn GridEncoding: scale = exp2(level * log2(per_level_scale)) * base - 1, resolution = ceil(scale) + 1, entries = min(round_up(res^3, 8), 2^19)
    // (the library evaluates this in float with the device's fast exp2f, whose last bits are not reproducible; here every transcendental is taken in double
    //  and rounded to float once, so that any host libm gives the same table: log2(1.3f) -> float, level * that in float, exp2 -> float, * 16 - 1 in float)
    const float log2_scale = (float)log2((double)1.3f);
    uint64_t off = 0;
    for (int l = 0; l < kNgpLevels; ++l) {
        const float scale = (float)exp2((double)((float)l * log2_scale)) * 16.f - 1.0f;
        const uint32_t res = (uint32_t)ceilf(scale) + 1u;
        uint64_t n = (uint64_t)res * res * res;
        n = std::min<uint64_t>(n, 0xFFFFFFFFull / 2);
        n = (n + 7) / 8 * 8;
        n = std::min<uint64_t>(n, 1ull << 19);
        lv.scale[l] = scale; lv.res[l] = res; lv.size[l] = (uint32_t)n; lv.offset[l] = (uint32_t)off;
        off += n;
    }
    return off;
}
static uint16_t f32_to_f16_bits(float f) { _Float16 h = (_Float16)f; uint16_t u; memcpy(&u, &h, 2); return u; }     // round to nearest even, as torch's .half()
extern "C" IRIS_API int64_t iris_ngp_n_params(void) {
    NgpLevels lv;
    return (int64_t)kNgpMlpParams + (int64_t)ngp_levels(lv) * 2;
}
extern "C" IRIS_API int iris_ngp_create(const float* params, int64_t n_params, double voxel_min, double voxel_max, int device, iris_ngp** out) {
    API_BEGIN
    if (!out || !params) return fail(IRIS_ERR_ARG, "iris_ngp_create: bad arguments");
    HIP_TRY(hipSetDevice(device));              // (before anything is allocated: an invalid device leaves nothing behind)
    iris_ngp* g = new iris_ngp();
    g->device = device;
    g->n_entries = ngp_levels(g->lv);
    if (n_params != (int64_t)kNgpMlpParams + (int64_t)g->n_entries * 2) {
        delete g;
        return fail(IRIS_ERR_ARG, "iris_ngp_create: mlp.params has " + std::to_string(n_params) + " entries, the NGPBRDF configuration has " + std::to_string(iris_ngp_n_params()));
    }
    std::vector<uint16_t> h((size_t)n_params);
    for (int64_t i = 0; i < n_params; ++i) h[(size_t)i] = f32_to_f16_bits(params[i]);
    auto cleanup = [&](const char* msg) { (void)hipFree(g->d_grid); (void)hipFree(g->d_w); (void)hipFree(g->d_feat); if (g->last_use) (void)hipEventDestroy(g->last_use); delete g; return fail(IRIS_ERR_HIP, msg); };
    if (hipMalloc(&g->d_w, (size_t)kNgpMlpParams * 2) != hipSuccess || hipMalloc(&g->d_grid, (size_t)g->n_entries * 4) != hipSuccess ||
        hipMalloc(&g->d_feat, (size_t)kNgpLevels * kNgpChunk * 4) != hipSuccess)
        return cleanup("iris_ngp_create: out of device memory");
    if (hipEventCreateWithFlags(&g->last_use, hipEventDisableTiming) != hipSuccess) return cleanup("iris_ngp_create: hipEventCreate failed");
    if (hipMemcpy(g->d_w, h.data(), (size_t)kNgpMlpParams * 2, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(g->d_grid, h.data() + kNgpMlpParams, (size_t)g->n_entries * 4, hipMemcpyHostToDevice) != hipSuccess)
        return cleanup("iris_ngp_create: upload failed");
    g->vmin = (float)voxel_min;
    g->den = (float)(voxel_max - voxel_min);           // (the difference of the two python floats, taken in double, enters the float32 tensor arithmetic as one scalar)
    *out = g;
    return IRIS_OK;
    API_END
}
extern "C" IRIS_API int iris_ngp_forward(const iris_ngp* g, const float* position, int64_t N, float* albedo, float* roughness, float* metallic, iris_stream_t stream) {
    API_BEGIN
    if (!g || N < 0 || (N > 0 && (!position || !albedo || !roughness || !metallic))) return fail(IRIS_ERR_ARG, "iris_ngp_forward: bad arguments");
    HIP_TRY(hipSetDevice(g->device));
    hipStream_t st = (hipStream_t)stream;
    if (N == 0) return IRIS_OK;
    std::lock_guard<std::mutex> lock(g->mu);
    if (g->used && g->last_stream != st) HIP_TRY(hipStreamWaitEvent(st, g->last_use, 0));      // the previous forward of this handle still owns d_feat
    NgpArgs a{};
    a.lv = g->lv; a.grid = (const uint32_t*)g->d_grid; a.w = (const _Float16*)g->d_w; a.pos = position; a.feat = (uint32_t*)g->d_feat;
    a.albedo = albedo; a.rough = roughness; a.metal = metallic; a.n_chunk = kNgpChunk; a.vmin = g->vmin; a.den = g->den;
    for (int64_t n0 = 0; n0 < N; n0 += kNgpChunk) {          // (stream-ordered: the feature planes of a chunk are consumed before the next chunk's encode overwrites them)
        a.n0 = n0; a.n = (int)std::min<int64_t>(kNgpChunk, N - n0);
        hipLaunchKernelGGL(ngp_encode_kernel, dim3((a.n + 255) / 256, kNgpLevels), dim3(256), 0, st, a);
        const int tiles = (a.n + 31) / 32;
        hipLaunchKernelGGL(ngp_mlp_kernel, dim3(std::min(std::max((tiles + 3) / 4, 1), 2048)), dim3(256), 0, st, a);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(g->last_use, st));
    g->last_stream = st; g->used = true;
    return IRIS_OK;
    API_END
}
extern "C" IRIS_API int iris_debug_ngp_encode(const iris_ngp* g, const float* position, int64_t N, uint32_t* feat, iris_stream_t stream) {
    API_BEGIN
    if (!g || N < 0 || N > kNgpChunk || (N > 0 && (!position || !feat))) return fail(IRIS_ERR_ARG, "iris_debug_ngp_encode: bad arguments");
    if (N == 0) return IRIS_OK;
    HIP_TRY(hipSetDevice(g->device));
    NgpArgs a{};
    a.lv = g->lv; a.grid = (const uint32_t*)g->d_grid; a.w = (const _Float16*)g->d_w; a.pos = position; a.feat = feat;
    a.n_chunk = (int)N; a.vmin = g->vmin; a.den = g->den; a.n0 = 0; a.n = (int)N;
    hipLaunchKernelGGL(ngp_encode_kernel, dim3((a.n + 255) / 256, kNgpLevels), dim3(256), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
    API_END
}
extern "C" IRIS_API void iris_ngp_destroy(iris_ngp* g) {
    if (!g) return;
    (void)hipFree(g->d_grid); (void)hipFree(g->d_w); (void)hipFree(g->d_feat);
    if (g->last_use) (void)hipEventDestroy(g->last_use);
    delete g;
}

extern "C" IRIS_API int iris_emitter_create(const uint8_t* is_emitter, int64_t nf, const float* radiance, int64_t n_rad, const float* area,
                                   int64_t k, const float* verts, const float* cdf, int device, iris_emitter** out) {
    API_BEGIN
    if (!out || nf < 0 || k < 0 || n_rad < 0 || (nf > 0 && !is_emitter)) return fail(IRIS_ERR_ARG, "iris_emitter_create: bad arguments");
    HIP_TRY(hipSetDevice(device));
    std::vector<int32_t> ord((size_t)std::max<int64_t>(nf, 1), -1);
    int64_t c = 0;
    for (int64_t i = 0; i < nf; ++i) ord[(size_t)i] = is_emitter[i] ? (int32_t)c++ : -1;
    if (c != k) return fail(IRIS_ERR_ARG, "iris_emitter_create: is_emitter.sum() != len(emitter_area)");
    if (c > n_rad) return fail(IRIS_ERR_ARG, "iris_emitter_create: radiance has fewer rows than emitters");
    std::vector<float> rad((size_t)std::max<int64_t>(n_rad, 1) * 4, 0.f);
    for (int64_t i = 0; i < n_rad; ++i) { rad[i * 4] = radiance[i * 3]; rad[i * 4 + 1] = radiance[i * 3 + 1]; rad[i * 4 + 2] = radiance[i * 3 + 2]; }
    iris_emitter* e = new iris_emitter();
    e->device = device; e->n_rad = n_rad; e->k = k;
    HIP_TRY(hipMalloc(&e->d_ord, ord.size() * 4));
    HIP_TRY(hipMalloc(&e->d_rad, rad.size() * 4));
    HIP_TRY(hipMalloc(&e->d_area, (size_t)std::max<int64_t>(k, 1) * 4));
    HIP_TRY(hipMemcpy(e->d_ord, ord.data(), ord.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(e->d_rad, rad.data(), rad.size() * 4, hipMemcpyHostToDevice));
    if (k > 0) HIP_TRY(hipMemcpy(e->d_area, area, (size_t)k * 4, hipMemcpyHostToDevice));
    e->dev.emit_ord = (const int32_t*)e->d_ord; e->dev.radiance = (const float4*)e->d_rad; e->dev.area = (const float*)e->d_area;
    e->dev.nf = nf;
    float kf = (float)k; if (kf < 1e-12f) kf = 1e-12f;  // NF.normalize(ones(k), p=1)
    e->dev.emitter_pdf = 1.0f / kf;
    if (verts && cdf && k > 0) {                          // tables of sample_emitter (model/emitter.py:224-255)
        std::vector<int32_t> o2t((size_t)k);
        for (int64_t i = 0; i < nf; ++i) if (ord[(size_t)i] >= 0) o2t[(size_t)ord[(size_t)i]] = (int32_t)i;
        HIP_TRY(hipMalloc(&e->d_verts, (size_t)k * 36));
        HIP_TRY(hipMalloc(&e->d_cdf, (size_t)k * 4));
        HIP_TRY(hipMalloc(&e->d_ord2tri, (size_t)k * 4));
        HIP_TRY(hipMemcpy(e->d_verts, verts, (size_t)k * 36, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(e->d_cdf, cdf, (size_t)k * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(e->d_ord2tri, o2t.data(), (size_t)k * 4, hipMemcpyHostToDevice));
        e->sample.cdf = (const float*)e->d_cdf; e->sample.verts = (const float*)e->d_verts; e->sample.area = (const float*)e->d_area;
        e->sample.ord2tri = (const int32_t*)e->d_ord2tri; e->sample.k = k; e->sample.emitter_pdf = e->dev.emitter_pdf;
        e->can_sample = true;
    }
    *out = e;
    return IRIS_OK;
    API_END
}
extern "C" IRIS_API void iris_emitter_destroy(iris_emitter* e) {
    if (!e) return;
    (void)hipFree(e->d_ord); (void)hipFree(e->d_rad); (void)hipFree(e->d_area);
    (void)hipFree(e->d_verts); (void)hipFree(e->d_cdf); (void)hipFree(e->d_ord2tri);
    for (const auto* v : {&e->fused, &e->retired}) for (const auto& f : *v) { (void)hipFree(f.d_tris); if (f.ready) (void)hipEventDestroy(f.ready); }
    delete e;
}

__global__ void pad_rows_kernel(const float* __restrict__ src, float4* __restrict__ dst, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = make_float4(src[i * 3], src[i * 3 + 1], src[i * 3 + 2], 0.f);
}
extern "C" IRIS_API int iris_slf_set_radiance(iris_slf* s, const float* radiance_dev, int64_t kv, iris_stream_t stream) {
    if (!s || kv != s->kv || (kv > 0 && !radiance_dev)) return fail(IRIS_ERR_ARG, "iris_slf_set_radiance: bad arguments");
    if (kv == 0) return IRIS_OK;
    hipLaunchKernelGGL(pad_rows_kernel, dim3(grid_for(kv, 256, 1024)), dim3(256), 0, (hipStream_t)stream, radiance_dev, (float4*)s->d_rad, kv);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}
extern "C" IRIS_API int iris_emitter_set_radiance(iris_emitter* e, const float* radiance_dev, int64_t n_rad, iris_stream_t stream) {
    if (!e || !radiance_dev || n_rad != e->n_rad) return fail(IRIS_ERR_ARG, "iris_emitter_set_radiance: bad arguments");
    hipLaunchKernelGGL(pad_rows_kernel, dim3(grid_for(n_rad, 256, 1024)), dim3(256), 0, (hipStream_t)stream, radiance_dev, (float4*)e->d_rad, n_rad);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}

// ======================================================================================================
// a1 ray generation
// ======================================================================================================
struct RaygenArgs { float K[9]; float c2w[12]; float focal; int H, W, ray_diff, synthetic; };

__global__ void raygen_kernel(RaygenArgs a, float* __restrict__ rays_o, float* __restrict__ rays_d, float* __restrict__ dxdu,
                              float* __restrict__ dydv) {
    const int64_t n = (int64_t)a.H * a.W;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int y = (int)(i / a.W), x = (int)(i - (int64_t)y * a.W);
        float dc0, dc1, ifx, ify;
        if (a.synthetic) {  // utils/dataset/synthetic_ldr.py:21-34
            const float hw = (float)((double)a.W / 2.0), hh = (float)((double)a.H / 2.0);
            dc0 = -(((float)x + 0.5f) - hw) / a.focal;
            dc1 = -(((float)y + 0.5f) - hh) / a.focal;
            ifx = ify = 1.0f / a.focal;
        } else {  // utils/dataset/real_ldr.py:49-61
            dc0 = ((float)x + 0.5f - a.K[2]) / a.K[0];
            dc1 = ((float)y + 0.5f - a.K[5]) / a.K[4];
            ifx = 1.0f / a.K[0]; ify = 1.0f / a.K[4];
        }
        f3 d = mk3((dc0 * a.c2w[0] + dc1 * a.c2w[1]) + a.c2w[2], (dc0 * a.c2w[4] + dc1 * a.c2w[5]) + a.c2w[6],
                   (dc0 * a.c2w[8] + dc1 * a.c2w[9]) + a.c2w[10]);
        rays_o[i * 3] = a.c2w[3]; rays_o[i * 3 + 1] = a.c2w[7]; rays_o[i * 3 + 2] = a.c2w[11];
        if (a.ray_diff) {
            st3(rays_d + i * 3, d);
            dxdu[i * 3] = ifx * a.c2w[0]; dxdu[i * 3 + 1] = ifx * a.c2w[4]; dxdu[i * 3 + 2] = ifx * a.c2w[8];
            dydv[i * 3] = ify * a.c2w[1]; dydv[i * 3 + 1] = ify * a.c2w[5]; dydv[i * 3 + 2] = ify * a.c2w[9];
        } else if (a.synthetic) {
            float nrm = sqrtf((d.x * d.x + d.y * d.y) + d.z * d.z);  // rays_d / torch.norm(rays_d)
            st3(rays_d + i * 3, mk3(d.x / nrm, d.y / nrm, d.z / nrm));
        } else {
            st3(rays_d + i * 3, t_normalize(d));
        }
    }
}
static int raygen_launch(RaygenArgs a, float* o, float* d, float* dx, float* dy, iris_stream_t stream) {
    if (a.H <= 0 || a.W <= 0 || !o || !d || (a.ray_diff && (!dx || !dy))) return fail(IRIS_ERR_ARG, "iris_raygen: bad arguments");
    hipLaunchKernelGGL(raygen_kernel, dim3(grid_for((int64_t)a.H * a.W, 256, 4096)), dim3(256), 0, (hipStream_t)stream, a, o, d, dx, dy);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}
extern "C" IRIS_API int iris_raygen_real(const float K[9], const float c2w[12], int H, int W, int ray_diff, float* rays_o, float* rays_d,
                                float* dxdu, float* dydv, iris_stream_t stream) {
    if (!K || !c2w) return fail(IRIS_ERR_ARG, "iris_raygen_real: null K/c2w");
    RaygenArgs a{};
    std::memcpy(a.K, K, 36); std::memcpy(a.c2w, c2w, 48);
    a.H = H; a.W = W; a.ray_diff = ray_diff; a.synthetic = 0; a.focal = 1.f;
    return raygen_launch(a, rays_o, rays_d, dxdu, dydv, stream);
}
extern "C" IRIS_API int iris_raygen_synthetic(float focal, const float c2w[12], int H, int W, int ray_diff, float* rays_o, float* rays_d,
                                     float* dxdu, float* dydv, iris_stream_t stream) {
    if (!c2w) return fail(IRIS_ERR_ARG, "iris_raygen_synthetic: null c2w");
    RaygenArgs a{};
    std::memcpy(a.c2w, c2w, 48);
    a.H = H; a.W = W; a.ray_diff = ray_diff; a.synthetic = 1; a.focal = focal;
    return raygen_launch(a, rays_o, rays_d, dxdu, dydv, stream);
}

// ======================================================================================================
// a2 ray_intersect
// ======================================================================================================
template <int LAYOUT, bool JOINT = false>
__global__ __launch_bounds__(kBlock) void intersect_kernel(SceneDev sc, const float* __restrict__ xs, const float* __restrict__ ds,
                                                           int64_t B, float* __restrict__ pos, float* __restrict__ nrm,
                                                           float* __restrict__ uv, int64_t* __restrict__ idx, uint8_t* __restrict__ valid) {
    __shared__ uint32_t s_stack[kStackLds * kBlock];
    for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < B; i += (int64_t)gridDim.x * kBlock) {
        f3 o = ld3(xs + i * 3), d = ld3(ds + i * 3);
        Hit h = trace_bvh4<LAYOUT, false, kStackLds, false, JOINT>(sc, o, d, s_stack + threadIdx.x);
        if (h.slot >= 0) {
            f3 p0, p1, p2;
            hit_vertices(sc, h, p0, p1, p2);
            if (pos) st3(pos + i * 3, hit_position(h, p0, p1, p2));
            if (nrm) {
                f3 n = t_normalize(hit_normal(p0, p1, p2));                    // NF.normalize(ret.n)
                if (t_dot(n, mk3(-d.x, -d.y, -d.z)) < 0.f) n = mk3(-n.x, -n.y, -n.z);  // double_sided(-ds, normals)
                st3(nrm + i * 3, n);
            }
            if (uv) { uv[i * 2] = h.u; uv[i * 2 + 1] = h.v; }
            if (idx) idx[i] = h.id;
            if (valid) valid[i] = 1;
        } else {
            if (pos) st3(pos + i * 3, mk3(0.f, 0.f, 0.f));
            if (nrm) st3(nrm + i * 3, mk3(0.f, 0.f, 0.f));
            if (uv) { uv[i * 2] = 0.f; uv[i * 2 + 1] = 0.f; }
            if (idx) idx[i] = -1;
            if (valid) valid[i] = 0;
        }
    }
}
extern "C" IRIS_API int iris_intersect(const iris_scene* s, const float* xs, const float* ds, int64_t B, float* pos, float* nrm, float* uv,
                              int64_t* idx, uint8_t* valid, iris_stream_t stream) {
    if (!s || B < 0 || (B > 0 && (!xs || !ds))) return fail(IRIS_ERR_ARG, "iris_intersect: bad arguments");
    if (B == 0) return IRIS_OK;
    if (s->dev.layout == kLayoutQ8 && joint_launch(B))
        hipLaunchKernelGGL((intersect_kernel<kLayoutQ8, true>), dim3(grid_for(B, kBlock, num_cus() * 6)), dim3(kBlock), 0, (hipStream_t)stream, s->dev, xs, ds, B,
                           pos, nrm, uv, idx, valid);
    else if (s->dev.layout == kLayoutQ8)
        hipLaunchKernelGGL(intersect_kernel<kLayoutQ8>, dim3(grid_for(B, kBlock, num_cus() * 6)), dim3(kBlock), 0, (hipStream_t)stream, s->dev, xs, ds, B,
                           pos, nrm, uv, idx, valid);
    else
        hipLaunchKernelGGL(intersect_kernel<kLayoutF32>, dim3(grid_for(B, kBlock, num_cus() * 6)), dim3(kBlock), 0, (hipStream_t)stream, s->dev, xs, ds, B,
                           pos, nrm, uv, idx, valid);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}

// ======================================================================================================
// a3/a4 samplers, a5 lookups, a10 lerp (unfused entry points of the call surface)
// ======================================================================================================
__global__ void sample_diffuse_kernel(const float* __restrict__ u2, const float* __restrict__ normal, int64_t B, float* __restrict__ wi,
                                      float* __restrict__ pdf, float* __restrict__ weight) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < B; i += (int64_t)gridDim.x * blockDim.x) {
        f3 n = ld3(normal + i * 3), t, b;
        normal_space(n, t, b);
        f3 d = diffuse_sampler(u2[i * 2], u2[i * 2 + 1], n, t, b);
        st3(wi + i * 3, d);
        if (pdf) pdf[i] = relu(t_dot(n, d)) / kPi;
        if (weight) st3(weight + i * 3, mk3(1.f, 1.f, 1.f));
    }
}
extern "C" IRIS_API int iris_sample_diffuse(const float* u2, const float* normal, int64_t B, float* wi, float* pdf, float* weight,
                                   iris_stream_t stream) {
    if (B < 0 || (B > 0 && (!u2 || !normal || !wi))) return fail(IRIS_ERR_ARG, "iris_sample_diffuse: bad arguments");
    if (B == 0) return IRIS_OK;
    hipLaunchKernelGGL(sample_diffuse_kernel, dim3(grid_for(B, 256, 8192)), dim3(256), 0, (hipStream_t)stream, u2, normal, B, wi, pdf, weight);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}

__global__ void sample_specular_kernel(const float* __restrict__ u2, const float* __restrict__ wo, const float* __restrict__ normal,
                                       float rough_all, const float* __restrict__ rough_each, int64_t B, float* __restrict__ wi, float* __restrict__ pdf,
                                       float* __restrict__ w0, float* __restrict__ w1) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < B; i += (int64_t)gridDim.x * blockDim.x) {
        f3 n = ld3(normal + i * 3), o = ld3(wo + i * 3), t, b;
        const float rough = rough_each ? rough_each[i] : rough_all;
        normal_space(n, t, b);
        f3 d = specular_sampler(u2[i * 2], u2[i * 2 + 1], rough, o, n, t, b);
        SpecW w = specular_weights(d, o, n, rough, pdf != nullptr);
        st3(wi + i * 3, d);
        if (pdf) pdf[i] = w.pdf;
        if (w0) w0[i] = w.g0;
        if (w1) w1[i] = w.g1;
    }
}
extern "C" IRIS_API int iris_sample_specular(const float* u2, const float* wo, const float* normal, float roughness, int64_t B, float* wi,
                                    float* pdf, float* w0, float* w1, iris_stream_t stream) {
    if (B < 0 || (B > 0 && (!u2 || !wo || !normal || !wi))) return fail(IRIS_ERR_ARG, "iris_sample_specular: bad arguments");
    if (B == 0) return IRIS_OK;
    hipLaunchKernelGGL(sample_specular_kernel, dim3(grid_for(B, 256, 8192)), dim3(256), 0, (hipStream_t)stream, u2, wo, normal, roughness,
                       (const float*)nullptr, B, wi, pdf, w0, w1);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}
extern "C" IRIS_API int iris_sample_specular_v(const float* u2, const float* wo, const float* normal, const float* roughness, int64_t B, float* wi,
                                      float* pdf, float* w0, float* w1, iris_stream_t stream) {
    if (B < 0 || (B > 0 && (!u2 || !wo || !normal || !roughness || !wi))) return fail(IRIS_ERR_ARG, "iris_sample_specular_v: bad arguments");
    if (B == 0) return IRIS_OK;
    hipLaunchKernelGGL(sample_specular_kernel, dim3(grid_for(B, 256, 8192)), dim3(256), 0, (hipStream_t)stream, u2, wo, normal, 0.f, roughness, B,
                       wi, pdf, w0, w1);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}

__global__ void slf_lookup_kernel(SlfDev s, const float* __restrict__ x, int64_t B, int64_t* __restrict__ idx, float* __restrict__ rgb) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < B; i += (int64_t)gridDim.x * blockDim.x) {
        f3 p = ld3(x + i * 3);
        int j = slf_index(s, p);
        if (idx) idx[i] = j;
        if (rgb) {
            f3 r = mk3(0.f, 0.f, 0.f);
            if (j >= 0) { float4 q = s.radiance[j]; r = mk3(q.x, q.y, q.z); }
            st3(rgb + i * 3, r);
        }
    }
}
extern "C" IRIS_API int iris_slf_lookup(const iris_slf* s, const float* x, int64_t B, int64_t* idx, float* rgb, iris_stream_t stream) {
    if (!s || B < 0 || (B > 0 && !x)) return fail(IRIS_ERR_ARG, "iris_slf_lookup: bad arguments");
    if (B == 0) return IRIS_OK;
    hipLaunchKernelGGL(slf_lookup_kernel, dim3(grid_for(B, 256, 8192)), dim3(256), 0, (hipStream_t)stream, s->dev, x, B, idx, rgb);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}

__global__ void eval_emitter_kernel(EmitDev e, SlfDev s, const float* __restrict__ pos, const int64_t* __restrict__ tri,
                                    const float* __restrict__ rough, float trace_rough, int64_t B, float* __restrict__ Le,
                                    float* __restrict__ emit_pdf, uint8_t* __restrict__ valid_next) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < B; i += (int64_t)gridDim.x * blockDim.x) {
        float pdf; bool vn;
        f3 l = eval_emitter1(e, s, ld3(pos + i * 3), tri[i], rough != nullptr, rough ? rough[i] : 0.f, trace_rough, pdf, vn);
        st3(Le + i * 3, l);
        if (emit_pdf) emit_pdf[i] = pdf;
        if (valid_next) valid_next[i] = vn ? 1 : 0;
    }
}
extern "C" IRIS_API int iris_eval_emitter(const iris_emitter* e, const iris_slf* s, const float* position, const int64_t* triangle_idx,
                                 const float* roughness, float trace_roughness, int64_t B, float* Le, float* emit_pdf,
                                 uint8_t* valid_next, iris_stream_t stream) {
    if (!e || !s || B < 0 || (B > 0 && (!position || !triangle_idx || !Le))) return fail(IRIS_ERR_ARG, "iris_eval_emitter: bad arguments");
    if (B == 0) return IRIS_OK;
    hipLaunchKernelGGL(eval_emitter_kernel, dim3(grid_for(B, 256, 8192)), dim3(256), 0, (hipStream_t)stream, e->dev, s->dev, position,
                       triangle_idx, roughness, trace_roughness, B, Le, emit_pdf, valid_next);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}

__global__ void lerp_specular_kernel(const float* __restrict__ spec, const float* __restrict__ rough, int64_t B, int R, float* __restrict__ out) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < B; i += (int64_t)gridDim.x * blockDim.x) {
        float r = (rough[i] - 0.02f) / (float)(1.0 - 0.02) * (float)(R - 1);
        int r1 = min(max((int)ceilf(r), 0), R - 1), r0 = min(max((int)floorf(r), 0), R - 1);
        float w = r - floorf(r);
        for (int c = 0; c < 3; ++c) out[i * 3 + c] = spec[(i * R + r0) * 3 + c] * (1.f - w) + spec[(i * R + r1) * 3 + c] * w;
    }
}
extern "C" IRIS_API int iris_lerp_specular(const float* specular, const float* roughness, int64_t B, int R, float* out, iris_stream_t stream) {
    if (B < 0 || R < 1 || (B > 0 && (!specular || !roughness || !out))) return fail(IRIS_ERR_ARG, "iris_lerp_specular: bad arguments");
    if (B == 0) return IRIS_OK;
    hipLaunchKernelGGL(lerp_specular_kernel, dim3(grid_for(B, 256, 8192)), dim3(256), 0, (hipStream_t)stream, specular, roughness, B, R, out);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}

// ---- multi-GPU: the gathered stripes of a view back into image order (iris_amd/sharding.py; the reference is single-GPU, bake_shading.py:41).
// gathered[r][m][j] is map m of rank r at its j-th local pixel (the rank's rows -- stripe s of `stripe` rows belongs to rank s % world -- in
// ascending order, row-major); full[m][row * W + col].  Pure index arithmetic: no index tensors, one pass, reads and writes coalesced along a row.
__global__ void unstripe_maps_kernel(const float* __restrict__ gathered, int world, int M, int64_t n_max, int H, int W, int stripe, float* __restrict__ full) {
    const int64_t n_px = (int64_t)H * W, total = n_px * M * 3;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = i / (n_px * 3), rem = i - m * n_px * 3, p = rem / 3;
        const int c = (int)(rem - p * 3);
        const int row = (int)(p / W), col = (int)(p - (int64_t)row * W);
        const int s = row / stripe, r = s % world;
        const int64_t j = ((int64_t)(s / world) * stripe + (row - s * stripe)) * W + col;
        full[i] = gathered[(((int64_t)r * M + m) * n_max + j) * 3 + c];
    }
}
extern "C" IRIS_API int iris_unstripe_maps(const float* gathered, int world, int n_maps, int64_t n_max, int H, int W, int stripe_rows, float* full, iris_stream_t stream) {
    if (world < 1 || n_maps < 0 || H < 0 || W < 0 || stripe_rows < 1 || n_max < 0) return fail(IRIS_ERR_ARG, "iris_unstripe_maps: bad arguments");
    // the largest share of a rank: rank 0 owns stripes 0, world, 2 world, ...
    const int n_stripes = (H + stripe_rows - 1) / stripe_rows;
    int64_t rows0 = 0;
    for (int s = 0; s < n_stripes; s += world) rows0 += std::min(stripe_rows, H - s * stripe_rows);
    if (n_max < rows0 * W) return fail(IRIS_ERR_ARG, "iris_unstripe_maps: n_max is smaller than the largest rank's pixel count");
    const int64_t total = (int64_t)H * W * n_maps * 3;
    if (total == 0) return IRIS_OK;
    if (!gathered || !full) return fail(IRIS_ERR_ARG, "iris_unstripe_maps: null pointer");
    hipLaunchKernelGGL(unstripe_maps_kernel, dim3(grid_for(total, 256, 16384)), dim3(256), 0, (hipStream_t)stream, gathered, world, n_maps, n_max, H, W, stripe_rows, full);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}

// ---- the small helpers of utils/ops.py as calls of their own (inside the bake / path-tracing kernels the same device functions are fused)
__global__ void normal_space_kernel(const float* normal, int64_t B, float* out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < B; i += (int64_t)gridDim.x * blockDim.x) {
        const f3 n = ld3(normal + i * 3);
        f3 t, b;
        normal_space(n, t, b);
        float* o = out + i * 9;                    // (B,3,3)[i][row][col], columns tangent, bitangent, normal
        o[0] = t.x; o[1] = b.x; o[2] = n.x; o[3] = t.y; o[4] = b.y; o[5] = n.y; o[6] = t.z; o[7] = b.z; o[8] = n.z;
    }
}
__global__ void double_sided_kernel(const float* V, float* N, int64_t B) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < B; i += (int64_t)gridDim.x * blockDim.x) {
        const f3 v = ld3(V + i * 3), n = ld3(N + i * 3);
        if (t_dot(n, v) < 0.f) st3(N + i * 3, mk3(-n.x, -n.y, -n.z));
    }
}
__global__ void angle2xyz_kernel(const float* theta, const float* phi, int64_t B, float* out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < B; i += (int64_t)gridDim.x * blockDim.x) {
        const float st = sinf(theta[i]), ct = cosf(theta[i]), sp = sinf(phi[i]), cp = cosf(phi[i]);
        st3(out + i * 3, t_normalize(mk3(st * cp, st * sp, ct)));
    }
}
// op 0: D_GGX(a = cos_h, b = eta)   1: G1_GGX_Schlick(a = NoV, b = eta)   2: G_Smith(a = NoV, b = NoL, c = eta)
//    3: fresnelSchlick(a = VoH, b = F0)   4: fresnelSchlick_sep(a = VoH) -> out = 1 - x, out2 = x      (x = (1 - VoH)^5)
__global__ void ggx_terms_kernel(int op, const float* a, const float* b, const float* c, int64_t B, float* out, float* out2) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < B; i += (int64_t)gridDim.x * blockDim.x) {
        const float x = a[i];
        if (op == 0) out[i] = D_GGX(x, b[i]);
        else if (op == 1) out[i] = G1_GGX_Schlick(x, b[i]);
        else if (op == 2) out[i] = G1_GGX_Schlick(b[i], c[i]) * G1_GGX_Schlick(x, c[i]);
        else {
            const float y = 1.f - x, y2 = y * y, p5 = y2 * y2 * y;
            if (op == 3) out[i] = b[i] + (1.f - b[i]) * p5;
            else { out[i] = 1.f - p5; out2[i] = p5; }
        }
    }
}
extern "C" IRIS_API int iris_get_normal_space(const float* normal, int64_t B, float* out, iris_stream_t stream) {
    if (B < 0 || (B > 0 && (!normal || !out))) return fail(IRIS_ERR_ARG, "iris_get_normal_space: bad arguments");
    if (B == 0) return IRIS_OK;
    hipLaunchKernelGGL(normal_space_kernel, dim3(grid_for(B, 256, 8192)), dim3(256), 0, (hipStream_t)stream, normal, B, out);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}
extern "C" IRIS_API int iris_double_sided(const float* V, float* N, int64_t B, iris_stream_t stream) {
    if (B < 0 || (B > 0 && (!V || !N))) return fail(IRIS_ERR_ARG, "iris_double_sided: bad arguments");
    if (B == 0) return IRIS_OK;
    hipLaunchKernelGGL(double_sided_kernel, dim3(grid_for(B, 256, 8192)), dim3(256), 0, (hipStream_t)stream, V, N, B);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}
extern "C" IRIS_API int iris_angle2xyz(const float* theta, const float* phi, int64_t B, float* out, iris_stream_t stream) {
    if (B < 0 || (B > 0 && (!theta || !phi || !out))) return fail(IRIS_ERR_ARG, "iris_angle2xyz: bad arguments");
    if (B == 0) return IRIS_OK;
    hipLaunchKernelGGL(angle2xyz_kernel, dim3(grid_for(B, 256, 8192)), dim3(256), 0, (hipStream_t)stream, theta, phi, B, out);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}
extern "C" IRIS_API int iris_ggx_terms(int op, const float* a, const float* b, const float* c, int64_t B, float* out, float* out2, iris_stream_t stream) {
    const bool need_b = op != 4, need_c = op == 2, need_2 = op == 4;
    if (op < 0 || op > 4 || B < 0 || (B > 0 && (!a || !out || (need_b && !b) || (need_c && !c) || (need_2 && !out2))))
        return fail(IRIS_ERR_ARG, "iris_ggx_terms: bad arguments");
    if (B == 0) return IRIS_OK;
    hipLaunchKernelGGL(ggx_terms_kernel, dim3(grid_for(B, 256, 8192)), dim3(256), 0, (hipStream_t)stream, op, a, b, c, B, out, out2);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}

// ---- 8(f)-3: packed shading cache + shading combine (iris_cache.h)
extern "C" IRIS_API int iris_cache_row_floats(int R) { return (R < 1 || R > kMaxLevels) ? 0 : cache_row_floats(R); }
extern "C" IRIS_API int iris_cache_pack(const float* diffuse, const float* const* spec0, const float* const* spec1, int64_t n, int R, float* rows,
                                        iris_stream_t stream) {
    if (n < 0 || R < 1 || R > kMaxLevels || (n > 0 && (!diffuse || !spec0 || !spec1 || !rows))) return fail(IRIS_ERR_ARG, "iris_cache_pack: bad arguments");
    if (n == 0) return IRIS_OK;
    CacheMaps m{};
    m.diffuse = diffuse;
    for (int j = 0; j < R; ++j) {
        if (!spec0[j] || !spec1[j]) return fail(IRIS_ERR_ARG, "iris_cache_pack: null map");
        m.s0[j] = spec0[j]; m.s1[j] = spec1[j];
    }
    hipLaunchKernelGGL(cache_pack_kernel, dim3(grid_for(n * (cache_row_floats(R) / 4), 256, 16384)), dim3(256), 0, (hipStream_t)stream, m, n, R, rows);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}
extern "C" IRIS_API int iris_cache_gather(const float* rows, const int64_t* idx, int64_t B, int R, float* out, iris_stream_t stream) {
    if (B < 0 || R < 1 || R > kMaxLevels || (B > 0 && (!rows || !out))) return fail(IRIS_ERR_ARG, "iris_cache_gather: bad arguments");
    if (B == 0) return IRIS_OK;
    hipLaunchKernelGGL(cache_gather_kernel, dim3(grid_for(B * (3 + 6 * R), 256, 16384)), dim3(256), 0, (hipStream_t)stream, rows, idx, B, R, out);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}
extern "C" IRIS_API int iris_shade_cached_fwd(const float* rows, const int64_t* idx, const float* albedo, const float* metallic,
                                              const float* roughness, int64_t B, int R, float* L, iris_stream_t stream) {
    if (B < 0 || R < 1 || R > kMaxLevels || (B > 0 && (!rows || !albedo || !metallic || !roughness || !L)))
        return fail(IRIS_ERR_ARG, "iris_shade_cached_fwd: bad arguments");
    if (B == 0) return IRIS_OK;
    hipLaunchKernelGGL(shade_cached_fwd_kernel, dim3(grid_for(B, 256, 16384)), dim3(256), 0, (hipStream_t)stream, rows, idx, albedo, metallic,
                       roughness, B, R, L);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}
extern "C" IRIS_API int iris_shade_cached_bwd(const float* rows, const int64_t* idx, const float* albedo, const float* metallic,
                                              const float* roughness, const float* gL, int64_t B, int R, float* g_albedo, float* g_metallic,
                                              float* g_roughness, iris_stream_t stream) {
    if (B < 0 || R < 1 || R > kMaxLevels || (B > 0 && (!rows || !albedo || !metallic || !roughness || !gL)))
        return fail(IRIS_ERR_ARG, "iris_shade_cached_bwd: bad arguments");
    if (B == 0) return IRIS_OK;
    hipLaunchKernelGGL(shade_cached_bwd_kernel, dim3(grid_for(B, 256, 16384)), dim3(256), 0, (hipStream_t)stream, rows, idx, albedo, metallic,
                       roughness, gL, B, R, g_albedo, g_metallic, g_roughness);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}

// ---- 8(f)-4: denoiser substitute (iris_denoise.h)
extern "C" IRIS_API uint64_t iris_denoise_workspace_bytes(int H, int W) {
    if (H < 1 || W < 1) return 0;
    return (uint64_t)H * W * (2 + 2 * kDnMaxMaps) * sizeof(float4);
}
template <int M>
static void denoise_group(const DnParams& P, const DnMaps& mp, const float4* g0, const float4* g1, int iterations, hipStream_t st) {
    const dim3 grid((P.W + 15) / 16, (P.H + 15) / 16), block(256);
    hipLaunchKernelGGL((dn_variance_kernel<M>), grid, block, 0, st, P, mp, g0, g1);
    DnMaps cur = mp;
    for (int it = 0; it < iterations; ++it) {
        if (it == iterations - 1) hipLaunchKernelGGL((dn_atrous_kernel<M, true>), grid, block, 0, st, P, cur, g0, g1, 1 << it);
        else hipLaunchKernelGGL((dn_atrous_kernel<M, false>), grid, block, 0, st, P, cur, g0, g1, 1 << it);
        for (int m = 0; m < M; ++m) std::swap(cur.a[m], cur.b[m]);
    }
}
extern "C" IRIS_API int iris_denoise(const float* normal, const float* position, const uint8_t* valid, int H, int W, int n_maps,
                                     const float* const* in, float* const* out, int iterations, float sigma_l, float sigma_n, float sigma_p,
                                     void* workspace, uint64_t workspace_bytes, iris_stream_t stream) {
    if (H < 1 || W < 1 || n_maps < 0 || iterations < 1 || iterations > 8 || !(sigma_l > 0.f) || !(sigma_n >= 0.f) || !(sigma_p > 0.f) ||
        (n_maps > 0 && (!in || !out)))
        return fail(IRIS_ERR_ARG, "iris_denoise: bad arguments");
    if (!workspace || workspace_bytes < iris_denoise_workspace_bytes(H, W))
        return fail(IRIS_ERR_ARG, "iris_denoise: workspace of iris_denoise_workspace_bytes() bytes required");
    if (n_maps == 0) return IRIS_OK;
    for (int m = 0; m < n_maps; ++m) if (!in[m] || !out[m]) return fail(IRIS_ERR_ARG, "iris_denoise: null map");
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = (int64_t)H * W;
    float4* g0 = (float4*)workspace;
    float4* g1 = g0 + n;
    float4* bufs = g1 + n;
    hipLaunchKernelGGL(dn_guides_kernel, dim3(grid_for(n, 256, 8192)), dim3(256), 0, st, normal, position, valid, n, g0, g1);
    DnParams P{H, W, sigma_l, sigma_n, sigma_p};
    for (int m0 = 0; m0 < n_maps; m0 += kDnMaxMaps) {
        const int M = std::min(kDnMaxMaps, n_maps - m0);
        DnMaps mp{};
        for (int m = 0; m < M; ++m) { mp.in[m] = in[m0 + m]; mp.out[m] = out[m0 + m]; mp.a[m] = bufs + (int64_t)(2 * m) * n; mp.b[m] = bufs + (int64_t)(2 * m + 1) * n; }
        switch (M) {
            case 1: denoise_group<1>(P, mp, g0, g1, iterations, st); break;
            case 2: denoise_group<2>(P, mp, g0, g1, iterations, st); break;
            case 3: denoise_group<3>(P, mp, g0, g1, iterations, st); break;
            default: denoise_group<4>(P, mp, g0, g1, iterations, st); break;
        }
    }
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}

__global__ void philox_kernel(uint64_t seed, uint64_t idx0, uint32_t stream_id, int64_t n, float* __restrict__ u2) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float a, b;
        philox_u2(seed, idx0 + (uint64_t)i, stream_id, a, b);
        u2[i * 2] = a; u2[i * 2 + 1] = b;
    }
}
extern "C" IRIS_API int iris_philox_u2(uint64_t seed, uint64_t idx0, uint32_t stream_id, int64_t n, float* u2, iris_stream_t stream) {
    if (n < 0 || (n > 0 && !u2)) return fail(IRIS_ERR_ARG, "iris_philox_u2: bad arguments");
    if (n == 0) return IRIS_OK;
    hipLaunchKernelGGL(philox_kernel, dim3(grid_for(n, 256, 8192)), dim3(256), 0, (hipStream_t)stream, seed, idx0, stream_id, n, u2);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}

// ======================================================================================================
// a3..a7 fused bake kernels (iris_bake.h)
// ======================================================================================================
// Pixels per tile.  ~5000-ray tiles are the optimum at 1080p x SPP 128 (with 32-B slots: 8192 rays: 324 ms per view, 6144: 322, 5120: 319,
// 4096: 308, 3072: 314, 2048: 326; with today's 24-B slots 4096: 6.99, 5120: 7.04, 6144: 6.98 Grays/s): larger tiles push the workgroups' slot
// slabs out of the 256 MB Infinity Cache, smaller ones pay more low-utilisation drains (one per wave and tile).  The LDS ray list is sized for
// exactly that (kTileRays = 5120), which is what lets 7 workgroups share a CU.  Round 3 (watertight leaf test, phase threshold 12): 3072-ray tiles
// 7.37, 3584: 7.42, 4096: 7.50, 4608: 7.45, 5120: 7.43 Grays/s -> the default target is 4096 (spp above it still get kTileRays).
// iris_debug_set("tile_target_rays") overrides.
constexpr int kTileTarget = 4096;
static int tile_pixels(int spp) {
    const int target = g_opt_tile_target_rays > 0 ? (int)std::min<long long>(kTileRays, std::max<long long>(64, g_opt_tile_target_rays)) : kTileTarget;
    return std::max(1, std::min(kTileRays, std::max(target, spp)) / spp);
}
#ifndef IRIS_TILE_GRID
#define IRIS_TILE_GRID IRIS_TILE_WAVES
#endif
static int bake_grid_blocks() { return num_cus() * IRIS_TILE_GRID; }  // resident 256-thread workgroups per CU (VGPR- and LDS-bound)
static int view_grid_blocks() { return bake_grid_blocks(); }
static uint64_t stack_ovf_bytes() {
    return (uint64_t)std::max(bake_grid_blocks(), view_grid_blocks()) * (kStackCapacity - IRIS_TILE_STACK) * kBlock * sizeof(uint32_t);
}

static uint64_t park_bytes() {      // IRIS_PARK: wave-private pools of parked rays (iris_trace.h)
    return IRIS_PARK ? (uint64_t)std::max(bake_grid_blocks(), view_grid_blocks()) * (kBlock / 64) * kParkCap * kParkWords4 * 16 : 0;
}
extern "C" IRIS_API int iris_bake_tile_max_spp(void) { return kTileRays; }
extern "C" IRIS_API uint64_t iris_bake_workspace_bytes(int64_t P, int spp, int specular) {
    if (spp < 1 || spp > kTileRays || P < 0) return 0;  // v1 kernel only
    // [256 B counters][blocks x kTileRays x (16|32) B per-ray slots]
    // (packing the pixel tensors into 48-B records was measured 7 % SLOWER than reading pos/nrm/wo directly: not done)
    // + [blocks x (96 - LDS depth) x 256 dwords: traversal-stack entries beyond the LDS part]   (specular sizing also serves iris_bake_view)
    const uint64_t blocks = (uint64_t)std::max(bake_grid_blocks(), view_grid_blocks());
    return 256 + blocks * kTileRays * (specular ? 2 : 1) * sizeof(float4) + park_bytes() + stack_ovf_bytes();
}

static const float4* fused_tris(const iris_scene* sc, const iris_emitter* em, hipStream_t st);
static int bake_launch(bool spec, const iris_scene* sc, const iris_emitter* em, const iris_slf* slf, const float* pos, const float* nrm,
                       const float* wo, float rough, int64_t P, int spp, const float* u2, uint64_t seed, uint32_t stream_id,
                       const int32_t* pix_id, float* out0, float* out1, int64_t* tri_next, int64_t* src_next, uint64_t* stats, int variant,
                       void* workspace, uint64_t workspace_bytes, iris_stream_t stream) {
    if (!sc || !em || !slf || P < 0 || spp < 1 || (P > 0 && (!pos || !nrm || !out0 || (spec && (!wo || !out1)))))
        return fail(IRIS_ERR_ARG, "iris_bake: bad arguments");
    if (variant < IRIS_BAKE_AUTO || variant > IRIS_BAKE_TILE_SORTED) return fail(IRIS_ERR_ARG, "iris_bake: unknown kernel variant");
    if (em->dev.nf != sc->info.n_triangles)      // eval_emitter indexes is_emitter / emitter_idx by the hit triangle (model/emitter.py:196-203)
        return fail(IRIS_ERR_ARG, "iris_bake: the emitter tables were built for a mesh with a different number of triangles than the scene");
    if (P == 0) return IRIS_OK;
    BakeArgs a{};
    a.sc = sc->dev; a.em = em->dev; a.slf = slf->dev;
    a.pos = pos; a.nrm = nrm; a.wo = wo; a.u2 = u2; a.pix_id = pix_id;
    a.P = P; a.spp = spp; a.seed = seed; a.stream_id = stream_id; a.rough = rough;
    a.out0 = out0; a.out1 = out1; a.tri_next = tri_next; a.src_next = src_next; a.stats = (unsigned long long*)stats;
    const uint64_t need = iris_bake_workspace_bytes(P, spp, spec ? 1 : 0);
    bool tiled = variant != IRIS_BAKE_PIXEL_PER_WAVE && need > 0 && workspace && workspace_bytes >= need;
    if (variant == IRIS_BAKE_TILE_SORTED && !tiled)
        return fail(IRIS_ERR_ARG, "iris_bake: the tile-sorted kernel needs spp <= iris_bake_tile_max_spp() and a workspace of iris_bake_workspace_bytes()");
    hipStream_t st = (hipStream_t)stream;
    if (tiled) {
        const int blocks = bake_grid_blocks();
        int tile_px = tile_pixels(spp);                      // ~4096 rays per tile (at most what fits the LDS ray list) ...
        int tiles_per_block = 4;                             // ... but at least ~4 tiles per workgroup, so that the dynamic tile queue
        if (g_opt_tiles_per_block > 0) tiles_per_block = (int)g_opt_tiles_per_block;                  // balances (iris_debug_set("tiles_per_block"))
        const int64_t even = (P + (int64_t)blocks * tiles_per_block - 1) / ((int64_t)blocks * tiles_per_block);
        if (even < tile_px) tile_px = (int)std::max<int64_t>(even, std::min(tile_px, 16));   // not below 16 px: the sort needs rays
        if (tile_px < 1) tile_px = 1;
        a.tile_px = tile_px;
        a.tile_counter = (unsigned int*)workspace;
        a.scratch = (float4*)((char*)workspace + 256);
        a.stack_ovf = (uint32_t*)((char*)workspace + need - stack_ovf_bytes());
#if IRIS_PARK
        a.park = (iris_u4v*)((char*)workspace + need - stack_ovf_bytes() - park_bytes());
#endif
        HIP_TRY(hipMemsetAsync(workspace, 0, 256, st));
        if (const float4* ft = fused_tris(sc, em, st)) { a.sc.tris = ft; a.em.emit_ord = nullptr; }      // (tile kernels only: their shading pass reads the ordinal from the record)
        const int64_t n_tiles = (P + tile_px - 1) / tile_px;
        const int grid = (int)std::min<int64_t>(blocks, n_tiles);
#define IRIS_LAUNCH_BAKE(KERNEL, GRID)                                                                                            \
    do {                                                                                                                         \
        const bool q8 = a.sc.layout == kLayoutQ8;                                                                                \
        if (stats) {                                                                                                             \
            if (spec) { if (q8) hipLaunchKernelGGL((KERNEL<true, true, kLayoutQ8>), dim3(GRID), dim3(kBlock), 0, st, a);          \
                        else hipLaunchKernelGGL((KERNEL<true, true, kLayoutF32>), dim3(GRID), dim3(kBlock), 0, st, a); }          \
            else      { if (q8) hipLaunchKernelGGL((KERNEL<false, true, kLayoutQ8>), dim3(GRID), dim3(kBlock), 0, st, a);         \
                        else hipLaunchKernelGGL((KERNEL<false, true, kLayoutF32>), dim3(GRID), dim3(kBlock), 0, st, a); }         \
        } else {                                                                                                                 \
            if (spec) { if (q8) hipLaunchKernelGGL((KERNEL<true, false, kLayoutQ8>), dim3(GRID), dim3(kBlock), 0, st, a);         \
                        else hipLaunchKernelGGL((KERNEL<true, false, kLayoutF32>), dim3(GRID), dim3(kBlock), 0, st, a); }         \
            else      { if (q8) hipLaunchKernelGGL((KERNEL<false, false, kLayoutQ8>), dim3(GRID), dim3(kBlock), 0, st, a);        \
                        else hipLaunchKernelGGL((KERNEL<false, false, kLayoutF32>), dim3(GRID), dim3(kBlock), 0, st, a); }        \
        }                                                                                                                        \
    } while (0)
        IRIS_LAUNCH_BAKE(bake_tile_kernel, grid);
    } else {
        const int ppw = (spp < 64 && (spp & (spp - 1)) == 0) ? 64 / spp : 1;
        const int64_t n_groups = (P + ppw - 1) / ppw;
        const int grid = grid_for(n_groups * 64, kBlock, num_cus() * 6);
        IRIS_LAUNCH_BAKE(bake_kernel, grid);
    }
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}
// All lobes of a view in one launch (see bake_view_kernel).  roughness[l] < 0 selects the diffuse lobe.
__global__ void fuse_ord_kernel(const float4* __restrict__ src, const int32_t* __restrict__ emit_ord, int64_t n_records, float4* __restrict__ dst) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_records; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 x = src[i * 4], y = src[i * 4 + 1], z = src[i * 4 + 2];
        const int id = __float_as_int(x.w);
        const int ord = id >= 0 ? emit_ord[id] : -1;              // (the degenerate record unused child slots point to carries id -1)
        dst[i * 4] = x; dst[i * 4 + 1] = y; dst[i * 4 + 2] = z; dst[i * 4 + 3] = make_float4(__int_as_float(ord), 0.f, 0.f, 0.f);
    }
}
// the scene's leaf records with the emitter's ordinals in the fourth plane; NULL when they cannot be had (memory): the caller then keeps the plain table + the ordinal gather
static const float4* fused_tris(const iris_scene* sc, const iris_emitter* em, hipStream_t st) {
    if (IRIS_NO_FUSED_RECORDS) return nullptr;
    std::lock_guard<std::mutex> lock(em->fused_mu);
    for (auto& f : em->fused) if (f.scene_uid == sc->uid) {
        // the table is filled by a kernel on the stream of the call that built it: a call on another stream waits for that kernel ON THE DEVICE (no host
        // synchronisation inside a bake call -- legal under stream capture) until the event has been seen complete once
        if (!f.done) {
            if (hipEventQuery(f.ready) == hipSuccess) f.done = true;
            else { (void)hipGetLastError(); if (hipStreamWaitEvent(st, f.ready, 0) != hipSuccess) { (void)hipGetLastError(); return nullptr; } }
        }
        return (const float4*)f.d_tris;
    }
    if (em->fused.size() >= 2) { em->retired.push_back(em->fused.front()); em->fused.erase(em->fused.begin()); }      // (a training loop uses one scene: the live set stays at two)
    int prev = 0;
    if (hipGetDevice(&prev) != hipSuccess || hipSetDevice(sc->device) != hipSuccess) { (void)hipGetLastError(); return nullptr; }     // allocate where the scene lives, whatever the caller's current device
    void* d = nullptr; hipEvent_t ev = nullptr;
    const int64_t n_rec = (int64_t)sc->dev.n_tris + 1;
    bool ok = hipMalloc(&d, (size_t)n_rec * 64) == hipSuccess && hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(fuse_ord_kernel, dim3(grid_for(n_rec, 256, 4096)), dim3(256), 0, st, sc->dev.tris, em->dev.emit_ord, n_rec, (float4*)d);
        ok = hipGetLastError() == hipSuccess && hipEventRecord(ev, st) == hipSuccess;
    }
    (void)hipSetDevice(prev);
    if (!ok) { (void)hipGetLastError(); if (d) (void)hipFree(d); if (ev) (void)hipEventDestroy(ev); return nullptr; }
    em->fused.push_back({sc->uid, d, ev, false});
    return (const float4*)d;            // (this call's own launch follows the fill on the same stream)
}
extern "C" IRIS_API int iris_bake_view(const iris_scene* sc, const iris_emitter* em, const iris_slf* slf, const float* pos, const float* nrm,
                              const float* wo, const int32_t* pix_id, int64_t P, int n_lobes, const float* roughness, const int32_t* spp,
                              const uint32_t* stream_ids, uint64_t seed, float* const* out0, float* const* out1, void* workspace,
                              uint64_t workspace_bytes, iris_stream_t stream) {
    if (!sc || !em || !slf || P < 0 || n_lobes < 1 || n_lobes > kMaxLobes || !roughness || !spp || !stream_ids || !out0 || !out1 ||
        (P > 0 && (!pos || !nrm || !wo)))
        return fail(IRIS_ERR_ARG, "iris_bake_view: bad arguments");
    if (em->dev.nf != sc->info.n_triangles)
        return fail(IRIS_ERR_ARG, "iris_bake_view: the emitter tables were built for a mesh with a different number of triangles than the scene");
    if (P == 0) return IRIS_OK;
    const uint64_t need = iris_bake_workspace_bytes(P, 1, 1);
    if (!workspace || workspace_bytes < need) return fail(IRIS_ERR_ARG, "iris_bake_view: workspace of iris_bake_workspace_bytes() bytes required");
    ViewArgs v{};
    v.base.sc = sc->dev; v.base.em = em->dev; v.base.slf = slf->dev;
    if (const float4* ft = fused_tris(sc, em, (hipStream_t)stream)) { v.base.sc.tris = ft; v.base.em.emit_ord = nullptr; }      // the shading pass reads the ordinal from the record
    v.base.pos = pos; v.base.nrm = nrm; v.base.wo = wo; v.base.pix_id = pix_id; v.base.P = P; v.base.seed = seed;
    v.base.tile_counter = (unsigned int*)workspace;
    v.base.scratch = (float4*)((char*)workspace + 256);
    v.base.stack_ovf = (uint32_t*)((char*)workspace + need - stack_ovf_bytes());
#if IRIS_PARK
    v.base.park = (iris_u4v*)((char*)workspace + need - stack_ovf_bytes() - park_bytes());
#endif
    v.n_lobes = n_lobes;
    const int blocks = view_grid_blocks();
    long long t = 0;
    int tile_px_min = kTileRays;
    for (int l = 0; l < n_lobes; ++l) {
        if (spp[l] < 1 || spp[l] > kTileRays) return fail(IRIS_ERR_ARG, "iris_bake_view: spp must be in [1, iris_bake_tile_max_spp()]");
        if (!out0[l] || (roughness[l] >= 0.f && !out1[l])) return fail(IRIS_ERR_ARG, "iris_bake_view: null output");
        int tile_px = tile_pixels(spp[l]);
        const int64_t even = (P * n_lobes + (int64_t)blocks * 4 - 1) / ((int64_t)blocks * 4);   // >= ~4 tiles per workgroup over the whole view
        if (even < tile_px) tile_px = (int)std::max<int64_t>(even, std::min(tile_px, 16));
        if (tile_px < 1) tile_px = 1;
        v.lobe[l].rough = roughness[l]; v.lobe[l].spp = spp[l]; v.lobe[l].stream_id = stream_ids[l]; v.lobe[l].spec = roughness[l] >= 0.f ? 1 : 0;
        v.lobe[l].tile_px = tile_px; v.lobe[l].out0 = out0[l]; v.lobe[l].out1 = out1[l];
        tile_px_min = std::min(tile_px_min, tile_px);
        t += (P + tile_px - 1) / tile_px;
    }
    // the queue's virtual numbering (ViewArgs): spans of kTileChunk tiles of the lobe with the smallest tiles
    v.span_px = kTileChunk * tile_px_min;
    for (int l = 0; l < n_lobes; ++l) v.lobe[l].tiles_per_span = (v.span_px + v.lobe[l].tile_px - 1) / v.lobe[l].tile_px;
    const long long n_spans = (P + v.span_px - 1) / v.span_px;
    hipStream_t st = (hipStream_t)stream;
    v.n_tiles = ((n_spans + 7) / 8) * 8 * n_lobes * kTileChunk;
    HIP_TRY(hipMemsetAsync(workspace, 0, 256, st));
    const int grid = (int)std::min<long long>(blocks, t);
    if (v.base.sc.layout == kLayoutQ8) hipLaunchKernelGGL(bake_view_kernel<kLayoutQ8>, dim3(grid), dim3(kBlock), 0, st, v);
    else hipLaunchKernelGGL(bake_view_kernel<kLayoutF32>, dim3(grid), dim3(kBlock), 0, st, v);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}
extern "C" IRIS_API int iris_bake_diffuse(const iris_scene* sc, const iris_emitter* em, const iris_slf* slf, const float* pos, const float* nrm,
                                 int64_t P, int spp, const float* u2, uint64_t seed, uint32_t stream_id, const int32_t* pix_id,
                                 float* Ld, int64_t* tri_next, void* workspace, uint64_t workspace_bytes, iris_stream_t stream) {
    return bake_launch(false, sc, em, slf, pos, nrm, nullptr, -1.f, P, spp, u2, seed, stream_id, pix_id, Ld, nullptr, tri_next, nullptr, nullptr,
                       IRIS_BAKE_AUTO, workspace, workspace_bytes, stream);
}
extern "C" IRIS_API int iris_bake_specular(const iris_scene* sc, const iris_emitter* em, const iris_slf* slf, const float* pos, const float* nrm,
                                  const float* wo, float roughness, int64_t P, int spp, const float* u2, uint64_t seed,
                                  uint32_t stream_id, const int32_t* pix_id, float* Ls0, float* Ls1, int64_t* tri_next,
                                  void* workspace, uint64_t workspace_bytes, iris_stream_t stream) {
    return bake_launch(true, sc, em, slf, pos, nrm, wo, roughness, P, spp, u2, seed, stream_id, pix_id, Ls0, Ls1, tri_next, nullptr, nullptr,
                       IRIS_BAKE_AUTO, workspace, workspace_bytes, stream);
}
// diagnostics twins (iris_hip_debug.h): kernel variant, instrumented build, per-sample source rows
extern "C" IRIS_API int iris_debug_bake_diffuse(const iris_scene* sc, const iris_emitter* em, const iris_slf* slf, const float* pos, const float* nrm,
                                       int64_t P, int spp, const float* u2, uint64_t seed, uint32_t stream_id, const int32_t* pix_id,
                                       float* Ld, int64_t* tri_next, int64_t* src_next, uint64_t* stats, int variant, void* workspace,
                                       uint64_t workspace_bytes, iris_stream_t stream) {
    return bake_launch(false, sc, em, slf, pos, nrm, nullptr, -1.f, P, spp, u2, seed, stream_id, pix_id, Ld, nullptr, tri_next, src_next, stats, variant,
                       workspace, workspace_bytes, stream);
}
extern "C" IRIS_API int iris_debug_bake_specular(const iris_scene* sc, const iris_emitter* em, const iris_slf* slf, const float* pos, const float* nrm,
                                        const float* wo, float roughness, int64_t P, int spp, const float* u2, uint64_t seed,
                                        uint32_t stream_id, const int32_t* pix_id, float* Ls0, float* Ls1, int64_t* tri_next, int64_t* src_next,
                                        uint64_t* stats, int variant, void* workspace, uint64_t workspace_bytes, iris_stream_t stream) {
    return bake_launch(true, sc, em, slf, pos, nrm, wo, roughness, P, spp, u2, seed, stream_id, pix_id, Ls0, Ls1, tri_next, src_next, stats, variant,
                       workspace, workspace_bytes, stream);
}

#ifdef IRIS_PHASE_TIMING
// diagnostic build only (tools/diag_phases.py): read / reset the per-phase cycle sums of the tile kernels
extern "C" IRIS_API int iris_debug_phase_cycles(unsigned long long* out8, int reset) {
    HIP_TRY(hipDeviceSynchronize());
    if (out8) HIP_TRY(hipMemcpyFromSymbol(out8, HIP_SYMBOL(iris::g_phase_cycles), 64));
    if (reset) { unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0}; HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(iris::g_phase_cycles), z, 64)); }
    return IRIS_OK;
}
#endif

// ======================================================================================================
// a9 (cfg 5): path_tracing_single building blocks and stages (iris_pt.h)
// ======================================================================================================
// Large batches go through the direction-sorted, persistent-lane tile kernel (iris_pt.h); small ones (cfg 5: 262 144 rays) keep the
// one-ray-per-thread kernels, which expose more parallelism.  iris_debug_set("pt_tile_min") overrides the switch-over (tests force the tile path).
static bool pt_tiling(int64_t N, int& tile_rays, int& grid) {
    const int64_t blocks = (int64_t)num_cus() * IRIS_PT_WAVES;
    int64_t min_n = 512 * blocks;
    if (g_opt_pt_tile_min >= 0) min_n = g_opt_pt_tile_min;
    if (N < min_n) return false;
    int64_t t = (N / (2 * blocks) + kBlock - 1) / kBlock * kBlock;          // >= 2 tiles per resident workgroup ...
    tile_rays = (int)std::min<int64_t>(kPtTileCap, std::max<int64_t>(kBlock, t));   // ... of 256 .. 4096 rays
    grid = (int)std::min<int64_t>(blocks, (N + tile_rays - 1) / tile_rays);
    return true;
}
#define LAUNCH1D(kernel, n, st, ...)                                                                          \
    do {                                                                                                      \
        hipLaunchKernelGGL(kernel, dim3(grid_for((n), 256, 8192)), dim3(256), 0, (hipStream_t)(st), __VA_ARGS__); \
        HIP_TRY(hipGetLastError());                                                                           \
    } while (0)

extern "C" IRIS_API int iris_sample_emitter(const iris_emitter* e, const float* s1, const float* s2, const float* position, int64_t N, float* wi,
                                   float* pdf, int64_t* tri, iris_stream_t stream) {
    if (!e || !e->can_sample) return fail(IRIS_ERR_ARG, "iris_sample_emitter: emitter was created without vertices / cdf");
    if (N < 0 || (N > 0 && (!s1 || !s2 || !position || !wi || !pdf || !tri))) return fail(IRIS_ERR_ARG, "iris_sample_emitter: bad arguments");
    if (N == 0) return IRIS_OK;
    LAUNCH1D(sample_emitter_kernel, N, stream, e->sample, s1, s2, position, N, wi, pdf, tri);
    return IRIS_OK;
}
extern "C" IRIS_API int iris_eval_brdf(const float* wi, const float* wo, const float* normal, const float* albedo, const float* roughness,
                              const float* metallic, int64_t N, float* brdf, float* pdf, iris_stream_t stream) {
    if (N < 0 || (N > 0 && (!wi || !wo || !normal || !albedo || !roughness || !metallic || !brdf || !pdf))) return fail(IRIS_ERR_ARG, "iris_eval_brdf: bad arguments");
    if (N == 0) return IRIS_OK;
    LAUNCH1D(eval_brdf_kernel, N, stream, wi, wo, normal, albedo, roughness, metallic, N, brdf, pdf);
    return IRIS_OK;
}
extern "C" IRIS_API int iris_sample_brdf(const float* s1, const float* s2, const float* wo, const float* normal, const float* albedo,
                                const float* roughness, const float* metallic, int64_t N, float* wi, float* pdf, float* weight,
                                iris_stream_t stream) {
    if (N < 0 || (N > 0 && (!s1 || !s2 || !wo || !normal || !albedo || !roughness || !metallic || !wi || !pdf || !weight)))
        return fail(IRIS_ERR_ARG, "iris_sample_brdf: bad arguments");
    if (N == 0) return IRIS_OK;
    LAUNCH1D(sample_brdf_kernel, N, stream, s1, s2, wo, normal, albedo, roughness, metallic, N, wi, pdf, weight);
    return IRIS_OK;
}
extern "C" IRIS_API int iris_pt_jitter(const float* rays_d, const float* dxdu, const float* dydv, const float* dudv, int64_t B, int spp, float* wi,
                              iris_stream_t stream) {
    if (B < 0 || spp < 1 || (B > 0 && (!rays_d || !dxdu || !dydv || !dudv || !wi))) return fail(IRIS_ERR_ARG, "iris_pt_jitter: bad arguments");
    if (B == 0) return IRIS_OK;
    LAUNCH1D(pt_jitter_kernel, B * spp, stream, rays_d, dxdu, dydv, dudv, B, spp, wi);
    return IRIS_OK;
}
__global__ void pt_primary_emit_kernel(EmitDev e, const int64_t* __restrict__ tri, int64_t N, int32_t* __restrict__ e0, uint8_t* __restrict__ valid_next) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
        const bool vis = tri[i] != -1;
        const int ord = vis ? e.emit_ord[tri[i]] : -1;
        e0[i] = ord;
        valid_next[i] = (vis && ord < 0) ? 1 : 0;
    }
}
extern "C" IRIS_API int iris_pt_primary_emit(const iris_emitter* e, const int64_t* tri, int64_t N, int32_t* e0, uint8_t* valid_next, iris_stream_t stream) {
    if (!e || N < 0 || (N > 0 && (!tri || !e0 || !valid_next))) return fail(IRIS_ERR_ARG, "iris_pt_primary_emit: bad arguments");
    if (N == 0) return IRIS_OK;
    LAUNCH1D(pt_primary_emit_kernel, N, stream, e->dev, tri, N, e0, valid_next);
    return IRIS_OK;
}
// The head of path_tracing_single's un-compacted mode as ONE launch (round 5): :338-340 jitter -> :343 ray_intersect(rays_o.repeat_interleave(spp), wi) -> :344 the primary hit's
// emitter ordinal -> which paths continue (path_of) -> wo = -wi.  The same arithmetic as iris_pt_jitter + iris_intersect + iris_pt_primary_emit and the torch glue between them
// (repeat_interleave, where, neg): six launches of a 0.5 ms call.
template <int LAYOUT, bool JOINT>
__global__ __launch_bounds__(kBlock) void pt_primary_kernel(SceneDev sc, EmitDev em, const float* __restrict__ rays_o, const float* __restrict__ rays_d, const float* __restrict__ dxdu,
                                                            const float* __restrict__ dydv, const float* __restrict__ dudv, int64_t B, int spp, float* __restrict__ wi_out,
                                                            float* __restrict__ wo_out, float* __restrict__ pos, float* __restrict__ nrm, int32_t* __restrict__ e0,
                                                            uint8_t* __restrict__ valid_next, int32_t* __restrict__ path_of) {
    __shared__ uint32_t s_stack[kStackLds * kBlock];
    const int64_t n = B * spp;
    for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const int64_t b = i / spp;
        const float du = dudv[i] - 0.5f, dv = dudv[n + i] - 0.5f;
        const f3 d0 = ld3(rays_d + b * 3), dx = ld3(dxdu + b * 3), dy = ld3(dydv + b * 3);
        const f3 d = t_normalize(mk3((d0.x + dx.x * du) + dy.x * dv, (d0.y + dx.y * du) + dy.y * dv, (d0.z + dx.z * du) + dy.z * dv));   // pt_jitter_kernel
        const f3 o = ld3(rays_o + b * 3);
        st3(wi_out + i * 3, d); st3(wo_out + i * 3, mk3(-d.x, -d.y, -d.z));
        const Hit h = trace_bvh4<LAYOUT, false, kStackLds, false, JOINT>(sc, o, d, s_stack + threadIdx.x);
        int ord = -1;
        if (h.slot >= 0) {                                                         // intersect_kernel's outputs
            f3 p0, p1, p2;
            hit_vertices(sc, h, p0, p1, p2);
            st3(pos + i * 3, hit_position(h, p0, p1, p2));
            f3 nn = t_normalize(hit_normal(p0, p1, p2));
            if (t_dot(nn, mk3(-d.x, -d.y, -d.z)) < 0.f) nn = mk3(-nn.x, -nn.y, -nn.z);
            st3(nrm + i * 3, nn);
            ord = em.emit_ord[h.id];                                               // pt_primary_emit_kernel
        } else {
            st3(pos + i * 3, mk3(0.f, 0.f, 0.f)); st3(nrm + i * 3, mk3(0.f, 0.f, 0.f));
        }
        const bool cont = h.slot >= 0 && ord < 0;
        e0[i] = ord; valid_next[i] = cont ? 1 : 0; path_of[i] = cont ? (int32_t)i : -1;
    }
}
extern "C" IRIS_API int iris_pt_primary(const iris_scene* sc, const iris_emitter* e, const float* rays_o, const float* rays_d, const float* dxdu, const float* dydv,
                               const float* dudv, int64_t B, int spp, float* wi, float* wo, float* pos, float* nrm, int32_t* e0, uint8_t* valid_next, int32_t* path_of,
                               iris_stream_t stream) {
    if (!sc || !e || B < 0 || spp < 1 || (B > 0 && (!rays_o || !rays_d || !dxdu || !dydv || !dudv || !wi || !wo || !pos || !nrm || !e0 || !valid_next || !path_of)))
        return fail(IRIS_ERR_ARG, "iris_pt_primary: bad arguments");
    if (B == 0) return IRIS_OK;
    if (B * spp >= ((int64_t)1 << 31)) return fail(IRIS_ERR_ARG, "iris_pt_primary: more than 2^31 paths in one call (path_of is int32)");
    if (e->dev.nf != sc->info.n_triangles) return fail(IRIS_ERR_ARG, "iris_pt_primary: the emitter tables are for a mesh of another size than the scene's");
    const int64_t N = B * spp;
    const dim3 grid(grid_for(N, kBlock, num_cus() * 6));
    if (sc->dev.layout == kLayoutQ8 && joint_launch(N))
        hipLaunchKernelGGL((pt_primary_kernel<kLayoutQ8, true>), grid, dim3(kBlock), 0, (hipStream_t)stream, sc->dev, e->dev, rays_o, rays_d, dxdu, dydv, dudv, B, spp, wi, wo, pos, nrm, e0, valid_next, path_of);
    else if (sc->dev.layout == kLayoutQ8)
        hipLaunchKernelGGL((pt_primary_kernel<kLayoutQ8, false>), grid, dim3(kBlock), 0, (hipStream_t)stream, sc->dev, e->dev, rays_o, rays_d, dxdu, dydv, dudv, B, spp, wi, wo, pos, nrm, e0, valid_next, path_of);
    else
        hipLaunchKernelGGL((pt_primary_kernel<kLayoutF32, false>), grid, dim3(kBlock), 0, (hipStream_t)stream, sc->dev, e->dev, rays_o, rays_d, dxdu, dydv, dudv, B, spp, wi, wo, pos, nrm, e0, valid_next, path_of);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}
extern "C" IRIS_API int iris_pt_nee(const iris_scene* sc, const iris_emitter* e, const float* pos, const float* nrm, const float* wo, const float* albedo,
                           const float* roughness, const float* metallic, const float* s1, const float* s2, int64_t N, float* coef1, int32_t* e1,
                           float g_eps, float pdf_eps, float mis_eps, iris_stream_t stream) {
    if (!sc || !e || !e->can_sample) return fail(IRIS_ERR_ARG, "iris_pt_nee: scene / emitter (with vertices + cdf) required");
    if (N < 0 || (N > 0 && (!pos || !nrm || !wo || !albedo || !roughness || !metallic || !s1 || !s2 || !coef1 || !e1))) return fail(IRIS_ERR_ARG, "iris_pt_nee: bad arguments");
    if (N == 0) return IRIS_OK;
    PtArgs a{};
    a.sc = sc->dev; a.em = e->dev; a.es = e->sample; a.N = N;
    a.pos = pos; a.nrm = nrm; a.wo = wo; a.albedo = albedo; a.rough = roughness; a.metal = metallic; a.s1 = s1; a.s2 = s2;
    a.coef1 = coef1; a.e1 = e1; a.g_eps = g_eps; a.pdf_eps = pdf_eps; a.mis_eps = mis_eps;
    int tile_rays, grid;
    if (pt_tiling(N, tile_rays, grid)) {
        if (a.sc.layout == kLayoutQ8) hipLaunchKernelGGL((pt_tiled_kernel<kLayoutQ8, true>), dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, a, tile_rays);
        else hipLaunchKernelGGL((pt_tiled_kernel<kLayoutF32, true>), dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, a, tile_rays);
    } else if (a.sc.layout == kLayoutQ8 && joint_launch(N)) hipLaunchKernelGGL((pt_nee_kernel<kLayoutQ8, true>), dim3(grid_for(N, kBlock, num_cus() * 6)), dim3(kBlock), 0, (hipStream_t)stream, a);
    else if (a.sc.layout == kLayoutQ8) hipLaunchKernelGGL(pt_nee_kernel<kLayoutQ8>, dim3(grid_for(N, kBlock, num_cus() * 6)), dim3(kBlock), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(pt_nee_kernel<kLayoutF32>, dim3(grid_for(N, kBlock, num_cus() * 6)), dim3(kBlock), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}
extern "C" IRIS_API int iris_pt_brdf_trace(const iris_scene* sc, const float* pos, const float* nrm, const float* wo, const float* albedo,
                                  const float* roughness, const float* metallic, const float* s1, const float* s2, int64_t N, float* wi,
                                  float* pdf, float* weight, float* pos_next, float* nrm_next, int64_t* tri_next, uint8_t* valid,
                                  int lobe, float lobe_roughness, iris_stream_t stream) {
    if (!sc || N < 0 || lobe < 0 || lobe > 2 ||
        (N > 0 && (!pos || !nrm || !wo || !s2 || !wi || !pdf || !weight || !pos_next || !nrm_next || !tri_next || !valid ||
                   (lobe == 0 && (!albedo || !roughness || !metallic || !s1)))))
        return fail(IRIS_ERR_ARG, "iris_pt_brdf_trace: bad arguments");
    if (N == 0) return IRIS_OK;
    PtArgs a{};
    a.sc = sc->dev; a.N = N;
    a.pos = pos; a.nrm = nrm; a.wo = wo; a.albedo = albedo; a.rough = roughness; a.metal = metallic; a.s1 = s1; a.s2 = s2;
    a.wi_out = wi; a.brdf_pdf = pdf; a.brdf_w = weight; a.pos_next = pos_next; a.nrm_next = nrm_next; a.tri_next = tri_next; a.valid_next_hit = valid;
    a.lobe = lobe; a.lobe_rough = lobe_roughness;
    int tile_rays, grid;
    if (pt_tiling(N, tile_rays, grid)) {
        if (a.sc.layout == kLayoutQ8) hipLaunchKernelGGL((pt_tiled_kernel<kLayoutQ8, false>), dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, a, tile_rays);
        else hipLaunchKernelGGL((pt_tiled_kernel<kLayoutF32, false>), dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, a, tile_rays);
    } else if (a.sc.layout == kLayoutQ8 && joint_launch(N)) hipLaunchKernelGGL((pt_brdf_trace_kernel<kLayoutQ8, true>), dim3(grid_for(N, kBlock, num_cus() * 6)), dim3(kBlock), 0, (hipStream_t)stream, a);
    else if (a.sc.layout == kLayoutQ8) hipLaunchKernelGGL(pt_brdf_trace_kernel<kLayoutQ8>, dim3(grid_for(N, kBlock, num_cus() * 6)), dim3(kBlock), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(pt_brdf_trace_kernel<kLayoutF32>, dim3(grid_for(N, kBlock, num_cus() * 6)), dim3(kBlock), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}
// Both tracing stages of a bounce in one launch (see pt_bounce_kernel); calls below the tiling threshold run the two stages one after the other.
extern "C" IRIS_API int iris_pt_bounce(const iris_scene* sc, const iris_emitter* e, const float* pos, const float* nrm, const float* wo, const float* albedo,
                              const float* roughness, const float* metallic, const float* s1, const float* s2, const float* s1b, const float* s2b, int64_t N,
                              float* coef1, int32_t* e1, float g_eps, float pdf_eps, float mis_eps, float* wi, float* pdf, float* weight, float* pos_next,
                              float* nrm_next, int64_t* tri_next, uint8_t* valid, iris_stream_t stream) {
    if (!sc || !e || !e->can_sample) return fail(IRIS_ERR_ARG, "iris_pt_bounce: scene / emitter (with vertices + cdf) required");
    if (N < 0 || (N > 0 && (!pos || !nrm || !wo || !albedo || !roughness || !metallic || !s1 || !s2 || !s1b || !s2b || !coef1 || !e1 || !wi || !pdf || !weight || !pos_next ||
                            !nrm_next || !tri_next || !valid)))
        return fail(IRIS_ERR_ARG, "iris_pt_bounce: bad arguments");
    if (N == 0) return IRIS_OK;
    int tile_rays, grid, t1 = 0, g1 = 0;
    // Together when that makes the tiles LARGER (a reference-size batch of refine_shading, 1.3 M paths, has 512-ray tiles per stage: 111.7 -> 120.4 Mpaths/s); a call whose
    // stages already run full 4096-ray tiles each gains nothing from mixing the two ray kinds in a tile (measured -3.4 % at 21 M paths) and keeps the two launches.
    const bool full_tiles_already = pt_tiling(N, t1, g1) && t1 >= kPtTileCap;
    if (full_tiles_already || !pt_tiling(2 * N, tile_rays, grid)) {
        int rc = iris_pt_nee(sc, e, pos, nrm, wo, albedo, roughness, metallic, s1, s2, N, coef1, e1, g_eps, pdf_eps, mis_eps, stream);
        if (rc != IRIS_OK) return rc;
        return iris_pt_brdf_trace(sc, pos, nrm, wo, albedo, roughness, metallic, s1b, s2b, N, wi, pdf, weight, pos_next, nrm_next, tri_next, valid, 0, 0.0f, stream);
    }
    PtArgs a{};
    a.sc = sc->dev; a.em = e->dev; a.es = e->sample; a.N = N;
    a.pos = pos; a.nrm = nrm; a.wo = wo; a.albedo = albedo; a.rough = roughness; a.metal = metallic; a.s1 = s1; a.s2 = s2; a.s1b = s1b; a.s2b = s2b;
    a.coef1 = coef1; a.e1 = e1; a.g_eps = g_eps; a.pdf_eps = pdf_eps; a.mis_eps = mis_eps;
    a.wi_out = wi; a.brdf_pdf = pdf; a.brdf_w = weight; a.pos_next = pos_next; a.nrm_next = nrm_next; a.tri_next = tri_next; a.valid_next_hit = valid;
    a.lobe = 0; a.lobe_rough = 0.f;
    const int tile_paths = std::max(kBlock / 2, tile_rays / 2);
    const int64_t n_tiles = (N + tile_paths - 1) / tile_paths;
    const int g2 = (int)std::min<int64_t>((int64_t)num_cus() * IRIS_PT_WAVES, n_tiles);
    if (a.sc.layout == kLayoutQ8) hipLaunchKernelGGL((pt_bounce_kernel<kLayoutQ8>), dim3(g2), dim3(kBlock), 0, (hipStream_t)stream, a, tile_paths);
    else hipLaunchKernelGGL((pt_bounce_kernel<kLayoutF32>), dim3(g2), dim3(kBlock), 0, (hipStream_t)stream, a, tile_paths);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}
extern "C" IRIS_API int iris_pt_brdf_finish(const iris_emitter* e, const iris_slf* slf, const float* pos, const float* pos_next, const float* nrm_next,
                                   const float* wi, const int64_t* tri_next, const float* roughness_next, const float* pdf, const float* weight,
                                   int64_t N, float* coef2, float* const2, int32_t* e2, uint8_t* valid_next, float trace_roughness, float g_eps,
                                   iris_stream_t stream) {
    if (!e || !slf || N < 0 || (N > 0 && (!pos || !pos_next || !nrm_next || !wi || !tri_next || !pdf || !weight || !coef2 || !const2 || !e2)))      // (roughness_next may be NULL: see iris_hip.h)
        return fail(IRIS_ERR_ARG, "iris_pt_brdf_finish: bad arguments");
    if (N == 0) return IRIS_OK;
    PtArgs a{};
    a.em = e->dev; a.slf = slf->dev; a.N = N;
    a.pos = pos; a.pos_n_in = pos_next; a.nrm_n_in = nrm_next; a.wi_in = wi; a.tri_n_in = tri_next; a.rough_next = roughness_next; a.pdf_in = pdf; a.w_in = weight;
    a.coef2 = coef2; a.const2 = const2; a.e2 = e2; a.valid_next_hit = valid_next; a.trace_rough = trace_roughness; a.g_eps = g_eps;
    LAUNCH1D(pt_brdf_finish_kernel, N, stream, a);
    return IRIS_OK;
}
extern "C" IRIS_API int iris_pt_accumulate_fwd(const float* radiance, const int32_t* e0, const int32_t* path_of, const int32_t* e1, const float* coef1,
                                      const int32_t* e2, const float* coef2, const float* const2, int64_t B, int spp, float* L,
                                      iris_stream_t stream) {
    if (B < 0 || spp < 1 || (B > 0 && (!radiance || !e0 || !path_of || !L))) return fail(IRIS_ERR_ARG, "iris_pt_accumulate_fwd: bad arguments");
    if (B == 0) return IRIS_OK;
    int lpp = 1;
    while (lpp < spp && lpp < 64) lpp <<= 1;                     // lanes per pixel: min(64, next power of two >= spp)
    const int64_t n_groups = (B + 64 / lpp - 1) / (64 / lpp);    // one wave per group of 64 / lpp pixels
    hipLaunchKernelGGL(pt_accumulate_fwd_kernel, dim3(grid_for(n_groups * 64, 256, 8192)), dim3(256), 0, (hipStream_t)stream, radiance, e0, path_of, e1, coef1, e2, coef2, const2, B, spp, lpp, L);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}
extern "C" IRIS_API int iris_pt_accumulate_bwd(const float* gL, const int32_t* e0, const int32_t* path_of, const int32_t* e1, const float* coef1,
                                      const int32_t* e2, const float* coef2, int64_t B, int spp, float* g_radiance, iris_stream_t stream) {
    if (B < 0 || spp < 1 || (B > 0 && (!gL || !e0 || !path_of || !g_radiance))) return fail(IRIS_ERR_ARG, "iris_pt_accumulate_bwd: bad arguments");
    if (B == 0) return IRIS_OK;
    LAUNCH1D(pt_accumulate_bwd_kernel, B * spp, stream, gL, e0, path_of, e1, coef1, e2, coef2, B, spp, g_radiance);
    return IRIS_OK;
}

extern "C" IRIS_API int iris_pt_apply(float* L, const int32_t* rows, float* throughput, const float* radiance, const int32_t* e, const float* coef,
                             const float* cst, const float* weight, int64_t N, int nan_to_zero, iris_stream_t stream) {
    if (N < 0 || (N > 0 && (!L || (e && (!radiance || !coef))))) return fail(IRIS_ERR_ARG, "iris_pt_apply: bad arguments");
    if (N == 0) return IRIS_OK;
    LAUNCH1D(pt_apply_kernel, N, stream, L, rows, throughput, radiance, e, coef, cst, weight, N, nan_to_zero);
    return IRIS_OK;
}

extern "C" IRIS_API uint64_t iris_pt_compact_workspace_bytes(int64_t N) {
    return N <= 0 ? 0 : (uint64_t)((N + kCompactItems - 1) / kCompactItems) * sizeof(int32_t);
}
extern "C" IRIS_API int iris_pt_compact(const uint8_t* keep, int64_t N, int n3, const float* const* src3, float* const* dst3, uint32_t negate3, int n1,
                               const float* const* src1, float* const* dst1, int ni, const int32_t* const* srci, int32_t* const* dsti, int32_t* count,
                               void* workspace, uint64_t workspace_bytes, iris_stream_t stream) {
    if (N < 0 || !count || n3 < 0 || n1 < 0 || ni < 0 || n3 > kCompactMax || n1 > kCompactMax || ni > kCompactMax || (N > 0 && (!keep || !workspace)) ||
        (n3 > 0 && (!src3 || !dst3)) || (n1 > 0 && (!src1 || !dst1)) || (ni > 0 && (!srci || !dsti)))
        return fail(IRIS_ERR_ARG, "iris_pt_compact: bad arguments");
    if (N >= ((int64_t)1 << 31)) return fail(IRIS_ERR_ARG, "iris_pt_compact: more than 2^31 rows");
    if (workspace_bytes < iris_pt_compact_workspace_bytes(N)) return fail(IRIS_ERR_ARG, "iris_pt_compact: workspace smaller than iris_pt_compact_workspace_bytes(N)");
    if (N == 0) { HIP_TRY(hipMemsetAsync(count, 0, sizeof(int32_t), (hipStream_t)stream)); return IRIS_OK; }
    CompactArgs a{};
    a.keep = keep; a.N = N; a.n3 = n3; a.n1 = n1; a.ni = ni; a.negate3 = negate3; a.block_counts = (int32_t*)workspace; a.count = count;
    for (int j = 0; j < n3; ++j) { if (!src3[j] || !dst3[j]) return fail(IRIS_ERR_ARG, "iris_pt_compact: null array"); a.src3[j] = src3[j]; a.dst3[j] = dst3[j]; }
    for (int j = 0; j < n1; ++j) { if (!src1[j] || !dst1[j]) return fail(IRIS_ERR_ARG, "iris_pt_compact: null array"); a.src1[j] = src1[j]; a.dst1[j] = dst1[j]; }
    for (int j = 0; j < ni; ++j) { if (!srci[j] || !dsti[j]) return fail(IRIS_ERR_ARG, "iris_pt_compact: null array"); a.srci[j] = srci[j]; a.dsti[j] = dsti[j]; }
    const int blocks = (int)((N + kCompactItems - 1) / kCompactItems);
    hipLaunchKernelGGL(pt_compact_count_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    hipLaunchKernelGGL(pt_compact_move_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}

// ======================================================================================================
// 8(f)-2: G-buffer pooling builders (the stages that produce vslf.npz / emitter.pth)
// ======================================================================================================
// VoxelSLF.scatter_add (model/slf.py:56-61): radiance[idx] += rgb, count[idx] += 1 with idx = spatial_idx(x).
// Samples that fall into an empty voxel (idx = -1) are dropped (torch would raise on the negative index).
__global__ void slf_scatter_add_kernel(SlfDev s, const float* __restrict__ x, const float* __restrict__ rgb, int64_t B, float* __restrict__ acc,
                                       unsigned long long* __restrict__ count) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < B; i += (int64_t)gridDim.x * blockDim.x) {
        const int j = slf_index(s, ld3(x + i * 3));
        if (j < 0) continue;
        atomicAdd(acc + (int64_t)j * 3, rgb[i * 3]); atomicAdd(acc + (int64_t)j * 3 + 1, rgb[i * 3 + 1]); atomicAdd(acc + (int64_t)j * 3 + 2, rgb[i * 3 + 2]);
        atomicAdd(count + j, 1ull);
    }
}
extern "C" IRIS_API int iris_slf_scatter_add(const iris_slf* s, const float* x, const float* rgb, int64_t B, float* radiance_acc, int64_t* count,
                                    iris_stream_t stream) {
    if (!s || B < 0 || (B > 0 && (!x || !rgb || !radiance_acc || !count))) return fail(IRIS_ERR_ARG, "iris_slf_scatter_add: bad arguments");
    if (B == 0) return IRIS_OK;
    LAUNCH1D(slf_scatter_add_kernel, B, stream, s->dev, x, rgb, B, radiance_acc, (unsigned long long*)count);
    return IRIS_OK;
}
// slf_bake.py:104-110: occupancy histogram of the voxel grid, hist[x + y*H + z*H*H] += 1 (float counts, exact below 2^24)
__global__ void voxel_histogram_kernel(const float* __restrict__ x, int64_t B, float vmin, float den, int H, float* __restrict__ hist) {
    SlfDev s{}; s.H = H; s.vmin = vmin; s.den = den;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < B; i += (int64_t)gridDim.x * blockDim.x) {
        const int cx = voxel_coord(x[i * 3], s), cy = voxel_coord(x[i * 3 + 1], s), cz = voxel_coord(x[i * 3 + 2], s);
        atomicAdd(hist + ((int64_t)cz * H + cy) * H + cx, 1.0f);
    }
}
extern "C" IRIS_API int iris_voxel_histogram(const float* x, int64_t B, double voxel_min, double voxel_max, int H, float* hist, iris_stream_t stream) {
    if (B < 0 || H <= 0 || (B > 0 && (!x || !hist))) return fail(IRIS_ERR_ARG, "iris_voxel_histogram: bad arguments");
    if (B == 0) return IRIS_OK;
    LAUNCH1D(voxel_histogram_kernel, B, stream, x, B, (float)voxel_min, (float)(voxel_max - voxel_min), H, hist);
    return IRIS_OK;
}
// extract_emitter_ldr.py:90-95 (torch_scatter.scatter(..., reduce='sum')): out[idx[i]] += values[i], count[idx[i]] += 1; idx < 0 skipped
__global__ void scatter_add_rows_kernel(const float* __restrict__ values, const int64_t* __restrict__ idx, int64_t B, int64_t F, float* __restrict__ out,
                                        float* __restrict__ count) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < B; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t j = idx[i];
        if (j < 0 || j >= F) continue;
        atomicAdd(out + j * 3, values[i * 3]); atomicAdd(out + j * 3 + 1, values[i * 3 + 1]); atomicAdd(out + j * 3 + 2, values[i * 3 + 2]);
        if (count) atomicAdd(count + j, 1.0f);
    }
}
extern "C" IRIS_API int iris_scatter_add_rows(const float* values, const int64_t* idx, int64_t B, int64_t F, float* out, float* count, iris_stream_t stream) {
    if (B < 0 || F < 0 || (B > 0 && (!values || !idx || !out))) return fail(IRIS_ERR_ARG, "iris_scatter_add_rows: bad arguments");
    if (B == 0) return IRIS_OK;
    LAUNCH1D(scatter_add_rows_kernel, B, stream, values, idx, B, F, out, count);
    return IRIS_OK;
}
