// Host-side quality check of the wide-BVH builder (iris_amd/csrc/bvh_build.cpp): builds the tree for a mesh dumped by
// tools/bvh_eval/dump_room.py and traces a sample of bake-like secondary rays (origins on the surface, cosine-distributed directions
// into the room) with the traversal rule of the kernels (iris_trace.h: ordered by entry distance, children culled by the best hit),
// counting node visits and triangle tests per ray.  Applies the 8-bit plane quantisation of the Q8 node layout (iris_hip.hip) so that
// the counts are those of the device tree.  Numbers agree with the instrumented GPU launches (bench.py roofline.nodes_per_ray).
//   g++ -O2 -std=c++17 -I iris_amd/csrc tools/bvh_eval/bvh_eval.cpp iris_amd/csrc/bvh_build.cpp -lpthread -o tools/bvh_eval/bvh_eval
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "bvh_build.h"

using namespace iris;

struct V3 { float x, y, z; };
static V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
static float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

struct QNode { float lo[kMaxWidth][3], hi[kMaxWidth][3]; };   // decoded (quantised, conservative) child boxes

int main(int argc, char** argv) {
    const char* path = argc > 1 ? argv[1] : "/tmp/room.bin";
    const int n_rays = argc > 2 ? atoi(argv[2]) : 200000;
    const int max_leaf = argc > 3 ? atoi(argv[3]) : 4;
    const bool quant = argc > 4 ? atoi(argv[4]) != 0 : true;
    const float tri_cost = argc > 5 ? (float)atof(argv[5]) : 0.7f;
    FILE* f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); return 1; }
    int64_t nv, nf;
    if (fread(&nv, 8, 1, f) != 1 || fread(&nf, 8, 1, f) != 1) return 1;
    std::vector<float> verts(nv * 3);
    std::vector<int32_t> faces(nf * 3);
    if (fread(verts.data(), 4, nv * 3, f) != (size_t)nv * 3 || fread(faces.data(), 4, nf * 3, f) != (size_t)nf * 3) return 1;
    fclose(f);
    auto t0 = std::chrono::steady_clock::now();
    const char* ps = getenv("BVH_EVAL_PRESPLIT");   // presplit factor (default 8, 0 = off)
    const char* wenv = getenv("BVH_EVAL_WIDTH");   // children per node: 4 (the kernels' layout) or 8 (round 5: per-lane BVH8 study)
    const int width = wenv ? atoi(wenv) : 4;
    const bool oct_order = getenv("BVH_EVAL_ORDER") && !strcmp(getenv("BVH_EVAL_ORDER"), "octant");   // children in the ray octant's split-axis order (the kernels since round 3) instead of by entry distance
    WideBvh bvh = build_wide_bvh(verts.data(), nv, faces.data(), nf, width, max_leaf, 2e-5f, tri_cost, ps ? (float)atof(ps) : 8.f);
    const double build_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const size_t nn = bvh.nodes.size();
    // Q8 decode: per node, origin = min of child lows, per axis the smallest power of two 2^e with 255 * 2^e >= extent
    std::vector<QNode> q(nn);
    const bool iso = getenv("BVH_EVAL_ISO") != nullptr;   // ONE plane scale per node (the largest axis extent) instead of one per axis: the 80-B node study of round 5
    for (size_t i = 0; i < nn; ++i) {
        const WideNode& w = bvh.nodes[i];
        double ext_max = 0;
        for (int k = 0; k < 3; ++k) {
            float org = INFINITY, hi3 = -INFINITY;
            for (int s = 0; s < w.n; ++s) { org = std::min(org, w.lo[s][k]); hi3 = std::max(hi3, w.hi[s][k]); }
            ext_max = std::max(ext_max, (double)hi3 - (double)org);
        }
        for (int k = 0; k < 3; ++k) {
            float org = INFINITY, hi3 = -INFINITY;
            for (int s = 0; s < w.n; ++s) { org = std::min(org, w.lo[s][k]); hi3 = std::max(hi3, w.hi[s][k]); }
            const double ext = iso ? ext_max : (double)hi3 - (double)org;
            int e = -126;
            if (ext > 0) e = std::max(-126, (int)std::ceil(std::log2(ext / 255.0)));
            while (std::ldexp(255.0, e) < ext) ++e;
            const double sc = std::ldexp(1.0, e);
            for (int s = 0; s < width; ++s) {
                if (s >= w.n) { q[i].lo[s][k] = INFINITY; q[i].hi[s][k] = -INFINITY; continue; }
                int lo = (int)std::floor(((double)w.lo[s][k] - org) / sc), hi = (int)std::ceil(((double)w.hi[s][k] - org) / sc);
                lo = std::min(255, std::max(0, lo)); hi = std::min(255, std::max(0, hi));
                q[i].lo[s][k] = quant ? (float)(org + lo * sc) : w.lo[s][k]; q[i].hi[s][k] = quant ? (float)(org + hi * sc) : w.hi[s][k];
            }
        }
    }
    // rays: BVH_EVAL_RAYS=<file from dump_rays.py>: the bake kernel's own rays, tile by tile in the order of its LDS counting sort; otherwise
    // area-weighted surface points, cosine hemisphere around the normal that faces the room centre
    std::mt19937_64 rng(1234);
    std::uniform_real_distribution<float> U(0.f, 1.f);
    std::vector<double> cdf(nf);
    auto vert = [&](int64_t fi, int k) { const float* p = verts.data() + (int64_t)faces[fi * 3 + k] * 3; return V3{p[0], p[1], p[2]}; };
    double acc = 0;
    V3 centre{0, 0, 0};
    for (int64_t i = 0; i < nv; ++i) { centre.x += verts[i * 3] / nv; centre.y += verts[i * 3 + 1] / nv; centre.z += verts[i * 3 + 2] / nv; }
    for (int64_t i = 0; i < nf; ++i) { V3 c = cross(sub(vert(i, 1), vert(i, 0)), sub(vert(i, 2), vert(i, 0))); acc += 0.5 * std::sqrt(dot(c, c)); cdf[i] = acc; }
    long long tot_nodes = 0, tot_tris = 0, tot_slabs = 0, hits = 0, max_sp = 0;
    std::vector<uint32_t> stack(256);
    std::vector<float> dstack(256);
    const char* cm = getenv("BVH_EVAL_CULL");   // unset: as the kernels (no distances on the stack); "exact": skip popped entries whose entry distance exceeds the best hit; "<m>": the same through a code of exponent + m mantissa bits
    const int cull_bits = !cm ? -2 : (!strcmp(cm, "exact") ? -1 : atoi(cm));
    long long skipped_nodes = 0, skipped_leaves = 0;
    const bool seed_pass = getenv("BVH_EVAL_SEED") != nullptr;
    auto code = [&](float t) -> uint32_t { uint32_t b; memcpy(&b, &t, 4); return b >> (23 - cull_bits); };
    auto pop = [&](int& sp, float best) -> uint32_t {
        while (sp) {
            --sp;
            const bool skip = cull_bits == -2 ? false : cull_bits == -1 ? dstack[sp] > best : code(dstack[sp]) > code(best);
            if (!skip) return stack[sp];
            if (stack[sp] & 0x80000000u) ++skipped_leaves; else ++skipped_nodes;
        }
        return 0xffffffffu;
    };
    // Moeller-Trumbore on triangle ti (statistics only: the kernels' watertight test decides the same hits up to edge cases)
    auto tri_hit = [&](int64_t ti, V3 o, V3 d, float& tt) -> bool {
        V3 q0 = vert(ti, 0), f1 = sub(vert(ti, 1), q0), f2 = sub(vert(ti, 2), q0);
        V3 pv = cross(d, f2); float det = dot(f1, pv); float inv = 1.f / det;
        V3 tv = sub(o, q0); float uu = dot(tv, pv) * inv; V3 qv = cross(tv, f1); float vv = dot(d, qv) * inv; tt = dot(f2, qv) * inv;
        return uu >= 0 && vv >= 0 && uu + vv <= 1 && tt >= 0;
    };
    // one traversal with the rule of the kernels; `best` enters as the initial bound; returns the hit triangle (-1: none found below the bound)
    auto trace = [&](V3 o, V3 d, float& best) -> int64_t {
        const float id[3] = {1.f / d.x, 1.f / d.y, 1.f / d.z}, oo[3] = {o.x, o.y, o.z};
        const int oct = (d.x < 0 ? 1 : 0) | (d.y < 0 ? 2 : 0) | (d.z < 0 ? 4 : 0);
        int64_t hit_tri = -1;
        int sp = 0;
        uint32_t cur = 0;   // node index, or 0x80000000 | start << 3 | count
        for (;;) {
            if (cur & 0x80000000u) {
                const int start = (cur & 0x7fffffffu) >> 3, cnt = cur & 7;
                for (int k = 0; k < cnt; ++k) {
                    ++tot_tris;
                    const int64_t ti = bvh.tri_order[start + k];
                    float tt;
                    if (tri_hit(ti, o, d, tt) && tt < best) { best = tt; hit_tri = ti; }
                }
                cur = pop(sp, best);
                if (cur == 0xffffffffu) break;
                continue;
            }
            ++tot_nodes;
            tot_slabs += width;
            const WideNode& w = bvh.nodes[cur];
            float key[kMaxWidth]; uint32_t ref[kMaxWidth]; int m = 0;
            for (int j = 0; j < w.n; ++j) {
                const int s = oct_order ? (int)w.order[oct][j] : j;
                float tn = 0.f, tf = best;
                for (int k = 0; k < 3; ++k) {
                    float t0 = (q[cur].lo[s][k] - oo[k]) * id[k], t1 = (q[cur].hi[s][k] - oo[k]) * id[k];
                    if (t0 > t1) std::swap(t0, t1);
                    tn = std::max(tn, t0); tf = std::min(tf, t1);
                }
                if (tn <= tf) { key[m] = tn; ref[m] = w.child[s] >= 0 ? (uint32_t)w.child[s] : (0x80000000u | (uint32_t)w.leaf_start[s] << 3 | (uint32_t)w.leaf_count[s]); ++m; }
            }
            if (!oct_order) for (int i = 1; i < m; ++i) for (int j = i; j > 0 && key[j] < key[j - 1]; --j) { std::swap(key[j], key[j - 1]); std::swap(ref[j], ref[j - 1]); }
            if (m == 0) { cur = pop(sp, best); if (cur == 0xffffffffu) break; continue; }
            for (int i = m - 1; i >= 1; --i) { dstack[sp] = key[i]; stack[sp++] = ref[i]; }
            max_sp = std::max<long long>(max_sp, sp);
            cur = ref[0];
        }
        return hit_tri;
    };
    long long n_traced = 0;
    const char* rays_path = getenv("BVH_EVAL_RAYS");
    if (rays_path) {
        // BVH_EVAL_PRED=<k>: ray i of a tile first tests the triangle ray i - k of the sorted list hit (k = 1: its list neighbour; k = 64: the ray the same lane
        // carried one refill round earlier -- the one whose result IS available when ray i starts) and, when that test hits, enters the tree with its distance
        // as the bound.  A valid upper bound: the closest hit is unchanged.  Reported per lobe: predictor hit rate, node visits / triangle tests with and without.
        const int pred = getenv("BVH_EVAL_PRED") ? atoi(getenv("BVH_EVAL_PRED")) : 0;
        FILE* rf = fopen(rays_path, "rb");
        if (!rf) { fprintf(stderr, "cannot open %s\n", rays_path); return 1; }
        int64_t n_groups, per;
        if (fread(&n_groups, 8, 1, rf) != 1 || fread(&per, 8, 1, rf) != 1) return 1;
        std::vector<float> buf((size_t)per * 6);
        std::vector<int64_t> hit_of((size_t)per);
        struct LobeStat { long long rays = 0, nodes = 0, tris = 0, nodes_base = 0, tris_base = 0, pred_tests = 0, pred_hits = 0, pred_exact = 0; } ls[8];
        for (int64_t g = 0; g < n_groups; ++g) {
            int32_t lobe;
            if (fread(&lobe, 4, 1, rf) != 1 || fread(buf.data(), 4, (size_t)per * 6, rf) != (size_t)per * 6) return 1;
            LobeStat& st = ls[lobe & 7];
            for (int64_t i = 0; i < per; ++i) {
                const float* r = buf.data() + i * 6;
                V3 o{r[0], r[1], r[2]}, d{r[3], r[4], r[5]};
                // baseline traversal (also gives the true hit, which the predictor of later rays uses)
                long long n0 = tot_nodes, t0c = tot_tris, s0 = tot_slabs;
                float best = INFINITY;
                hit_of[(size_t)i] = trace(o, d, best);
                st.nodes_base += tot_nodes - n0; st.tris_base += tot_tris - t0c;
                hits += best < INFINITY;
                if (pred > 0) {
                    tot_nodes = n0; tot_tris = t0c; tot_slabs = s0;
                    float b2 = INFINITY;
                    const int64_t pt = i >= pred ? hit_of[(size_t)(i - pred)] : -1;
                    if (pt >= 0) {
                        ++st.pred_tests; ++tot_tris;
                        float tt;
                        if (tri_hit(pt, o, d, tt)) { ++st.pred_hits; b2 = tt * 1.000001f; st.pred_exact += pt == hit_of[(size_t)i]; }    // (the bound must not cull the triangle itself)
                    }
                    (void)trace(o, d, b2);
                }
                st.nodes += tot_nodes - n0; st.tris += tot_tris - t0c; ++st.rays; ++n_traced;
            }
        }
        fclose(rf);
        printf("{\"rays_file\": \"%s\", \"predictor_offset\": %d, \"per_lobe\": [", rays_path, pred);
        for (int l = 0; l < 7; ++l) {
            const LobeStat& st = ls[l];
            if (!st.rays) continue;
            printf("%s{\"lobe\": %d, \"rays\": %lld, \"nodes_per_ray\": %.3f, \"tris_per_ray\": %.3f, \"nodes_per_ray_with_predictor\": %.3f, \"tris_per_ray_with_predictor\": %.3f, "
                   "\"predictor_hit_rate\": %.4f, \"predictor_is_the_closest_hit\": %.4f}", l ? ", " : "", l, st.rays, (double)st.nodes_base / st.rays, (double)st.tris_base / st.rays,
                   (double)st.nodes / st.rays, (double)st.tris / st.rays, st.pred_tests ? (double)st.pred_hits / st.rays : 0.0, st.pred_tests ? (double)st.pred_exact / st.rays : 0.0);
        }
        printf("], ");
    } else {
    for (int r = 0; r < n_rays; ++r) {
        const int64_t fi = std::lower_bound(cdf.begin(), cdf.end(), U(rng) * acc) - cdf.begin();
        float a = U(rng), b = U(rng);
        if (a + b > 1.f) { a = 1.f - a; b = 1.f - b; }
        V3 p0 = vert(fi, 0), e1 = sub(vert(fi, 1), p0), e2 = sub(vert(fi, 2), p0);
        V3 o{p0.x + a * e1.x + b * e2.x, p0.y + a * e1.y + b * e2.y, p0.z + a * e1.z + b * e2.z};
        V3 n = cross(e1, e2);
        float nl = std::sqrt(dot(n, n));
        n = {n.x / nl, n.y / nl, n.z / nl};
        if (dot(n, sub(centre, o)) < 0) n = {-n.x, -n.y, -n.z};
        V3 t = std::fabs(n.x) > 0.1f ? V3{0, 1, 0} : V3{1, 0, 0};
        V3 bx = cross(t, n); float bl = std::sqrt(dot(bx, bx)); bx = {bx.x / bl, bx.y / bl, bx.z / bl};
        V3 by = cross(n, bx);
        const float u0 = U(rng), u1 = U(rng), rr = std::sqrt(u0), ph = 6.2831853f * u1, cz = std::sqrt(std::max(0.f, 1.f - u0));
        V3 d{bx.x * rr * std::cos(ph) + by.x * rr * std::sin(ph) + n.x * cz, bx.y * rr * std::cos(ph) + by.y * rr * std::sin(ph) + n.y * cz,
             bx.z * rr * std::cos(ph) + by.z * rr * std::sin(ph) + n.z * cz};
        o = {o.x + 8.94e-5f * d.x, o.y + 8.94e-5f * d.y, o.z + 8.94e-5f * d.z};
        float best = INFINITY;
        // BVH_EVAL_SEED=1: upper bound for any hit predictor -- trace once uncounted, then count the traversal that STARTS with the true hit distance as its bound
        if (seed_pass) {
            const long long n0 = tot_nodes, t0c = tot_tris, s0 = tot_slabs;
            (void)trace(o, d, best);
            tot_nodes = n0; tot_tris = t0c; tot_slabs = s0;
            best = best < INFINITY ? best * 1.0001f : best;
        }
        (void)trace(o, d, best);
        hits += best < INFINITY;
        ++n_traced;
    }
    printf("{");
    }
    size_t leaves = 0, leaf_tris = 0, children = 0;
    for (const auto& w : bvh.nodes) for (int s = 0; s < w.n; ++s) { ++children; if (w.child[s] < 0) { ++leaves; leaf_tris += w.leaf_count[s]; } }
    printf("\"width\": %d, \"child_order\": \"%s\", \"triangles\": %lld, \"leaf_records\": %zu, \"nodes\": %zu, \"depth\": %d, \"sah\": %.3f, \"build_s\": %.2f, \"children_per_node\": %.3f, \"tris_per_leaf\": %.3f, "
           "\"rays\": %lld, \"hit_frac\": %.4f, \"nodes_per_ray\": %.3f, \"slab_tests_per_ray\": %.3f, \"tris_per_ray\": %.3f, \"max_stack\": %lld, \"cull\": \"%s\", \"skipped_nodes_per_ray\": %.3f, \"skipped_leaves_per_ray\": %.3f}\n",
           width, oct_order ? "octant" : "distance", (long long)nf, bvh.tri_order.size(), nn, bvh.depth, bvh.sah_cost, build_s, (double)children / nn, (double)leaf_tris / leaves, n_traced, (double)hits / n_traced,
           (double)tot_nodes / n_traced, (double)tot_slabs / n_traced, (double)tot_tris / n_traced, max_sp, cm ? cm : "none", (double)skipped_nodes / n_traced, (double)skipped_leaves / n_traced);
    return 0;
}
