// Host-only check of the BVH builder (iris_amd/csrc/bvh_build.cpp), meant to be compiled with
//   g++ -O1 -g -fsanitize=address,undefined bvh_build_check.cpp ../../iris_amd/csrc/bvh_build.cpp -lpthread
// (GPU AddressSanitizer is not available on the pool: sanitizers run on the CPU build only).  Builds trees over random
// triangle soups, degenerate inputs and a grid, and checks the structural invariants the traversal kernels rely on.
#include <algorithm>
#include <cmath>
#include <initializer_list>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../iris_amd/csrc/bvh_build.h"

using namespace iris;

static int fails = 0;
#define CHECK(c, ...) do { if (!(c)) { std::printf("FAIL %s:%d: ", __FILE__, __LINE__); std::printf(__VA_ARGS__); std::printf("\n"); ++fails; } } while (0)

static int64_t split_refs = 0;
static void check_tree(const char* name, const std::vector<float>& v, const std::vector<int32_t>& f, int width, int max_leaf, int expect_split = -1) {
    const int64_t nv = (int64_t)v.size() / 3, nf = (int64_t)f.size() / 3;
    WideBvh b = build_wide_bvh(v.data(), nv, f.data(), nf, width, max_leaf);
    CHECK(!b.nodes.empty(), "%s: no root", name);
    const int64_t nr = (int64_t)b.tri_order.size();      // leaf records: one per triangle, more where long triangles were split (presplit)
    CHECK(nr >= nf && nr <= nf + std::max<int64_t>(nf / 4, 1024), "%s: tri_order has %zu entries for %lld triangles", name, b.tri_order.size(), (long long)nf);
    CHECK(expect_split < 0 || (expect_split > 0) == (nr > nf), "%s: %lld extra references, expected %s", name, (long long)(nr - nf), expect_split ? "some" : "none");
    std::vector<int> seen((size_t)nf, 0);
    for (int32_t t : b.tri_order) { CHECK(t >= 0 && t < nf, "%s: triangle id %d out of range", name, t); if (t >= 0 && t < nf) seen[(size_t)t]++; }
    for (int64_t t = 0; t < nf; ++t) CHECK(seen[(size_t)t] >= 1, "%s: triangle %lld referenced %d times", name, (long long)t, seen[(size_t)t]);
    std::vector<int> node_ref(b.nodes.size(), 0);
    std::vector<char> leaf_cover((size_t)nr, 0);
    struct LeafBox { float lo[3], hi[3]; };
    std::vector<std::vector<LeafBox>> boxes_of((size_t)nf);     // split triangles: the leaf boxes of their references
    for (size_t i = 0; i < b.nodes.size(); ++i) {
        const WideNode& w = b.nodes[i];
        CHECK(w.n >= 0 && w.n <= width, "%s: node %zu has %d children", name, i, w.n);
        // per ray octant the slots in front-to-back order: a permutation of the children, octant 0 = the canonical (left-to-right) order, and two
        // octants that differ in every axis visit in exactly opposite order
        for (int o = 0; o < 8; ++o) {
            int mask = 0;
            for (int j = 0; j < w.n; ++j) { CHECK(w.order[o][j] < w.n, "%s: node %zu octant %d order entry %d", name, i, o, (int)w.order[o][j]); mask |= 1 << w.order[o][j]; }
            CHECK(mask == (1 << w.n) - 1, "%s: node %zu octant %d: order is not a permutation", name, i, o);
            for (int j = 0; j < w.n; ++j) {
                if (o == 0) CHECK(w.order[0][j] == j, "%s: node %zu: octant 0 is not the canonical order", name, i);
                CHECK(w.order[o][j] == w.order[7 - o][w.n - 1 - j], "%s: node %zu: octants %d and %d are not mirror images", name, i, o, 7 - o);
            }
        }
        int32_t prev_child = -1;
        for (int s = 0; s < w.n; ++s) {
            if (w.child[s] >= 0) {
                CHECK((size_t)w.child[s] < b.nodes.size() && (size_t)w.child[s] > i, "%s: node %zu child %d -> %d", name, i, s, w.child[s]);
                if ((size_t)w.child[s] < b.nodes.size()) node_ref[(size_t)w.child[s]]++;
                if (prev_child >= 0) CHECK(w.child[s] == prev_child + 1, "%s: internal children of node %zu are not consecutive", name, i);
                prev_child = w.child[s];
            } else {
                CHECK(w.leaf_count[s] >= 1 && w.leaf_count[s] <= 7, "%s: leaf of %d triangles", name, w.leaf_count[s]);   // 3-bit count in the leaf reference
                CHECK(w.leaf_start[s] >= 0 && (int64_t)w.leaf_start[s] + w.leaf_count[s] <= nr, "%s: leaf range", name);
                for (int32_t j = w.leaf_start[s]; j < w.leaf_start[s] + w.leaf_count[s] && j < nr; ++j) {
                    CHECK(!leaf_cover[(size_t)j], "%s: leaf slot %d covered twice", name, j);
                    leaf_cover[(size_t)j] = 1;
                    const int32_t t = b.tri_order[(size_t)j];
                    if (seen[(size_t)t] > 1) {
                        LeafBox lb; for (int a = 0; a < 3; ++a) { lb.lo[a] = w.lo[s][a]; lb.hi[a] = w.hi[s][a]; }
                        boxes_of[(size_t)t].push_back(lb);
                        continue;
                    }
                    for (int k = 0; k < 3; ++k) {
                        const float* p = v.data() + (int64_t)f[(size_t)t * 3 + k] * 3;
                        for (int a = 0; a < 3; ++a)
                            CHECK(p[a] >= w.lo[s][a] && p[a] <= w.hi[s][a], "%s: vertex outside its leaf box (node %zu slot %d axis %d)", name, i, s, a);
                    }
                }
            }
            for (int a = 0; a < 3; ++a) CHECK(w.lo[s][a] <= w.hi[s][a], "%s: inverted box in a used slot", name);
        }
    }
    for (size_t i = 1; i < b.nodes.size(); ++i) CHECK(node_ref[i] == 1, "%s: node %zu referenced %d times", name, i, node_ref[i]);
    for (int64_t j = 0; j < nr; ++j) CHECK(leaf_cover[(size_t)j], "%s: leaf slot %lld not covered", name, (long long)j);
    // a split triangle: every point of it lies in the (padded) leaf box of at least one of its references
    std::mt19937 prng(11);
    std::uniform_real_distribution<float> PU(0.f, 1.f);
    for (int64_t t = 0; t < nf; ++t) {
        if (boxes_of[(size_t)t].empty()) continue;
        const float* p0 = v.data() + (int64_t)f[(size_t)t * 3] * 3; const float* p1 = v.data() + (int64_t)f[(size_t)t * 3 + 1] * 3; const float* p2 = v.data() + (int64_t)f[(size_t)t * 3 + 2] * 3;
        for (int q = 0; q < 200; ++q) {
            float a = PU(prng), c = PU(prng);
            if (q < 3) { a = q == 1; c = q == 2; } else if (q < 40) { (q % 3 == 0 ? a : c) = 0.f; if (q % 3 == 2) c = 1.f - a; }   // vertices and edges first
            if (a + c > 1.f) { a = 1.f - a; c = 1.f - c; }
            double pt[3];
            for (int k = 0; k < 3; ++k) pt[k] = (double)p0[k] + (double)a * ((double)p1[k] - p0[k]) + (double)c * ((double)p2[k] - p0[k]);
            bool in = false;
            for (const LeafBox& lb : boxes_of[(size_t)t]) {
                bool ok = true;
                for (int k = 0; k < 3; ++k) ok = ok && pt[k] >= lb.lo[k] && pt[k] <= lb.hi[k];
                in = in || ok;
            }
            CHECK(in, "%s: a point of split triangle %lld (%d references) is in none of their leaf boxes", name, (long long)t, (int)boxes_of[(size_t)t].size());
        }
    }
    split_refs = nr - nf;
    CHECK(3 * b.depth + 4 <= 96, "%s: depth %d too deep for the traversal stack", name, b.depth);
    std::printf("ok %-28s nf=%-8lld refs=%-8lld nodes=%-8zu depth=%-3d sah=%.2f\n", name, (long long)nf, (long long)nr, b.nodes.size(), b.depth, b.sah_cost);
}

int main() {
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> U(0.f, 1.f);
    auto soup = [&](int nf, float size, float extent) {
        std::vector<float> v; std::vector<int32_t> f;
        for (int t = 0; t < nf; ++t) {
            float c[3] = {U(rng) * extent, U(rng) * extent, U(rng) * extent};
            for (int k = 0; k < 3; ++k) { for (int a = 0; a < 3; ++a) v.push_back(c[a] + (U(rng) - 0.5f) * size); f.push_back(t * 3 + k); }
        }
        return std::make_pair(v, f);
    };
    { auto s = soup(0, 1.f, 1.f); check_tree("empty", s.first, s.second, 4, 4); }
    { auto s = soup(1, 1.f, 1.f); check_tree("one triangle", s.first, s.second, 4, 4); }
    { auto s = soup(5, 1.f, 1.f); check_tree("five triangles", s.first, s.second, 4, 4); }
    { auto s = soup(20000, 0.05f, 4.f); check_tree("soup 20k", s.first, s.second, 4, 4); }
    { auto s = soup(200000, 0.02f, 4.f); check_tree("soup 200k (threaded build)", s.first, s.second, 4, 4); }
    { auto s = soup(3000, 0.05f, 4.f); check_tree("soup width 8 leaf 7", s.first, s.second, 8, 7); }
    { auto s = soup(3000, 0.05f, 4.f); check_tree("soup leaf 1", s.first, s.second, 4, 1); }
    {   // all triangles identical (no split plane exists), degenerate (zero-area) triangles, huge coordinates
        std::vector<float> v = {0, 0, 0, 1, 0, 0, 0, 1, 0}; std::vector<int32_t> f;
        for (int t = 0; t < 1000; ++t) { f.push_back(0); f.push_back(1); f.push_back(2); }
        check_tree("1000 identical", v, f, 4, 4);
        std::vector<float> v2; std::vector<int32_t> f2;
        for (int t = 0; t < 500; ++t) { float x = U(rng) * 1e6f; for (int k = 0; k < 3; ++k) { v2.push_back(x); v2.push_back(x); v2.push_back(x); f2.push_back(t * 3 + k); } }
        check_tree("degenerate points, 1e6", v2, f2, 4, 4);
    }
    {   // regular grid (many equal centroids per axis)
        std::vector<float> v; std::vector<int32_t> f; const int n = 64;
        for (int y = 0; y <= n; ++y) for (int x = 0; x <= n; ++x) { v.push_back((float)x); v.push_back((float)y); v.push_back(0.f); }
        for (int y = 0; y < n; ++y) for (int x = 0; x < n; ++x) {
            int a = y * (n + 1) + x; f.push_back(a); f.push_back(a + 1); f.push_back(a + n + 2); f.push_back(a); f.push_back(a + n + 2); f.push_back(a + n + 1);
        }
        check_tree("grid 64x64", v, f, 4, 4);
    }
    {   // long triangles among fine detail (a decimated floor and a wall under a soup): split into clipped references
        auto s = soup(20000, 0.03f, 4.f);
        auto add = [&](std::initializer_list<float> pts) { int base = (int)s.first.size() / 3; for (float x : pts) s.first.push_back(x); for (int k = 0; k < 3; ++k) s.second.push_back(base + k); };
        add({0, 0, 0, 4, 0, 0, 4, 4, 0}); add({0, 0, 0, 4, 4, 0, 0, 4, 0});           // floor
        add({0, 0, 0, 0, 4, 4, 0, 0, 4}); add({0.5f, 0.2f, 3.9f, 3.7f, 3.1f, 0.1f, 3.9f, 3.3f, 0.2f});   // wall, a long diagonal sliver
        check_tree("soup 20k + 4 long triangles", s.first, s.second, 4, 4, 1);
        CHECK(split_refs >= 100, "long triangles: only %lld extra references", (long long)split_refs);
    }
    { auto s = soup(20000, 0.05f, 4.f); check_tree("soup 20k (uniform: no split)", s.first, s.second, 4, 4, 0); }
    if (fails) { std::printf("%d check(s) failed\n", fails); return 1; }
    std::printf("all BVH builder checks passed\n");
    return 0;
}
