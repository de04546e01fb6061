// Binned-SAH BVH2 build (down to single triangles) + SAH-optimal collapse to a W-wide tree by dynamic programming (host only).
// See bvh_build.h.
#include "bvh_build.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <limits>
#include <thread>

namespace iris {
namespace {

constexpr int kBins = 32;  // measured: 32 bins -> 3.6 % fewer node visits per ray than 16; 64 bins and an exact sweep for small nodes: no further gain
constexpr float kInf = std::numeric_limits<float>::infinity();

struct Box {
    float lo[3], hi[3];
    void reset() { for (int k = 0; k < 3; ++k) { lo[k] = kInf; hi[k] = -kInf; } }
    void grow(const float* a, const float* b) {
        for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], a[k]); hi[k] = std::max(hi[k], b[k]); }
    }
    void grow(const Box& o) { grow(o.lo, o.hi); }
    float area() const {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (!(dx >= 0.f)) return 0.f;
        return 2.f * (dx * dy + dy * dz + dz * dx);
    }
};

struct TriInfo { Box b; float c[3]; };

struct Node2 {
    Box b;
    int32_t left = -1, right = -1;  // internal
    int32_t start = 0, count = 0;   // leaf if count > 0
    int32_t first = 0, ntri = 0;    // the subtree's triangles: order[first, first + ntri)
    int8_t axis = 0;                // split axis (the left child holds the lower centroids)
};

struct Builder {
    const TriInfo* tri;
    int32_t* order;
    Node2* nodes;
    std::atomic<int32_t> next{0};
    int max_leaf;

    int32_t build(int32_t start, int32_t count, int depth) {
        int32_t me = next.fetch_add(1);
        Node2& nd = nodes[me];
        Box b, cb; b.reset(); cb.reset();
        for (int32_t i = start; i < start + count; ++i) {
            const TriInfo& t = tri[order[i]];
            b.grow(t.b); cb.grow(t.c, t.c);
        }
        nd.b = b; nd.left = nd.right = -1; nd.start = start; nd.count = count; nd.first = start; nd.ntri = count;
        if (count == 1) return me;

        int best_axis = -1, best_bin = -1;
        float best_cost = kInf;
        // one pass over the triangles fills the bins of all three axes (the gather tri[order[i]] is what a large node pays for)
        Box bins[3][kBins]; int cnt[3][kBins];
        float scale3[3]; bool use[3];
        for (int ax = 0; ax < 3; ++ax) {
            const float ext = cb.hi[ax] - cb.lo[ax];
            use[ax] = ext > 0.f;
            scale3[ax] = use[ax] ? (float)kBins / ext : 0.f;
            for (int k = 0; k < kBins; ++k) { bins[ax][k].reset(); cnt[ax][k] = 0; }
        }
        for (int32_t i = start; i < start + count; ++i) {
            const TriInfo& t = tri[order[i]];
            for (int ax = 0; ax < 3; ++ax) {
                if (!use[ax]) continue;
                const int k = std::min(kBins - 1, std::max(0, (int)((t.c[ax] - cb.lo[ax]) * scale3[ax])));
                bins[ax][k].grow(t.b); cnt[ax][k]++;
            }
        }
        for (int ax = 0; ax < 3; ++ax) {
            if (!use[ax]) continue;
            const Box* bb = bins[ax]; const int* bc = cnt[ax];
            float ra[kBins]; int rc[kBins];
            Box acc; acc.reset(); int n = 0;
            for (int k = kBins - 1; k > 0; --k) { acc.grow(bb[k]); n += bc[k]; ra[k] = acc.area(); rc[k] = n; }
            acc.reset(); n = 0;
            for (int k = 0; k < kBins - 1; ++k) {
                acc.grow(bb[k]); n += bc[k];
                if (n == 0 || rc[k + 1] == 0) continue;
                float cost = acc.area() * (float)n + ra[k + 1] * (float)rc[k + 1];
                if (cost < best_cost) { best_cost = cost; best_axis = ax; best_bin = k; }
            }
        }
        // SAH termination for small nodes: leaf cost = count * A ; split cost = A (one traversal step) + children.
        // (max_leaf <= 0: split down to single triangles; the wide collapse decides where the leaves are)
        if (max_leaf > 0 && count <= max_leaf) {
            float a = b.area();
            if (best_axis < 0 || (float)count * a <= 1.0f * a + best_cost) return me;
        }
        int32_t mid;
        if (best_axis < 0) {
            mid = start + count / 2;
        } else {
            const float ext = cb.hi[best_axis] - cb.lo[best_axis], scale = (float)kBins / ext;
            int32_t i = start, j = start + count - 1;
            while (i <= j) {
                const TriInfo& t = tri[order[i]];
                int k = std::min(kBins - 1, std::max(0, (int)((t.c[best_axis] - cb.lo[best_axis]) * scale)));
                if (k <= best_bin) ++i; else { std::swap(order[i], order[j]); --j; }
            }
            mid = i;
            if (mid == start || mid == start + count) mid = start + count / 2;
        }
        int32_t l, r;
        if (count > (1 << 16) && depth < 4) {
            int32_t lres = -1;
            std::thread th([&] { lres = build(start, mid - start, depth + 1); });
            r = build(mid, start + count - mid, depth + 1);
            th.join();
            l = lres;
        } else {
            l = build(start, mid - start, depth + 1);
            r = build(mid, start + count - mid, depth + 1);
        }
        Node2& nd2 = nodes[me];
        nd2.left = l; nd2.right = r; nd2.count = 0; nd2.axis = (int8_t)std::max(best_axis, 0);
        return me;
    }
};

// Bounds of (triangle  intersected with  box), Sutherland-Hodgman in double, rounded outward to float.  false: empty.
bool clipped_bounds(const float* v0, const float* v1, const float* v2, const Box& box, Box& out) {
    double poly[16][3], tmp[16][3];
    int n = 3;
    for (int k = 0; k < 3; ++k) { poly[0][k] = v0[k]; poly[1][k] = v1[k]; poly[2][k] = v2[k]; }
    for (int ax = 0; ax < 3 && n > 0; ++ax) {
        for (int side = 0; side < 2 && n > 0; ++side) {
            const double plane = side == 0 ? (double)box.lo[ax] : (double)box.hi[ax], sgn = side == 0 ? 1.0 : -1.0;
            int m = 0;
            for (int i = 0; i < n; ++i) {
                const double* a = poly[i]; const double* b = poly[(i + 1) % n];
                const double da = sgn * (a[ax] - plane), db = sgn * (b[ax] - plane);
                if (da >= 0.0) { for (int k = 0; k < 3; ++k) tmp[m][k] = a[k]; ++m; }
                if ((da > 0.0 && db < 0.0) || (da < 0.0 && db > 0.0)) {
                    const double t = da / (da - db);
                    for (int k = 0; k < 3; ++k) tmp[m][k] = a[k] + t * (b[k] - a[k]);
                    tmp[m][ax] = plane;
                    ++m;
                }
            }
            n = m;
            for (int i = 0; i < n; ++i) for (int k = 0; k < 3; ++k) poly[i][k] = tmp[i][k];
        }
    }
    if (n == 0) return false;
    out.reset();
    for (int i = 0; i < n; ++i) {
        float lo[3], hi[3];
        for (int k = 0; k < 3; ++k) {
            lo[k] = (float)poly[i][k]; if ((double)lo[k] > poly[i][k]) lo[k] = std::nextafter(lo[k], -kInf);
            hi[k] = (float)poly[i][k]; if ((double)hi[k] < poly[i][k]) hi[k] = std::nextafter(hi[k], kInf);
        }
        out.grow(lo, hi);
    }
    for (int k = 0; k < 3; ++k) { out.lo[k] = std::max(out.lo[k], box.lo[k]); out.hi[k] = std::min(out.hi[k], box.hi[k]); }   // (never outside the box it was cut from)
    for (int k = 0; k < 3; ++k) if (out.lo[k] > out.hi[k]) return false;
    return true;
}

float sah_of(const Node2* nodes, int32_t i, float root_area) {
    const Node2& n = nodes[i];
    float a = n.b.area() / root_area;
    if (n.count > 0) return a * (float)n.count;
    return a + sah_of(nodes, n.left, root_area) + sah_of(nodes, n.right, root_area);
}

}  // namespace

WideBvh build_wide_bvh(const float* verts, int64_t /*nv*/, const int32_t* faces, int64_t nf, int width, int leaf_tris,
                       float pad_rel, float tri_cost, float presplit) {
    leaf_tris = std::min(7, std::max(1, leaf_tris));
    WideBvh out;
    out.width = width;
    for (int k = 0; k < 3; ++k) { out.root_lo[k] = 0.f; out.root_hi[k] = 0.f; }
    auto empty_node = [&] {
        WideNode w; w.n = 0;
        for (int s = 0; s < kMaxWidth; ++s) {
            for (int k = 0; k < 3; ++k) { w.lo[s][k] = kInf; w.hi[s][k] = -kInf; }
            w.child[s] = -1; w.leaf_start[s] = 0; w.leaf_count[s] = 0;
        }
        for (int o = 0; o < 8; ++o) for (int s = 0; s < kMaxWidth; ++s) w.order[o][s] = (uint8_t)s;
        return w;
    };
    if (nf <= 0) { out.nodes.push_back(empty_node()); out.depth = 1; return out; }

    std::vector<TriInfo> tri((size_t)nf);
    std::vector<int32_t> order((size_t)nf);
    Box g; g.reset();
    for (int64_t f = 0; f < nf; ++f) {
        order[(size_t)f] = (int32_t)f;
        TriInfo& t = tri[(size_t)f];
        t.b.reset();
        for (int k = 0; k < 3; ++k) { const float* p = verts + (int64_t)faces[f * 3 + k] * 3; t.b.grow(p, p); }
        for (int k = 0; k < 3; ++k) t.c[k] = 0.5f * (t.b.lo[k] + t.b.hi[k]);
        g.grow(t.b);
    }
    float ext = 0.f;
    for (int k = 0; k < 3; ++k) ext = std::max({ext, g.hi[k] - g.lo[k], std::fabs(g.lo[k]), std::fabs(g.hi[k])});
    const float pad = pad_rel * ext + 1e-30f;
    out.pad = pad;

    // ---- early split clipping: a triangle much longer than the typical one (decimated walls and floors next to fine detail) is
    // referenced through several boxes, each the bounds of the triangle clipped to a piece of its box, so that it does not force one
    // large leaf box across everything beside it.  A reference whose longest side exceeds presplit x (median longest side) is cut at the
    // middle of that side, largest first, until none is left or the references have grown by a quarter.  The leaf records of a split
    // triangle are copies with the same index: the closest hit is a minimum over (t, index), so duplicates cannot change it.
    std::vector<int32_t> ref_tri;              // reference -> triangle (empty: identity)
    if (presplit > 0.f && nf >= 2) {
        std::vector<float> side((size_t)nf);
        auto longest = [](const Box& b, int& ax) { float m = -1.f; ax = 0; for (int k = 0; k < 3; ++k) if (b.hi[k] - b.lo[k] > m) { m = b.hi[k] - b.lo[k]; ax = k; } return m; };
        int ax;
        for (int64_t f = 0; f < nf; ++f) side[(size_t)f] = longest(tri[(size_t)f].b, ax);
        std::vector<float> sorted(side);
        std::nth_element(sorted.begin(), sorted.begin() + nf / 2, sorted.end());
        const float limit = presplit * sorted[(size_t)(nf / 2)];
        if (limit > 0.f) {
            struct Cand { float side; int32_t ref; bool operator<(const Cand& o) const { return side < o.side || (side == o.side && ref > o.ref); } };
            std::vector<Cand> heap;
            for (int64_t f = 0; f < nf; ++f) if (side[(size_t)f] > limit) heap.push_back({side[(size_t)f], (int32_t)f});
            if (!heap.empty()) {
                std::make_heap(heap.begin(), heap.end());
                ref_tri.resize((size_t)nf);
                for (int64_t f = 0; f < nf; ++f) ref_tri[(size_t)f] = (int32_t)f;
                const size_t max_refs = (size_t)nf + std::max<size_t>((size_t)nf / 4, 1024);
                while (!heap.empty() && tri.size() < max_refs) {
                    std::pop_heap(heap.begin(), heap.end());
                    const Cand c = heap.back(); heap.pop_back();
                    const Box b = tri[(size_t)c.ref].b;
                    longest(b, ax);
                    const float mid = 0.5f * (b.lo[ax] + b.hi[ax]);
                    if (!(mid > b.lo[ax] && mid < b.hi[ax])) continue;
                    Box lb = b, rb = b; lb.hi[ax] = mid; rb.lo[ax] = mid;
                    const int32_t f = ref_tri[(size_t)c.ref];
                    const float *v0 = verts + (int64_t)faces[(int64_t)f * 3] * 3, *v1 = verts + (int64_t)faces[(int64_t)f * 3 + 1] * 3, *v2 = verts + (int64_t)faces[(int64_t)f * 3 + 2] * 3;
                    Box lc, rc;
                    if (!clipped_bounds(v0, v1, v2, lb, lc) || !clipped_bounds(v0, v1, v2, rb, rc)) continue;   // (a sliver: leave the reference whole)
                    TriInfo l, r;
                    l.b = lc; r.b = rc;
                    for (int k = 0; k < 3; ++k) { l.c[k] = 0.5f * (lc.lo[k] + lc.hi[k]); r.c[k] = 0.5f * (rc.lo[k] + rc.hi[k]); }
                    tri[(size_t)c.ref] = l;
                    const int32_t nr = (int32_t)tri.size();
                    tri.push_back(r); ref_tri.push_back(f); order.push_back(nr);
                    float sl = longest(lc, ax), sr = longest(rc, ax);
                    if (sl > limit) { heap.push_back({sl, c.ref}); std::push_heap(heap.begin(), heap.end()); }
                    if (sr > limit) { heap.push_back({sr, nr}); std::push_heap(heap.begin(), heap.end()); }
                }
            }
        }
    }
    const int64_t nref = (int64_t)tri.size();

    std::vector<Node2> nodes((size_t)(2 * nref));
    Builder bld;
    bld.tri = tri.data(); bld.order = order.data(); bld.nodes = nodes.data(); bld.max_leaf = 0;   // down to single triangles
    bld.build(0, (int32_t)nref, 0);
    for (int k = 0; k < 3; ++k) { out.root_lo[k] = nodes[0].b.lo[k] - pad; out.root_hi[k] = nodes[0].b.hi[k] + pad; }

    // ---- collapse to `width`-wide nodes: SAH-optimal for the given binary topology (dynamic programming over "a subtree represented
    // by at most i children of one wide node", Ylitie et al. 2017, section 3.1), breadth first; the internal children of a node get
    // consecutive indices.  cost = sum over wide nodes A * 1 + sum over leaves A * triangles * tri_cost (areas relative to the root).
    //   C(n,1) = min(leaf(n), A_n + D(n,W));   D(n,j) = min_k C(l,k) + C(r,j-k);   C(n,i) = min(D(n,i), C(n,i-1))
    const int W = std::min(std::max(width, 2), kMaxWidth);
    const int32_t n2 = bld.next.load();
    struct Dp { float c[kMaxWidth]; uint8_t k[kMaxWidth + 1]; uint8_t leaf; };   // c[i-1] = C(n,i), i = 1..W-1; k[j] = left share of D(n,j), 0 = "use C(n,j-1)"
    std::vector<Dp> dp((size_t)n2);
    const float root_area = std::max(nodes[0].b.area(), 1e-30f);
    for (int32_t i = n2 - 1; i >= 0; --i) {          // children have larger indices than their parent
        const Node2& nd = nodes[(size_t)i];
        Dp& d = dp[(size_t)i];
        const float a = nd.b.area() / root_area;
        const float leaf = nd.ntri <= leaf_tris ? a * (float)nd.ntri * tri_cost : kInf;
        for (int j = 0; j <= W; ++j) d.k[j] = 0;
        if (nd.count > 0) { d.leaf = 1; for (int j = 0; j < W; ++j) d.c[j] = leaf; continue; }
        const Dp &l = dp[(size_t)nd.left], &r = dp[(size_t)nd.right];
        float dist[kMaxWidth + 1];
        for (int j = 2; j <= W; ++j) {
            dist[j] = kInf;
            for (int k = 1; k < j; ++k) {
                const float c = l.c[k - 1] + r.c[j - k - 1];
                if (c < dist[j]) { dist[j] = c; d.k[j] = (uint8_t)k; }
            }
        }
        const float internal = a + dist[W];
        d.leaf = leaf <= internal;
        d.c[0] = std::min(leaf, internal);
        for (int j = 2; j < W; ++j) {
            if (dist[j] < d.c[j - 2]) d.c[j - 1] = dist[j];
            else { d.c[j - 1] = d.c[j - 2]; d.k[j] = 0; }
        }
        d.c[W - 1] = 0.f;
    }
    out.sah_cost = dp[0].c[0];

    out.tri_order.reserve((size_t)nref);
    struct Child { int32_t n2; bool leaf; };
    // the children a subtree contributes when it is given `slots` slots of its parent's wide node
    // oct < 0: canonical (left to right); oct in 0..7: at every binary split the child a ray of that octant enters first comes first
    auto emit = [&](auto&& self, int32_t n, int slots, std::vector<Child>& outc, int oct) -> void {
        const Node2& nd = nodes[(size_t)n];
        const Dp& d = dp[(size_t)n];
        if (nd.count > 0 || slots == 1) { outc.push_back({n, nd.count > 0 || d.leaf != 0}); return; }
        if (d.k[slots] == 0) { self(self, n, slots - 1, outc, oct); return; }
        const bool flip = oct >= 0 && ((oct >> nd.axis) & 1);
        if (!flip) { self(self, nd.left, d.k[slots], outc, oct); self(self, nd.right, slots - d.k[slots], outc, oct); }
        else       { self(self, nd.right, slots - d.k[slots], outc, oct); self(self, nd.left, d.k[slots], outc, oct); }
    };
    struct Item { int32_t n2; int32_t wide; int depth; };
    std::vector<Item> queue;
    out.nodes.push_back(empty_node());
    queue.push_back({0, 0, 1});
    std::vector<Child> ch;
    for (size_t qi = 0; qi < queue.size(); ++qi) {
        const Item it = queue[qi];
        out.depth = std::max(out.depth, it.depth);
        ch.clear();
        const Node2& r = nodes[(size_t)it.n2];
        if (r.count > 0 || (it.n2 == 0 && dp[0].leaf)) ch.push_back({it.n2, true});      // a root that is itself a leaf: one leaf child
        else { emit(emit, r.left, dp[(size_t)it.n2].k[W], ch, -1); emit(emit, r.right, W - dp[(size_t)it.n2].k[W], ch, -1); }
        WideNode w = empty_node();
        w.n = (int)ch.size();
        for (int o = 0; o < 8; ++o) {
            for (int i = 0; i < kMaxWidth; ++i) w.order[o][i] = (uint8_t)i;
            if (w.n < 2) continue;
            std::vector<Child> oc;
            const bool flip = (o >> r.axis) & 1;
            const int kl = dp[(size_t)it.n2].k[W];
            if (!flip) { emit(emit, r.left, kl, oc, o); emit(emit, r.right, W - kl, oc, o); }
            else       { emit(emit, r.right, W - kl, oc, o); emit(emit, r.left, kl, oc, o); }
            for (size_t j = 0; j < oc.size() && j < (size_t)kMaxWidth; ++j)
                for (int i = 0; i < w.n; ++i) if (ch[(size_t)i].n2 == oc[j].n2) w.order[o][j] = (uint8_t)i;
        }
        for (int i = 0; i < w.n; ++i) {          // internal children first get consecutive wide indices
            const Node2& c = nodes[(size_t)ch[(size_t)i].n2];
            for (int k = 0; k < 3; ++k) { w.lo[i][k] = c.b.lo[k] - pad; w.hi[i][k] = c.b.hi[k] + pad; }
            if (!ch[(size_t)i].leaf) {
                w.child[i] = (int32_t)out.nodes.size();
                out.nodes.push_back(empty_node());
                queue.push_back({ch[(size_t)i].n2, w.child[i], it.depth + 1});
            }
        }
        for (int i = 0; i < w.n; ++i) {
            const Node2& c = nodes[(size_t)ch[(size_t)i].n2];
            if (ch[(size_t)i].leaf) {
                w.child[i] = -1;
                w.leaf_start[i] = (int32_t)out.tri_order.size();
                w.leaf_count[i] = c.ntri;
                for (int32_t j = c.first; j < c.first + c.ntri; ++j) out.tri_order.push_back(ref_tri.empty() ? order[(size_t)j] : ref_tri[(size_t)order[(size_t)j]]);
            }
        }
        out.nodes[(size_t)it.wide] = w;
    }
    return out;
}

}  // namespace iris
