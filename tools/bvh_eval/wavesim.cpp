// Wave-level simulator of the bake kernels' traversal (iris_trace.h trace_stream as tile_sort_trace drives it): the kernel's OWN rays
// (tools/bvh_eval/dump_block_rays.py), its BVH (iris_amd/csrc/bvh_build.cpp, 8-bit planes, per-octant child order) and its SCHEDULE -- 64 persistent lanes per
// wave, four waves of a workgroup drawing from one sorted ray list, refill when 48 lanes are idle, alternating node / leaf phases with the 12-lane early exit,
// the shared scalar visit when >= 44 of the lanes at a node sit at one node (test gated off after two misses) -- replayed on the host.  What it answers
// (round 6): how do SIMD lane utilisation, the share of node visits taken through the shared path and the wave-level step counts change when the rays are
// sorted over a WINDOW of many tiles instead of one tile (5120 rays / 256 bins = 20 rays per bin today)?
//
//   g++ -O2 -std=c++17 -I iris_amd/csrc tools/bvh_eval/wavesim.cpp iris_amd/csrc/bvh_build.cpp -lpthread -o /tmp/wavesim
//   wavesim room.bin block_rays.bin <scheme> [<scheme> ...]
// scheme:
//   tile                        today's kernel: tiles of 32 pixels (8 x 4 of the host's 8 x 8 block order) x spp rays, key = dir_bin (octant | 8 x 4 cells)
//   win:WX:WY:NU:NV:OX:OY:ord   windows of WX x WY pixels; key = octant | NU x NV direction cells inside the octant | origin cells of OX x OY pixels;
//                               ord = d (direction cell major, origin cell minor) | o (origin cell major) ; one workgroup (4 waves) per window
// Output: one JSON object per (scheme, lobe) with the counts per ray and a modelled vector-instruction count per ray
// (regular node step 87, shared 74, leaf step 60, refill round 45 vector instructions per wave-level step: DESIGN.md 5e).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "bvh_build.h"

using namespace iris;

struct V3 { float x, y, z; };
static V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
static float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

static constexpr uint32_t kLeaf = 0x80000000u, kEmpty = 0xffffffffu;
static int kRefillMin = 48, kRefillTop = 0 /* 0: = kRefillMin */, kPhaseMin = 12, kPhaseMinLeaf = 0 /* 0: = kPhaseMin */, kScalarTop = 44, kSharedTries = 2;   // the kernel's constants (iris_trace.h); WAVESIM_REFILL / _PHASE / _TOP / _TRIES override them

struct QNode { float lo[4][3], hi[4][3]; uint32_t ref[4]; };   // decoded (quantised, conservative) child boxes; ref: node index or kLeaf | start << 3 | count; unused: inverted box

struct Scene {
    std::vector<float> verts; std::vector<int32_t> faces; WideBvh bvh; std::vector<QNode> q;
    V3 vert(int64_t fi, int k) const { const float* p = verts.data() + (int64_t)faces[fi * 3 + k] * 3; return V3{p[0], p[1], p[2]}; }
};

static bool load_scene(const char* path, Scene& sc) {
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    int64_t nv, nf;
    if (fread(&nv, 8, 1, f) != 1 || fread(&nf, 8, 1, f) != 1) return false;
    sc.verts.resize(nv * 3); sc.faces.resize(nf * 3);
    if (fread(sc.verts.data(), 4, nv * 3, f) != (size_t)nv * 3 || fread(sc.faces.data(), 4, nf * 3, f) != (size_t)nf * 3) return false;
    fclose(f);
    sc.bvh = build_wide_bvh(sc.verts.data(), nv, sc.faces.data(), nf, 4, 4, 2e-5f, 0.7f, 8.f);
    const size_t nn = sc.bvh.nodes.size();
    sc.q.resize(nn);
    for (size_t i = 0; i < nn; ++i) {
        const WideNode& w = sc.bvh.nodes[i];
        for (int k = 0; k < 3; ++k) {
            float org = INFINITY, hi3 = -INFINITY;
            for (int s = 0; s < w.n; ++s) { org = std::min(org, w.lo[s][k]); hi3 = std::max(hi3, w.hi[s][k]); }
            const double ext = (double)hi3 - (double)org;
            int e = -126;
            if (ext > 0) e = std::max(-126, (int)std::ceil(std::log2(ext / 255.0)));
            while (std::ldexp(255.0, e) < ext) ++e;
            const double scl = std::ldexp(1.0, e);
            for (int s = 0; s < 4; ++s) {
                if (s >= w.n) { sc.q[i].lo[s][k] = INFINITY; sc.q[i].hi[s][k] = -INFINITY; continue; }
                int lo = (int)std::floor(((double)w.lo[s][k] - org) / scl), hi = (int)std::ceil(((double)w.hi[s][k] - org) / scl);
                lo = std::min(255, std::max(0, lo)); hi = std::min(255, std::max(0, hi));
                sc.q[i].lo[s][k] = (float)(org + lo * scl); sc.q[i].hi[s][k] = (float)(org + hi * scl);
            }
        }
        for (int s = 0; s < 4; ++s)
            sc.q[i].ref[s] = s >= w.n ? kEmpty : (w.child[s] >= 0 ? (uint32_t)w.child[s] : (kLeaf | (uint32_t)w.leaf_start[s] << 3 | (uint32_t)w.leaf_count[s]));
    }
    return true;
}

struct Ray { float o[3], d[3]; int px, py, s; };

struct Lane {
    uint32_t cur = kEmpty; uint32_t pend = kEmpty; int sp = 0; uint32_t stack[128]; float best = INFINITY; float o[3], d[3], id[3]; int oct = 0; bool live = false;
};

struct Stats {
    long long rays = 0, node_visits = 0, tri_tests = 0, node_iters = 0, shared_iters = 0, shared_visits = 0, leaf_iters = 0, refill_rounds = 0, drain_node_iters = 0, drain_visits = 0;
    long long node_lines = 0, leaf_lines = 0;        // distinct records touched by the regular node steps / the leaf steps
    long long ge32_iters = 0, ge32_visits = 0;       // node steps with >= 32 lanes at one node (instrumented-launch statistic of round 4)
    long long shared_tests = 0;
    long long ns_leaf_lanes = 0, ns_idle_lanes = 0, ls_node_lanes = 0, ls_idle_lanes = 0, shared_out_lanes = 0;   // who waits: lanes at a leaf / idle during node steps, at a node / idle during leaf steps, lanes at ANOTHER node sitting out a shared step
};

struct Wave {
    Lane l[64]; std::vector<Lane> own; bool more = true, done = false; int shared_tries = kSharedTries; int phase = 0;   // 0 top, 1 node, 2 leaf
};

struct Sim {
    const Scene& sc; Stats st; int parking = 0; int pool_take = 64; int park_max_sp = 12; std::vector<Lane> pool;   // parking: 0 off, 1 one pool per workgroup, 2 one pool per WAVE (no exchange between waves)
     long long parked = 0, pool_waves = 0; bool postpone = false;   // postpone: a lane that reaches a leaf in the node phase parks it (one per lane) and goes on with its stack until the phase ends
    explicit Sim(const Scene& s) : sc(s) {}

    void begin(Lane& L, const Ray& r) {
        for (int k = 0; k < 3; ++k) { L.o[k] = r.o[k]; L.d[k] = r.d[k]; L.id[k] = 1.f / r.d[k]; }
        L.oct = (r.d[0] < 0 ? 1 : 0) | (r.d[1] < 0 ? 2 : 0) | (r.d[2] < 0 ? 4 : 0);
        L.best = INFINITY; L.sp = 0; L.cur = 0; L.pend = kEmpty; L.live = true;
    }
    void node_step(Lane& L) {
        const QNode& n = sc.q[L.cur];
        const WideNode& w = sc.bvh.nodes[L.cur];
        uint32_t hit[4]; int m = 0;
        for (int j = 0; j < w.n; ++j) {
            const int s = (int)w.order[L.oct][j];
            float tn = 0.f, tf = L.best;
            for (int k = 0; k < 3; ++k) {
                float t0 = (n.lo[s][k] - L.o[k]) * L.id[k], t1 = (n.hi[s][k] - L.o[k]) * L.id[k];
                if (t0 > t1) std::swap(t0, t1);
                tn = std::max(tn, t0); tf = std::min(tf, t1);
            }
            if (tn <= tf) hit[m++] = n.ref[s];
        }
        if (m == 0) { L.cur = L.sp ? L.stack[--L.sp] : kEmpty; return; }
        for (int i = m - 1; i >= 1; --i) L.stack[L.sp++] = hit[i];
        L.cur = hit[0];
    }
    void park(Lane& L) {      // after a node step / pop: a leaf reference is parked when the slot is free
        if (postpone && L.pend == kEmpty && L.cur != kEmpty && (L.cur & kLeaf)) { L.pend = L.cur; L.cur = L.sp ? L.stack[--L.sp] : kEmpty; }
    }
    // tests the first triangle of leaf reference `ref`; returns what is left of the leaf (kEmpty: finished)
    uint32_t tri_one(Lane& L, uint32_t ref) {
        const int start = (int)((ref & 0x7fffffffu) >> 3);
        const int64_t ti = sc.bvh.tri_order[start];
        V3 o{L.o[0], L.o[1], L.o[2]}, d{L.d[0], L.d[1], L.d[2]};
        V3 q0 = sc.vert(ti, 0), f1 = sub(sc.vert(ti, 1), q0), f2 = sub(sc.vert(ti, 2), q0);
        V3 pv = cross(d, f2); float det = dot(f1, pv); float inv = 1.f / det;
        V3 tv = sub(o, q0); float uu = dot(tv, pv) * inv; V3 qv = cross(tv, f1); float vv = dot(d, qv) * inv; float tt = dot(f2, qv) * inv;
        if (uu >= 0 && vv >= 0 && uu + vv <= 1 && tt >= 0 && tt < L.best) L.best = tt;
        ref += 7u;
        return (ref & 7u) == 0u ? kEmpty : ref;
    }
    void leaf_step(Lane& L) {
        if (L.pend != kEmpty) { L.pend = tri_one(L, L.pend); return; }
        L.cur = tri_one(L, L.cur);
        if (L.cur == kEmpty) L.cur = L.sp ? L.stack[--L.sp] : kEmpty;
    }
    static bool at_node(const Lane& L) { return L.cur != kEmpty && !(L.cur & kLeaf); }
    static bool at_leaf(const Lane& L) { return L.pend != kEmpty || (L.cur != kEmpty && (L.cur & kLeaf)); }
    static bool is_idle(const Lane& L) { return L.cur == kEmpty && L.pend == kEmpty; }

    // one wave-level step (refill round, node step or leaf step); false when the wave has finished its share of the list
    bool step(Wave& w, const std::vector<Ray>& rays, const std::vector<uint32_t>& list, size_t& cursor) {
        for (;;) {
            if (w.done) return false;
            if (w.phase == 0) {
                int n_idle = 0;
                for (auto& L : w.l) n_idle += is_idle(L);
                bool stepped = false;
                if (w.more && (n_idle >= (kRefillTop ? kRefillTop : kRefillMin) || n_idle == 64)) {
                    bool any = false;
                    if (parking) {
                        // PARKING (round 6 study): a refilling wave hands the rays it still carries to a pool and starts 64 fresh rays together; when the pool holds a
                        // wave's worth, a refilling wave takes those instead (the stragglers run with each other); nothing is parked once the list is exhausted;
                        // a ray whose stack has entries beyond the LDS part stays where it is
                        std::vector<Lane>& pl = parking == 2 ? w.own : pool;
                        if (cursor < list.size()) for (auto& L : w.l) if (!is_idle(L) && L.sp <= park_max_sp) { pl.push_back(L); ++parked; L.cur = kEmpty; L.pend = kEmpty; }
                        const bool from_pool = (int)pl.size() >= pool_take || cursor >= list.size();
                        if (from_pool && !pl.empty()) ++pool_waves;
                        for (auto& L : w.l) if (is_idle(L)) {
                            if (from_pool) { if (!pl.empty()) { L = pl.back(); pl.pop_back(); any = true; } }
                            else if (cursor < list.size()) { begin(L, rays[list[cursor++]]); st.rays++; any = true; }
                        }
                    } else
                    for (auto& L : w.l) if (is_idle(L)) {
                        L.live = false;
                        if (cursor < list.size()) { begin(L, rays[list[cursor++]]); st.rays++; any = true; }
                    }
                    if (!any) w.more = false;
                    w.shared_tries = kSharedTries;
                    st.refill_rounds++;
                    stepped = true;
                }
                bool work = false;
                for (auto& L : w.l) work |= !is_idle(L);
                if (!work) { if (!w.more) { w.done = true; return false; } if (stepped) return true; continue; }
                w.phase = 1;
                if (stepped) return true;
            }
            if (w.phase == 1) {
                int n_node = 0, n_leaf = 0, n_idle = 0, first = -1;
                for (int i = 0; i < 64; ++i) { const Lane& L = w.l[i]; if (at_node(L)) { if (first < 0) first = i; ++n_node; } else if (at_leaf(L)) ++n_leaf; else ++n_idle; }
                if (n_node == 0 || (n_node < kPhaseMin && n_leaf >= kPhaseMin) || (n_node <= 64 - kRefillMin && w.more && n_idle >= kRefillMin)) { w.phase = 2; continue; }
                // statistic: lanes at the first lane's node
                int n_same = 0;
                for (auto& L : w.l) n_same += at_node(L) && L.cur == w.l[first].cur && L.oct == w.l[first].oct;
                if (n_same >= 32) { st.ge32_iters++; st.ge32_visits += n_same; }
                if (w.shared_tries > 0) {
                    st.shared_tests++;
                    if (n_same >= kScalarTop) {
                        const uint32_t c0 = w.l[first].cur; const int o0 = w.l[first].oct;
                        for (auto& L : w.l) if (at_node(L) && L.cur == c0 && L.oct == o0) { node_step(L); park(L); st.node_visits++; st.shared_visits++; if (!w.more) st.drain_visits++; }
                        st.node_iters++; st.shared_iters++; if (!w.more) st.drain_node_iters++;
                        st.ns_leaf_lanes += n_leaf; st.ns_idle_lanes += n_idle; st.shared_out_lanes += n_node - n_same;
                        w.shared_tries = kSharedTries;
                        return true;
                    }
                    --w.shared_tries;
                }
                {   // distinct node records of the step
                    uint64_t keys[64]; int nk = 0;
                    for (auto& L : w.l) if (at_node(L)) keys[nk++] = ((uint64_t)L.oct << 32) | L.cur;
                    std::sort(keys, keys + nk);
                    st.node_lines += std::unique(keys, keys + nk) - keys;
                }
                for (auto& L : w.l) if (at_node(L)) { node_step(L); park(L); st.node_visits++; if (!w.more) st.drain_visits++; }
                st.node_iters++; if (!w.more) st.drain_node_iters++;
                st.ns_leaf_lanes += n_leaf; st.ns_idle_lanes += n_idle;
                return true;
            }
            {   // leaf phase
                int n_node = 0, n_leaf = 0, n_idle = 0;
                for (auto& L : w.l) { if (at_leaf(L)) ++n_leaf; else if (at_node(L)) ++n_node; else ++n_idle; }
                if (n_leaf == 0 || (n_leaf < (kPhaseMinLeaf ? kPhaseMinLeaf : kPhaseMin) && n_node >= (kPhaseMinLeaf ? kPhaseMinLeaf : kPhaseMin)) || (n_leaf <= 64 - kRefillMin && w.more && n_idle >= kRefillMin)) { w.phase = 0; continue; }
                uint32_t keys[64]; int nk = 0;
                for (auto& L : w.l) if (at_leaf(L)) keys[nk++] = (L.cur & 0x7fffffffu) >> 3;
                std::sort(keys, keys + nk);
                st.leaf_lines += std::unique(keys, keys + nk) - keys;
                for (auto& L : w.l) if (at_leaf(L)) { leaf_step(L); st.tri_tests++; }
                st.leaf_iters++; st.ls_node_lanes += n_node; st.ls_idle_lanes += n_idle;
                return true;
            }
        }
    }
    // one workgroup: four waves, one sorted list, steps interleaved round-robin
    void workgroup(const std::vector<Ray>& rays, const std::vector<uint32_t>& list) {
        std::vector<Wave> w(4);
        size_t cursor = 0;
        pool.clear();
        for (bool any = true; any;) {
            any = false;
            for (auto& x : w) any |= step(x, rays, list, cursor);
        }
    }
};

static int dir_cell(const float* d, int NU, int NV) {
    const float ax = std::fabs(d[0]), ay = std::fabs(d[1]), az = std::fabs(d[2]);
    const float inv = 1.f / (ax + ay + az + 1e-30f);
    const float a = ax * inv, b = ay * inv;
    const float v = b / (1.f - a + 1e-30f);
    const int iu = std::min(NU - 1, (int)(a * NU)), iv0 = std::min(NV - 1, (int)(v * NV));
    const int iv = (iu & 1) ? NV - 1 - iv0 : iv0;
    return iu * NV + iv;
}
static int octant(const float* d) { return (d[0] < 0 ? 1 : 0) | (d[1] < 0 ? 2 : 0) | (d[2] < 0 ? 4 : 0); }
static uint32_t morton2(uint32_t x, uint32_t y) {
    auto sp = [](uint32_t v) { v &= 0xffff; v = (v | (v << 8)) & 0x00ff00ff; v = (v | (v << 4)) & 0x0f0f0f0f; v = (v | (v << 2)) & 0x33333333; v = (v | (v << 1)) & 0x55555555; return v; };
    return sp(x) | (sp(y) << 1);
}

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: wavesim room.bin block_rays.bin scheme...\n"); return 1; }
    if (getenv("WAVESIM_REFILL")) kRefillMin = atoi(getenv("WAVESIM_REFILL"));
    if (getenv("WAVESIM_PHASE")) kPhaseMin = atoi(getenv("WAVESIM_PHASE"));
    if (getenv("WAVESIM_REFILL_TOP")) kRefillTop = atoi(getenv("WAVESIM_REFILL_TOP"));        // the refill test at the top of a round (where the wave is anyway) against the early exit from a phase
    if (getenv("WAVESIM_PHASE_LEAF")) kPhaseMinLeaf = atoi(getenv("WAVESIM_PHASE_LEAF"));     // the early exit of the LEAF phase on its own
    if (getenv("WAVESIM_TOP")) kScalarTop = atoi(getenv("WAVESIM_TOP"));
    if (getenv("WAVESIM_TRIES")) kSharedTries = atoi(getenv("WAVESIM_TRIES"));
    Scene sc;
    if (!load_scene(argv[1], sc)) { fprintf(stderr, "cannot load %s\n", argv[1]); return 1; }
    FILE* rf = fopen(argv[2], "rb");
    if (!rf) { fprintf(stderr, "cannot open %s\n", argv[2]); return 1; }
    int64_t n_groups, block, spp;
    if (fread(&n_groups, 8, 1, rf) != 1 || fread(&block, 8, 1, rf) != 1 || fread(&spp, 8, 1, rf) != 1) return 1;
    struct Group { int lobe, x0, y0; std::vector<Ray> rays; };
    std::vector<Group> groups((size_t)n_groups);
    for (auto& g : groups) {
        int32_t hdr[3];
        if (fread(hdr, 4, 3, rf) != 3) return 1;
        g.lobe = hdr[0]; g.x0 = hdr[1]; g.y0 = hdr[2];
        const size_t n = (size_t)(block * block * spp);
        std::vector<float> buf(n * 6);
        if (fread(buf.data(), 4, n * 6, rf) != n * 6) return 1;
        g.rays.resize(n);
        for (size_t i = 0; i < n; ++i) {
            Ray& r = g.rays[i];
            memcpy(r.o, &buf[i * 6], 12); memcpy(r.d, &buf[i * 6 + 3], 12);
            const size_t p = i / (size_t)spp;
            r.py = (int)(p / (size_t)block); r.px = (int)(p % (size_t)block); r.s = (int)(i % (size_t)spp);
        }
    }
    fclose(rf);
    for (int a = 3; a < argc; ++a) {
        std::string scheme = argv[a];
        const bool postpone = scheme.rfind("pend+", 0) == 0;
        if (postpone) scheme = scheme.substr(5);
        const int parking = scheme.rfind("park+", 0) == 0 ? 1 : scheme.rfind("wpark+", 0) == 0 ? 2 : 0;
        if (parking) scheme = scheme.substr(parking == 1 ? 5 : 6);
        int WX = 8, WY = 4, NU = 8, NV = 4, OX = 0, OY = 0; char ord = 'd';
        const bool tile = scheme == "tile";
        if (!tile) {
            if (sscanf(scheme.c_str(), "win:%d:%d:%d:%d:%d:%d:%c", &WX, &WY, &NU, &NV, &OX, &OY, &ord) != 7) { fprintf(stderr, "bad scheme %s\n", scheme.c_str()); return 1; }
        }
        if (OX <= 0) OX = WX;
        if (OY <= 0) OY = WY;
        Stats per_lobe[8]; long long parked_tot[8] = {0};
        for (const auto& g : groups) {
            Sim sim(sc); sim.postpone = postpone; sim.parking = parking; if (getenv("WAVESIM_POOL_TAKE")) sim.pool_take = atoi(getenv("WAVESIM_POOL_TAKE"));
            // windows of WX x WY pixels (the host's block-ordered pixel list cuts them; a tile = 8 x 4: half an 8 x 8 block)
            for (int wy = 0; wy + WY <= block; wy += WY)
                for (int wx = 0; wx + WX <= block; wx += WX) {
                    std::vector<std::pair<uint64_t, uint32_t>> keyed;
                    keyed.reserve((size_t)WX * WY * spp);
                    for (int y = wy; y < wy + WY; ++y)
                        for (int x = wx; x < wx + WX; ++x)
                            for (int s = 0; s < spp; ++s) {
                                const uint32_t i = (uint32_t)(((size_t)y * block + x) * spp + s);
                                const Ray& r = g.rays[i];
                                const uint64_t dc = (uint64_t)octant(r.d) * (uint64_t)(NU * NV) + (uint64_t)dir_cell(r.d, NU, NV);
                                const uint64_t oc = morton2((uint32_t)((x - wx) / OX), (uint32_t)((y - wy) / OY));
                                const uint64_t key = ord == 'd' ? (dc << 20) | oc : (oc << 40) | dc;
                                keyed.emplace_back(key, i);
                            }
                    std::stable_sort(keyed.begin(), keyed.end(), [](const auto& p, const auto& q) { return p.first < q.first; });
                    std::vector<uint32_t> list(keyed.size());
                    for (size_t i = 0; i < keyed.size(); ++i) list[i] = keyed[i].second;
                    sim.workgroup(g.rays, list);
                }
            Stats& t = per_lobe[g.lobe & 7];
            const Stats& s = sim.st;
            t.rays += s.rays; t.node_visits += s.node_visits; t.tri_tests += s.tri_tests; t.node_iters += s.node_iters; t.shared_iters += s.shared_iters; t.shared_visits += s.shared_visits;
            t.leaf_iters += s.leaf_iters; t.refill_rounds += s.refill_rounds; t.drain_node_iters += s.drain_node_iters; t.drain_visits += s.drain_visits; t.node_lines += s.node_lines;
            t.leaf_lines += s.leaf_lines; t.ge32_iters += s.ge32_iters; t.ge32_visits += s.ge32_visits; t.shared_tests += s.shared_tests;
            t.shared_tests += 0; parked_tot[g.lobe & 7] += sim.parked;
            t.ns_leaf_lanes += s.ns_leaf_lanes; t.ns_idle_lanes += s.ns_idle_lanes; t.ls_node_lanes += s.ls_node_lanes; t.ls_idle_lanes += s.ls_idle_lanes; t.shared_out_lanes += s.shared_out_lanes;
        }
        for (int l = 0; l < 8; ++l) {
            const Stats& s = per_lobe[l];
            if (!s.rays) continue;
            const double R = (double)s.rays, reg = (double)(s.node_iters - s.shared_iters);
            const double model = (reg * 87.0 + s.shared_iters * 74.0 + s.leaf_iters * 60.0 + s.refill_rounds * 45.0) / R;
            printf("{\"scheme\": \"%s\", \"lobe\": %d, \"rays\": %lld, \"rays_per_window\": %lld, \"node_visits_per_ray\": %.3f, \"tri_tests_per_ray\": %.3f, "
                   "\"node_step_lane_util\": %.4f, \"leaf_step_lane_util\": %.4f, \"shared_visit_share\": %.4f, \"shared_step_share\": %.4f, \"ge32_step_share\": %.4f, \"ge32_visit_share\": %.4f, "
                   "\"node_steps_per_ray\": %.4f, \"leaf_steps_per_ray\": %.4f, \"refill_rounds_per_ray\": %.5f, \"drain_node_step_share\": %.4f, \"drain_lane_util\": %.4f, "
                   "\"node_step_lanes_waiting_at_a_leaf\": %.4f, \"node_step_lanes_idle\": %.4f, \"node_step_lanes_sitting_out_a_shared_step\": %.4f, \"leaf_step_lanes_waiting_at_a_node\": %.4f, \"leaf_step_lanes_idle\": %.4f, "
                   "\"records_per_regular_node_step\": %.2f, \"records_per_leaf_step\": %.2f, \"vector_loads_per_ray\": %.3f, \"modelled_traversal_vector_instructions_per_ray\": %.2f, \"parked_per_ray\": %.4f}\n",
                   (std::string(postpone ? "pend+" : parking == 1 ? "park+" : parking == 2 ? "wpark+" : "") + scheme).c_str(), l, s.rays, (long long)WX * WY * spp, s.node_visits / R, s.tri_tests / R, (double)s.node_visits / (64.0 * s.node_iters), (double)s.tri_tests / (64.0 * s.leaf_iters),
                   (double)s.shared_visits / s.node_visits, (double)s.shared_iters / s.node_iters, (double)s.ge32_iters / s.node_iters, (double)s.ge32_visits / s.node_visits,
                   s.node_iters / R, s.leaf_iters / R, s.refill_rounds / R, (double)s.drain_node_iters / s.node_iters, s.drain_node_iters ? (double)s.drain_visits / (64.0 * s.drain_node_iters) : 0.0,
                   s.ns_leaf_lanes / (64.0 * s.node_iters), s.ns_idle_lanes / (64.0 * s.node_iters), s.shared_out_lanes / (64.0 * s.node_iters), s.ls_node_lanes / (64.0 * s.leaf_iters), s.ls_idle_lanes / (64.0 * s.leaf_iters),
                   reg > 0 ? s.node_lines / reg : 0.0, (double)s.leaf_lines / s.leaf_iters, (reg * 4.0 + s.leaf_iters * 3.0) / R, model, parked_tot[l] / R);
            fflush(stdout);
        }
    }
    return 0;
}
