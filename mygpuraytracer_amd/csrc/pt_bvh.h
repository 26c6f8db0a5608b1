// pt_bvh.h -- bounding-volume hierarchy over the triangles of one OBJ geom (what the reference only gestures at:
// `//TODO BVH` src/pathtrace.cu:289, BOUNDING_BOX :40,306-311).  It changes WHICH triangles meshIntersectionTest
// (src/intersections.h:207-233) looks at, never what it computes for one: every visited triangle goes through the
// same rayTriangle + barycentric point + distance arithmetic, and the winner is the minimum of (distance, face
// index) -- the reference's strict `t < tmin` in face order.
//
// Why the tree equals the loop (a bound, not a sample).  A subtree is skipped in two cases only.
//  (1) Its box starts beyond the best distance so far.  The best distance is |o - p|, p = w v0 + b0 p1 + b1 p2 with the
//      ACCEPTED barycentrics 0 <= b0 <= 1, b1 >= 0, b0 + b1 <= 1: a point of the triangle up to one rounding, hence of every
//      box around it.  A box whose entry parameter exceeds 1.0001 x that distance cannot hold a triangle with a nearer
//      such point.  No assumption about how well the triangle test is conditioned.
//  (2) The ray misses its box.  Here the argument needs the rounding error of glm::intersectRayTriangle in binary32.
//      With u = 2^-24, D = |o - v0|, |d| = 1, the computed quantities satisfy (standard dot / cross product bounds)
//          |a - a*| <= 6.5 u |e1||e2|,   |s.p - (s.p)*| <= 7.5 u D |e2|,   |d.q - (d.q)*| <= 8 u D |e1|,   |e2.q - (e2.q)*| <= 8 u D |e1||e2|
//      so when the test accepts (a >= FLT_EPSILON, 0 <= bx <= 1, by >= 0, bx + by <= 1, bz >= 0) the EXACT barycentrics and
//      ray parameter of the line's intersection with the triangle's plane obey
//          -eu <= u* <= 1 + eu,  -ev <= v*,  u* + v* <= 1 + eu + ev,  t* >= -et,
//          eu <= (13 u |e1||e2| + 15 u D |e2|) / a + 4u,   ev <= (13 u |e1||e2| + 16 u D |e1|) / a + 4u,   et <= 16 u D |e1||e2| / a.
//      The exact intersection point therefore lies within  eu |e1| + ev |e2|  of the triangle, and with the conditioning
//      kappa = a / (|e1||e2|)  (= sin(angle e1,e2) x |cos(angle ray, normal)|, at most 1)
//          distance(ray, triangle) + (how far behind the origin)  <=  (31 u D + 26 u L + 16 u D) / kappa + 8 u L  <  64 u (D + L) / kappa,
//      L = max(|e1|, |e2|) -- the aspect ratio of the triangle cancels, only kappa remains.  So a ray the loop accepts a
//      triangle for reaches every box around that triangle WIDENED BY 64 u (D + L) / kappa.  The traversal widens every
//      slab by  slack = 2^11 u (|o - c| + 4 R)  (c, R: centre and half diagonal of the root box; D <= |o - c| + R, L <= 2R;
//      bvhSlack / slabEntry in pt_device.h), on top of the static inflation below and of the slab test's own rounding
//      (4 u (|o| + |box|) per plane, inside the 4 R term; the four-wide walk adds the slack to the origin term before the fused
//      multiply-add instead of after it -- one more rounding of that size per plane, 5 u, inside the same term).  Hence:
//
//          for EVERY ray, at ANY distance, the tree returns the loop's face and distance whenever each triangle the loop
//          accepts is conditioned no worse than kappa >= 2^-5 (about 1.8 degrees off grazing for a right-angled triangle).
//
//      Below that conditioning the reference's own test decides by rounding noise over a band of width ~64 u D / kappa along
//      the triangle's edges; the static inflation (2e-3 of the node's diagonal + 1e-5 of the mesh's) still covers most of
//      it, but equality there is measured (tests/test_bvh.py: grazing rays, needles, far origins up to 10^6 mesh sizes),
//      not proven.  The slack grows with distance: a ray from 10^4 mesh sizes away prunes little -- as it must, its
//      triangle tests being accurate to a fraction of the mesh only.
//
// The traversal (bvhNearest) lives in pt_device.h next to the triangle test; this header is the host-side builder.
// Layout ("threaded" preorder, no stack): node = 2 x 16 bytes {lo.xyz, skip}{hi.xyz, leaf}.  skip = the next node
// when this one is missed or is a leaf (-1 = done); an inner node that is hit continues at n + 1.  leaf = count << 28
// | first (count 0 = inner, the low bits then name its right child).  Leaf triangles are stored in leaf order, BVH_TRI = 12 words each: the
// three vertices as loaded (the walk forms e1 = p1 - v0, e2 = p2 - v0 itself: the same subtractions as the upload-time table, and the
// reference interpolates the hit point from v0, p1, p2), the face index inside the geom, two pads.
#pragma once
#include "pt_device.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

namespace ptd {

// ---- host-side builder: binned SAH, leaves of <= BVH_LEAF_MAX triangles ------------------------------------------
struct BvhBuild {
    std::vector<BvhQuad> nodes;           // 2 per node, appended to whatever is already there
    std::vector<float> tris;              // BVH_TRI per leaf triangle, appended likewise
    std::vector<BvhWide4> wide;           // 4 per four-wide quantised node = 16 words (bvhNearestWide), appended likewise
};

namespace bvh_detail {
struct Prim { double lo[3], hi[3], c[3]; int face; };
struct Node { double lo[3], hi[3]; int left = -1, right = -1, first = 0, count = 0, size = 1; };

inline void boundsOf(const std::vector<Prim> &pr, int a, int b, double lo[3], double hi[3]) {
    for (int k = 0; k < 3; k++) { lo[k] = 1e300; hi[k] = -1e300; }
    for (int i = a; i < b; i++)
        for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], pr[i].lo[k]); hi[k] = std::max(hi[k], pr[i].hi[k]); }
}
inline double area(const double lo[3], const double hi[3]) {
    const double x = hi[0] - lo[0], y = hi[1] - lo[1], z = hi[2] - lo[2];
    return x * y + y * z + z * x;
}

inline int buildRec(std::vector<Prim> &pr, std::vector<Node> &out, int a, int b, int depth, int &maxdepth) {
    const int id = (int)out.size();
    if (depth > maxdepth) maxdepth = depth;
    out.emplace_back();
    {
        Node &nd = out[id];
        boundsOf(pr, a, b, nd.lo, nd.hi);
    }
    const int n = b - a;
    if (n <= BVH_LEAF_MAX || depth > 48) {
        // (depth guard: more than BVH_LEAF_MAX triangles may end in one leaf only if count still fits 4 bits)
        if (n <= 15) { out[id].first = a; out[id].count = n; return id; }
    }
    double clo[3] = {1e300, 1e300, 1e300}, chi[3] = {-1e300, -1e300, -1e300};
    for (int i = a; i < b; i++)
        for (int k = 0; k < 3; k++) { clo[k] = std::min(clo[k], pr[i].c[k]); chi[k] = std::max(chi[k], pr[i].c[k]); }
    constexpr int NB = 16;
    double bestCost = 1e300;
    int bestAxis = -1, bestBin = -1;
    for (int ax = 0; ax < 3; ax++) {
        const double ext = chi[ax] - clo[ax];
        if (!(ext > 0.0)) continue;
        int cnt[NB] = {0};
        double blo[NB][3], bhi[NB][3];
        for (int q = 0; q < NB; q++) for (int k = 0; k < 3; k++) { blo[q][k] = 1e300; bhi[q][k] = -1e300; }
        for (int i = a; i < b; i++) {
            int q = (int)((pr[i].c[ax] - clo[ax]) / ext * NB);
            q = std::min(std::max(q, 0), NB - 1);
            cnt[q]++;
            for (int k = 0; k < 3; k++) { blo[q][k] = std::min(blo[q][k], pr[i].lo[k]); bhi[q][k] = std::max(bhi[q][k], pr[i].hi[k]); }
        }
        double rArea[NB]; int rCnt[NB];
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
        int c = 0;
        for (int q = NB - 1; q >= 1; q--) {
            for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], blo[q][k]); hi[k] = std::max(hi[k], bhi[q][k]); }
            c += cnt[q];
            rArea[q] = c ? area(lo, hi) : 0.0; rCnt[q] = c;
        }
        for (int k = 0; k < 3; k++) { lo[k] = 1e300; hi[k] = -1e300; }
        c = 0;
        for (int q = 0; q < NB - 1; q++) {
            for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], blo[q][k]); hi[k] = std::max(hi[k], bhi[q][k]); }
            c += cnt[q];
            if (c == 0 || rCnt[q + 1] == 0) continue;
            const double cost = area(lo, hi) * c + rArea[q + 1] * rCnt[q + 1];
            if (cost < bestCost) { bestCost = cost; bestAxis = ax; bestBin = q; }
        }
    }
    int mid;
    if (bestAxis >= 0) {
        const double ext = chi[bestAxis] - clo[bestAxis], lo0 = clo[bestAxis];
        const int ax = bestAxis, bb = bestBin;
        auto it = std::stable_partition(pr.begin() + a, pr.begin() + b, [&](const Prim &p) {
            int q = (int)((p.c[ax] - lo0) / ext * NB);
            q = std::min(std::max(q, 0), NB - 1);
            return q <= bb;
        });
        mid = (int)(it - pr.begin());
    } else {
        mid = a + n / 2;                  // all centroids coincide: split by position in the list
    }
    if (mid <= a || mid >= b) mid = a + n / 2;
    const int l = buildRec(pr, out, a, mid, depth + 1, maxdepth);
    const int r = buildRec(pr, out, mid, b, depth + 1, maxdepth);
    out[id].left = l; out[id].right = r; out[id].size = 1 + out[l].size + out[r].size;
    return id;
}
}  // namespace bvh_detail

// Appends the tree of faces [faceStart, faceStart + faceCount) (15 floats each: 3 x (pos xyz, uv)) to `out`;
// tri9 holds v0, e1, e2 per face as uploaded for the plain loop.  Returns the root's node index (and the tree's depth).
// wroot_out / wneed_out: the tree's four-wide root (index into out.wide / 8; -1 if the root is a leaf) and the stack entries a walk of
// the wide nodes can need (3 per level below the root + 1).
inline int bvhBuild(const float *faces15, const float *tri9, int faceStart, int faceCount, BvhBuild &out, int *depth_out = nullptr,
                    int *wroot_out = nullptr, int *wneed_out = nullptr) {
    using namespace bvh_detail;
    std::vector<Prim> pr((size_t)faceCount);
    for (int j = 0; j < faceCount; j++) {
        const float *f = faces15 + (size_t)(faceStart + j) * 15;
        Prim &p = pr[j];
        p.face = j;
        for (int k = 0; k < 3; k++) {
            const double a = f[k], b = f[5 + k], c = f[10 + k];
            p.lo[k] = std::min(a, std::min(b, c)); p.hi[k] = std::max(a, std::max(b, c));
            p.c[k] = (a + b + c) / 3.0;
            if (!(p.lo[k] == p.lo[k]) || !(p.hi[k] == p.hi[k])) { p.lo[k] = -1e30; p.hi[k] = 1e30; p.c[k] = 0.0; }   // NaN vertex: never culled
        }
    }
    std::vector<Node> tree;
    tree.reserve((size_t)faceCount * 2);
    int maxdepth = 0;
    buildRec(pr, tree, 0, faceCount, 0, maxdepth);
    if (depth_out) *depth_out = maxdepth;          // root = depth 0; the ordered traversal needs a stack of depth + 1 entries
    double mdiag = 0.0;
    for (int k = 0; k < 3; k++) mdiag += (tree[0].hi[k] - tree[0].lo[k]) * (tree[0].hi[k] - tree[0].lo[k]);
    mdiag = std::sqrt(mdiag);
    const int base = (int)(out.nodes.size() / 2), tbase = (int)(out.tris.size() / BVH_TRI);
    out.nodes.resize(out.nodes.size() + 2 * tree.size());
    // preorder == creation order of buildRec (node, left subtree, right subtree), so node i sits at base + i
    std::vector<int> skip(tree.size(), -1);
    for (size_t i = 0; i < tree.size(); i++)
        if (tree[i].count == 0) { skip[tree[i].left] = base + tree[i].right; skip[tree[i].right] = skip[i]; }
    std::vector<float> flo(tree.size() * 3), fhi(tree.size() * 3);      // the boxes as stored (inflated, rounded outwards)
    for (size_t i = 0; i < tree.size(); i++) {
        const Node &nd = tree[i];
        double diag = 0.0;
        for (int k = 0; k < 3; k++) diag += (nd.hi[k] - nd.lo[k]) * (nd.hi[k] - nd.lo[k]);
        const double m = 2e-3 * std::sqrt(diag) + 1e-5 * mdiag + 1e-30;
        float lo[3], hi[3];
        for (int k = 0; k < 3; k++) {
            lo[k] = nextafterf((float)(nd.lo[k] - m), -INFINITY);
            hi[k] = nextafterf((float)(nd.hi[k] + m), INFINITY);
        }
        for (int k = 0; k < 3; k++) { flo[i * 3 + k] = lo[k]; fhi[i * 3 + k] = hi[k]; }
        BvhQuad A{lo[0], lo[1], lo[2], skip[i]};
        // leaf: count << 28 | first triangle; inner node: its right child (count bits 0) -- the left child is n + 1, so a visit
        // can request both children's boxes at once instead of learning the right child from the left child's skip link
        BvhQuad B{hi[0], hi[1], hi[2], nd.count ? (int32_t)(((uint32_t)nd.count << 28) | (uint32_t)(tbase + nd.first)) : (int32_t)(base + nd.right)};
        out.nodes[2 * (base + i)] = A; out.nodes[2 * (base + i) + 1] = B;
    }
    // Four-wide quantised nodes over the same tree (layout and walk: bvhNearestWide in pt_device.h): a wide node per inner node
    // reached in an even number of steps from the root; its entries are that node's grandchildren, or a child where the child is a
    // leaf (leaves first), each with the binary node's own stored box rounded OUTWARDS onto the wide node's 8-bit grid.
    {
        const int wbase = (int)(out.wide.size() / 4);
        auto alloc = [&]() { out.wide.resize(out.wide.size() + 4); return (int)(out.wide.size() / 4) - 1; };
        auto bits = [](float f) { int32_t v; memcpy(&v, &f, 4); return v; };
        int wroot = -1;
        if (tree[0].count == 0 && tbase + faceCount < (1 << 24)) {
            wroot = alloc();
            std::vector<int> level{0}, wide_of{wroot};
            for (size_t q = 0; q < level.size(); q++) {
                const int i = level[q], w = wide_of[q];
                int ent[4], ne = 0;
                for (int pass = 0; pass < 2; pass++)                     // leaves into the first slots
                    for (int c : {tree[i].left, tree[i].right}) {
                        if (tree[c].count) { if (pass == 0) ent[ne++] = c; }
                        else for (int g : {tree[c].left, tree[c].right})
                            if ((tree[g].count != 0) == (pass == 0)) ent[ne++] = g;
                    }
                // the grid: origin = the entries' common lower corner, step = extent / 255 rounded up until 255 steps reach the top
                float org[3], step[3];
                for (int k = 0; k < 3; k++) {
                    float lo = INFINITY, hi = -INFINITY;
                    for (int e = 0; e < ne; e++) { lo = std::min(lo, flo[ent[e] * 3 + k]); hi = std::max(hi, fhi[ent[e] * 3 + k]); }
                    org[k] = lo;
                    float st = (hi - lo) / 255.0f;
                    if (!(st > 0.0f)) st = 1e-30f;
                    while (fmaf(255.0f, st, lo) < hi) st = nextafterf(st, INFINITY);
                    step[k] = st;
                }
                uint32_t ql[3] = {0, 0, 0}, qh[3] = {0, 0, 0};
                int32_t ref[4] = {-1, -1, -1, -1};
                for (int e = 0; e < ne; e++) {
                    const int b = ent[e];
                    for (int k = 0; k < 3; k++) {
                        // largest grid coordinate whose point is <= the box's lower bound / smallest whose point is >= its upper
                        // bound, decided with the arithmetic the walk uses (one fused multiply-add)
                        int a = (int)floorf((flo[b * 3 + k] - org[k]) / step[k]);
                        a = std::min(255, std::max(0, a));
                        while (a > 0 && fmaf((float)a, step[k], org[k]) > flo[b * 3 + k]) a--;
                        int z = (int)ceilf((fhi[b * 3 + k] - org[k]) / step[k]);
                        z = std::min(255, std::max(0, z));
                        while (z < 255 && fmaf((float)z, step[k], org[k]) < fhi[b * 3 + k]) z++;
                        // (a = 0 gives the origin itself, <= every lower bound; z = 255 reaches the top by the choice of step)
                        ql[k] |= (uint32_t)a << (8 * e); qh[k] |= (uint32_t)z << (8 * e);
                    }
                    if (tree[b].count) ref[e] = (int32_t)(0x80000000u | ((uint32_t)tree[b].count << 24) | (uint32_t)(tbase + tree[b].first));
                    else {
                        const int cw = alloc();
                        ref[e] = cw;
                        level.push_back(b); wide_of.push_back(cw);
                    }
                }
                BvhWide4 *W = &out.wide[(size_t)w * 4];
                W[0] = BvhWide4{bits(org[0]), bits(org[1]), bits(org[2]), bits(step[0])};
                W[1] = BvhWide4{bits(step[1]), bits(step[2]), ref[0], ref[1]};
                W[2] = BvhWide4{ref[2], ref[3], (int32_t)ql[0], (int32_t)ql[1]};
                W[3] = BvhWide4{(int32_t)ql[2], (int32_t)qh[0], (int32_t)qh[1], (int32_t)qh[2]};
            }
        }
        // the stack a walk can need, exactly: taking one entry of a wide node (leaf or inner -- the walk defers leaves like inner
        // entries) leaves at most its other entries on the stack -- need(w) = (entries of w - 1) + max over inner entries c of need(c);
        // wide nodes were created parents first, so a sweep from the back sees every child before its parent
        const int nw = (int)(out.wide.size() / 4) - wbase;
        std::vector<int> need((size_t)std::max(nw, 1), 0);
        for (int w = nw - 1; w >= 0; w--) {
            const BvhWide4 *W = &out.wide[(size_t)(wbase + w) * 4];
            const int32_t refs[4] = {W[1].c, W[1].d, W[2].a, W[2].b};
            int entries = 0, deepest = 0;
            for (int k = 0; k < 4; k++) {
                if (refs[k] != -1) entries++;
                if (refs[k] >= 0) deepest = std::max(deepest, need[(size_t)(refs[k] - wbase)]);
            }
            need[(size_t)w] = entries ? entries - 1 + deepest : 0;
        }
        if (wroot_out) *wroot_out = wroot;
        if (wneed_out) *wneed_out = (nw ? need[0] : 0) + 1;
    }
    out.tris.resize(out.tris.size() + (size_t)faceCount * BVH_TRI);
    for (int i = 0; i < faceCount; i++) {
        const int j = pr[i].face;
        const float *f = faces15 + (size_t)(faceStart + j) * 15;
        float *o = &out.tris[(size_t)(tbase + i) * BVH_TRI];
        for (int k = 0; k < 3; k++) { o[k] = f[k]; o[3 + k] = f[5 + k]; o[6 + k] = f[10 + k]; }
        memcpy(&o[9], &j, 4);
        o[10] = o[11] = 0.f;
    }
    (void)tri9;
    return base;
}

// The reference's loop over all faces (src/intersections.h:213-233) on the host, for the CPU check of the tree.
inline float loopNearestHost(const float *faces15, const float *tri9, int nfaces, vec3 o, vec3 d, int &face) {
    float tmin = 3.402823466e+38f;
    face = -1;
    for (int j = 0; j < nfaces; j++) {
        const float *t9 = tri9 + (size_t)j * 9, *f = faces15 + (size_t)j * 15;
        const vec3 v0 = V3(t9[0], t9[1], t9[2]), e1 = V3(t9[3], t9[4], t9[5]), e2 = V3(t9[6], t9[7], t9[8]);
        float b0, b1;
        if (rayTriangle(o, d, v0, e1, e2, b0, b1)) {
            const vec3 p1 = V3(f[5], f[6], f[7]), p2 = V3(f[10], f[11], f[12]);
            const float w = 1 - b0 - b1;
            const vec3 p = add(add(scale(v0, w), scale(p1, b0)), scale(p2, b1));
            const float t = length(sub(o, p));
            if (t < tmin) { tmin = t; face = j; }
        }
    }
    return tmin;
}

}  // namespace ptd
