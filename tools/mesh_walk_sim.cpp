// mesh_walk_sim.cpp -- CPU experiment behind k_mesh's ray ordering (round 4): what does a WAVE of 64 four-wide BVH walks cost when its
// lanes hold these rays?  Builds the product's own tree (pt_bvh.h) over a face list, runs the product's own walk (bvhNearestWide,
// compiled for the host) on every ray and records each walk's sequence of steps -- node visits and leaves with their triangle counts --
// then replays the rays in groups of 64 the way the device executes the "while-while" loop: all lanes that hold an inner node step
// together until none does, then the lanes that hold a leaf test its triangles together (the triangle loop runs to the longest
// count in the wave), and so on until every lane is done.  Prints the instruction-slot cost of the given ORDER of rays:
//     wave cost = C_NODE x (node rounds) + C_TRI x (triangle rounds),  lane utilisation = useful lane-steps / (64 x rounds).
//   build: hipcc -O2 -std=c++17 -ffp-contract=off -o /tmp/mesh_walk_sim tools/mesh_walk_sim.cpp      (host code only)
//   run:   mesh_walk_sim faces15.f32 rays6.f32 [order.i32]     rays = object-space origin + direction, order = permutation (default identity)
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../mygpuraytracer_amd/csrc/pt_bvh.h"
using namespace ptd;
template <class T> static std::vector<T> slurp(const char *p) {
    FILE *f = fopen(p, "rb"); if (!f) { perror(p); exit(1); }
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<T> v(n / sizeof(T)); if (fread(v.data(), sizeof(T), v.size(), f) != v.size()) exit(1); fclose(f); return v;
}
int main(int argc, char **argv) {
    if (argc < 3) return 2;
    const double C_NODE = 190, C_TRI = 180;      // vector instructions per node step / per triangle test
    std::vector<float> faces = slurp<float>(argv[1]), rays = slurp<float>(argv[2]);
    const int nf = (int)(faces.size() / 15), nr = (int)(rays.size() / 6);
    std::vector<int32_t> order;
    if (argc > 3) order = slurp<int32_t>(argv[3]); else { order.resize(nr); for (int i = 0; i < nr; i++) order[i] = i; }
    std::vector<float> tri9((size_t)nf * 9);
    for (int j = 0; j < nf; j++) for (int k = 0; k < 3; k++) { tri9[j * 9 + k] = faces[j * 15 + k]; tri9[j * 9 + 3 + k] = faces[j * 15 + 5 + k] - faces[j * 15 + k]; tri9[j * 9 + 6 + k] = faces[j * 15 + 10 + k] - faces[j * 15 + k]; }
    BvhBuild bb; int depth = 0, wroot = -1, wneed = 0;
    const int root = bvhBuild(faces.data(), tri9.data(), 0, nf, bb, &depth, &wroot, &wneed);
    std::vector<int32_t> stack((size_t)wneed + 2);
    std::vector<std::vector<int>> tr(nr);
    long long nodes = 0, tris = 0, hits = 0;
    for (int i = 0; i < nr; i++) {
        const vec3 o = V3(rays[i * 6], rays[i * 6 + 1], rays[i * 6 + 2]), d = normalize(V3(rays[i * 6 + 3], rays[i * 6 + 4], rays[i * 6 + 5]));
        int face; float b0, b1; std::vector<int> t(4096, 0);
        bvhNearestWide(bb.nodes.data(), bb.wide.data(), bb.tris.data(), root, wroot, o, d, face, b0, b1, stack.data(), 1, nullptr, t.data());
        t.resize(t[0] + 1); tr[i].assign(t.begin() + 1, t.end());
        for (int s : tr[i]) { if (s == 0) nodes++; else tris += s; }
        hits += face >= 0;
    }
    double cost = 0, useful = 0, slots = 0; long long node_rounds = 0, tri_rounds = 0;
    for (int g = 0; g < (int)order.size(); g += 64) {
        const int n = std::min(64, (int)order.size() - g);
        size_t pos[64] = {0};
        for (;;) {
            bool any_node = true;
            while (any_node) {                                       // inner loop: lanes that hold an inner node step together
                any_node = false; int act = 0;
                for (int l = 0; l < n; l++) { const auto &t = tr[order[g + l]]; if (pos[l] < t.size() && t[pos[l]] == 0) { pos[l]++; act++; any_node = true; } }
                if (any_node) { node_rounds++; cost += C_NODE; useful += act * C_NODE; slots += 64 * C_NODE; }
            }
            int maxc = 0, sumc = 0;                                   // leaves: the triangle loop runs to the longest count
            for (int l = 0; l < n; l++) { const auto &t = tr[order[g + l]]; if (pos[l] < t.size()) { maxc = std::max(maxc, t[pos[l]]); sumc += t[pos[l]]; pos[l]++; } }
            if (!maxc) break;
            tri_rounds += maxc; cost += C_TRI * maxc; useful += C_TRI * sumc; slots += 64.0 * C_TRI * maxc;
        }
    }
    // ---- the same rays with a wave that REFILLS its lanes: a wave owns a batch of `batch` consecutive rays; a lane whose walk has
    // ended waits until at least `thresh` lanes are idle (or nothing else is left to do), then the idle lanes take the batch's next rays
    // together (C_SETUP wave-instructions: load, object-space transform, slab terms, root box).  One node step or ONE triangle per
    // round: node rounds run while at least `nmin` lanes hold a node or no lane holds a triangle.  Environment: SIM_BATCH,
    // SIM_THRESH, SIM_NMIN, SIM_SETUP, SIM_OVERHEAD (wave-instructions of scheduling per round).
    if (getenv("SIM_BATCH")) {
        const int batch = atoi(getenv("SIM_BATCH")), thresh = getenv("SIM_THRESH") ? atoi(getenv("SIM_THRESH")) : 16, nmin = getenv("SIM_NMIN") ? atoi(getenv("SIM_NMIN")) : 24;
        const double C_SETUP = getenv("SIM_SETUP") ? atof(getenv("SIM_SETUP")) : 220, C_OVER = getenv("SIM_OVERHEAD") ? atof(getenv("SIM_OVERHEAD")) : 25;
        double c2 = 0, useful2 = 0, slots2 = 0; long long waves2 = 0;
        for (int g = 0; g < (int)order.size(); g += batch) {
            const int n = std::min(batch, (int)order.size() - g);
            waves2++;
            int next = 0, ray[64]; size_t pos[64]; int left[64];      // left = triangles left of the leaf in hand
            for (int l = 0; l < 64; l++) { ray[l] = -1; pos[l] = 0; left[l] = 0; }
            for (;;) {
                int idle = 0, nodes_ = 0, tris_ = 0;
                for (int l = 0; l < 64; l++) {
                    if (ray[l] >= 0 && left[l] == 0 && pos[l] >= tr[ray[l]].size()) ray[l] = -1;
                    if (ray[l] < 0) idle++;
                    else if (left[l] > 0) tris_++;
                    else if (tr[ray[l]][pos[l]] == 0) nodes_++;
                    else { left[l] = tr[ray[l]][pos[l]]; pos[l]++; tris_++; }
                }
                if (next < n && (idle >= thresh || idle == 64 || nodes_ + tris_ == 0)) {
                    int took = 0;
                    for (int l = 0; l < 64 && next < n; l++) if (ray[l] < 0) { ray[l] = order[g + next++]; pos[l] = 0; left[l] = 0; took++; }
                    c2 += C_SETUP; useful2 += took * C_SETUP; slots2 += 64 * C_SETUP;
                    continue;
                }
                if (nodes_ + tris_ == 0) break;
                c2 += C_OVER; slots2 += 64 * C_OVER;
                if (nodes_ >= nmin || tris_ == 0) {
                    for (int l = 0; l < 64; l++) if (ray[l] >= 0 && left[l] == 0 && pos[l] < tr[ray[l]].size() && tr[ray[l]][pos[l]] == 0) pos[l]++;
                    c2 += C_NODE; useful2 += nodes_ * C_NODE; slots2 += 64 * C_NODE;
                } else {
                    for (int l = 0; l < 64; l++) if (ray[l] >= 0 && left[l] > 0) left[l]--;
                    c2 += C_TRI; useful2 += tris_ * C_TRI; slots2 += 64 * C_TRI;
                }
            }
        }
        printf("{\"refill\": {\"batch\": %d, \"thresh\": %d, \"nmin\": %d, \"instr_per_64_rays\": %.0f, \"lane_utilisation\": %.3f}}\n", batch, thresh, nmin,
               c2 / order.size() * 64, useful2 / slots2);
    }
    printf("{\"rays\": %d, \"hit\": %.3f, \"nodes_per_ray\": %.2f, \"tris_per_ray\": %.2f, \"waves\": %d, \"node_rounds_per_wave\": %.2f, \"tri_rounds_per_wave\": %.2f, "
           "\"instr_per_wave\": %.0f, \"lane_utilisation\": %.3f}\n", nr, (double)hits / nr, (double)nodes / nr, (double)tris / nr, (int)((order.size() + 63) / 64),
           (double)node_rounds / ((order.size() + 63) / 64), (double)tri_rounds / ((order.size() + 63) / 64), cost / ((order.size() + 63) / 64), useful / slots);
    return 0;
}
