// scene_loader.cpp -- scenes/*.txt loader behind ptx_scene_* (include/mi355x_pathtracer.h).
//
// Reads the reference's scene text format (grammar: SURVEY.md appendix B) and produces the same POD the
// reference's Scene class holds after construction (src/scene.cpp:10-420, src/utilities.cpp:65-112):
// geoms with transform / inverseTransform / invTranspose, materials, per-geom triangle lists, camera.
// Differences from the reference, all on inputs where the reference itself is undefined:
//   * malformed input returns PTX_ERR_INVALID with a message instead of indexing an empty token vector,
//     exit()ing, or silently dropping an object whose id is out of sequence;
//   * every geom owns four texture slots (an OBJ whose .mtl names no map gets empty ones), instead of the
//     reference's four scene-wide vectors that fall out of step with geoms (src/scene.cpp:138-218);
//   * OBJ/MTL parsing is a small own parser (the reference vendors tinyobjloader 2.0): v / vt / f records,
//     triangles and quads (quads split along the shorter diagonal, ties -> [0,1,3],[1,2,3], as
//     tiny_obj_loader.h:1511-1553 does); polygons with more corners by its ear clipping (triangulate_ngon below);
//   * textures are read from binary PPM (P6), PNG (pt_png.h) and JPEG (pt_jpeg.h) files, flipped vertically like
//     stbi_set_flip_vertically_on_load (src/scene.cpp:133); any other format counts as "failed to load" => empty
//     texture, the reference's own fallback.
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/mi355x_pathtracer.h"
#include "pt_hostmath.h"
#include "pt_png.h"
#include "pt_jpeg.h"

// the error string lives in pt_engine.hip next to ptx_last_error()
extern "C" void ptx_internal_set_error(const char *msg);

struct ptx_scene {
    std::vector<ptx_geom> geoms;
    std::vector<ptx_material> materials;
    std::vector<std::vector<float>> faces;              // per geom, 15 floats per triangle
    std::vector<std::vector<uint8_t>> texdata;          // 4 per geom: kd, ks, bump, ke (struct Geom order)
    ptx_camera camera{};
    float fovy = 0.f;
    int iterations = 0, traceDepth = 0;
    std::string imageName;
    void fix_pointers() {
        for (size_t i = 0; i < geoms.size(); i++) {
            geoms[i].faces = faces[i].empty() ? nullptr : faces[i].data();
            ptx_texture *tx[4] = {&geoms[i].kd, &geoms[i].ks, &geoms[i].bump, &geoms[i].ke};
            for (int k = 0; k < 4; k++) tx[k]->image = texdata[i * 4 + k].empty() ? nullptr : texdata[i * 4 + k].data();
        }
    }
};

namespace {

int fail(int code, const std::string &msg) { ptx_internal_set_error(msg.c_str()); return code; }

// utilityCore::safeGetline (src/utilities.cpp:82-112): \n, \r\n, \r; last line may lack a newline.
// Returns false once nothing more can be read (the reference's eofbit + empty line).
struct LineReader {
    std::string data;
    size_t pos = 0;
    bool good = true;
    bool getline(std::string &t) {
        t.clear();
        if (pos >= data.size()) { good = false; return false; }
        while (pos < data.size()) {
            char c = data[pos++];
            if (c == '\n') return true;
            if (c == '\r') { if (pos < data.size() && data[pos] == '\n') pos++; return true; }
            t += c;
        }
        return true;
    }
};

// utilityCore::tokenizeString (src/utilities.cpp:74-80)
std::vector<std::string> tokenize(const std::string &s) {
    std::istringstream ss(s);
    std::vector<std::string> out;
    std::string tok;
    while (ss >> tok) out.push_back(tok);
    return out;
}

float to_f(const std::string &s) { return (float)atof(s.c_str()); }

bool read_file(const std::string &path, std::string &out) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::ostringstream ss;
    ss << f.rdbuf();
    out = ss.str();
    return true;
}

std::string join_path(const std::string &base, const std::string &rel) {
    if (!rel.empty() && rel[0] == '/') return rel;
    if (base.empty()) return rel;
    return base + "/" + rel;
}

std::string normalise_separators(std::string p) {      // "..\\textures\\a b.jpg" (Windows .mtl) -> "../textures/a b.jpg"
    std::string out;
    for (size_t i = 0; i < p.size(); i++) {
        if (p[i] == '\\') { if (out.empty() || out.back() != '/') out += '/'; }
        else out += p[i];
    }
    return out;
}

// binary PPM reader; rows flipped so that row 0 is the bottom of the picture (stbi flip-vertically)
bool load_ppm_flipped(const std::string &path, int &w, int &h, int &ch, std::vector<uint8_t> &pixels) {
    std::string d;
    if (!read_file(path, d)) return false;
    size_t p = 0;
    auto token = [&](std::string &tok) {
        tok.clear();
        while (p < d.size()) {
            if (d[p] == '#') { while (p < d.size() && d[p] != '\n') p++; }
            else if (isspace((unsigned char)d[p])) p++;
            else break;
        }
        while (p < d.size() && !isspace((unsigned char)d[p])) tok += d[p++];
        return !tok.empty();
    };
    std::string magic, sw, sh, smax;
    if (!token(magic) || magic != "P6" || !token(sw) || !token(sh) || !token(smax)) return false;
    w = atoi(sw.c_str()); h = atoi(sh.c_str());
    if (w <= 0 || h <= 0 || atoi(smax.c_str()) != 255) return false;
    p++;                                      // single whitespace after maxval
    size_t need = (size_t)w * h * 3;
    if (d.size() < p + need) return false;
    ch = 3;
    pixels.resize(need);
    for (int y = 0; y < h; y++) memcpy(&pixels[(size_t)(h - 1 - y) * w * 3], &d[p + (size_t)y * w * 3], (size_t)w * 3);
    return true;
}

// a texture file as stbi_load(name, &w, &h, &n, 0) with the vertical flip would deliver it: binary PPM, PNG (pt_png.h) or
// JPEG (pt_jpeg.h); anything else is "failed to load" => empty texture, the reference's own fallback (src/scene.cpp:152-156)
bool load_texture_flipped(const std::string &path, int &w, int &h, int &ch, std::vector<uint8_t> &pixels) {
    if (load_ppm_flipped(path, w, h, ch, pixels)) return true;
    std::string d;
    if (!read_file(path, d)) return false;
    if (ptpng::load_png_flipped(d, w, h, ch, pixels)) return true;
    return ptjpeg::load_jpeg_flipped(d, w, h, ch, pixels);
}

struct MtlInfo {
    float kd[3] = {0, 0, 0}, ks[3] = {0, 0, 0}, ke[3] = {0, 0, 0};
    float ior = 1.0f;
    std::string map_kd, map_ks, map_ke, map_bump;
    bool found = false;
};

// first material of the .mtl (the reference uses objMaterials[0] only, src/scene.cpp:68,134,221-231)
bool parse_mtl_first(const std::string &path, MtlInfo &out) {
    std::string d;
    if (!read_file(path, d)) return false;
    LineReader lr; lr.data = d;
    std::string line;
    int nmat = 0;
    while (lr.getline(line)) {
        std::vector<std::string> t = tokenize(line);
        if (t.empty() || t[0][0] == '#') continue;
        if (t[0] == "newmtl") { nmat++; if (nmat > 1) break; out.found = true; continue; }
        if (nmat != 1) continue;
        auto rest = [&]() {                     // texture name = rest of the line, verbatim (tiny_obj_loader.h:1316)
            size_t k = line.find(t[0]);
            std::string r = line.substr(k + t[0].size());
            size_t a = r.find_first_not_of(" \t");
            if (a == std::string::npos) return std::string();
            size_t b = r.find_last_not_of(" \t\r");
            return r.substr(a, b - a + 1);
        };
        if (t[0] == "Kd" && t.size() >= 4) for (int k = 0; k < 3; k++) out.kd[k] = to_f(t[1 + k]);
        else if (t[0] == "Ks" && t.size() >= 4) for (int k = 0; k < 3; k++) out.ks[k] = to_f(t[1 + k]);
        else if (t[0] == "Ke" && t.size() >= 4) for (int k = 0; k < 3; k++) out.ke[k] = to_f(t[1 + k]);
        else if (t[0] == "Ni" && t.size() >= 2) out.ior = to_f(t[1]);
        else if (t[0] == "map_Kd") out.map_kd = rest();
        else if (t[0] == "map_Ks") out.map_ks = rest();
        else if (t[0] == "map_Ke") out.map_ke = rest();
        else if (t[0] == "map_Bump" || t[0] == "map_bump" || t[0] == "bump") out.map_bump = rest();
    }
    return out.found;
}

struct ObjIndex { int v = -1, vt = -1; };

bool parse_face_corner(const std::string &tok, int nv, int nvt, ObjIndex &out) {
    // v, v/vt, v//vn, v/vt/vn ; 1-based, negative = relative to the end
    const char *s = tok.c_str();
    char *end = nullptr;
    long v = strtol(s, &end, 10);
    if (end == s) return false;
    out.v = v > 0 ? (int)v - 1 : nv + (int)v;
    out.vt = -1;
    if (*end == '/') {
        const char *s2 = end + 1;
        if (*s2 != '/' && *s2 != 0) {
            long vt = strtol(s2, &end, 10);
            if (end != s2) out.vt = vt > 0 ? (int)vt - 1 : nvt + (int)vt;
        }
    }
    return out.v >= 0 && out.v < nv;
}

// point-in-polygon by crossing number, as tiny_obj_loader.h:1414-1426 evaluates it (float arithmetic, same expression)
bool pnpoly3(const float *vertx, const float *verty, float testx, float testy) {
    bool c = false;
    for (int i = 0, j = 2; i < 3; j = i++)
        if (((verty[i] > testy) != (verty[j] > testy)) &&
            (testx < (vertx[j] - vertx[i]) * (testy - verty[i]) / (verty[j] - verty[i]) + vertx[i]))
            c = !c;
    return c;
}

// Polygons with more than four corners, cut into triangles the way tinyobjloader's built-in ear clipping does
// (tiny_obj_loader.h:1566-1852): project on the two axes chosen from the first non-degenerate corner, walk a candidate
// corner around the polygon, emit it as an ear unless it turns the wrong way ("cross * area < 0", where area is the
// quirky 0.5 * (x0 * y1 - y0 * x1) of the candidate's first two vertices) or another vertex lies inside it, remove its
// middle vertex; give up after a full round without progress, and emit what is left if it is a triangle.
template <class PushTri>
void triangulate_ngon(const std::vector<ObjIndex> &face, const std::vector<float> &v, PushTri push_tri) {
    size_t npolys = face.size();
    size_t axes[2] = {1, 2};
    for (size_t k = 0; k < npolys; ++k) {
        const float *a = &v[(size_t)face[(k + 0) % npolys].v * 3], *b = &v[(size_t)face[(k + 1) % npolys].v * 3],
                    *c = &v[(size_t)face[(k + 2) % npolys].v * 3];
        const float e0x = b[0] - a[0], e0y = b[1] - a[1], e0z = b[2] - a[2];
        const float e1x = c[0] - b[0], e1y = c[1] - b[1], e1z = c[2] - b[2];
        const float cx = fabsf(e0y * e1z - e0z * e1y), cy = fabsf(e0z * e1x - e0x * e1z), cz = fabsf(e0x * e1y - e0y * e1x);
        const float epsilon = 1.1920928955078125e-07f;
        if (cx > epsilon || cy > epsilon || cz > epsilon) {
            if (!(cx > cy && cx > cz)) {
                axes[0] = 0;
                if (cz > cx && cz > cy) axes[1] = 1;
            }
            break;
        }
    }
    std::vector<ObjIndex> rem = face;
    size_t guess_vert = 0;
    size_t remainingIterations = face.size(), previousRemainingVertices = rem.size();
    while (rem.size() > 3 && remainingIterations > 0) {
        npolys = rem.size();
        if (guess_vert >= npolys) guess_vert -= npolys;
        if (previousRemainingVertices != npolys) { previousRemainingVertices = npolys; remainingIterations = npolys; }
        else remainingIterations--;
        ObjIndex ind[3];
        float vx[3], vy[3];
        for (size_t k = 0; k < 3; k++) {
            ind[k] = rem[(guess_vert + k) % npolys];
            vx[k] = v[(size_t)ind[k].v * 3 + axes[0]];
            vy[k] = v[(size_t)ind[k].v * 3 + axes[1]];
        }
        const float e0x = vx[1] - vx[0], e0y = vy[1] - vy[0], e1x = vx[2] - vx[1], e1y = vy[2] - vy[1];
        const float cross = e0x * e1y - e0y * e1x;
        const float area = (vx[0] * vy[1] - vy[0] * vx[1]) * 0.5f;
        if (cross * area < 0.0f) { guess_vert += 1; continue; }
        bool overlap = false;
        for (size_t otherVert = 3; otherVert < npolys; ++otherVert) {
            const size_t idx = (guess_vert + otherVert) % npolys;
            const float tx = v[(size_t)rem[idx].v * 3 + axes[0]], ty = v[(size_t)rem[idx].v * 3 + axes[1]];
            if (pnpoly3(vx, vy, tx, ty)) { overlap = true; break; }
        }
        if (overlap) { guess_vert += 1; continue; }
        push_tri(ind);
        size_t removed = (guess_vert + 1) % npolys;
        while (removed + 1 < npolys) { rem[removed] = rem[removed + 1]; removed += 1; }
        rem.pop_back();
    }
    if (rem.size() == 3) push_tri(rem.data());
}

// Scene::loadObj (src/scene.cpp:38-234): triangles in file order, Vertex{position, texcoord}
int load_obj(const std::string &base_dir, const std::string &objpath, ptx_scene &sc, ptx_geom &g, std::vector<float> &faces,
             std::vector<uint8_t> tex[4]) {
    std::string d;
    std::string full = join_path(base_dir, objpath);
    if (!read_file(full, d)) return fail(PTX_ERR_IO, "cannot read OBJ file " + full);
    std::vector<float> v, vt;
    std::string mtllib;
    LineReader lr; lr.data = d;
    std::string line;
    auto push_tri = [&](const ObjIndex c[3]) {
        for (int k = 0; k < 3; k++) {
            faces.push_back(v[c[k].v * 3 + 0]); faces.push_back(v[c[k].v * 3 + 1]); faces.push_back(v[c[k].v * 3 + 2]);
            if (c[k].vt >= 0 && (size_t)c[k].vt * 2 + 1 < vt.size()) { faces.push_back(vt[c[k].vt * 2]); faces.push_back(vt[c[k].vt * 2 + 1]); }
            else { faces.push_back(0.f); faces.push_back(0.f); }
        }
    };
    while (lr.getline(line)) {
        std::vector<std::string> t = tokenize(line);
        if (t.empty() || t[0][0] == '#') continue;
        if (t[0] == "v" && t.size() >= 4) { for (int k = 0; k < 3; k++) v.push_back(to_f(t[1 + k])); }
        else if (t[0] == "vt" && t.size() >= 3) { vt.push_back(to_f(t[1])); vt.push_back(to_f(t[2])); }
        else if (t[0] == "mtllib" && t.size() >= 2) { if (mtllib.empty()) mtllib = t[1]; }
        else if (t[0] == "f") {
            int nc = (int)t.size() - 1;
            if (nc < 3) continue;                                   // "Degenerated face", tinyobj skips it
            std::vector<ObjIndex> cv((size_t)nc);
            for (int k = 0; k < nc; k++)
                if (!parse_face_corner(t[1 + k], (int)v.size() / 3, (int)vt.size() / 2, cv[k]))
                    return fail(PTX_ERR_INVALID, "bad face record in " + full + ": " + line);
            const ObjIndex *c = cv.data();
            if (nc == 3) { push_tri(c); }
            else if (nc > 4) { triangulate_ngon(cv, v, push_tri); }
            else {
                const float *p0 = &v[c[0].v * 3], *p1 = &v[c[1].v * 3], *p2 = &v[c[2].v * 3], *p3 = &v[c[3].v * 3];
                float e02[3] = {p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2]};
                float e13[3] = {p3[0] - p1[0], p3[1] - p1[1], p3[2] - p1[2]};
                float sqr02 = e02[0] * e02[0] + e02[1] * e02[1] + e02[2] * e02[2];
                float sqr13 = e13[0] * e13[0] + e13[1] * e13[1] + e13[2] * e13[2];
                if (sqr02 < sqr13) {
                    ObjIndex a[3] = {c[0], c[1], c[2]}, b[3] = {c[0], c[2], c[3]};
                    push_tri(a); push_tri(b);
                } else {
                    ObjIndex a[3] = {c[0], c[1], c[3]}, b[3] = {c[1], c[2], c[3]};
                    push_tri(a); push_tri(b);
                }
            }
        }
    }
    g.faceSize = (int)(faces.size() / 15);
    // material of the object = first material of its .mtl, searched under ../models/materials (scene.cpp:41)
    MtlInfo mi;
    if (mtllib.empty() || !parse_mtl_first(join_path(base_dir, "../models/materials/" + mtllib), mi))
        return fail(PTX_ERR_INVALID, "OBJ " + full + " has no readable material library (the reference indexes objMaterials[0])");
    const std::string *names[4] = {&mi.map_kd, &mi.map_ks, &mi.map_bump, &mi.map_ke};    // struct Geom order
    ptx_texture *slots[4] = {&g.kd, &g.ks, &g.bump, &g.ke};
    for (int k = 0; k < 4; k++) {
        slots[k]->width = slots[k]->height = slots[k]->channels = 0; slots[k]->image = nullptr;
        if (names[k]->empty()) continue;
        int w, h, ch;
        if (load_texture_flipped(join_path(base_dir, normalise_separators(*names[k])), w, h, ch, tex[k])) {
            slots[k]->width = w; slots[k]->height = h; slots[k]->channels = ch;
        }
    }
    ptx_material m;
    memset(&m, 0, sizeof m);
    for (int k = 0; k < 3; k++) { m.specular_color[k] = mi.ks[k]; m.color[k] = mi.kd[k]; }
    m.specular_exponent = 0.0f;
    m.indexOfRefraction = mi.ior;
    m.emittance = mi.ke[0];
    m.hasReflective = 0.0f; m.hasRefractive = 0.0f;
    sc.materials.push_back(m);
    g.materialid = (int)sc.materials.size() - 1;
    return PTX_OK;
}

void camera_derive(ptx_scene &s) {         // src/scene.cpp:364-372
    ptx_camera &c = s.camera;
    float yscaled = tanf(s.fovy * (PTH_PI / 180));
    float xscaled = (yscaled * c.resolution[0]) / c.resolution[1];
    float fovx = (atanf(xscaled) * 180) / PTH_PI;
    c.fov[0] = fovx; c.fov[1] = s.fovy;
    c.pixelLength[0] = 2 * xscaled / (float)c.resolution[0];
    c.pixelLength[1] = 2 * yscaled / (float)c.resolution[1];
}

}  // namespace

extern "C" {

int ptx_scene_load(const char *scene_path, const char *base_dir_c, ptx_scene **out) {
    if (!scene_path || !out) return fail(PTX_ERR_INVALID, "null argument");
    *out = nullptr;
    std::string path = scene_path, base;
    if (base_dir_c) base = base_dir_c;
    else { size_t k = path.find_last_of('/'); base = k == std::string::npos ? "." : path.substr(0, k); }
    LineReader lr;
    if (!read_file(path, lr.data)) return fail(PTX_ERR_IO, "cannot read scene file " + path);
    ptx_scene *s = new ptx_scene;
    auto bail = [&](int code, const std::string &msg) { delete s; return fail(code, path + ": " + msg); };
    bool have_camera = false;
    std::string line;
    while (lr.getline(line)) {
        if (line.empty()) continue;
        std::vector<std::string> tokens = tokenize(line);
        if (tokens.empty()) continue;
        if (tokens[0] == "MATERIAL") {                                       // Scene::loadMaterial, scene.cpp:385-420
            if (tokens.size() < 2 || atoi(tokens[1].c_str()) != (int)s->materials.size())
                return bail(PTX_ERR_INVALID, "MATERIAL id does not match the number of materials so far");
            ptx_material m;
            memset(&m, 0, sizeof m);
            for (int i = 0; i < 7; i++) {
                if (!lr.getline(line)) return bail(PTX_ERR_INVALID, "MATERIAL block needs 7 property lines");
                std::vector<std::string> t = tokenize(line);
                if (t.empty()) return bail(PTX_ERR_INVALID, "empty line inside a MATERIAL block");
                auto need = [&](size_t n) { return t.size() >= n; };
                if (t[0] == "RGB" && need(4)) for (int k = 0; k < 3; k++) m.color[k] = to_f(t[1 + k]);
                else if (t[0] == "SPECEX" && need(2)) m.specular_exponent = to_f(t[1]);
                else if (t[0] == "SPECRGB" && need(4)) for (int k = 0; k < 3; k++) m.specular_color[k] = to_f(t[1 + k]);
                else if (t[0] == "REFL" && need(2)) m.hasReflective = to_f(t[1]);
                else if (t[0] == "REFR" && need(2)) m.hasRefractive = to_f(t[1]);
                else if (t[0] == "REFRIOR" && need(2)) m.indexOfRefraction = to_f(t[1]);
                else if (t[0] == "EMITTANCE" && need(2)) m.emittance = to_f(t[1]);
            }
            s->materials.push_back(m);
        } else if (tokens[0] == "CAMERA") {                                  // Scene::loadCamera, scene.cpp:324-383
            ptx_camera &c = s->camera;
            memset(&c, 0, sizeof c);
            for (int i = 0; i < 5; i++) {
                if (!lr.getline(line)) return bail(PTX_ERR_INVALID, "CAMERA block needs 5 property lines");
                std::vector<std::string> t = tokenize(line);
                if (t.empty()) return bail(PTX_ERR_INVALID, "empty line inside the CAMERA block");
                if (t[0] == "RES" && t.size() >= 3) { c.resolution[0] = atoi(t[1].c_str()); c.resolution[1] = atoi(t[2].c_str()); }
                else if (t[0] == "FOVY" && t.size() >= 2) s->fovy = to_f(t[1]);
                else if (t[0] == "ITERATIONS" && t.size() >= 2) s->iterations = atoi(t[1].c_str());
                else if (t[0] == "DEPTH" && t.size() >= 2) s->traceDepth = atoi(t[1].c_str());
                else if (t[0] == "FILE" && t.size() >= 2) s->imageName = t[1];
            }
            while (lr.getline(line) && !line.empty()) {
                std::vector<std::string> t = tokenize(line);
                if (t.size() < 4) continue;
                float *dst = t[0] == "EYE" ? c.position : t[0] == "LOOKAT" ? c.lookAt : t[0] == "UP" ? c.up : nullptr;
                if (dst) for (int k = 0; k < 3; k++) dst[k] = to_f(t[1 + k]);
            }
            if (c.resolution[0] <= 0 || c.resolution[1] <= 0) return bail(PTX_ERR_INVALID, "RES must be positive");
            camera_derive(*s);
            // camera.right is taken from the still-zero view (scene.cpp:370 runs before :374) => NaN, kept as is;
            // every caller overwrites it through the runCuda recompute before tracing.
            float zero[3] = {0.f, 0.f, 0.f}, cr[3];
            pth::cross3(zero, c.up, cr);
            pth::norm3(cr, c.right);
            float dv[3] = {c.lookAt[0] - c.position[0], c.lookAt[1] - c.position[1], c.lookAt[2] - c.position[2]};
            pth::norm3(dv, c.view);
            have_camera = true;
        } else if (tokens[0] == "OBJECT") {                                  // Scene::loadGeom, scene.cpp:236-322
            if (tokens.size() < 2 || atoi(tokens[1].c_str()) != (int)s->geoms.size())
                return bail(PTX_ERR_INVALID, "OBJECT id does not match the number of geoms so far");
            ptx_geom g;
            memset(&g, 0, sizeof g);
            std::string objfile;
            if (!lr.getline(line) || line.empty()) return bail(PTX_ERR_INVALID, "OBJECT needs a type line");
            if (line == "sphere") g.type = PTX_SPHERE;
            else if (line == "cube") g.type = PTX_CUBE;
            else if (line == "triangle") g.type = PTX_TRIANGLE;
            else if (line == "obj") {
                g.type = PTX_OBJ;
                if (!lr.getline(line) || line.empty()) return bail(PTX_ERR_INVALID, "obj needs a file name line");
                objfile = line;
            } else return bail(PTX_ERR_INVALID, "unknown object type '" + line + "'");
            if (g.type != PTX_OBJ) {
                if (!lr.getline(line) || line.empty()) return bail(PTX_ERR_INVALID, "OBJECT needs a material line");
                std::vector<std::string> t = tokenize(line);
                if (t.size() < 2) return bail(PTX_ERR_INVALID, "bad material line");
                g.materialid = atoi(t[1].c_str());
            } else g.materialid = -1;
            while (lr.getline(line) && !line.empty()) {
                std::vector<std::string> t = tokenize(line);
                if (t.size() < 4) continue;
                float *dst = t[0] == "TRANS" ? g.translation : t[0] == "ROTAT" ? g.rotation : t[0] == "SCALE" ? g.scale : nullptr;
                if (dst) for (int k = 0; k < 3; k++) dst[k] = to_f(t[1 + k]);
            }
            pth::Mat4 xf = pth::buildTransformationMatrix(g.translation, g.rotation, g.scale);
            pth::Mat4 inv = pth::inverse(xf), invT = pth::inverseTranspose(xf);
            memcpy(g.transform, xf.m, 64); memcpy(g.inverseTransform, inv.m, 64); memcpy(g.invTranspose, invT.m, 64);
            std::vector<float> faces;
            std::vector<uint8_t> tex[4];
            if (g.type == PTX_OBJ) {
                int rc = load_obj(base, objfile, *s, g, faces, tex);
                if (rc != PTX_OK) { delete s; return rc; }
            }
            s->geoms.push_back(g);
            s->faces.push_back(std::move(faces));
            for (int k = 0; k < 4; k++) s->texdata.push_back(std::move(tex[k]));
        }
    }
    if (!have_camera) return bail(PTX_ERR_INVALID, "scene has no CAMERA block");
    for (size_t i = 0; i < s->geoms.size(); i++)
        if (s->geoms[i].materialid < 0 || s->geoms[i].materialid >= (int)s->materials.size())
            return bail(PTX_ERR_INVALID, "OBJECT " + std::to_string(i) + " refers to a material that does not exist");
    s->fix_pointers();
    *out = s;
    return PTX_OK;
}

void ptx_scene_free(ptx_scene *s) { delete s; }
int ptx_scene_num_geoms(const ptx_scene *s) { return s ? (int)s->geoms.size() : 0; }
int ptx_scene_num_materials(const ptx_scene *s) { return s ? (int)s->materials.size() : 0; }
const ptx_geom *ptx_scene_geoms(const ptx_scene *s) { return s && !s->geoms.empty() ? s->geoms.data() : nullptr; }
const ptx_material *ptx_scene_materials(const ptx_scene *s) { return s && !s->materials.empty() ? s->materials.data() : nullptr; }
ptx_camera *ptx_scene_camera(ptx_scene *s) { return s ? &s->camera : nullptr; }
int ptx_scene_iterations(const ptx_scene *s) { return s ? s->iterations : 0; }
int ptx_scene_trace_depth(const ptx_scene *s) { return s ? s->traceDepth : 0; }
void ptx_scene_set_trace_depth(ptx_scene *s, int depth) { if (s) s->traceDepth = depth; }
const char *ptx_scene_image_name(const ptx_scene *s) { return s ? s->imageName.c_str() : ""; }

void ptx_scene_set_resolution(ptx_scene *s, int w, int h) {
    if (!s || w <= 0 || h <= 0) return;
    s->camera.resolution[0] = w; s->camera.resolution[1] = h;
    camera_derive(*s);
}

// ---- camera controls of src/main.cpp as plain functions (no window: a script of mouse events drives them) ----------
// main.cpp:56-70: phi (horizontal) and theta (vertical) of the loader's view vector, zoom = |position - lookAt|
void ptx_orbit_init(const ptx_scene *s, ptx_orbit *o) {
    if (!s || !o) return;
    const ptx_camera &cam = s->camera;
    float viewXZ[3] = {cam.view[0], 0.0f, cam.view[2]}, viewZY[3] = {0.0f, cam.view[1], cam.view[2]};
    float nxz[3], nzy[3];
    pth::norm3(viewXZ, nxz); pth::norm3(viewZY, nzy);
    o->phi = acosf(nxz[0] * 0.f + nxz[1] * 0.f + nxz[2] * -1.f);
    o->theta = acosf(nzy[0] * 0.f + nzy[1] * 1.f + nzy[2] * 0.f);
    float d[3] = {cam.position[0] - cam.lookAt[0], cam.position[1] - cam.lookAt[1], cam.position[2] - cam.lookAt[2]};
    o->zoom = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    for (int k = 0; k < 3; k++) o->og_look_at[k] = cam.lookAt[k];
}
// left button drag, main.cpp:184-189 (xpos - lastX and ypos - lastY are doubles there)
void ptx_orbit_left_drag(ptx_orbit *o, double dx, double dy, int width, int height) {
    if (!o) return;
    o->phi = (float)((double)o->phi - dx / width);
    o->theta = (float)((double)o->theta - dy / height);
    o->theta = fmaxf(0.001f, fminf(o->theta, 3.1415926535897932384626422832795028841971f));
}
// right button drag, main.cpp:190-194
void ptx_orbit_right_drag(ptx_orbit *o, double dy, int height) {
    if (!o) return;
    o->zoom = (float)((double)o->zoom + dy / height);
    o->zoom = fmaxf(0.1f, o->zoom);
}
// middle button drag, main.cpp:195-209: lookAt slides in the ground plane
void ptx_orbit_middle_drag(ptx_scene *s, double dx, double dy) {
    if (!s) return;
    ptx_camera &cam = s->camera;
    float f[3] = {cam.view[0], 0.0f, cam.view[2]}, r[3] = {cam.right[0], 0.0f, cam.right[2]}, fn[3], rn[3];
    pth::norm3(f, fn); pth::norm3(r, rn);
    const float fx = (float)dx, fy = (float)dy;
    for (int k = 0; k < 3; k++) cam.lookAt[k] -= fx * rn[k] * 0.01f;
    for (int k = 0; k < 3; k++) cam.lookAt[k] += fy * fn[k] * 0.01f;
}
// SPACE, main.cpp:166-171
void ptx_orbit_recenter(ptx_scene *s, const ptx_orbit *o) {
    if (!s || !o) return;
    for (int k = 0; k < 3; k++) s->camera.lookAt[k] = o->og_look_at[k];
}
// runCuda's recompute when camchanged, main.cpp:105-123
void ptx_orbit_apply(ptx_scene *s, const ptx_orbit *o) {
    if (!s || !o) return;
    ptx_camera &cam = s->camera;
    const float phi = o->phi, theta = o->theta, zoom = o->zoom;
    float cp[3];
    cp[0] = zoom * sinf(phi) * sinf(theta);
    cp[1] = zoom * cosf(theta);
    cp[2] = zoom * cosf(phi) * sinf(theta);
    float ncp[3];
    pth::norm3(cp, ncp);
    float v[3] = {-ncp[0], -ncp[1], -ncp[2]};
    float u[3] = {0, 1, 0}, r[3], up[3];
    pth::cross3(v, u, r);
    pth::cross3(r, v, up);
    for (int k = 0; k < 3; k++) { cam.view[k] = v[k]; cam.up[k] = up[k]; cam.right[k] = r[k]; cam.position[k] = cp[k] + cam.lookAt[k]; }
}

// what the first runCuda() of a session does: main.cpp:56-70 then :105-123
void ptx_scene_apply_runcuda_camera(ptx_scene *s) {
    if (!s) return;
    ptx_orbit o;
    ptx_orbit_init(s, &o);
    ptx_orbit_apply(s, &o);
}

}  // extern "C"
