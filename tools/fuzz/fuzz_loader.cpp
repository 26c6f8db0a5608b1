// fuzz_loader.cpp -- sanitizer run of the host-side file code (CPU build only): the scene / OBJ / MTL loader and the
// PPM / PNG / JPEG texture decoders on the repository's scenes and the tests' image fixtures, each also truncated at many
// lengths and with bytes flipped (deterministic LCG).  Built with -fsanitize=address,undefined by tools/fuzz/run.sh; a
// finding aborts the run.  What is checked is "no memory error, no undefined behaviour, no hang" -- results are the
// parity tests' business.
//   fuzz_loader DIR_WITH_IMAGE_FILES SCENE_DIR [mutations per file]
#include <dirent.h>
#include <sys/stat.h>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../mygpuraytracer_amd/csrc/scene_loader.cpp"

static std::string g_err;
extern "C" void ptx_internal_set_error(const char *msg) { g_err = msg ? msg : ""; }

static uint32_t lcg(uint32_t &s) { s = s * 1664525u + 1013904223u; return s >> 8; }

static std::vector<std::string> list_dir(const std::string &d) {
    std::vector<std::string> out;
    if (DIR *dir = opendir(d.c_str())) {
        while (dirent *e = readdir(dir)) if (e->d_name[0] != '.') out.push_back(d + "/" + e->d_name);
        closedir(dir);
    }
    return out;
}

static std::string slurp(const std::string &p) {
    std::string d;
    if (FILE *f = fopen(p.c_str(), "rb")) { char buf[65536]; size_t n; while ((n = fread(buf, 1, sizeof buf, f)) > 0) d.append(buf, n); fclose(f); }
    return d;
}

static std::string mutate(const std::string &d, uint32_t &seed) {
    std::string m = d;
    const uint32_t kind = lcg(seed) % 4;
    if (m.empty()) return m;
    if (kind == 0) m.resize(lcg(seed) % m.size());                                   // truncate
    else if (kind == 1) for (int k = 0, n = 1 + lcg(seed) % 8; k < n; k++) m[lcg(seed) % m.size()] ^= (char)(1u << (lcg(seed) % 8));
    else if (kind == 2) for (int k = 0, n = 1 + lcg(seed) % 4; k < n; k++) m[lcg(seed) % m.size()] = (char)lcg(seed);
    else { const size_t a = lcg(seed) % m.size(), n = lcg(seed) % 64; m.insert(a, m.substr(a, n)); }   // duplicate a stretch
    return m;
}

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s IMAGE_DIR SCENE_DIR [mutations]\n", argv[0]); return 2; }
    const int nmut = argc > 3 ? atoi(argv[3]) : 200;
    uint32_t seed = getenv("FUZZ_SEED") ? (uint32_t)strtoul(getenv("FUZZ_SEED"), nullptr, 10) : 12345u;
    long decoded = 0, refused = 0;
    // ---- decoders on memory buffers ---------------------------------------------------------------------------------
    for (const std::string &f : list_dir(argv[1])) {
        const std::string d = slurp(f);
        for (int k = 0; k <= nmut; k++) {
            const std::string m = k ? mutate(d, seed) : d;
            int w = 0, h = 0, ch = 0;
            std::vector<uint8_t> px;
            bool ok = ptpng::load_png_flipped(m, w, h, ch, px) || ptjpeg::load_jpeg_flipped(m, w, h, ch, px);
            if (ok) {
                if (w <= 0 || h <= 0 || ch <= 0 || px.size() != (size_t)w * h * ch) { fprintf(stderr, "%s: inconsistent result\n", f.c_str()); return 1; }
                volatile uint8_t sink = 0;
                for (uint8_t v : px) sink ^= v;                                       // touch every byte (ASan)
                decoded++;
            } else refused++;
        }
    }
    printf("decoders: %ld decoded, %ld refused\n", decoded, refused);
    // ---- scene / OBJ / MTL / PPM through the loader, from a scratch copy of the tree ----------------------------------
    const std::string scratch = std::string(argc > 4 ? argv[4] : "/tmp") + "/fuzz_scene";
    long loaded = 0, failed = 0;
    for (const std::string &f : list_dir(argv[2])) {
        if (f.size() < 4 || f.substr(f.size() - 4) != ".txt") continue;
        const std::string d = slurp(f);
        for (int k = 0; k <= nmut; k++) {
            const std::string m = k ? mutate(d, seed) : d;
            const std::string path = scratch + ".txt";
            if (FILE *o = fopen(path.c_str(), "wb")) { fwrite(m.data(), 1, m.size(), o); fclose(o); }
            ptx_scene *s = nullptr;
            const int rc = ptx_scene_load(path.c_str(), argv[2], &s);
            if (rc == PTX_OK && s) { loaded++; ptx_scene_free(s); } else failed++;
        }
    }
    printf("scenes: %ld loaded, %ld refused\n", loaded, failed);
    // ---- the files a scene pulls in (OBJ, MTL, PPM maps): a scratch copy of the tree with one file mutated at a time -----
    const std::string root = std::string(argv[2]) + "/..";
    const char *assets[] = {"models/cube.obj", "models/standin_ship.obj", "models/materials/cube.mtl", "models/materials/standin_ship.mtl",
                            "textures/standin_kd.ppm", "textures/standin_ks.ppm", "textures/standin_ke.ppm", "textures/standin_bump.ppm"};
    const char *users[] = {"cornellObj.txt", "cornellSpaceship.txt", "cornellObj.txt", "cornellSpaceship.txt",
                           "cornellSpaceship.txt", "cornellSpaceship.txt", "cornellSpaceship.txt", "cornellSpaceship.txt"};
    const std::string tree = scratch + "_tree";
    for (const char *d : {"", "/scenes", "/models", "/models/materials", "/textures"}) mkdir((tree + d).c_str(), 0755);
    auto put = [&](const std::string &rel, const std::string &data) {
        if (FILE *o = fopen((tree + "/" + rel).c_str(), "wb")) { fwrite(data.data(), 1, data.size(), o); fclose(o); }
    };
    for (const char *a : assets) put(a, slurp(root + "/" + a));
    for (const char *u : {"cornellObj.txt", "cornellSpaceship.txt"}) put(std::string("scenes/") + u, slurp(std::string(argv[2]) + "/" + u));
    long aloaded = 0, afailed = 0;
    for (size_t i = 0; i < sizeof assets / sizeof *assets; i++) {
        const std::string orig = slurp(root + "/" + assets[i]);
        for (int k = 0; k < nmut; k++) {
            put(assets[i], mutate(orig, seed));
            ptx_scene *s = nullptr;
            const int rc = ptx_scene_load((tree + "/scenes/" + users[i]).c_str(), (tree + "/scenes").c_str(), &s);
            if (rc == PTX_OK && s) { aloaded++; ptx_scene_free(s); } else afailed++;
        }
        put(assets[i], orig);
    }
    printf("assets: %ld loaded, %ld refused\n", aloaded, afailed);
    return 0;
}
