// pt_image.h -- image output with the semantics of the reference's saveImage + image::savePNG
// (src/main.cpp:81-102, src/image.cpp:22-39), without stb_image_write: a self-contained PNG encoder (zlib stream
// of stored blocks -- valid PNG, no compression) and a PFM writer for the raw fp32 frame.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <string>
#include <cstring>
#include <vector>

namespace ptimg {

inline uint32_t crc32(const uint8_t *d, size_t n, uint32_t crc = 0) {
    static uint32_t table[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; i++) { uint32_t c = i; for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1; table[i] = c; }
        init = true;
    }
    crc = ~crc;
    for (size_t i = 0; i < n; i++) crc = table[(crc ^ d[i]) & 0xff] ^ (crc >> 8);
    return ~crc;
}

inline void put32(std::vector<uint8_t> &v, uint32_t x) { v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x); }

inline void chunk(std::vector<uint8_t> &png, const char type[4], const std::vector<uint8_t> &data) {
    put32(png, (uint32_t)data.size());
    std::vector<uint8_t> td(type, type + 4);
    td.insert(td.end(), data.begin(), data.end());
    png.insert(png.end(), td.begin(), td.end());
    put32(png, crc32(td.data(), td.size()));
}

// rgb8: h rows of w*3 bytes, top row first
inline bool write_png_rgb8(const std::string &path, int w, int h, const uint8_t *rgb8) {
    std::vector<uint8_t> raw;
    raw.reserve((size_t)h * (w * 3 + 1));
    for (int y = 0; y < h; y++) { raw.push_back(0); raw.insert(raw.end(), rgb8 + (size_t)y * w * 3, rgb8 + (size_t)(y + 1) * w * 3); }
    std::vector<uint8_t> z = {0x78, 0x01};
    uint32_t a = 1, b = 0;
    for (uint8_t c : raw) { a = (a + c) % 65521; b = (b + a) % 65521; }
    for (size_t off = 0; off < raw.size() || off == 0; off += 65535) {
        size_t n = raw.size() - off < 65535 ? raw.size() - off : 65535;
        z.push_back(off + n >= raw.size() ? 1 : 0);
        z.push_back(n & 0xff); z.push_back(n >> 8); z.push_back(~n & 0xff); z.push_back((~n >> 8) & 0xff);
        z.insert(z.end(), raw.begin() + off, raw.begin() + off + n);
        if (raw.empty()) break;
    }
    put32(z, (b << 16) | a);
    std::vector<uint8_t> png = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    std::vector<uint8_t> ihdr;
    put32(ihdr, (uint32_t)w); put32(ihdr, (uint32_t)h);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    chunk(png, "IHDR", ihdr);
    chunk(png, "IDAT", z);
    chunk(png, "IEND", {});
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return false;
    bool ok = fwrite(png.data(), 1, png.size(), f) == png.size();
    fclose(f);
    return ok;
}

// saveImage (src/main.cpp:81-92) + savePNG (src/image.cpp:22-33): pixel (x, y) of the accumulation buffer, divided by
// the sample count, goes to column W-1-x; clamp to [0,1], times 255, truncate.
inline void to_rgb8_mirrored(int w, int h, const float *sum_rgb, float samples, std::vector<uint8_t> &out) {
    out.resize((size_t)w * h * 3);
    for (int x = 0; x < w; x++)
        for (int y = 0; y < h; y++) {
            const float *p = sum_rgb + ((size_t)x + (size_t)y * w) * 3;
            uint8_t *o = &out[((size_t)(w - 1 - x) + (size_t)y * w) * 3];
            for (int k = 0; k < 3; k++) {
                float v = p[k] / samples;
                v = v < 0.f ? 0.f : (v > 1.f ? 1.f : v);
                o[k] = (uint8_t)(v * 255.f);
            }
        }
}

// raw fp32 frame (PFM, bottom row first as the format demands), mean radiance = sum / samples, not mirrored
inline bool write_pfm(const std::string &path, int w, int h, const float *sum_rgb, float samples) {
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return false;
    fprintf(f, "PF\n%d %d\n-1.0\n", w, h);
    std::vector<float> row((size_t)w * 3);
    for (int y = h - 1; y >= 0; y--) {
        for (int i = 0; i < w * 3; i++) row[i] = sum_rgb[(size_t)y * w * 3 + i] / samples;
        fwrite(row.data(), sizeof(float), row.size(), f);
    }
    fclose(f);
    return true;
}

// image::saveHDR (src/image.cpp:41-45; the call at src/main.cpp:101 is commented out there, --hdr enables it here):
// Radiance RGBE file of the same float pixels savePNG gets, i.e. mean radiance, x-mirrored by saveImage.  The bytes are
// those of the stb_image_write the reference vendors (external/include/stb_image_write.h:250-390): shared exponent of
// the largest channel via frexp, mantissas truncated; scanlines of width 8..32767 are stored channel-planar with the
// "2 2 hi lo" marker and per-channel RLE where a run is >= 3 equal bytes (<= 127 per packet) and literals go out in
// packets of <= 128; other widths are flat RGBE.
inline void rgbe_of(const float rgb[3], uint8_t out[4]) {
    float m = rgb[1] > rgb[2] ? rgb[1] : rgb[2];
    m = rgb[0] > m ? rgb[0] : m;
    if (m < 1e-32) { out[0] = out[1] = out[2] = out[3] = 0; return; }      // compared in double, as the vendored code does
    int e;
    const float scale = (float)frexp(m, &e) * 256.0f / m;
    for (int k = 0; k < 3; k++) out[k] = (uint8_t)(rgb[k] * scale);
    out[3] = (uint8_t)(e + 128);
}

inline void hdr_scanline(std::vector<uint8_t> &o, int w, const float *rgb) {
    uint8_t q[4];
    if (w < 8 || w >= 32768) {
        for (int x = 0; x < w; x++) { rgbe_of(rgb + 3 * x, q); o.insert(o.end(), q, q + 4); }
        return;
    }
    std::vector<uint8_t> plane((size_t)w * 4);
    for (int x = 0; x < w; x++) { rgbe_of(rgb + 3 * x, q); for (int c = 0; c < 4; c++) plane[(size_t)c * w + x] = q[c]; }
    o.push_back(2); o.push_back(2); o.push_back((uint8_t)((w >> 8) & 0xff)); o.push_back((uint8_t)(w & 0xff));
    for (int c = 0; c < 4; c++) {
        const uint8_t *p = &plane[(size_t)c * w];
        int x = 0;
        while (x < w) {
            int r = x;                                   // first position where three equal bytes start, else the end
            while (r + 2 < w && !(p[r] == p[r + 1] && p[r] == p[r + 2])) r++;
            const bool run = r + 2 < w;
            if (!run) r = w;
            while (x < r) {                              // literals
                const int len = r - x > 128 ? 128 : r - x;
                o.push_back((uint8_t)len); o.insert(o.end(), p + x, p + x + len);
                x += len;
            }
            if (run) {
                while (r < w && p[r] == p[x]) r++;
                while (x < r) {
                    const int len = r - x > 127 ? 127 : r - x;
                    o.push_back((uint8_t)(len + 128)); o.push_back(p[x]);
                    x += len;
                }
            }
        }
    }
}

// pixels: W*H*3 floats, row-major, top row first, exactly what is to be stored (no division, no mirroring here)
inline bool write_hdr(const std::string &path, int w, int h, const float *pixels) {
    if (w <= 0 || h <= 0 || !pixels) return false;
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return false;
    fprintf(f, "#?RADIANCE\n# Written by stb_image_write.h\nFORMAT=32-bit_rle_rgbe\n");
    fprintf(f, "EXPOSURE=          1.0000000000000\n\n-Y %d +X %d\n", h, w);
    std::vector<uint8_t> o;
    bool ok = true;
    for (int y = 0; y < h && ok; y++) {
        o.clear();
        hdr_scanline(o, w, pixels + (size_t)y * w * 3);
        ok = fwrite(o.data(), 1, o.size(), f) == o.size();
    }
    return fclose(f) == 0 && ok;
}

// mean radiance, x-mirrored: the float image saveImage hands to savePNG / saveHDR (src/main.cpp:86-92)
inline void to_mean_mirrored(int w, int h, const float *sum_rgb, float samples, std::vector<float> &out) {
    out.resize((size_t)w * h * 3);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            for (int k = 0; k < 3; k++)
                out[((size_t)(w - 1 - x) + (size_t)y * w) * 3 + k] = sum_rgb[((size_t)x + (size_t)y * w) * 3 + k] / samples;
}

// Checkpoint of a render in progress: the accumulation buffer (sum over iterations, as ptx_read_image returns it) and
// how many iterations are in it.  "PTXCKPT1", int32 W, int32 H, int64 iterations, W*H*3 float32 (little endian).
// Resuming with ptx_write_image + ptx_render(iterations + 1, ...) continues bit-identically: nothing else carries over
// between iterations (the RNG is seeded from the iteration number, src/pathtrace.cu:62-66).
inline bool write_checkpoint(const std::string &path, int w, int h, long long iterations, const float *sum_rgb) {
    const std::string tmp = path + ".tmp";
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) return false;
    const int32_t wh[2] = {w, h};
    const int64_t it = iterations;
    bool ok = fwrite("PTXCKPT1", 1, 8, f) == 8 && fwrite(wh, 4, 2, f) == 2 && fwrite(&it, 8, 1, f) == 1 &&
              fwrite(sum_rgb, sizeof(float), (size_t)w * h * 3, f) == (size_t)w * h * 3;
    ok = (fclose(f) == 0) && ok;
    if (ok) ok = rename(tmp.c_str(), path.c_str()) == 0;      // never leaves a half-written checkpoint under `path`
    return ok;
}
inline bool read_checkpoint(const std::string &path, int w, int h, long long &iterations, float *sum_rgb, std::string &why) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) { why = "cannot open " + path; return false; }
    char magic[8];
    int32_t wh[2];
    int64_t it = 0;
    bool ok = fread(magic, 1, 8, f) == 8 && memcmp(magic, "PTXCKPT1", 8) == 0 && fread(wh, 4, 2, f) == 2 && fread(&it, 8, 1, f) == 1;
    if (!ok) why = path + " is not a checkpoint file";
    else if (wh[0] != w || wh[1] != h) { ok = false; why = path + " has another resolution"; }
    else if (fread(sum_rgb, sizeof(float), (size_t)w * h * 3, f) != (size_t)w * h * 3) { ok = false; why = path + " is truncated"; }
    fclose(f);
    iterations = it;
    return ok;
}

}  // namespace ptimg
