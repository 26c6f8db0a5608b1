// pt_image.h -- image output with the semantics of the reference's saveImage + image::savePNG / saveHDR
// (src/main.cpp:81-102, src/image.cpp:22-45) without stb_image_write: own PNG and Radiance encoders that take the same
// decisions as the writer the reference vendors (external/include/stb_image_write.h v0.98), so the files are the same
// bytes (tests/golden/png_files.npz, hdr_files.npz were written by that code), plus a PFM writer for the raw fp32 frame
// and the checkpoint format.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdlib>
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

// ---- PNG as stbi_write_png writes it (stb_image_write.h:448-711) ------------------------------------------------------
// zlib stream: one fixed-Huffman block; matches are found through a hash of the next three bytes into 16384 buckets, each
// keeping its last 8..16 positions (at 16 the older 8 are dropped); the longest match wins, the later position on ties;
// a match is given up for a literal when the next byte starts a longer one; positions inside a match are not entered.
inline void zlib_like_stb(const std::vector<uint8_t> &in, std::vector<uint8_t> &out) {
    static const uint16_t len_base[] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258, 259};
    static const uint8_t len_extra[] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dist_base[] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577, 32768};
    static const uint8_t dist_extra[] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    constexpr int BUCKETS = 16384, KEEP = 8;
    const long long n = (long long)in.size();
    const uint8_t *d = in.data();
    uint32_t bitbuf = 0;
    int bitcount = 0;
    auto add = [&](uint32_t code, int bits) {                       // LSB first
        bitbuf |= code << bitcount;
        bitcount += bits;
        while (bitcount >= 8) { out.push_back((uint8_t)bitbuf); bitbuf >>= 8; bitcount -= 8; }
    };
    auto rev = [](int code, int bits) { int r = 0; while (bits--) { r = (r << 1) | (code & 1); code >>= 1; } return (uint32_t)r; };
    auto sym = [&](int v) {                                         // the fixed literal/length code of RFC 1951 3.2.6
        if (v <= 143) add(rev(0x30 + v, 8), 8);
        else if (v <= 255) add(rev(0x190 + v - 144, 9), 9);
        else if (v <= 279) add(rev(v - 256, 7), 7);
        else add(rev(0xc0 + v - 280, 8), 8);
    };
    auto hash3 = [&](long long i) {
        uint32_t h = d[i] + ((uint32_t)d[i + 1] << 8) + ((uint32_t)d[i + 2] << 16);
        h ^= h << 3; h += h >> 5; h ^= h << 4; h += h >> 17; h ^= h << 25; h += h >> 6;
        return (int)(h & (BUCKETS - 1));
    };
    auto run = [&](long long a, long long b, long long limit) {     // common prefix, at most 258
        long long k = 0;
        while (k < limit && k < 258 && d[a + k] == d[b + k]) k++;
        return (int)k;
    };
    out.push_back(0x78); out.push_back(0x5e);
    add(1, 1); add(1, 2);                                           // last block, fixed Huffman
    std::vector<std::vector<long long>> table(BUCKETS);
    long long i = 0;
    while (i < n - 3) {
        std::vector<long long> &bucket = table[hash3(i)];
        int best = 3;
        long long where = -1;
        for (long long pos : bucket)
            if (pos > i - 32768) {
                const int m = run(pos, i, n - i);
                if (m >= best) { best = m; where = pos; }
            }
        if ((int)bucket.size() == 2 * KEEP) bucket.erase(bucket.begin(), bucket.begin() + KEEP);
        bucket.push_back(i);
        if (where >= 0) {
            for (long long pos : table[hash3(i + 1)])
                if (pos > i - 32767 && run(pos, i + 1, n - i - 1) > best) { where = -1; break; }
        }
        if (where >= 0) {
            const int dist = (int)(i - where);
            int j = 0;
            while (best > len_base[j + 1] - 1) j++;
            sym(j + 257);
            if (len_extra[j]) add((uint32_t)(best - len_base[j]), len_extra[j]);
            j = 0;
            while (dist > dist_base[j + 1] - 1) j++;
            add(rev(j, 5), 5);
            if (dist_extra[j]) add((uint32_t)(dist - dist_base[j]), dist_extra[j]);
            i += best;
        } else {
            sym(d[i]);
            i++;
        }
    }
    for (; i < n; i++) sym(d[i]);
    sym(256);
    while (bitcount) add(0, 1);
    uint32_t s1 = 1, s2 = 0;                                        // Adler-32
    for (long long k = 0; k < n; k++) { s1 = (s1 + d[k]) % 65521; s2 = (s2 + s1) % 65521; }
    out.push_back((uint8_t)(s2 >> 8)); out.push_back((uint8_t)s2); out.push_back((uint8_t)(s1 >> 8)); out.push_back((uint8_t)s1);
}

inline int paeth_pick(int a, int b, int c) {
    const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    if (pa <= pb && pa <= pc) return a;
    return pb <= pc ? b : c;
}

// rgb8: h rows of w*3 bytes, top row first.  Per row the filter with the smallest sum of |signed byte| is chosen, the first
// such in the order none, sub, up, average, Paeth; the first row has no row above it, so its "up" is "none" and its
// average and Paeth use only the pixel to the left -- but are still labelled 2, 3, 4.
inline bool write_png_rgb8(const std::string &path, int w, int h, const uint8_t *rgb8) {
    const int n = 3, rowb = w * n;
    std::vector<uint8_t> raw((size_t)h * (rowb + 1));
    std::vector<int8_t> line((size_t)rowb);
    for (int y = 0; y < h; y++) {
        const uint8_t *z = rgb8 + (size_t)y * rowb, *up = z - rowb;
        auto filter_row = [&](int k) {
            for (int i = 0; i < rowb; i++) {
                const int left = i >= n ? z[i - n] : 0, above = y ? up[i] : 0, diag = (y && i >= n) ? up[i - n] : 0;
                int v;
                switch (k) {
                    case 0: v = z[i]; break;
                    case 1: v = z[i] - left; break;
                    case 2: v = z[i] - above; break;
                    case 3: v = z[i] - ((left + above) >> 1); break;
                    default: v = z[i] - paeth_pick(left, above, diag); break;
                }
                line[i] = (int8_t)v;
            }
        };
        int best = 0, bestval = 0x7fffffff;
        for (int k = 0; k < 5; k++) {
            filter_row(k);
            int est = 0;
            for (int i = 0; i < rowb; i++) est += abs((int)line[i]);
            if (est < bestval) { bestval = est; best = k; }
        }
        filter_row(best);
        raw[(size_t)y * (rowb + 1)] = (uint8_t)best;
        memcpy(&raw[(size_t)y * (rowb + 1) + 1], line.data(), (size_t)rowb);
    }
    std::vector<uint8_t> z;
    zlib_like_stb(raw, z);
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
