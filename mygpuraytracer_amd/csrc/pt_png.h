// pt_png.h -- PNG textures for the scene loader (host only).
//
// The reference loads every map_* of an OBJ's .mtl with stbi_load(name, &w, &h, &n, 0) after
// stbi_set_flip_vertically_on_load(true) (src/scene.cpp:133-215).  PNG is lossless, so decoding it by the book gives the
// texels stb_image gives; what has to match is stb's conventions for the channel count and the odd formats:
//   colour type 0 / 4 (grey, grey+alpha)  -> 1 / 2 channels      2 / 6 (RGB, RGBA) -> 3 / 4 channels
//   colour type 3 (palette)               -> 3 channels, 4 when the file has a tRNS chunk
//   tRNS on grey / RGB (colour key)       -> one more channel, 0 where the pixel equals the key, else 255
//   16 bits per sample                    -> the high byte        1/2/4-bit grey -> scaled by 255 / 85 / 17
//   Adam7 interlace                       -> de-interlaced         CRCs and the zlib Adler-32 are not checked (stb does not)
// Rows come out bottom-up (the vertical flip).  tests/test_loader.py pins this against the reference's own loader
// (oracle/_ref) on the hand-written files of tests/pngcases.py.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace ptpng {

// ---- zlib inflate (RFC 1950/1951) ----------------------------------------------------------------------------------
struct BitReader {
    const uint8_t *p, *end;
    uint32_t buf = 0;
    int n = 0;
    bool ok = true;
    uint32_t bits(int k) {
        while (n < k) {
            if (p >= end) { ok = false; return 0; }
            buf |= (uint32_t)*p++ << n;
            n += 8;
        }
        uint32_t v = k ? (buf & ((1u << k) - 1u)) : 0;
        buf >>= k; n -= k;
        return v;
    }
};

struct Huffman {
    uint16_t count[16] = {0}, symbol[288] = {0};
    bool build(const uint8_t *lengths, int nsym) {
        memset(count, 0, sizeof count);
        for (int i = 0; i < nsym; i++) count[lengths[i]]++;
        if (count[0] == nsym) return true;                     // no codes at all (legal for an unused distance tree)
        int left = 1;
        for (int len = 1; len < 16; len++) { left <<= 1; left -= count[len]; if (left < 0) return false; }
        uint16_t offs[16];
        offs[1] = 0;
        for (int len = 1; len < 15; len++) offs[len + 1] = offs[len] + count[len];
        for (int i = 0; i < nsym; i++) if (lengths[i]) symbol[offs[lengths[i]]++] = (uint16_t)i;
        return true;
    }
    int decode(BitReader &br) const {
        int code = 0, first = 0, index = 0;
        for (int len = 1; len < 16; len++) {
            code |= (int)br.bits(1);
            if (!br.ok) return -1;
            const int c = count[len];
            if (code - c < first) return symbol[index + (code - first)];
            index += c; first += c; first <<= 1; code <<= 1;
        }
        return -1;
    }
};

inline bool inflate_zlib(const std::vector<uint8_t> &z, std::vector<uint8_t> &out, size_t expect) {
    if (z.size() < 2) return false;
    if ((z[0] & 15) != 8 || ((z[0] << 8) | z[1]) % 31 != 0 || (z[1] & 32)) return false;      // deflate, header check, no preset dictionary
    BitReader br{z.data() + 2, z.data() + z.size()};
    static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint8_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    out.clear();
    out.reserve(expect);
    for (;;) {
        const int last = (int)br.bits(1), type = (int)br.bits(2);
        if (!br.ok) return false;
        if (type == 0) {
            br.buf = 0; br.n = 0;                              // to the byte boundary
            if (br.end - br.p < 4) return false;
            const unsigned len = br.p[0] | (br.p[1] << 8), nlen = br.p[2] | (br.p[3] << 8);
            br.p += 4;
            if ((len ^ 0xffffu) != nlen || (size_t)(br.end - br.p) < len) return false;
            out.insert(out.end(), br.p, br.p + len);
            br.p += len;
        } else if (type == 1 || type == 2) {
            Huffman lit, dist;
            uint8_t lengths[320];
            if (type == 1) {
                int i = 0;
                for (; i < 144; i++) lengths[i] = 8;
                for (; i < 256; i++) lengths[i] = 9;
                for (; i < 280; i++) lengths[i] = 7;
                for (; i < 288; i++) lengths[i] = 8;
                lit.build(lengths, 288);
                for (i = 0; i < 30; i++) lengths[i] = 5;
                dist.build(lengths, 30);
            } else {
                const int nlen = (int)br.bits(5) + 257, ndist = (int)br.bits(5) + 1, ncode = (int)br.bits(4) + 4;
                if (!br.ok || nlen > 286 || ndist > 30) return false;
                static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                uint8_t cl[19] = {0};
                for (int i = 0; i < ncode; i++) cl[order[i]] = (uint8_t)br.bits(3);
                Huffman lencode;
                if (!br.ok || !lencode.build(cl, 19)) return false;
                int idx = 0;
                while (idx < nlen + ndist) {
                    const int sym = lencode.decode(br);
                    if (sym < 0) return false;
                    if (sym < 16) lengths[idx++] = (uint8_t)sym;
                    else {
                        int rep, val = 0;
                        if (sym == 16) { if (idx == 0) return false; val = lengths[idx - 1]; rep = 3 + (int)br.bits(2); }
                        else if (sym == 17) rep = 3 + (int)br.bits(3);
                        else rep = 11 + (int)br.bits(7);
                        if (!br.ok || idx + rep > nlen + ndist) return false;
                        while (rep--) lengths[idx++] = (uint8_t)val;
                    }
                }
                if (lengths[256] == 0) return false;
                if (!lit.build(lengths, nlen) || !dist.build(lengths + nlen, ndist)) return false;
            }
            for (;;) {
                const int sym = lit.decode(br);
                if (sym < 0) return false;
                if (out.size() > expect + (1u << 20)) return false;      // far more data than the picture needs: not a texture
                if (sym < 256) out.push_back((uint8_t)sym);
                else if (sym == 256) break;
                else {
                    const int s = sym - 257;
                    if (s >= 29) return false;
                    const int len = lbase[s] + (int)br.bits(lext[s]);
                    const int ds = dist.decode(br);
                    if (ds < 0 || ds >= 30) return false;
                    const size_t d = dbase[ds] + br.bits(dext[ds]);
                    if (!br.ok || d > out.size()) return false;
                    const size_t from = out.size() - d;
                    for (int k = 0; k < len; k++) out.push_back(out[from + k]);
                }
            }
        } else {
            return false;
        }
        if (last) break;
    }
    return true;
}

// ---- PNG ----------------------------------------------------------------------------------------------------------
inline uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

inline int paeth(int a, int b, int c) {
    const int p = a + b - c, pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// one (sub)image of w x h pixels out of the filtered stream at `src`; returns the bytes consumed or 0 on error.
// out: samples as stored (bit depth `depth`, `nch` samples per pixel), one byte per 8-bit sample, two per 16-bit
inline size_t unfilter(const uint8_t *src, size_t avail, int w, int h, int nch, int depth, std::vector<uint8_t> &rows) {
    const size_t bpp = (size_t)(nch * depth + 7) / 8;                   // filter unit in bytes (>= 1)
    const size_t stride = ((size_t)w * nch * depth + 7) / 8;
    if (avail < (stride + 1) * (size_t)h) return 0;
    rows.assign(stride * h, 0);
    for (int y = 0; y < h; y++) {
        const uint8_t *in = src + (stride + 1) * (size_t)y;
        const int filter = in[0];
        if (filter > 4) return 0;
        in++;
        uint8_t *cur = &rows[stride * y];
        const uint8_t *up = y ? cur - stride : nullptr;
        for (size_t x = 0; x < stride; x++) {
            const int a = x >= bpp ? cur[x - bpp] : 0, b = up ? up[x] : 0, c = (up && x >= bpp) ? up[x - bpp] : 0;
            int v = in[x];
            if (filter == 1) v += a;
            else if (filter == 2) v += b;
            else if (filter == 3) v += (a + b) >> 1;
            else if (filter == 4) v += paeth(a, b, c);
            cur[x] = (uint8_t)v;
        }
    }
    return (stride + 1) * (size_t)h;
}

// Decodes a PNG the way stbi_load(..., req_comp = 0) does and flips it vertically.  false = "failed to load".
inline bool load_png_flipped(const std::string &data, int &W, int &H, int &CH, std::vector<uint8_t> &pixels) {
    static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    if (data.size() < 8 + 25 || memcmp(data.data(), sig, 8) != 0) return false;
    const uint8_t *p = (const uint8_t *)data.data() + 8, *end = (const uint8_t *)data.data() + data.size();
    int w = 0, h = 0, depth = 0, color = 0, interlace = 0;
    bool first = true, has_trans = false;
    uint8_t palette[256 * 4];
    int pal_len = 0;
    uint16_t key[3] = {0, 0, 0};
    std::vector<uint8_t> z;
    for (;;) {
        if (end - p < 8) return false;
        const uint32_t len = be32(p), type = be32(p + 4);
        p += 8;
        if ((size_t)(end - p) < (size_t)len) return false;
        if (first && type != 0x49484452u) return false;                 // IHDR must come first
        if (type == 0x49484452u) {
            if (!first || len != 13) return false;
            first = false;
            w = (int)be32(p); h = (int)be32(p + 4); depth = p[8]; color = p[9]; interlace = p[12];
            if (w <= 0 || h <= 0 || w > (1 << 24) || h > (1 << 24)) return false;
            if (depth != 1 && depth != 2 && depth != 4 && depth != 8 && depth != 16) return false;
            if (color > 6 || color == 1 || color == 5) return false;
            if (color == 3 && depth == 16) return false;
            if ((color == 2 || color == 4 || color == 6) && depth < 8) return false;
            if (p[10] != 0 || p[11] != 0 || interlace > 1) return false;
            if ((uint64_t)w * h > (1u << 27)) return false;                 // (11585 x 11585: bounds the working memory at ~1 GB)
            for (int i = 0; i < 256; i++) palette[i * 4 + 3] = 255;
        } else if (type == 0x504c5445u) {                               // PLTE
            if (len > 256 * 3 || len % 3) return false;
            pal_len = (int)len / 3;
            for (int i = 0; i < pal_len; i++) { palette[i * 4] = p[i * 3]; palette[i * 4 + 1] = p[i * 3 + 1]; palette[i * 4 + 2] = p[i * 3 + 2]; palette[i * 4 + 3] = 255; }
        } else if (type == 0x74524e53u) {                               // tRNS
            if (!z.empty()) return false;                               // after IDAT: stb rejects
            if (color == 3) {
                if (pal_len == 0 || (int)len > pal_len) return false;
                for (uint32_t i = 0; i < len; i++) palette[i * 4 + 3] = p[i];
                has_trans = true;
            } else {
                if (color & 4) return false;                            // tRNS with an alpha channel
                const int nc = (color & 2) ? 3 : 1;
                if (len != (uint32_t)nc * 2) return false;
                for (int k = 0; k < nc; k++) key[k] = (uint16_t)((p[k * 2] << 8) | p[k * 2 + 1]);
                has_trans = true;
            }
        } else if (type == 0x49444154u) {                               // IDAT
            if (color == 3 && pal_len == 0) return false;
            z.insert(z.end(), p, p + len);
        } else if (type == 0x49454e44u) {                               // IEND
            break;
        } else if (!(type & 0x20000000u)) {
            return false;                                               // unknown critical chunk
        }
        p += len;
        if (end - p < 4) return false;
        p += 4;                                                         // CRC, not checked
    }
    if (first || z.empty()) return false;
    const int nch = color == 3 ? 1 : ((color & 2) ? 3 : 1) + ((color & 4) ? 1 : 0);     // samples per pixel in the stream
    std::vector<uint8_t> raw;
    const size_t stride_full = ((size_t)w * nch * depth + 7) / 8;
    if (!inflate_zlib(z, raw, (stride_full + 1) * h + 64)) return false;
    // samples, one (8-bit) or two (16-bit) bytes each, for the whole picture
    const size_t bps = depth == 16 ? 2 : 1;
    std::vector<uint8_t> samples((size_t)w * h * nch * bps);
    auto expand_row = [&](const uint8_t *row, int pw, uint8_t *dst, size_t dst_step) {   // row of pw pixels -> samples at dst, dst_step bytes apart
        if (depth >= 8) {
            for (int x = 0; x < pw; x++) memcpy(dst + x * dst_step, row + (size_t)x * nch * bps, (size_t)nch * bps);
        } else {
            const int mask = (1 << depth) - 1;
            for (int x = 0; x < pw; x++) {                      // nch == 1 here (grey or palette)
                const int bit = x * depth;
                dst[x * dst_step] = (uint8_t)((row[bit >> 3] >> (8 - depth - (bit & 7))) & mask);
            }
        }
    };
    const size_t px = (size_t)nch * bps;
    if (!interlace) {
        std::vector<uint8_t> rows;
        if (!unfilter(raw.data(), raw.size(), w, h, nch, depth, rows)) return false;
        const size_t stride = ((size_t)w * nch * depth + 7) / 8;
        for (int y = 0; y < h; y++) expand_row(&rows[stride * y], w, &samples[(size_t)y * w * px], px);
    } else {
        static const int xo[7] = {0, 4, 0, 2, 0, 1, 0}, yo[7] = {0, 0, 4, 0, 2, 0, 1}, xs[7] = {8, 8, 4, 4, 2, 2, 1}, ys[7] = {8, 8, 8, 4, 4, 2, 2};
        size_t pos = 0;
        for (int pass = 0; pass < 7; pass++) {
            const int pw = (w - xo[pass] + xs[pass] - 1) / xs[pass], ph = (h - yo[pass] + ys[pass] - 1) / ys[pass];
            if (pw <= 0 || ph <= 0) continue;
            std::vector<uint8_t> rows;
            const size_t used = unfilter(raw.data() + pos, raw.size() - pos, pw, ph, nch, depth, rows);
            if (!used) return false;
            pos += used;
            const size_t stride = ((size_t)pw * nch * depth + 7) / 8;
            for (int y = 0; y < ph; y++)
                expand_row(&rows[stride * y], pw, &samples[((size_t)(yo[pass] + y * ys[pass]) * w + xo[pass]) * px], px * xs[pass]);
        }
    }
    // to 8-bit channels, stb's way
    const int out_n = color == 3 ? (has_trans ? 4 : 3) : nch + ((has_trans && !(color & 4)) ? 1 : 0);
    std::vector<uint8_t> img((size_t)w * h * out_n);
    static const uint8_t depth_scale[9] = {0, 0xff, 0x55, 0, 0x11, 0, 0, 0, 0x01};
    for (size_t i = 0; i < (size_t)w * h; i++) {
        const uint8_t *s = &samples[i * px];
        uint8_t *o = &img[i * out_n];
        if (color == 3) {
            const int idx = s[0];                                // (an index beyond PLTE reads the zero-initialised entry, as in stb)
            o[0] = palette[idx * 4]; o[1] = palette[idx * 4 + 1]; o[2] = palette[idx * 4 + 2];
            if (out_n == 4) o[3] = palette[idx * 4 + 3];
        } else {
            bool is_key = has_trans && !(color & 4);
            for (int c = 0; c < nch; c++) {
                const unsigned v16 = depth == 16 ? (unsigned)((s[c * 2] << 8) | s[c * 2 + 1]) : s[c];
                if (has_trans && !(color & 4) && c < ((color & 2) ? 3 : 1)) {
                    // stb compares 16-bit samples as stored, lower depths after scaling to 8 bits
                    const unsigned k = depth == 16 ? key[c] : (unsigned)((key[c] & 0xff) * depth_scale[depth]);
                    const unsigned v = depth == 16 ? v16 : (unsigned)(s[c] * depth_scale[depth]);
                    if (v != k) is_key = false;
                }
                o[c] = depth == 16 ? (uint8_t)(v16 >> 8) : (uint8_t)(s[c] * depth_scale[depth]);
            }
            if (has_trans && !(color & 4)) o[nch] = is_key ? 0 : 255;
        }
    }
    // vertical flip (stbi_set_flip_vertically_on_load)
    pixels.resize(img.size());
    const size_t rowb = (size_t)w * out_n;
    for (int y = 0; y < h; y++) memcpy(&pixels[(size_t)(h - 1 - y) * rowb], &img[(size_t)y * rowb], rowb);
    W = w; H = h; CH = out_n;
    return true;
}

}  // namespace ptpng
