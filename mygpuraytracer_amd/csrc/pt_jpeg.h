// pt_jpeg.h -- JPEG textures for the scene loader (host only).
//
// The reference's own textured scene (scenes/cornellSpaceship.txt) names four 4096x4096 JPEG maps, loaded with
// stbi_load(name, &w, &h, &n, 0) (stb_image v2.27, vendored as src/stb_image.h; src/scene.cpp:133-215).  JPEG is lossy:
// which bytes come out depends on the decoder's arithmetic, not only on the standard.  This is a restatement of the
// arithmetic stb_image uses, so that the texels -- hence colours -- are the reference's:
//   * Huffman / progressive entropy decoding as in ITU T.81 (baseline, extended and progressive 8-bit frames, restart
//     intervals); baseline coefficients are dequantised as they are decoded, progressive ones at the end, both in
//     16-bit arithmetic (stb_image.h:2180-2228, 2234-2407, 3058-3085);
//   * an integer IDCT that produces stb_image's values (12-bit constants, +512 >> 10 after the column pass and
//     +65536 + (128 << 17) >> 17 after the row pass; stb_image.h:2392-2490) in an own formulation: the 8-point transform as
//     two 4x4 integer matrices (idct1d below), whose entries are the sums the factored network forms of its twelve rounded
//     constants -- exact integer arithmetic, hence the same bytes, pinned by tests/golden/jpeg_textures.npz;
//   * chroma upsampling by the "3:1" triangle filters, per row pair as the decoder walks down the picture
//     (stb_image.h:3397-3590, 3840-3880), nearest neighbour for factors other than 2;
//   * YCbCr -> RGB in 20-bit fixed point with the Cb term of green masked to 16 bits (stb_image.h:3596-3622); frames whose
//     component ids are 'R','G','B', or Adobe-tagged with transform 0 and no JFIF header, are taken as RGB; four-component
//     frames as CMYK / YCCK by the Adobe transform flag (stb_image.h:3802-3807, 3895-3925);
//   * n = 3 channels for 3- and 4-component files, 1 for greyscale; rows bottom-up (the vertical flip).
// Files stb_image rejects (arithmetic coding, 12-bit, lossless, no EOI, unknown markers) are failed loads here as well.
// Pinned against the reference's loader on the files of tests/jpegcases (tests/golden/jpeg_textures.npz).
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace ptjpeg {

static const uint8_t kDezigzag[64 + 15] = {
    0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
    63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};     // (runs past 63 land on 63, as in stb_image)

struct Huff {
    uint8_t size[257];          // code length of the k-th code (canonical order), 0-terminated
    uint16_t code[256];
    uint8_t value[256];
    int32_t delta[17];
    uint32_t maxcode[18];
    bool build(const int count[16]) {
        int k = 0;
        for (int i = 0; i < 16; i++)
            for (int j = 0; j < count[i]; j++) { if (k >= 256) return false; size[k++] = (uint8_t)(i + 1); }
        size[k] = 0;
        unsigned c = 0;
        k = 0;
        for (int j = 1; j <= 16; j++) {
            delta[j] = k - (int)c;
            if (size[k] == j) {
                while (size[k] == j) code[k++] = (uint16_t)(c++);
                if (c - 1 >= (1u << j)) return false;
            }
            maxcode[j] = c << (16 - j);
            c <<= 1;
        }
        maxcode[17] = 0xffffffffu;
        return true;
    }
};

struct Comp {
    int id = 0, h = 0, v = 0, tq = 0, hd = 0, ha = 0, dc_pred = 0;
    int x = 0, y = 0, w2 = 0, h2 = 0, coeff_w = 0;
    std::vector<uint8_t> data;      // w2 x h2 samples
    std::vector<int16_t> coeff;     // progressive: 64 per block
};

struct Decoder {
    const uint8_t *p, *end;
    // entropy bit reader: bits are taken from the top of `buf`
    uint32_t buf = 0;
    int nbits = 0;
    int marker = 0xff;              // marker met inside entropy data (0xff = none)
    bool nomore = false;
    Huff hdc[4], hac[4];
    uint16_t dequant[4][64];
    Comp comp[4];
    int img_x = 0, img_y = 0, img_n = 0, h_max = 1, v_max = 1, mcu_x = 0, mcu_y = 0;
    bool progressive = false, jfif = false;
    int app14 = -1, rgb = 0;
    int scan_n = 0, order[4] = {0, 0, 0, 0}, spec_start = 0, spec_end = 0, succ_high = 0, succ_low = 0, eob_run = 0;
    int restart_interval = 0, todo = 0;

    int get8() { return p < end ? *p++ : 0; }
    int get16() { int a = get8(); return (a << 8) | get8(); }
    bool eof() const { return p >= end; }

    void grow() {
        do {
            unsigned b = nomore ? 0u : (unsigned)get8();
            if (b == 0xff) {
                int c = get8();
                while (c == 0xff) c = get8();
                if (c != 0) { marker = c; nomore = true; return; }
            }
            // nbits can be negative once a marker has ended the data (a read took more bits than were left); b is 0 from
            // then on, so the out-of-range shift stb performs there contributes nothing: skip it instead (tools/fuzz, UBSan)
            if (nbits >= -7) buf |= b << (24 - nbits);
            nbits += 8;
        } while (nbits <= 24);
    }
    int huff(const Huff &h) {
        if (nbits < 16) grow();
        const unsigned temp = buf >> 16;
        int k;
        for (k = 1;; k++)
            if (temp < h.maxcode[k]) break;
        if (k == 17) { nbits -= 16; return -1; }
        if (k > nbits) return -1;
        const int c = (int)((buf >> (32 - k)) & ((1u << k) - 1u)) + h.delta[k];
        if (c < 0 || c > 255) return -1;
        nbits -= k;
        buf <<= k;
        return h.value[c];
    }
    int extend_receive(int n) {                     // T.81 RECEIVE + EXTEND
        if (nbits < n) grow();
        const int sgn = (int)(buf >> 31);
        unsigned k = (buf << n) | (buf >> (32 - n));
        const unsigned mask = (1u << n) - 1u;
        buf = k & ~mask;
        k &= mask;
        nbits -= n;
        const int bias = (int)(0xffffffffu << n) + 1;    // (-1 << n) + 1
        return (int)k + (bias & (sgn - 1));
    }
    int get_bits(int n) {
        if (nbits < n) grow();
        unsigned k = (buf << n) | (buf >> (32 - n));
        const unsigned mask = (1u << n) - 1u;
        buf = k & ~mask;
        k &= mask;
        nbits -= n;
        return (int)k;
    }
    int get_bit() {
        if (nbits < 1) grow();
        const unsigned k = buf;
        buf <<= 1;
        --nbits;
        return (int)(k & 0x80000000u);
    }
    void reset() {
        nbits = 0; buf = 0; nomore = false;
        for (Comp &c : comp) c.dc_pred = 0;
        marker = 0xff;
        todo = restart_interval ? restart_interval : 0x7fffffff;
        eob_run = 0;
    }

    bool decode_block(int16_t data[64], const Huff &dc, const Huff &ac, int b, const uint16_t *dq) {
        const int t = huff(dc);
        if (t < 0 || t > 15) return false;
        memset(data, 0, 64 * sizeof(int16_t));
        const int diff = t ? extend_receive(t) : 0;
        const int d = (int)((unsigned)comp[b].dc_pred + (unsigned)diff);
        comp[b].dc_pred = d;
        data[0] = (int16_t)((unsigned)d * (unsigned)dq[0]);
        int k = 1;
        do {
            const int rs = huff(ac);
            if (rs < 0) return false;
            const int s = rs & 15, r = rs >> 4;
            if (s == 0) {
                if (rs != 0xf0) break;
                k += 16;
            } else {
                k += r;
                const unsigned zig = kDezigzag[k++];
                data[zig] = (int16_t)(extend_receive(s) * dq[zig]);
            }
        } while (k < 64);
        return true;
    }
    bool decode_prog_dc(int16_t data[64], const Huff &dc, int b) {
        if (spec_end != 0) return false;
        if (succ_high == 0) {
            memset(data, 0, 64 * sizeof(int16_t));
            const int t = huff(dc);
            if (t < 0 || t > 15) return false;
            const int diff = t ? extend_receive(t) : 0;
            const int d = (int)((unsigned)comp[b].dc_pred + (unsigned)diff);
            comp[b].dc_pred = d;
            data[0] = (int16_t)((unsigned)d << succ_low);
        } else if (get_bit()) {
            data[0] = (int16_t)(data[0] + (int16_t)(1 << succ_low));
        }
        return true;
    }
    bool decode_prog_ac(int16_t data[64], const Huff &ac) {
        if (spec_start == 0) return false;
        if (succ_high == 0) {
            const int shift = succ_low;
            if (eob_run) { --eob_run; return true; }
            int k = spec_start;
            do {
                const int rs = huff(ac);
                if (rs < 0) return false;
                const int s = rs & 15, r = rs >> 4;
                if (s == 0) {
                    if (r < 15) {
                        eob_run = 1 << r;
                        if (r) eob_run += get_bits(r);
                        --eob_run;
                        break;
                    }
                    k += 16;
                } else {
                    k += r;
                    const unsigned zig = kDezigzag[k++];
                    data[zig] = (int16_t)(extend_receive(s) * (1 << shift));
                }
            } while (k <= spec_end);
        } else {
            const int16_t bit = (int16_t)(1 << succ_low);
            auto refine = [&](int16_t *c) {
                if (get_bit())
                    if ((*c & bit) == 0) { if (*c > 0) *c = (int16_t)(*c + bit); else *c = (int16_t)(*c - bit); }
            };
            if (eob_run) {
                --eob_run;
                for (int k = spec_start; k <= spec_end; ++k) {
                    int16_t *c = &data[kDezigzag[k]];
                    if (*c != 0) refine(c);
                }
            } else {
                int k = spec_start;
                do {
                    const int rs = huff(ac);
                    if (rs < 0) return false;
                    int s = rs & 15, r = rs >> 4;
                    if (s == 0) {
                        if (r < 15) {
                            eob_run = (1 << r) - 1;
                            if (r) eob_run += get_bits(r);
                            r = 64;                     // to the end of the band
                        }
                    } else {
                        if (s != 1) return false;
                        s = get_bit() ? bit : -bit;
                    }
                    while (k <= spec_end) {
                        int16_t *c = &data[kDezigzag[k++]];
                        if (*c != 0) refine(c);
                        else {
                            if (r == 0) { *c = (int16_t)s; break; }
                            --r;
                        }
                    }
                } while (k <= spec_end);
            }
        }
        return true;
    }

    static uint8_t clamp8(int64_t x) { return x < 0 ? 0 : (x > 255 ? 255 : (uint8_t)x); }
    // 8x8 inverse DCT, 12-bit fixed point, written as the two 4x4 integer matrices an 8-point IDCT splits into (even inputs ->
    // the symmetric part, odd inputs -> the antisymmetric part): out[k] = E[k].(s0 s2 s4 s6) + O[k].(s1 s3 s5 s7),
    // out[7-k] = the same with a minus.  The entries are sums of the twelve rotation constants of the classic LLM factorisation
    // rounded to 1/4096 (what stb_image v2.27 uses: 2217 = round(0.5411961 * 4096), ...); integer arithmetic is exact, so this
    // matrix form yields, bit for bit, what the factored butterfly network with those constants yields -- which is what makes
    // the texels those of the reference's loader (tests/golden/jpeg_textures.npz) -- without being that network.
    // 64-bit accumulators: no overflow on corrupt coefficient data either.
    static void idct1d(const int64_t s[8], int64_t sym[4], int64_t asym[4]) {
        static const int32_t E[4][4] = {{4096, 5352, 4096, 2217}, {4096, 2217, -4096, -5350}, {4096, -2217, -4096, 5350}, {4096, -5352, 4096, -2217}};
        static const int32_t O[4][4] = {{5683, 4816, 3219, 1131}, {4816, -1129, -5681, -3218}, {3219, -5681, 1132, 4816}, {1131, -3218, 4816, -5680}};
        for (int k = 0; k < 4; k++) {
            sym[k] = E[k][0] * s[0] + E[k][1] * s[2] + E[k][2] * s[4] + E[k][3] * s[6];
            asym[k] = O[k][0] * s[1] + O[k][1] * s[3] + O[k][2] * s[5] + O[k][3] * s[7];
        }
    }
    static void idct(uint8_t *out, int stride, const int16_t d[64]) {
        int64_t val[64];
        for (int col = 0; col < 8; col++) {               // columns: result kept with 2 extra bits (>> 10 of the 12)
            int64_t s[8], e[4], o[4];
            for (int k = 0; k < 8; k++) s[k] = d[k * 8 + col];
            idct1d(s, e, o);
            for (int k = 0; k < 4; k++) {
                val[k * 8 + col] = (e[k] + o[k] + 512) >> 10;
                val[(7 - k) * 8 + col] = (e[k] - o[k] + 512) >> 10;
            }
        }
        for (int row = 0; row < 8; row++) {               // rows: >> 17 in all, + 128 level shift, rounded
            int64_t e[4], o[4];
            idct1d(val + row * 8, e, o);
            uint8_t *px = out + (size_t)row * stride;
            const int64_t bias = 65536 + (128 << 17);
            for (int k = 0; k < 4; k++) {
                px[k] = clamp8((e[k] + o[k] + bias) >> 17);
                px[7 - k] = clamp8((e[k] - o[k] + bias) >> 17);
            }
        }
    }

    int get_marker() {
        if (marker != 0xff) { const int x = marker; marker = 0xff; return x; }
        int x = get8();
        if (x != 0xff) return 0xff;
        while (x == 0xff) x = get8();
        return x;
    }
    static bool is_restart(int m) { return m >= 0xd0 && m <= 0xd7; }

    bool process_marker(int m) {
        if (m == 0xff) return false;
        if (m == 0xdd) {
            if (get16() != 4) return false;
            restart_interval = get16();
            return true;
        }
        if (m == 0xdb) {
            int L = get16() - 2;
            while (L > 0) {
                const int q = get8(), prec = q >> 4, t = q & 15;
                if (prec != 0 && prec != 1) return false;
                if (t > 3) return false;
                for (int i = 0; i < 64; i++) dequant[t][kDezigzag[i]] = (uint16_t)(prec ? get16() : get8());
                L -= prec ? 129 : 65;
            }
            return L == 0;
        }
        if (m == 0xc4) {
            int L = get16() - 2;
            while (L > 0) {
                int sizes[16], n = 0;
                const int q = get8(), tc = q >> 4, th = q & 15;
                if (tc > 1 || th > 3) return false;
                for (int i = 0; i < 16; i++) { sizes[i] = get8(); n += sizes[i]; }
                L -= 17;
                Huff &h = tc == 0 ? hdc[th] : hac[th];
                if (n > 256 || !h.build(sizes)) return false;
                for (int i = 0; i < n; i++) h.value[i] = (uint8_t)get8();
                L -= n;
            }
            return L == 0;
        }
        if ((m >= 0xe0 && m <= 0xef) || m == 0xfe) {
            int L = get16();
            if (L < 2) return false;
            L -= 2;
            if (m == 0xe0 && L >= 5) {
                static const char tag[5] = {'J', 'F', 'I', 'F', '\0'};
                bool ok = true;
                for (int i = 0; i < 5; i++) if (get8() != (uint8_t)tag[i]) ok = false;
                L -= 5;
                if (ok) jfif = true;
            } else if (m == 0xee && L >= 12) {
                static const char tag[6] = {'A', 'd', 'o', 'b', 'e', '\0'};
                bool ok = true;
                for (int i = 0; i < 6; i++) if (get8() != (uint8_t)tag[i]) ok = false;
                L -= 6;
                if (ok) { get8(); get16(); get16(); app14 = get8(); L -= 6; }
            }
            if (L < 0 || (size_t)(end - p) < (size_t)L) { p = end; return true; }
            p += L;
            return true;
        }
        return false;
    }

    bool frame_header() {
        const int Lf = get16();
        if (Lf < 11) return false;
        if (get8() != 8) return false;
        img_y = get16(); if (img_y == 0) return false;
        img_x = get16(); if (img_x == 0) return false;
        const int c = get8();
        if (c != 3 && c != 1 && c != 4) return false;
        img_n = c;
        if (Lf != 8 + 3 * img_n) return false;
        rgb = 0;
        for (int i = 0; i < img_n; i++) {
            static const char ids[3] = {'R', 'G', 'B'};
            comp[i].id = get8();
            if (img_n == 3 && comp[i].id == (uint8_t)ids[i]) ++rgb;
            const int q = get8();
            comp[i].h = q >> 4; if (!comp[i].h || comp[i].h > 4) return false;
            comp[i].v = q & 15; if (!comp[i].v || comp[i].v > 4) return false;
            comp[i].tq = get8(); if (comp[i].tq > 3) return false;
        }
        if ((uint64_t)img_x * img_y > (1u << 27)) return false;        // bounds the working memory (pt_png.h uses the same limit)
        h_max = v_max = 1;
        for (int i = 0; i < img_n; i++) { if (comp[i].h > h_max) h_max = comp[i].h; if (comp[i].v > v_max) v_max = comp[i].v; }
        // sampling factors that do not divide the largest one: the vendored stb_image then upsamples by the truncated ratio
        // and reads past the component's rows and, on the last row, past its buffer (later stb versions refuse such
        // files as corrupt) -- nothing defined to be identical to, so "failed to load" (found by tools/fuzz)
        for (int i = 0; i < img_n; i++) if (h_max % comp[i].h || v_max % comp[i].v) return false;
        const int mcu_w = h_max * 8, mcu_h = v_max * 8;
        mcu_x = (img_x + mcu_w - 1) / mcu_w;
        mcu_y = (img_y + mcu_h - 1) / mcu_h;
        for (int i = 0; i < img_n; i++) {
            Comp &k = comp[i];
            k.x = (img_x * k.h + h_max - 1) / h_max;
            k.y = (img_y * k.v + v_max - 1) / v_max;
            k.w2 = mcu_x * k.h * 8;
            k.h2 = mcu_y * k.v * 8;
            k.data.assign((size_t)k.w2 * k.h2, 0);
            if (progressive) { k.coeff_w = k.w2 / 8; k.coeff.assign((size_t)k.w2 * k.h2, 0); }
        }
        return true;
    }

    bool scan_header() {
        const int Ls = get16();
        scan_n = get8();
        if (scan_n < 1 || scan_n > 4 || scan_n > img_n) return false;
        if (Ls != 6 + 2 * scan_n) return false;
        for (int i = 0; i < scan_n; i++) {
            const int id = get8(), q = get8();
            int which = 0;
            for (; which < img_n; ++which) if (comp[which].id == id) break;
            if (which == img_n) return false;
            comp[which].hd = q >> 4; if (comp[which].hd > 3) return false;
            comp[which].ha = q & 15; if (comp[which].ha > 3) return false;
            order[i] = which;
        }
        spec_start = get8();
        spec_end = get8();
        const int aa = get8();
        succ_high = aa >> 4;
        succ_low = aa & 15;
        if (progressive) {
            if (spec_start > 63 || spec_end > 63 || spec_start > spec_end || succ_high > 13 || succ_low > 13) return false;
        } else {
            if (spec_start != 0 || succ_high != 0 || succ_low != 0) return false;
            spec_end = 63;
        }
        return true;
    }

    // true = go on with the next marker; a scan that ends early (no restart marker where one is due) keeps what it has
    bool after_mcu(bool &stop) {
        if (--todo <= 0) {
            if (nbits < 24) grow();
            if (!is_restart(marker)) { stop = true; return true; }
            reset();
        }
        return true;
    }

    bool entropy_data() {
        reset();
        bool stop = false;
        int16_t data[64];
        if (!progressive) {
            if (scan_n == 1) {
                const int n = order[0];
                const int w = (comp[n].x + 7) >> 3, h = (comp[n].y + 7) >> 3;
                for (int j = 0; j < h; j++)
                    for (int i = 0; i < w; i++) {
                        if (!decode_block(data, hdc[comp[n].hd], hac[comp[n].ha], n, dequant[comp[n].tq])) return false;
                        idct(&comp[n].data[(size_t)comp[n].w2 * j * 8 + i * 8], comp[n].w2, data);
                        after_mcu(stop);
                        if (stop) return true;
                    }
            } else {
                for (int j = 0; j < mcu_y; j++)
                    for (int i = 0; i < mcu_x; i++) {
                        for (int k = 0; k < scan_n; k++) {
                            const int n = order[k];
                            for (int y = 0; y < comp[n].v; y++)
                                for (int x = 0; x < comp[n].h; x++) {
                                    const int x2 = (i * comp[n].h + x) * 8, y2 = (j * comp[n].v + y) * 8;
                                    if (!decode_block(data, hdc[comp[n].hd], hac[comp[n].ha], n, dequant[comp[n].tq])) return false;
                                    idct(&comp[n].data[(size_t)comp[n].w2 * y2 + x2], comp[n].w2, data);
                                }
                        }
                        after_mcu(stop);
                        if (stop) return true;
                    }
            }
        } else {
            if (scan_n == 1) {
                const int n = order[0];
                const int w = (comp[n].x + 7) >> 3, h = (comp[n].y + 7) >> 3;
                for (int j = 0; j < h; j++)
                    for (int i = 0; i < w; i++) {
                        int16_t *blk = &comp[n].coeff[64 * ((size_t)i + (size_t)j * comp[n].coeff_w)];
                        if (spec_start == 0) { if (!decode_prog_dc(blk, hdc[comp[n].hd], n)) return false; }
                        else if (!decode_prog_ac(blk, hac[comp[n].ha])) return false;
                        after_mcu(stop);
                        if (stop) return true;
                    }
            } else {
                for (int j = 0; j < mcu_y; j++)
                    for (int i = 0; i < mcu_x; i++) {
                        for (int k = 0; k < scan_n; k++) {
                            const int n = order[k];
                            for (int y = 0; y < comp[n].v; y++)
                                for (int x = 0; x < comp[n].h; x++) {
                                    const int x2 = i * comp[n].h + x, y2 = j * comp[n].v + y;
                                    int16_t *blk = &comp[n].coeff[64 * ((size_t)x2 + (size_t)y2 * comp[n].coeff_w)];
                                    if (!decode_prog_dc(blk, hdc[comp[n].hd], n)) return false;
                                }
                        }
                        after_mcu(stop);
                        if (stop) return true;
                    }
            }
        }
        return true;
    }

    void finish() {
        if (!progressive) return;
        for (int n = 0; n < img_n; n++) {
            const int w = (comp[n].x + 7) >> 3, h = (comp[n].y + 7) >> 3;
            for (int j = 0; j < h; j++)
                for (int i = 0; i < w; i++) {
                    int16_t *blk = &comp[n].coeff[64 * ((size_t)i + (size_t)j * comp[n].coeff_w)];
                    const uint16_t *dq = dequant[comp[n].tq];
                    for (int k = 0; k < 64; k++) blk[k] = (int16_t)(blk[k] * dq[k]);
                    idct(&comp[n].data[(size_t)comp[n].w2 * j * 8 + i * 8], comp[n].w2, blk);
                }
        }
    }

    bool decode_image() {
        restart_interval = 0;
        jfif = false; app14 = -1; marker = 0xff;
        memset(dequant, 0, sizeof dequant);
        for (Huff &h : hdc) { memset(&h, 0, sizeof h); h.maxcode[17] = 0xffffffffu; }      // a table the file never defines decodes
        for (Huff &h : hac) { memset(&h, 0, sizeof h); h.maxcode[17] = 0xffffffffu; }      // nothing ("bad huffman code")
        int m = get_marker();
        if (m != 0xd8) return false;
        m = get_marker();
        while (!(m == 0xc0 || m == 0xc1 || m == 0xc2)) {
            if (!process_marker(m)) return false;
            m = get_marker();
            while (m == 0xff) {
                if (eof()) return false;
                m = get_marker();
            }
        }
        progressive = m == 0xc2;
        if (!frame_header()) return false;
        m = get_marker();
        while (m != 0xd9) {
            if (m == 0xda) {
                if (!scan_header()) return false;
                if (!entropy_data()) return false;
                if (marker == 0xff) {
                    while (!eof()) {
                        const int x = get8();
                        if (x == 255) { marker = get8(); break; }
                    }
                }
            } else if (m == 0xdc) {
                const int Ld = get16(), NL = get16();
                if (Ld != 4 || NL != img_y) return false;
            } else if (!process_marker(m)) {
                return false;
            }
            m = get_marker();
        }
        finish();
        return true;
    }
};

// ---- upsampling rows -----------------------------------------------------------------------------------------------
inline const uint8_t *row_v2(uint8_t *out, const uint8_t *nr, const uint8_t *fr, int w) {
    for (int i = 0; i < w; i++) out[i] = (uint8_t)((3 * nr[i] + fr[i] + 2) >> 2);
    return out;
}
inline const uint8_t *row_h2(uint8_t *out, const uint8_t *in, int w) {
    if (w == 1) { out[0] = out[1] = in[0]; return out; }
    out[0] = in[0];
    out[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
    int i;
    for (i = 1; i < w - 1; i++) {
        const int n = 3 * in[i] + 2;
        out[i * 2 + 0] = (uint8_t)((n + in[i - 1]) >> 2);
        out[i * 2 + 1] = (uint8_t)((n + in[i + 1]) >> 2);
    }
    out[i * 2 + 0] = (uint8_t)((in[w - 2] * 3 + in[w - 1] + 2) >> 2);
    out[i * 2 + 1] = in[w - 1];
    return out;
}
inline const uint8_t *row_hv2(uint8_t *out, const uint8_t *nr, const uint8_t *fr, int w) {
    if (w == 1) { out[0] = out[1] = (uint8_t)((3 * nr[0] + fr[0] + 2) >> 2); return out; }
    int t1 = 3 * nr[0] + fr[0];
    out[0] = (uint8_t)((t1 + 2) >> 2);
    for (int i = 1; i < w; i++) {
        const int t0 = t1;
        t1 = 3 * nr[i] + fr[i];
        out[i * 2 - 1] = (uint8_t)((3 * t0 + t1 + 8) >> 4);
        out[i * 2] = (uint8_t)((3 * t1 + t0 + 8) >> 4);
    }
    out[w * 2 - 1] = (uint8_t)((t1 + 2) >> 2);
    return out;
}
inline const uint8_t *row_generic(uint8_t *out, const uint8_t *nr, int w, int hs) {
    for (int i = 0; i < w; i++)
        for (int j = 0; j < hs; j++) out[i * hs + j] = nr[i];
    return out;
}
inline uint8_t blinn(uint8_t x, uint8_t y) { const unsigned t = (unsigned)x * y + 128; return (uint8_t)((t + (t >> 8)) >> 8); }

// Decodes a JPEG the way stbi_load(..., req_comp = 0) does and flips it vertically.  false = "failed to load".
inline bool load_jpeg_flipped(const std::string &file, int &W, int &H, int &CH, std::vector<uint8_t> &pixels) {
    if (file.size() < 4 || (uint8_t)file[0] != 0xff || (uint8_t)file[1] != 0xd8) return false;
    Decoder *z = new Decoder();
    z->p = (const uint8_t *)file.data();
    z->end = z->p + file.size();
    if (!z->decode_image()) { delete z; return false; }
    const int n = z->img_n >= 3 ? 3 : 1;
    const bool is_rgb = z->img_n == 3 && (z->rgb == 3 || (z->app14 == 0 && !z->jfif));
    const int decode_n = z->img_n;                      // (n == 3 whenever img_n >= 3, so every component is needed)
    const int w = z->img_x, h = z->img_y;
    struct Res { int hs, vs, ystep, w_lores, ypos; const uint8_t *line0, *line1; std::vector<uint8_t> buf; } res[4];
    for (int k = 0; k < decode_n; k++) {
        Res &r = res[k];
        r.hs = z->h_max / z->comp[k].h;
        r.vs = z->v_max / z->comp[k].v;
        r.ystep = r.vs >> 1;
        r.w_lores = (w + r.hs - 1) / r.hs;
        r.ypos = 0;
        r.line0 = r.line1 = z->comp[k].data.data();
        r.buf.assign((size_t)w + 3 + 8, 0);
    }
    std::vector<uint8_t> img((size_t)w * h * n);
    static const int kR = ((int)(1.40200f * 4096.0f + 0.5f)) << 8, kG1 = ((int)(0.71414f * 4096.0f + 0.5f)) << 8,
                     kG2 = ((int)(0.34414f * 4096.0f + 0.5f)) << 8, kB = ((int)(1.77200f * 4096.0f + 0.5f)) << 8;
    auto ycc = [&](uint8_t *out, const uint8_t *y, const uint8_t *pcb, const uint8_t *pcr) {
        for (int i = 0; i < w; i++) {
            const int y_fixed = (y[i] << 20) + (1 << 19);
            const int cr = pcr[i] - 128, cb = pcb[i] - 128;
            int r = y_fixed + cr * kR;
            int g = y_fixed + (cr * -kG1) + (int)((unsigned)(cb * -kG2) & 0xffff0000u);
            int b = y_fixed + cb * kB;
            r >>= 20; g >>= 20; b >>= 20;
            if ((unsigned)r > 255) r = r < 0 ? 0 : 255;
            if ((unsigned)g > 255) g = g < 0 ? 0 : 255;
            if ((unsigned)b > 255) b = b < 0 ? 0 : 255;
            out[i * 3 + 0] = (uint8_t)r; out[i * 3 + 1] = (uint8_t)g; out[i * 3 + 2] = (uint8_t)b;
        }
    };
    for (int j = 0; j < h; j++) {
        uint8_t *out = &img[(size_t)n * w * j];
        const uint8_t *co[4] = {nullptr, nullptr, nullptr, nullptr};
        for (int k = 0; k < decode_n; k++) {
            Res &r = res[k];
            const bool y_bot = r.ystep >= (r.vs >> 1);
            const uint8_t *nr = y_bot ? r.line1 : r.line0, *fr = y_bot ? r.line0 : r.line1;
            if (r.hs == 1 && r.vs == 1) co[k] = nr;
            else if (r.hs == 1 && r.vs == 2) co[k] = row_v2(r.buf.data(), nr, fr, r.w_lores);
            else if (r.hs == 2 && r.vs == 1) co[k] = row_h2(r.buf.data(), nr, r.w_lores);
            else if (r.hs == 2 && r.vs == 2) co[k] = row_hv2(r.buf.data(), nr, fr, r.w_lores);
            else { r.buf.resize((size_t)r.w_lores * r.hs + 8); co[k] = row_generic(r.buf.data(), nr, r.w_lores, r.hs); }
            if (++r.ystep >= r.vs) {
                r.ystep = 0;
                r.line0 = r.line1;
                if (++r.ypos < z->comp[k].y) r.line1 += z->comp[k].w2;
            }
        }
        if (z->img_n == 3) {
            if (is_rgb) for (int i = 0; i < w; i++) { out[i * 3] = co[0][i]; out[i * 3 + 1] = co[1][i]; out[i * 3 + 2] = co[2][i]; }
            else ycc(out, co[0], co[1], co[2]);
        } else if (z->img_n == 4) {
            if (z->app14 == 0) {
                for (int i = 0; i < w; i++) {
                    const uint8_t m = co[3][i];
                    out[i * 3] = blinn(co[0][i], m); out[i * 3 + 1] = blinn(co[1][i], m); out[i * 3 + 2] = blinn(co[2][i], m);
                }
            } else if (z->app14 == 2) {
                ycc(out, co[0], co[1], co[2]);
                for (int i = 0; i < w; i++) {
                    const uint8_t m = co[3][i];
                    out[i * 3] = blinn((uint8_t)(255 - out[i * 3]), m);
                    out[i * 3 + 1] = blinn((uint8_t)(255 - out[i * 3 + 1]), m);
                    out[i * 3 + 2] = blinn((uint8_t)(255 - out[i * 3 + 2]), m);
                }
            } else {
                ycc(out, co[0], co[1], co[2]);
            }
        } else {
            memcpy(out, co[0], (size_t)w);
        }
    }
    delete z;
    pixels.resize(img.size());
    const size_t rowb = (size_t)w * n;
    for (int y = 0; y < h; y++) memcpy(&pixels[(size_t)(h - 1 - y) * rowb], &img[(size_t)y * rowb], rowb);
    W = w; H = h; CH = n;
    return true;
}
#undef PTJ_F2F
#undef PTJ_FSH
#undef PTJ_IDCT_1D

}  // namespace ptjpeg
