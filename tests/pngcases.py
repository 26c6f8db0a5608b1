"""PNG files for the texture-loader tests, written by hand (zlib + chunks) so that every colour type, bit depth, filter
type, interlacing, palette / colour-key transparency and a stored (uncompressed) stream occur.  cases() is deterministic."""
import struct
import zlib

import numpy as np


def chunk(t, d):
    return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)

def pack_rows(samples, depth):
    """samples: (h, w*nch) ints -> list of packed row bytes"""
    rows = []
    for r in samples:
        if depth == 8: rows.append(bytes(r.astype(np.uint8)))
        elif depth == 16: rows.append(r.astype('>u2').tobytes())
        else:
            bits = ''.join(format(int(v), '0%db' % depth) for v in r)
            bits += '0' * (-len(bits) % 8)
            rows.append(bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8)))
    return rows

def filt(rows, bpp, rng):
    out = b''
    prev = None
    for r in rows:
        f = int(rng.integers(0, 5))
        cur = np.frombuffer(r, np.uint8).astype(int)
        up = np.frombuffer(prev, np.uint8).astype(int) if prev is not None else np.zeros_like(cur)
        o = np.zeros_like(cur)
        for x in range(len(cur)):
            a = cur[x - bpp] if x >= bpp else 0; b = up[x]; c = up[x - bpp] if x >= bpp else 0
            if f == 0: p = 0
            elif f == 1: p = a
            elif f == 2: p = b
            elif f == 3: p = (a + b) >> 1
            else:
                pp = a + b - c; pa, pb, pc = abs(pp - a), abs(pp - b), abs(pp - c)
                p = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
            o[x] = (cur[x] - p) & 255
        out += bytes([f]) + bytes(o.astype(np.uint8))
        prev = r
    return out

def make_png(w, h, color, depth, samples, interlace=0, palette=None, trns=None, rng=None, level=6):
    nch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[color]
    bpp = max(1, nch * depth // 8)
    rng = rng or np.random.default_rng(0)
    s = samples.reshape(h, w, nch)
    if not interlace:
        raw = filt(pack_rows(s.reshape(h, w * nch), depth), bpp, rng)
    else:
        xo = [0, 4, 0, 2, 0, 1, 0]; yo = [0, 0, 4, 0, 2, 0, 1]; xs = [8, 8, 4, 4, 2, 2, 1]; ys = [8, 8, 8, 4, 4, 2, 2]
        raw = b''
        for p in range(7):
            sub = s[yo[p]::ys[p], xo[p]::xs[p]]
            if sub.shape[0] == 0 or sub.shape[1] == 0: continue
            raw += filt(pack_rows(sub.reshape(sub.shape[0], -1), depth), bpp, rng)
    png = b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', struct.pack(">IIBBBBB", w, h, depth, color, 0, 0, interlace))
    if palette is not None: png += chunk(b'PLTE', bytes(np.asarray(palette, np.uint8).reshape(-1)))
    if trns is not None: png += chunk(b'tRNS', trns)
    z = zlib.compress(raw, level)
    half = len(z) // 2
    png += chunk(b'tEXt', b'Comment\0hello') + chunk(b'IDAT', z[:half]) + chunk(b'IDAT', z[half:]) + chunk(b'IEND', b'')
    return png

def cases():
    rng = np.random.default_rng(42)
    out = []
    def add(name, *a, **k): out.append((name, make_png(*a, rng=rng, **k)))
    w, h = 13, 9
    add("rgb8", w, h, 2, 8, rng.integers(0, 256, (h, w, 3)))
    add("rgba8", w, h, 6, 8, rng.integers(0, 256, (h, w, 4)))
    add("rgb8_adam7", w, h, 2, 8, rng.integers(0, 256, (h, w, 3)), interlace=1)
    add("rgb16", w, h, 2, 16, rng.integers(0, 65536, (h, w, 3)))
    add("rgba16_adam7", w, h, 6, 16, rng.integers(0, 65536, (h, w, 4)), interlace=1)
    pal = rng.integers(0, 256, (16, 3))
    add("pal4", w, h, 3, 4, rng.integers(0, 16, (h, w, 1)), palette=pal)
    add("pal8_trns", w, h, 3, 8, rng.integers(0, 16, (h, w, 1)), palette=pal, trns=bytes(rng.integers(0, 256, 7).astype(np.uint8)))
    add("pal1_adam7", w, h, 3, 1, rng.integers(0, 2, (h, w, 1)), palette=pal[:2], interlace=1)
    add("pal2", w, h, 3, 2, rng.integers(0, 4, (h, w, 1)), palette=pal[:4])
    s = rng.integers(0, 4, (h, w, 3)) * 60
    add("rgb8_key", w, h, 2, 8, s, trns=struct.pack(">HHH", 60, 120, 0))
    s16 = rng.integers(0, 3, (h, w, 3)) * 30000
    add("rgb16_key", w, h, 2, 16, s16, trns=struct.pack(">HHH", 30000, 0, 60000))
    add("grey8", w, h, 0, 8, rng.integers(0, 256, (h, w, 1)))
    add("grey4", w, h, 0, 4, rng.integers(0, 16, (h, w, 1)))
    add("greya8", w, h, 4, 8, rng.integers(0, 256, (h, w, 2)))
    add("grey2_key", w, h, 0, 2, rng.integers(0, 4, (h, w, 1)), trns=struct.pack(">H", 2))
    add("rgb8_stored", 300, 40, 2, 8, rng.integers(0, 256, (40, 300, 3)), level=0)
    add("rgb8_big", 257, 129, 2, 8, (np.arange(257 * 129 * 3) % 251).reshape(129, 257, 3), level=9)
    return out
