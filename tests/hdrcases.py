"""Float frames for the Radiance .hdr writer test: widths below 8 (flat RGBE), at 8, odd, and past the 127 / 128 packet
limits; flat areas, ramps, noise, zeros, values below the 1e-32 cutoff and far above 1.  cases() is deterministic."""
import numpy as np


def cases():
    rng = np.random.default_rng(77)
    out = {}

    def put(name, w, h, img):
        out[name] = np.ascontiguousarray(img, np.float32).reshape(h, w, 3)

    for w, h in ((1, 1), (5, 3), (7, 2), (8, 2), (9, 4), (37, 5), (129, 3), (300, 4), (640, 3)):
        put("noise_%dx%d" % (w, h), w, h, rng.random((h, w, 3)) * 1.5)
        flat = np.tile(rng.random((h, 1, 3)), (1, w, 1))
        put("flat_%dx%d" % (w, h), w, h, flat)
        img = flat.copy()                                       # runs broken by single pixels, pairs and short noise bursts
        for _ in range(max(1, w // 6)):
            x = int(rng.integers(0, w)); n = int(rng.integers(1, 4))
            img[:, x:x + n] = rng.random((h, min(n, w - x), 3))
        put("runs_%dx%d" % (w, h), w, h, img)
    w, h = 64, 6
    img = rng.random((h, w, 3))
    img[0] = 0.0
    img[1] = 1e-33                                              # under the cutoff: stored as 0 0 0 0
    img[2] *= 1e-30
    img[3] *= 1e6
    img[4, :, 0] = 0.0; img[4, :, 1] = np.linspace(0, 4, w); img[4, :, 2] = 2.0 ** np.arange(-32, 32)
    img[5] = np.repeat(rng.random((w // 4, 3)), 4, axis=0)       # runs of four
    put("range_64x6", w, h, img)
    return out
