"""8-bit RGB frames for the PNG writer test: noise (no matches), flat and striped areas (long matches, every length and
distance code range), smooth ramps (each row filter wins somewhere), repeats farther apart than the 32 KiB window, tiny
frames.  cases() is deterministic."""
import numpy as np


def cases():
    rng = np.random.default_rng(4096)
    out = {}

    def put(name, img):
        out[name] = np.ascontiguousarray(img, np.uint8)

    for w, h in ((1, 1), (2, 3), (3, 2), (17, 9), (64, 48), (300, 41)):
        put("noise_%dx%d" % (w, h), rng.integers(0, 256, (h, w, 3)))
        put("flat_%dx%d" % (w, h), np.full((h, w, 3), 77))
        x, y = np.meshgrid(np.arange(w), np.arange(h))
        put("ramp_%dx%d" % (w, h), np.stack([(x * 3) % 256, (y * 5) % 256, ((x + y) * 2) % 256], 2))
    w, h = 256, 200
    x, y = np.meshgrid(np.arange(w), np.arange(h))
    smooth = 127 + 90 * np.sin(x / 23.0)[..., None] * np.cos(y / 17.0)[..., None] * np.array([1.0, 0.7, 0.4])
    put("smooth_256x200", np.clip(smooth + rng.normal(0, 2.5, (h, w, 3)), 0, 255))
    stripes = np.zeros((h, w, 3), np.uint8)
    stripes[:, :, 0] = (x // 7 % 2) * 200; stripes[:, :, 1] = (y // 3 % 5) * 50; stripes[:, :, 2] = ((x * y) % 11 == 0) * 255
    put("stripes_256x200", stripes)
    far = rng.integers(0, 256, (120, 128, 3))                    # a block that comes back 40 KiB later: outside the window
    put("far_128x240", np.concatenate([far[:10], rng.integers(0, 256, (110, 128, 3)), far[:10], far[10:120]], 0))
    frame = np.zeros((90, 160, 3))                               # a Cornell-like frame: dark, a bright patch, coloured walls
    frame[:, :40] = (160, 30, 30); frame[:, 120:] = (30, 160, 30); frame[10:20, 60:100] = 255
    frame[20:, 40:120] = np.linspace(200, 60, 70)[:, None, None]
    put("frame_160x90", np.clip(frame + rng.normal(0, 6, frame.shape), 0, 255))
    return out
