"""JPEG files for the texture-loader tests: written by PIL (baseline / progressive, 4:4:4 / 4:2:2 / 4:2:0 / 4:1:1, optimised
Huffman tables, restart intervals, 16-bit quantisation tables, greyscale, CMYK, RGB without colour transform), plus
synthetic ones (4:4:0 by patching the sampling factors, entropy data cut short).  Only tests/golden/make_golden.py calls
cases(): encoders differ between PIL versions, so the tests use the bytes stored in tests/golden/jpeg_textures.npz."""
import io

import numpy as np


def pictures():
    rng = np.random.default_rng(5)
    out = {}
    def grad(w, h):
        y, x = np.mgrid[0:h, 0:w]
        r = (x * 255 // max(w - 1, 1)); g = (y * 255 // max(h - 1, 1)); b = ((x + y) * 7 % 256)
        return np.stack([r, g, b], 2).astype(np.uint8)
    def noise(w, h): return rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    def blobs(w, h):
        a = grad(w, h).astype(int)
        for _ in range(12):
            cx, cy, rr = rng.integers(0, w), rng.integers(0, h), rng.integers(2, max(3, min(w, h) // 2))
            y, x = np.mgrid[0:h, 0:w]
            m = (x - cx) ** 2 + (y - cy) ** 2 < rr * rr
            a[m] = rng.integers(0, 256, 3)
        return a.astype(np.uint8)
    out["grad"] = grad(64, 48); out["noise"] = noise(37, 29); out["blobs"] = blobs(75, 51); out["thin"] = blobs(1, 40); out["wide"] = noise(50, 1); out["big"] = blobs(120, 67)
    return out

def cases():
    from PIL import Image
    pics = pictures()
    res = []
    def add(name, arr, mode="RGB", **kw):
        im = Image.fromarray(arr if mode != "L" else arr[:, :, 0], "L" if mode == "L" else "RGB")
        if mode == "CMYK": im = im.convert("CMYK")
        b = io.BytesIO(); im.save(b, "JPEG", **kw); res.append((name, b.getvalue()))
    for pn, arr in pics.items():
        add(pn + "_420", arr, quality=85, subsampling=2)
        add(pn + "_444", arr, quality=93, subsampling=0)
        add(pn + "_422", arr, quality=70, subsampling=1)
        add(pn + "_prog420", arr, quality=80, subsampling=2, progressive=True)
        add(pn + "_prog444_opt", arr, quality=60, subsampling=0, progressive=True, optimize=True)
    add("blobs_q10", pics["blobs"], quality=10, subsampling=2)
    add("blobs_q100", pics["blobs"], quality=100, subsampling=0)
    add("blobs_opt", pics["blobs"], quality=75, subsampling=2, optimize=True)
    add("blobs_grey", pics["blobs"], mode="L", quality=80)
    add("blobs_cmyk", pics["blobs"], mode="CMYK", quality=80)
    try:
        add("big_restart", pics["big"], quality=80, subsampling=2, restart_marker_blocks=3)
        add("big_restart_prog", pics["big"], quality=80, subsampling=0, progressive=True, restart_marker_rows=1)
    except Exception as e:
        print("no restart marker support in this PIL:", e)
    for sub in ("4:1:1", "4:4:0"):
        try:
            add("blobs_" + sub.replace(":", ""), pics["blobs"], quality=80, subsampling=sub)
            add("big_prog_" + sub.replace(":", ""), pics["big"], quality=70, subsampling=sub, progressive=True)
        except Exception as e:
            print("no subsampling", sub, e)
    try:
        q = [min(255 * 4, 16 + 40 * i) for i in range(64)]
        add("blobs_qtab16", pics["blobs"], qtables=[q, q], subsampling=2)
    except Exception as e:
        print("no 16-bit qtables:", e)
    try:
        add("blobs_keeprgb", pics["blobs"], quality=90, keep_rgb=True)
    except Exception as e:
        print("no keep_rgb:", e)
    d = dict(res)
    # 4:4:0 (h1v2), which this PIL cannot write: the 4:2:2 stream read with the luma sampling factors swapped -- a valid
    # file with a scrambled picture and too few MCUs, so the decoder also runs into the EOI marker early
    j = bytearray(d["blobs_422"])
    k = j.find(b"\xff\xc0")
    assert j[k + 11] == 0x21
    j[k + 11] = 0x12
    res.append(("blobs_440_patched", bytes(j)))
    # entropy data cut short (EOI re-attached): the rest of the picture decodes from zero bits
    for nm in ("big_420", "big_prog444_opt"):       # (with restart intervals the scan stops early and stb_image leaves the
                                                      # remaining blocks uninitialised: nothing to compare)
        b = d[nm]
        res.append((nm + "_cut", b[: len(b) * 2 // 3] + b"\xff\xd9"))
    return res
