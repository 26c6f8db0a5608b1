"""An OBJ file full of polygons with more than four corners (convex, concave, star-shaped, non-planar, in every coordinate
plane, with collinear and repeated vertices) for the triangulation test.  obj_text() is deterministic."""
import numpy as np


def obj_text():
    rng = np.random.default_rng(31)
    lines = ["mtllib cube.mtl"]
    nv = 0
    faces = []

    def poly(pts):
        nonlocal nv
        for p in pts:
            lines.append("v %.6f %.6f %.6f" % tuple(p))
            lines.append("vt %.6f %.6f" % (rng.random(), rng.random()))
        faces.append("f " + " ".join("%d/%d" % (nv + k + 1, nv + k + 1) for k in range(len(pts))))
        nv += len(pts)

    def ring(n, radii, plane, off, wobble=0.0, reverse=False):
        ang = np.linspace(0, 2 * np.pi, n, endpoint=False) + rng.uniform(0, 1)
        r = np.resize(np.asarray(radii, float), n)
        a, b = r * np.cos(ang), r * np.sin(ang)
        c = rng.normal(scale=wobble, size=n) if wobble else np.zeros(n)
        pts = {"xy": np.stack([a, b, c], 1), "yz": np.stack([c, a, b], 1), "xz": np.stack([a, c, b], 1)}[plane] + off
        return pts[::-1] if reverse else pts

    k = 0
    for plane in ("xy", "yz", "xz"):
        for n in (5, 6, 7, 8, 12):
            for radii, wob, rev in (((1.0,), 0.0, False), ((1.0, 0.45), 0.0, False), ((1.0, 0.8, 0.3), 0.05, True), ((0.7, 1.2), 0.3, False)):
                poly(ring(n, radii, plane, np.array([3.0 * (k % 7), 3.0 * (k // 7), 0.0]), wob, rev))
                k += 1
    # an L, a comb, collinear runs, a repeated vertex, a tilted pentagon
    poly(np.array([[0, 0, 0], [3, 0, 0], [3, 1, 0], [1, 1, 0], [1, 3, 0], [0, 3, 0]], float) + [30, 0, 0])
    poly(np.array([[0, 0, 0], [5, 0, 0], [5, 2, 0], [4, 2, 0], [4, 1, 0], [3, 1, 0], [3, 2, 0], [2, 2, 0], [2, 1, 0], [1, 1, 0], [1, 2, 0], [0, 2, 0]], float) + [30, 5, 0])
    poly(np.array([[0, 0, 0], [1, 0, 0], [2, 0, 0], [3, 0, 0], [3, 2, 0], [0, 2, 0]], float) + [30, 10, 0])
    poly(np.array([[0, 0, 0], [2, 0, 0], [2, 0, 0], [2, 2, 0], [1, 3, 0], [0, 2, 0]], float) + [30, 15, 0])
    t = ring(5, (1.0,), "xy", np.zeros(3))
    rot = np.array([[1, 0, 0], [0, 0.6, -0.8], [0, 0.8, 0.6]])
    poly(t @ rot.T + [30, 20, 0])
    lines += faces
    return "\n".join(lines) + "\n"
