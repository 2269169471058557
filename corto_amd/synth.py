"""Deterministic synthetic meshes / point clouds for tests and bench.

No sample meshes ship with the reference (bun_zipper.ply is absent), so every input is
synthetic.  Everything here uses only IEEE +,-,*,/ on float64 (own polynomial sin/cos, integer LCG
noise) so the same arrays come out bit-identical on any host: the GPU box regenerates the very
same inputs the fixtures were made from.
"""
from __future__ import annotations

import hashlib

import numpy as np

_TWO_PI = 6.283185307179586


def _sincos(x):
    """sin, cos of float64 array via range reduction to [-pi/4, pi/4] + fixed polynomials
    (only + - * : reproducible across libm versions)."""
    x = np.asarray(x, dtype=np.float64)
    k = np.floor(x * (2.0 / np.pi) + 0.5)
    r = x - k * 1.5707963267948966 - k * 6.123233995736766e-17
    r2 = r * r
    s = r * (1.0 + r2 * (-1.0 / 6 + r2 * (1.0 / 120 + r2 * (-1.0 / 5040 + r2 * (1.0 / 362880 + r2 * (-1.0 / 39916800))))))
    c = 1.0 + r2 * (-0.5 + r2 * (1.0 / 24 + r2 * (-1.0 / 720 + r2 * (1.0 / 40320 + r2 * (-1.0 / 3628800 + r2 * (1.0 / 479001600))))))
    q = k.astype(np.int64) & 3
    sin = np.where(q == 0, s, np.where(q == 1, c, np.where(q == 2, -s, -c)))
    cos = np.where(q == 0, c, np.where(q == 1, -s, np.where(q == 2, -c, s)))
    return sin, cos


def _lcg_fast(seed: int, n: int) -> np.ndarray:
    """Counter-based hash noise (splitmix64 finaliser) - vectorised, reproducible."""
    with np.errstate(over="ignore"):
        z = (np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
             + np.uint64((seed * 0xBF58476D1CE4E5B9 + 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


class Mesh:
    """Plain container: float32 positions/normals/uvs, uint8 colours, uint32 index (all C-contiguous)."""

    def __init__(self, position, index=None, normal=None, color=None, uv=None, radius=None, groups=None):
        self.position = np.ascontiguousarray(position, dtype=np.float32)
        self.index = None if index is None else np.ascontiguousarray(index, dtype=np.uint32)
        self.normal = None if normal is None else np.ascontiguousarray(normal, dtype=np.float32)
        self.color = None if color is None else np.ascontiguousarray(color, dtype=np.uint8)
        self.uv = None if uv is None else np.ascontiguousarray(uv, dtype=np.float32)
        self.radius = None if radius is None else np.ascontiguousarray(radius, dtype=np.float32)
        self.groups = groups
        self.group_props = None      # optional: one {key: value} dict per group (Group::properties)

    @property
    def nvert(self):
        return self.position.shape[0]

    @property
    def nface(self):
        return 0 if self.index is None else self.index.shape[0]


def _grid_attrs(nu, nv1, px, py, pz, seed, color_components):
    """analytic-ish normals, uv, smooth colour gradients for a (nv1 x nu) grid of points."""
    # normals: normalised radial direction perturbed a little (they only need to be unit-ish vectors)
    n = np.stack([px, py, pz], axis=-1)
    ln = np.sqrt((n * n).sum(-1, keepdims=True))
    n = n / ln
    ii = np.arange(nu, dtype=np.float64)[None, :].repeat(nv1, 0)
    jj = np.arange(nv1, dtype=np.float64)[:, None].repeat(nu, 1)
    uv = np.stack([ii / nu, jj / max(nv1 - 1, 1)], axis=-1)
    col = np.empty((nv1, nu, 4), dtype=np.float64)
    col[..., 0] = 40 + 170 * (ii / nu)
    col[..., 1] = 30 + 190 * (jj / max(nv1 - 1, 1))
    col[..., 2] = 128 + 100 * n[..., 2]
    col[..., 3] = 200 + 40 * n[..., 0]
    col = np.clip(np.floor(col), 0, 255).astype(np.uint8)
    return n.reshape(-1, 3), uv.reshape(-1, 2), col.reshape(-1, 4)[:, :color_components]


def bumpy_sphere(nu=64, nv=32, seed=0, color_components=4, noise=0.01):
    """Genus-0 lat-long grid, open at both poles: nu*(nv+1) verts, 2*nu*nv tris.
    nu=64,nv=32 -> 2112 verts / 4096 tris (the C4/C5 unit); nu=512,nv=250 -> C2."""
    nv1 = nv + 1
    i = np.arange(nu, dtype=np.float64)
    j = np.arange(nv1, dtype=np.float64)
    phi = (i / nu) * _TWO_PI                                 # longitude, wraps
    theta = 0.15 + (j / nv) * (np.pi - 0.30)                 # latitude, poles cut off
    sp, cp = _sincos(phi)
    st, ct = _sincos(theta)
    s5, _ = _sincos(5.0 * theta)
    _, c7 = _sincos(7.0 * phi)
    r = 1.0 + 0.1 * s5[:, None] * c7[None, :]
    r = r + noise * (_lcg_fast(seed, nv1 * nu).reshape(nv1, nu) - 0.5)
    px = r * st[:, None] * cp[None, :]
    py = r * st[:, None] * sp[None, :]
    pz = r * ct[:, None] * np.ones_like(cp)[None, :]
    pos = np.stack([px, py, pz], axis=-1).reshape(-1, 3)
    nrm, uv, col = _grid_attrs(nu, nv1, px, py, pz, seed, color_components)
    # triangles: quad (i,j)-(i+1,j)-(i+1,j+1)-(i,j+1), i wraps
    jj, ii = np.meshgrid(np.arange(nv), np.arange(nu), indexing="ij")
    a = jj * nu + ii
    b = jj * nu + (ii + 1) % nu
    c = (jj + 1) * nu + (ii + 1) % nu
    d = (jj + 1) * nu + ii
    tris = np.stack([np.stack([a, b, c], -1), np.stack([a, c, d], -1)], axis=2).reshape(-1, 3)
    return Mesh(pos, tris, nrm, col, uv)


def bumpy_sphere_flipped(nu=64, nv=32, seed=0, color_components=4, flip=0.5):
    """bumpy_sphere with the diagonal of every grid quad flipped with probability `flip` (per-seed): the same vertex and triangle counts,
    but vertex valences from 4 to 8 instead of 6 everywhere, so that no two seeds share a CLERS stream and the (VERTEX LEFT) runs, the
    delta scans' blocks and the incident-face lists are those of an irregular mesh."""
    m = bumpy_sphere(nu, nv, seed, color_components)
    jj, ii = np.meshgrid(np.arange(nv), np.arange(nu), indexing="ij")
    a = jj * nu + ii
    b = jj * nu + (ii + 1) % nu
    c = (jj + 1) * nu + (ii + 1) % nu
    d = (jj + 1) * nu + ii
    f = (_lcg_fast(seed * 7919 + 13, nv * nu).reshape(nv, nu) < flip)[..., None]
    t0 = np.where(f, np.stack([a, b, d], -1), np.stack([a, b, c], -1))
    t1 = np.where(f, np.stack([b, c, d], -1), np.stack([a, c, d], -1))
    m.index = np.stack([t0, t1], axis=2).reshape(-1, 3).astype(m.index.dtype)
    return m


def closed_sphere(nu=24, nv=12, seed=0, color_components=4):
    """Closed genus-0 UV sphere with two pole vertices (triangle fans): 2 + nu*(nv-1) verts, 2*nu*(nv-1) tris.
    Closed surfaces end with END symbols and have no boundary."""
    i = np.arange(nu, dtype=np.float64)
    j = np.arange(1, nv, dtype=np.float64)
    sp, cp = _sincos((i / nu) * _TWO_PI)
    st, ct = _sincos((j / nv) * np.pi)
    r = 1.0 + 0.02 * (_lcg_fast(seed, (nv - 1) * nu).reshape(nv - 1, nu) - 0.5)
    px = r * st[:, None] * cp[None, :]
    py = r * st[:, None] * sp[None, :]
    pz = r * ct[:, None] * np.ones_like(cp)[None, :]
    nrm, uv, col = _grid_attrs(nu, nv - 1, px, py, pz, seed, color_components)
    pos = np.concatenate([np.stack([px, py, pz], -1).reshape(-1, 3), [[0, 0, 1.0], [0, 0, -1.0]]])
    nrm = np.concatenate([nrm, [[0, 0, 1.0], [0, 0, -1.0]]])
    uv = np.concatenate([uv, [[0.5, 0.0], [0.5, 1.0]]])
    col = np.concatenate([col, np.full((2, color_components), 200, dtype=np.uint8)])
    jj, ii = np.meshgrid(np.arange(nv - 2), np.arange(nu), indexing="ij")
    a = jj * nu + ii
    b = jj * nu + (ii + 1) % nu
    c = (jj + 1) * nu + (ii + 1) % nu
    d = (jj + 1) * nu + ii
    tris = np.stack([np.stack([a, b, c], -1), np.stack([a, c, d], -1)], axis=2).reshape(-1, 3)
    top, bot = nu * (nv - 1), nu * (nv - 1) + 1
    ar = np.arange(nu)
    fan_t = np.stack([np.full(nu, top), (ar + 1) % nu, ar], -1)
    base = (nv - 2) * nu
    fan_b = np.stack([np.full(nu, bot), base + ar, base + (ar + 1) % nu], -1)
    return Mesh(pos, np.concatenate([tris, fan_t, fan_b]), nrm, col, uv)


def torus(nu=48, nv=24, seed=0, color_components=4):
    """Closed genus-1 surface (wraps both ways): nu*nv verts, 2*nu*nv tris -> forces SPLIT symbols."""
    i = np.arange(nu, dtype=np.float64)
    j = np.arange(nv, dtype=np.float64)
    sp, cp = _sincos((i / nu) * _TWO_PI)
    st, ct = _sincos((j / nv) * _TWO_PI)
    R, r0 = 1.0, 0.35
    r = r0 + 0.01 * (_lcg_fast(seed, nv * nu).reshape(nv, nu) - 0.5)
    px = (R + r * ct[:, None]) * cp[None, :]
    py = (R + r * ct[:, None]) * sp[None, :]
    pz = r * st[:, None] * np.ones_like(cp)[None, :]
    pos = np.stack([px, py, pz], axis=-1).reshape(-1, 3)
    nx = ct[:, None] * cp[None, :]
    ny = ct[:, None] * sp[None, :]
    nz = st[:, None] * np.ones_like(cp)[None, :]
    nrm, uv, col = _grid_attrs(nu, nv, nx, ny, nz, seed, color_components)
    jj, ii = np.meshgrid(np.arange(nv), np.arange(nu), indexing="ij")
    a = jj * nu + ii
    b = jj * nu + (ii + 1) % nu
    c = ((jj + 1) % nv) * nu + (ii + 1) % nu
    d = ((jj + 1) % nv) * nu + ii
    tris = np.stack([np.stack([a, b, c], -1), np.stack([a, c, d], -1)], axis=2).reshape(-1, 3)
    return Mesh(pos, tris, nrm, col, uv)


def holey_disc(n=40, seed=0, hole_frac=0.12, color_components=3):
    """Planar-ish height field grid with random quads removed: many BOUNDARY / DELAY / SPLIT symbols and
    several connected components possible."""
    g = n + 1
    x = np.arange(g, dtype=np.float64) / n
    X, Y = np.meshgrid(x, x, indexing="xy")
    s3, _ = _sincos(3.0 * X * _TWO_PI / 2)
    _, c2 = _sincos(2.0 * Y * _TWO_PI / 2)
    Z = 0.15 * s3 * c2 + 0.004 * (_lcg_fast(seed, g * g).reshape(g, g) - 0.5)
    pos = np.stack([X, Y, Z], -1).reshape(-1, 3)
    nrm = np.stack([-0.3 * s3, -0.3 * c2, np.ones_like(Z)], -1)
    nrm = (nrm / np.sqrt((nrm * nrm).sum(-1, keepdims=True))).reshape(-1, 3)
    uv = np.stack([X, Y], -1).reshape(-1, 2)
    col = np.stack([255 * X, 255 * Y, 128 + 500 * Z, 255 * np.ones_like(Z)], -1)
    col = np.clip(np.floor(col), 0, 255).astype(np.uint8).reshape(-1, 4)[:, :color_components]
    jj, ii = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    a = jj * g + ii
    b = a + 1
    c = a + g + 1
    d = a + g
    keep = _lcg_fast(seed + 7919, n * n).reshape(n, n) >= hole_frac
    tris = np.stack([np.stack([a, b, c], -1), np.stack([a, c, d], -1)], axis=2)[keep].reshape(-1, 3)
    return Mesh(pos, tris, nrm, col, uv)


def strip(n=400, seed=0, color_components=4):
    """A 2 x n ribbon: every edge but the rungs is a boundary edge, so nearly every CLERS chain ends on the boundary
    after one triangle (the automaton's worst case for edge records per vertex)."""
    j = np.arange(2 * n, dtype=np.float32)
    col_i, row_i = np.floor(j / 2), j % 2
    ang = col_i * np.float32(0.05)
    s, c = _sincos(ang)
    r = np.float32(1.0) + row_i * np.float32(0.2)
    px, py, pz = r * c, r * s, col_i * np.float32(0.01) + _lcg_fast(seed, 2 * n) * np.float32(0.002)
    nrm, uv, col = _grid_attrs(2, n, px.reshape(n, 2), py.reshape(n, 2), pz.reshape(n, 2), seed, color_components)
    pos = np.stack([px, py, pz], axis=1)
    k = np.arange(n - 1, dtype=np.uint32) * 2
    tris = np.concatenate([np.stack([k, k + 2, k + 1], axis=1), np.stack([k + 1, k + 2, k + 3], axis=1)])
    return Mesh(pos, tris, nrm, col, uv)


def merge(meshes):
    """Concatenate meshes into one multi-component mesh."""
    off = 0
    P, I, N, C, U = [], [], [], [], []
    for m in meshes:
        P.append(m.position); I.append(m.index + off); N.append(m.normal); C.append(m.color); U.append(m.uv)
        off += m.nvert
    return Mesh(np.concatenate(P), np.concatenate(I), np.concatenate(N), np.concatenate(C), np.concatenate(U))


def shuffled(mesh: Mesh, seed=1, faces=True, verts=True):
    """Randomly permute vertex ids and/or face order (the encoder's traversal depends on both)."""
    m = mesh
    pos, idx, nrm, col, uv = m.position, m.index, m.normal, m.color, m.uv
    if verts:
        perm = np.argsort(_lcg_fast(seed, m.nvert), kind="stable")       # new -> old
        inv = np.empty_like(perm); inv[perm] = np.arange(m.nvert)
        pos, nrm, uv = pos[perm], nrm[perm], uv[perm]
        col = None if col is None else col[perm]
        idx = inv[idx].astype(np.uint32)
    if faces:
        fp = np.argsort(_lcg_fast(seed + 1, idx.shape[0]), kind="stable")
        idx = idx[fp]
    return Mesh(pos, idx, nrm, col, uv)


def point_cloud(nu=578, nv=289, seed=0, color_components=4):
    """Point cloud sampled on the bumpy sphere: nu*nv points, no faces (C3: 578*289 = 167 042)."""
    m = bumpy_sphere(nu, nv - 1, seed, color_components)
    return Mesh(m.position, None, m.normal, m.color, m.uv)


# ---------------------------------------------------------------------------------------------------------------------------------
# Non-lattice connectivity (round 5): what scanned / remeshed / decimated models look like to the CLERS automaton - valence 3..100+,
# fans, many boundary loops, hundreds of tiny components.  None of these is a quad grid with cut diagonals.
# ---------------------------------------------------------------------------------------------------------------------------------

def _surface_attrs(pos, seed, color_components):
    """normals = normalised position (+ a little hash noise), uv = an affine image of x,y, colours = smooth gradients"""
    p = np.asarray(pos, dtype=np.float64)
    n = p + 0.05 * (_lcg_fast(seed + 101, p.size).reshape(p.shape) - 0.5)
    ln = np.sqrt((n * n).sum(-1, keepdims=True))
    n = n / np.where(ln > 0, ln, 1.0)
    lo, hi = p.min(0), p.max(0)
    ext = np.where(hi > lo, hi - lo, 1.0)
    t = (p - lo) / ext
    uv = t[:, :2] * 0.999
    col = np.stack([40 + 170 * t[:, 0], 30 + 190 * t[:, 1], 128 + 100 * n[:, 2], 200 + 40 * n[:, 0]], -1)
    col = np.clip(np.floor(col), 0, 255).astype(np.uint8)[:, :color_components]
    return n, uv, col


def _canonical_faces(tris):
    """each triangle rotated so that its smallest vertex id comes first (orientation kept), rows sorted lexicographically:
    the same array whichever triangulator produced the set"""
    t = np.asarray(tris, dtype=np.int64).reshape(-1, 3)
    k = np.argmin(t, axis=1)
    r = np.arange(len(t))
    t = np.stack([t[r, k], t[r, (k + 1) % 3], t[r, (k + 2) % 3]], -1)
    return t[np.lexsort((t[:, 2], t[:, 1], t[:, 0]))]


def _compact(pos, tris):
    """drop unreferenced vertices, keep the order of the others"""
    used = np.zeros(len(pos), dtype=bool)
    used[tris.reshape(-1)] = True
    remap = np.cumsum(used) - 1
    return pos[used], remap[tris]


def icosphere(level=3, seed=0, color_components=4, noise=0.02):
    """Subdivided icosahedron: 10*4^level + 2 verts, 20*4^level tris, twelve vertices of valence 5 and the rest 6, no lattice
    anywhere (level 3: 642 / 1280, level 4: 2562 / 5120).  Closed, genus 0."""
    g = (1.0 + np.sqrt(5.0)) / 2.0
    v = np.array([[-1, g, 0], [1, g, 0], [-1, -g, 0], [1, -g, 0], [0, -1, g], [0, 1, g], [0, -1, -g], [0, 1, -g],
                  [g, 0, -1], [g, 0, 1], [-g, 0, -1], [-g, 0, 1]], dtype=np.float64)
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6], [7, 1, 8],
                  [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5], [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]], dtype=np.int64)
    for _ in range(level):
        e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
        key = np.minimum(e[:, 0], e[:, 1]) * (len(v) + 1) + np.maximum(e[:, 0], e[:, 1])
        uk, inv = np.unique(key, return_inverse=True)
        a, b = uk // (len(v) + 1), uk % (len(v) + 1)
        mid = len(v) + inv.reshape(3, -1).T                     # (nface, 3): midpoints of edges 01, 12, 20
        v = np.concatenate([v, 0.5 * (v[a] + v[b])])
        m01, m12, m20 = mid[:, 0], mid[:, 1], mid[:, 2]
        f = np.concatenate([np.stack([f[:, 0], m01, m20], -1), np.stack([f[:, 1], m12, m01], -1),
                            np.stack([f[:, 2], m20, m12], -1), np.stack([m01, m12, m20], -1)])
    v = v / np.sqrt((v * v).sum(-1, keepdims=True))
    s5, _ = _sincos(5.0 * v[:, 2])
    _, c7 = _sincos(7.0 * v[:, 0])
    r = 1.0 + 0.08 * s5 * c7 + noise * (_lcg_fast(seed, len(v)) - 0.5)
    pos = v * r[:, None]
    nrm, uv, col = _surface_attrs(pos, seed, color_components)
    return Mesh(pos, f, nrm, col, uv)


def _delaunay_2d(pts):
    """Delaunay triangles of 2-D points, counter-clockwise, canonical order.  scipy (qhull) when present; else a plain
    Bowyer-Watson (slow: thousands of points take seconds) - for points in general position both give the same set."""
    pts = np.ascontiguousarray(pts, dtype=np.float64)
    key = hashlib.sha256(pts.tobytes()).hexdigest()[:16]
    if key in _DELAUNAY_STORE:                                   # a fixture's point set: the triangulation its golden blob was made from
        return _DELAUNAY_STORE[key].astype(np.int64)
    try:
        from scipy.spatial import Delaunay
        t = Delaunay(pts).simplices.astype(np.int64)
    except ImportError:                                          # pragma: no cover  (the image has scipy)
        t = _bowyer_watson(pts)
    a, b, c = pts[t[:, 0]], pts[t[:, 1]], pts[t[:, 2]]
    area2 = (b[:, 0] - a[:, 0]) * (c[:, 1] - a[:, 1]) - (b[:, 1] - a[:, 1]) * (c[:, 0] - a[:, 0])
    t = np.where((area2 < 0)[:, None], t[:, [0, 2, 1]], t)
    t = _canonical_faces(t)
    if _DELAUNAY_RECORD is not None:
        _DELAUNAY_RECORD[key] = t.astype(np.int32)
    return t


# The point sets above come out bit-identical on any host (IEEE + - * / only), a triangulator's answer for nearly cocircular points need not:
# the golden fixtures made from Delaunay meshes carry their triangulations (tests/golden/delaunay_tris.npz, written by make_golden.py with
# _DELAUNAY_RECORD set, loaded by tests/conftest.py), so a test that REBUILDS such a mesh encodes the mesh the reference encoded whatever
# scipy / qhull build is installed.  bench.py and the stress tools triangulate afresh: they compare with the oracle, not with stored bytes.
_DELAUNAY_STORE = {}
_DELAUNAY_RECORD = None


def load_delaunay_store(path):
    z = np.load(path)
    for k in z.files:
        _DELAUNAY_STORE[k] = z[k]


def _bowyer_watson(pts):
    n = len(pts)
    lo, hi = pts.min(0), pts.max(0)
    c, d = 0.5 * (lo + hi), float((hi - lo).max()) * 20.0 + 1.0
    P = np.concatenate([pts, [[c[0] - d, c[1] - d], [c[0] + d, c[1] - d], [c[0], c[1] + d]]])
    tris = {(n, n + 1, n + 2)}

    def circ(t):
        ax, ay = P[t[0]]; bx, by = P[t[1]]; cx, cy = P[t[2]]
        dd = 2.0 * (ax * (by - cy) + bx * (cy - ay) + cx * (ay - by))
        ux = ((ax * ax + ay * ay) * (by - cy) + (bx * bx + by * by) * (cy - ay) + (cx * cx + cy * cy) * (ay - by)) / dd
        uy = ((ax * ax + ay * ay) * (cx - bx) + (bx * bx + by * by) * (ax - cx) + (cx * cx + cy * cy) * (bx - ax)) / dd
        return ux, uy, (ax - ux) ** 2 + (ay - uy) ** 2
    cc = {(n, n + 1, n + 2): circ((n, n + 1, n + 2))}
    for i in range(n):
        x, y = P[i]
        bad = [t for t in tris if (x - cc[t][0]) ** 2 + (y - cc[t][1]) ** 2 < cc[t][2]]
        edges = {}
        for t in bad:
            for e in ((t[0], t[1]), (t[1], t[2]), (t[2], t[0])):
                k = (min(e), max(e))
                edges[k] = None if k in edges else e
            tris.discard(t); cc.pop(t)
        for e in edges.values():
            if e is not None:
                t = (e[0], e[1], i)
                tris.add(t); cc[t] = circ(t)
    return np.array([t for t in tris if max(t) < n], dtype=np.int64).reshape(-1, 3)


def delaunay_disc(n=1100, seed=0, holes=6, color_components=4, drop=0.25):
    """Delaunay triangulation of a jittered point set in a disc, with circular holes punched out: valence 3..10, a boundary ring, one more
    boundary loop per hole, the odd pinched vertex where two holes nearly touch - connectivity with no lattice structure at all.
    About n vertices and 1.9 n triangles (n = 2200 -> ~4 100 triangles, the C4 unit's size).  The jitter keeps every pair of points at least
    a tenth of the grid pitch apart (no zero-area quantised triangle)."""
    side = int(np.ceil(np.sqrt(n / (0.7854 * (1.0 - drop))))) + 1
    gx, gy = np.meshgrid(np.arange(side, dtype=np.float64), np.arange(side, dtype=np.float64), indexing="xy")
    pitch = 2.0 / (side - 1)
    jit = _lcg_fast(seed * 3 + 1, 2 * side * side).reshape(2, side, side) - 0.5
    x = -1.0 + (gx + 0.5 * (np.floor(gy) % 2) + 0.8 * jit[0]) * pitch           # staggered rows: no square cells to begin with
    y = -1.0 + (gy + 0.8 * jit[1]) * pitch
    keep = (_lcg_fast(seed * 3 + 2, side * side).reshape(side, side) >= drop) & (x * x + y * y < (1.0 - 1.2 * pitch) ** 2)
    nb = max(12, int(3.2 / pitch))
    sb, cb = _sincos(np.arange(nb, dtype=np.float64) * (_TWO_PI / nb))
    pts = np.concatenate([np.stack([x[keep], y[keep]], -1), np.stack([cb, sb], -1)])
    tris = _delaunay_2d(pts)
    if holes:
        h = _lcg_fast(seed * 3 + 3, 3 * holes).reshape(holes, 3)
        hc = (h[:, :2] - 0.5) * 1.5
        hr = 0.06 + 0.16 * h[:, 2]
        cen = pts[tris].mean(1)
        inside = (((cen[:, None, :] - hc[None]) ** 2).sum(-1) < (hr * hr)[None]).any(1)
        tris = tris[~inside]
    s3, _ = _sincos(3.0 * pts[:, 0])
    _, c2 = _sincos(2.5 * pts[:, 1])
    z = 0.25 * s3 * c2 + 0.3 * (1.0 - pts[:, 0] ** 2 - pts[:, 1] ** 2)
    pos, tris = _compact(np.concatenate([pts, z[:, None]], 1), tris)
    nrm, uv, col = _surface_attrs(pos + [0, 0, 1.0], seed, color_components)
    return Mesh(pos, tris, nrm, col, uv)


def cone_fan(k=96, rings=3, seed=0, color_components=4, closed=True, flip=0.3):
    """A cone: an apex of valence k (>= 64 asks more of the automaton's chain ends and of K-NRM's incidence lists than any grid), `rings`
    rings of k vertices under it with randomly flipped diagonals, and (closed) a second valence-k fan at the base."""
    sa, ca_ = _sincos(np.arange(k, dtype=np.float64) * (_TWO_PI / k))
    P = [[0.0, 0.0, 1.0]]
    for r in range(1, rings + 1):
        rad = r / rings + 0.1 * (_lcg_fast(seed + r, k) - 0.5) / rings
        P += np.stack([rad * ca_, rad * sa, np.full(k, 1.0 - r / rings) + 0.05 * (_lcg_fast(seed + 50 + r, k) - 0.5)], -1).tolist()
    ar = np.arange(k)
    T = [np.stack([np.zeros(k, dtype=np.int64), 1 + ar, 1 + (ar + 1) % k], -1)]
    for r in range(rings - 1):
        a, b = 1 + r * k + ar, 1 + r * k + (ar + 1) % k
        d, c = a + k, b + k
        f = (_lcg_fast(seed * 31 + r, k) < flip)[:, None]
        T.append(np.where(f, np.stack([a, d, b], -1), np.stack([a, d, c], -1)))
        T.append(np.where(f, np.stack([b, d, c], -1), np.stack([a, c, b], -1)))
    if closed:
        base = len(P)
        P.append([0.0, 0.0, -0.3])
        a, b = 1 + (rings - 1) * k + ar, 1 + (rings - 1) * k + (ar + 1) % k
        T.append(np.stack([np.full(k, base), b, a], -1))
    pos = np.array(P, dtype=np.float64)
    nrm, uv, col = _surface_attrs(pos, seed, color_components)
    return Mesh(pos, np.concatenate(T), nrm, col, uv)


def decimated(mesh: Mesh, keep=0.5, seed=0, max_valence=24):
    """Random half-edge collapses (link condition: the edge's end points share exactly their two opposite vertices) until `keep` of
    the vertices are left: valences spread to 3..max_valence the way a decimated scan's do.  Input must be a closed manifold."""
    faces = [list(map(int, f)) for f in mesh.index]
    nv = mesh.nvert
    vf = [set() for _ in range(nv)]
    for i, f in enumerate(faces):
        for v in f:
            vf[v].add(i)
    alive = np.ones(nv, dtype=bool)
    dead_face = set()

    def ring(v):
        s = set()
        for i in vf[v]:
            s.update(faces[i])
        s.discard(v)
        return s
    target = max(4, int(nv * keep))
    left = nv
    order = np.argsort(_lcg_fast(seed + 17, nv * 8), kind="stable")
    for t in order:
        if left <= target:
            break
        v = int(t % nv)
        if not alive[v] or not vf[v]:
            continue
        rv = ring(v)
        cand = sorted(rv)
        u = cand[int(t // nv) % len(cand)]
        ru = ring(u)
        shared = rv & ru
        if len(shared) != 2 or len(ru) + len(rv) - 4 > max_valence or len(rv) <= 3 or len(ru) <= 3:
            continue
        if any(len(ring(s)) <= 3 for s in shared):            # the opposite vertices lose one neighbour each: keep valence >= 3
            continue
        for i in list(vf[v]):                                     # v -> u
            f = faces[i]
            if u in f:
                dead_face.add(i)
                for w in f:
                    vf[w].discard(i)
            else:
                f[f.index(v)] = u
                vf[u].add(i)
        vf[v] = set()
        alive[v] = False
        left -= 1
    tris = np.array([f for i, f in enumerate(faces) if i not in dead_face], dtype=np.int64).reshape(-1, 3)
    pos, tris = _compact(mesh.position.astype(np.float64), tris)
    sel = alive
    nrm = None if mesh.normal is None else mesh.normal[sel]
    col = None if mesh.color is None else mesh.color[sel]
    uv = None if mesh.uv is None else mesh.uv[sel]
    assert len(pos) == int(sel.sum())
    return Mesh(pos, tris, nrm, col, uv)


def confetti(ncomp=240, seed=0, color_components=4, max_faces=12):
    """Hundreds of tiny components of 1..max_faces faces each (open fans, closed fans, strips, tetrahedra, octahedra) scattered over a
    plane: every component costs the automaton a seed face and its chain ends, and the front never grows past a dozen edges."""
    side = int(np.ceil(np.sqrt(ncomp)))
    kind = (_lcg_fast(seed + 1, ncomp) * 5).astype(np.int64)
    size = 1 + (_lcg_fast(seed + 2, ncomp) * max_faces).astype(np.int64)
    jit = _lcg_fast(seed + 3, ncomp * 64).reshape(ncomp, 64) - 0.5
    P, T, off = [], [], 0
    for c in range(ncomp):
        cx, cy = float(c % side), float(c // side)
        f = int(min(size[c], max_faces))
        k = int(kind[c])
        if k == 3 and f < 4:
            k = 0
        if k == 4 and f < 8:
            k = 1
        if k == 2 and f < 3:
            k = 0
        if k == 0:                                   # open fan: centre + f+1 rim points over 300 degrees
            s, co = _sincos(np.arange(f + 1, dtype=np.float64) * (5.2 / max(f, 1)))
            p = np.concatenate([[[0, 0, 0.1]], np.stack([co, s, np.zeros(f + 1)], -1)])
            t = np.stack([np.zeros(f, dtype=np.int64), 1 + np.arange(f), 2 + np.arange(f)], -1)
        elif k == 1:                                 # strip of f triangles
            i = np.arange(f + 2, dtype=np.float64)
            p = np.stack([np.floor(i / 2) * (2.0 / max(f // 2 + 1, 1)) - 1.0, (i % 2) * 0.8 - 0.4, 0.05 * (i % 3)], -1)
            a = np.arange(f)
            t = np.where((a % 2 == 0)[:, None], np.stack([a, a + 1, a + 2], -1), np.stack([a + 1, a, a + 2], -1))
        elif k == 2:                                 # closed fan (a disc): hub of valence f
            s, co = _sincos(np.arange(f, dtype=np.float64) * (_TWO_PI / f))
            p = np.concatenate([[[0, 0, 0.2]], np.stack([co, s, np.zeros(f)], -1)])
            a = np.arange(f)
            t = np.stack([np.zeros(f, dtype=np.int64), 1 + a, 1 + (a + 1) % f], -1)
        elif k == 3:                                 # tetrahedron (closed, 4 faces)
            p = np.array([[1, 1, 1], [1, -1, -1], [-1, 1, -1], [-1, -1, 1]], dtype=np.float64) * 0.6
            t = np.array([[0, 1, 2], [0, 3, 1], [0, 2, 3], [1, 3, 2]], dtype=np.int64)
        else:                                        # octahedron (closed, 8 faces)
            p = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], dtype=np.float64) * 0.8
            t = np.array([[0, 2, 4], [2, 1, 4], [1, 3, 4], [3, 0, 4], [2, 0, 5], [1, 2, 5], [3, 1, 5], [0, 3, 5]], dtype=np.int64)
        p = p * 0.4 + 0.06 * jit[c, :p.size].reshape(p.shape) + [cx, cy, 0.0]
        P.append(p); T.append(t + off); off += len(p)
    pos = np.concatenate(P)
    nrm, uv, col = _surface_attrs(pos + [0, 0, 5.0], seed, color_components)
    return Mesh(pos, np.concatenate(T), nrm, col, uv)


def full_width_values(mesh: Mesh, seed=0, magnitude=2.0 ** 30):
    """Positions and uvs whose quantised values (q = 1) alternate around +-`magnitude`.  2^30: neighbour differences need all 32 bits of a
    bit field (upstream's needed() returns 32 for |v| >= 2^30, cstream.h:105-112) - the edge of the bit reader and of `(1<<diff)>>1`.
    2^28.6: differences of 2^29..2^30 need 31 bits, where that int expression is INT_MIN >> 1 = -2^30 (fixture fields31)."""
    with np.errstate(over="ignore"):
        h = _lcg_fast(seed + 5, mesh.nvert * 10).reshape(mesh.nvert, 10)
    sign = np.where(h[:, :5] < 0.5, -1.0, 1.0)
    mesh.position = np.ascontiguousarray((sign[:, :3] * (magnitude + np.floor(h[:, 5:8] * 2.0 ** 20))).astype(np.float32))
    if mesh.uv is not None:
        mesh.uv = np.ascontiguousarray((sign[:, 3:5] * 2.0 ** 18 * (1.0 + np.floor(h[:, 8:10] * 4095.0))).astype(np.float32))
    return mesh


def non_manifold(base: Mesh, seed=0, fins=8, dups=6, reversed_dups=6, bowties=3, glue=4, shuffle_faces=True):
    """`base` with what scanned / merged / badly exported models carry: **fins** (a third face on an edge, to a new vertex or to an
    existing one), **duplicated** faces, **reversed duplicates** (the same three vertices wound the other way), **bow-tie** vertices
    (a triangle that touches the surface in one vertex only) and **glued** pairs (two new faces back to back on an existing edge).
    Upstream's encoder pairs at most two faces on an edge, by the order std::sort leaves them in (src/encoder.cpp:450-504), and
    writes BOUNDARY where the face across an edge has been visited already ("glue", src/encoder.cpp:633-636) - so the decoder sees
    chain ends in the middle of a surface, components of one or two faces and vertices re-emitted under new ids."""
    rng_i = lambda k, n, hi: (_lcg_fast(seed * 131 + k, n) * hi).astype(np.int64)
    pos = [base.position.astype(np.float64)]
    nv = base.nvert
    idx = base.index.astype(np.int64)
    nf = len(idx)
    extra = []
    ext = float((base.position.max(0) - base.position.min(0)).max()) or 1.0
    newp = []

    def new_vertex(p):
        newp.append(p)
        return nv + len(newp) - 1
    jit = _lcg_fast(seed * 131 + 99, 3 * (fins + bowties * 2 + glue) + 3).reshape(-1, 3) - 0.5
    j = 0
    for t, f in enumerate(rng_i(1, fins, nf)):
        a, b, c = idx[f]
        if t % 3 == 2:                                  # to an existing vertex (whichever the hash picks: often far away)
            w = int(rng_i(2 + t, 1, nv)[0])
            if w in (a, b):
                continue
            extra.append([a, b, w])
        else:
            p = (pos[0][a] + pos[0][b]) * 0.5 + (0.03 + 0.05 * (t % 4)) * ext * (jit[j] + [0, 0, 0.7]); j += 1
            extra.append([a, b, new_vertex(p)] if t % 2 else [b, a, new_vertex(p)])
    for f in rng_i(3, dups, nf):
        extra.append(list(idx[f]))
    for f in rng_i(4, reversed_dups, nf):
        extra.append(list(idx[f][::-1]))
    for t, v in enumerate(rng_i(5, bowties, nv)):
        p = pos[0][v] + 0.04 * ext * (jit[j] + [0.6, 0, 0.6]); q = pos[0][v] + 0.04 * ext * (jit[j + 1] + [0, 0.6, 0.6]); j += 2
        extra.append([int(v), new_vertex(p), new_vertex(q)])
    for t, f in enumerate(rng_i(6, glue, nf)):
        a, b, c = idx[f]
        w = new_vertex((pos[0][a] + pos[0][b]) * 0.5 + 0.05 * ext * (jit[j] + [0, 0, -0.8])); j += 1
        extra += [[a, b, w], [b, a, w]]
    extra = np.array([e for e in extra if len(set(int(x) for x in e)) == 3], dtype=np.int64).reshape(-1, 3)
    P = np.concatenate([pos[0], np.array(newp, dtype=np.float64).reshape(-1, 3)])
    I = np.concatenate([idx, extra])
    if shuffle_faces:
        I = I[np.argsort(_lcg_fast(seed * 131 + 7, len(I)), kind="stable")]
    cc = 4 if base.color is None else base.color.shape[1]
    n2, uv2, col2 = _surface_attrs(P, seed, cc)
    k = len(newp)

    def grown(old, new):
        return None if old is None else np.concatenate([old, new[len(new) - k:].astype(old.dtype)]) if k else old
    return Mesh(P, I, grown(base.normal, n2), grown(base.color, col2), grown(base.uv, uv2))
