"""Deterministic synthetic meshes / point clouds for tests and bench.

No sample meshes ship with the reference (bun_zipper.ply is absent), so every input is
synthetic.  Everything here uses only IEEE +,-,*,/ on float64 (own polynomial sin/cos, integer LCG
noise) so the same arrays come out bit-identical on any host: the GPU box regenerates the very
same inputs the fixtures were made from.
"""
from __future__ import annotations

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
