import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# fixtures made from Delaunay meshes carry their triangulations: a test that rebuilds such a mesh gets the one the golden blob was made from
from corto_amd import synth as _synth  # noqa: E402
if os.path.exists(os.path.join(GOLDEN, "delaunay_tris.npz")):
    _synth.load_delaunay_store(os.path.join(GOLDEN, "delaunay_tris.npz"))

MESH_CASES = ["pos_only", "nrm_diff", "nrm_estimated_rgb", "c4_unit", "two_groups", "group_props", "holey_disc",
              "multi_component", "torus", "closed_sphere", "radius_attr", "entropy_none",
              "icosphere", "delaunay_holes", "delaunay_shuffled", "cone_fan", "decimated", "confetti", "fields32", "fields31",
              "nonmanifold_fins", "nonmanifold_glued", "nonmanifold_border"]
CLOUD_CASES = ["cloud_diff", "cloud_border"]
ALL_CASES = MESH_CASES + CLOUD_CASES


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def aligned(b: np.ndarray, align=16) -> np.ndarray:
    raw = np.zeros(len(b) + align, dtype=np.uint8)
    off = (-raw.ctypes.data) % align
    v = raw[off:off + len(b)]
    v[:] = b
    return v


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    d = {k: z[k] for k in z.files}
    if "crt" in d:
        d["crt"] = aligned(d["crt"])
    return d


@pytest.fixture(scope="session")
def have_ref():
    from oracle import refcodec
    return refcodec.available()
