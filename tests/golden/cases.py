"""The named fixture cases: (name, mesh, encoder keyword arguments).  Shared by make_golden.py (which runs the
REFERENCE encoder/decoder on them) and by the tests (which rebuild the same meshes for the repo's own encoder)."""
import numpy as np

from corto_amd import synth as S

DIFF, ESTIMATED, BORDER = 0, 1, 2


def cases():
    two_groups = S.bumpy_sphere(32, 16, seed=11)
    two_groups.groups = [400, two_groups.nface]
    group_props = S.bumpy_sphere(24, 12, seed=12, color_components=3)
    group_props.groups = [100, 101, 400, group_props.nface]
    group_props.group_props = [{"material": "skin", "texture": "t0.png"}, {}, {"z": "last", "a": "first", "m": ""}, {"material": "cloth"}]
    multi = S.merge([S.closed_sphere(20, 10, seed=1), S.torus(16, 8, seed=2), S.holey_disc(14, seed=3, color_components=4)])
    radius = S.bumpy_sphere(24, 12, seed=21)
    radius.radius = (0.25 + np.arange(radius.nvert, dtype=np.float32) % 17).reshape(-1, 1)
    return [
        ("pos_only", S.bumpy_sphere(64, 32, seed=1), dict(with_normal=False, with_color=False, with_uv=False)),
        ("nrm_diff", S.bumpy_sphere(64, 32, seed=2), dict(normal_prediction=DIFF, with_color=False)),
        ("nrm_estimated_rgb", S.bumpy_sphere(64, 32, seed=3, color_components=3), dict(normal_prediction=ESTIMATED)),
        ("c4_unit", S.bumpy_sphere(64, 32, seed=0), dict(normal_prediction=BORDER)),
        ("two_groups", two_groups, dict(normal_prediction=BORDER)),
        ("group_props", group_props, dict(normal_prediction=ESTIMATED)),
        ("holey_disc", S.shuffled(S.holey_disc(40, seed=5), seed=3), dict(normal_prediction=BORDER)),
        ("multi_component", multi, dict(normal_prediction=ESTIMATED)),
        ("torus", S.torus(48, 24, seed=4), dict(normal_prediction=BORDER)),
        ("closed_sphere", S.closed_sphere(32, 16, seed=6), dict(normal_prediction=DIFF)),
        ("radius_attr", radius, dict(normal_prediction=BORDER, exif={"mtllib": "a.mtl", "note": "x"})),
        ("entropy_none", S.bumpy_sphere(32, 16, seed=9), dict(normal_prediction=BORDER, entropy=0)),
        # non-lattice connectivity (round 5): valence 5/6 without a grid, Delaunay with holes, a valence-96 apex, a decimated sphere, confetti
        ("icosphere", S.icosphere(3, seed=31), dict(normal_prediction=BORDER)),
        ("delaunay_holes", S.delaunay_disc(1100, seed=32, holes=7), dict(normal_prediction=BORDER)),
        ("delaunay_shuffled", S.shuffled(S.delaunay_disc(500, seed=33, holes=4, color_components=3), seed=5), dict(normal_prediction=ESTIMATED, position_bits=12)),
        ("cone_fan", S.cone_fan(96, 3, seed=34), dict(normal_prediction=ESTIMATED)),
        ("decimated", S.decimated(S.icosphere(3, seed=35), keep=0.45, seed=35), dict(normal_prediction=DIFF)),
        ("confetti", S.confetti(240, seed=36), dict(normal_prediction=BORDER)),
        # bit fields of the full 32 bits (|values| >= 2^30), correlated (position) and per component (uv)
        ("fields32", S.full_width_values(S.bumpy_sphere(9, 7, seed=37), seed=37), dict(normal_prediction=DIFF, position_bits=0, position_q=1.0, uv_bits=0)),
        ("cloud_diff", S.point_cloud(96, 64, seed=7), dict(normal_prediction=DIFF)),
        ("cloud_border", S.point_cloud(40, 20, seed=8), dict(normal_prediction=BORDER)),
    ]
