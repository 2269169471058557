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
        # 31-bit fields: `(1<<diff)>>1` in int is INT_MIN >> 1 = -2^30 there (cstream.h:343) - upstream's bytes, not a round trip
        ("fields31", S.full_width_values(S.bumpy_sphere(9, 7, seed=38), seed=38, magnitude=2.0 ** 28.6), dict(normal_prediction=DIFF, position_bits=0, position_q=1.0, uv_bits=0)),
        ("fields32", S.full_width_values(S.bumpy_sphere(9, 7, seed=37), seed=37), dict(normal_prediction=DIFF, position_bits=0, position_q=1.0, uv_bits=0)),
        # non-manifold input (round 6): fins, duplicated and reversed faces, bow-tie vertices, glued pairs - upstream's encoder pairs two faces an
        # edge by std::sort's order and writes BOUNDARY for "glue" (src/encoder.cpp:450-504,633-636).  DIFF keeps every normal in contract; the
        # ESTIMATED one carries vertices whose face normals cancel (0/0 inside the estimate, finite output); BORDER without back-to-back pairs
        ("nonmanifold_fins", S.non_manifold(S.delaunay_disc(700, seed=41, holes=4), seed=41, fins=12, dups=8, reversed_dups=8, bowties=4, glue=5), dict(normal_prediction=DIFF)),
        ("nonmanifold_glued", S.non_manifold(S.bumpy_sphere_flipped(28, 14, seed=42), seed=42, fins=9, dups=10, reversed_dups=10, bowties=3, glue=8), dict(normal_prediction=ESTIMATED)),
        ("nonmanifold_border", S.non_manifold(S.merge([S.icosphere(2, seed=43), S.cone_fan(40, 2, seed=43)]), seed=43, fins=10, dups=6, reversed_dups=6, bowties=5, glue=0, shuffle_faces=False), dict(normal_prediction=BORDER, position_bits=12)),
        ("cloud_diff", S.point_cloud(96, 64, seed=7), dict(normal_prediction=DIFF)),
        ("cloud_border", S.point_cloud(40, 20, seed=8), dict(normal_prediction=BORDER)),
    ]
