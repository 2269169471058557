"""Shared by the CLI parity tests: PLY writers for synthetic meshes and runners for the two command line tools."""
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_CLI = os.path.join(ROOT, "oracle", "_ref", "corto_ref_cli")


def our_cli():
    from corto_amd import build
    if not os.path.exists(build.CLI):
        build.build()
    return build.CLI


def write_ply(path, m, binary=True, with_normal=True, with_color=True, with_uv=True, uv_names=("texture_u", "texture_v"),
              radius=None, faces=True, index_name="vertex_indices", double_xyz=False):
    """binary_little_endian or ascii PLY with the property names upstream's loader asks for (src/meshloader.cpp:52-64)"""
    nv = m.nvert
    nf = m.nface if faces else 0
    xyz = "double" if double_xyz else "float"
    hdr = ["ply", "format %s 1.0" % ("binary_little_endian" if binary else "ascii"), "comment made by tests/cli_common.py",
           "element vertex %d" % nv, "property %s x" % xyz, "property %s y" % xyz, "property %s z" % xyz]
    fields = [("p", "<f8" if double_xyz else "<f4", 3)]
    cols = [m.position.astype(np.float64 if double_xyz else np.float32)]
    if with_normal:
        hdr += ["property float nx", "property float ny", "property float nz"]; fields.append(("n", "<f4", 3)); cols.append(m.normal)
    if with_color:
        hdr += ["property uchar red", "property uchar green", "property uchar blue", "property uchar alpha"]
        fields.append(("c", "u1", 4)); cols.append(m.color)
    if with_uv:
        hdr += ["property float %s" % uv_names[0], "property float %s" % uv_names[1]]; fields.append(("t", "<f4", 2)); cols.append(m.uv)
    if radius is not None:
        hdr += ["property float radius"]; fields.append(("r", "<f4", 1)); cols.append(radius.reshape(-1, 1))
    if nf:
        hdr += ["element face %d" % nf, "property list uchar int %s" % index_name]
    hdr += ["end_header"]
    with open(path, "wb") as f:
        f.write(("\n".join(hdr) + "\n").encode())
        if binary:
            v = np.zeros(nv, dtype=fields)
            for (name, _, k), c in zip(fields, cols):
                v[name] = np.asarray(c).reshape(v[name].shape)
            f.write(v.tobytes())
            if nf:
                fa = np.zeros(nf, dtype=[("k", "u1"), ("i", "<i4", 3)]); fa["k"] = 3; fa["i"] = m.index
                f.write(fa.tobytes())
        else:
            for i in range(nv):
                row = []
                for (name, t, k), c in zip(fields, cols):
                    row += [repr(float(x)) if t != "u1" else str(int(x)) for x in np.asarray(c).reshape(nv, -1)[i]]
                f.write((" ".join(row) + "\n").encode())
            for i in range(nf):
                f.write(("3 %d %d %d\n" % tuple(int(x) for x in m.index[i])).encode())


def run(cli, args, cwd):
    r = subprocess.run([cli] + list(args), cwd=cwd, capture_output=True, text=True)
    assert r.returncode == 0, (cli, args, r.stdout[-400:], r.stderr[-400:])
    return r.stdout
