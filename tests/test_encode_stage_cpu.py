"""CPU: the formulation behind the GPU Tunstall ENCODER stage (k_encode.hip: 64 start positions per window walk the trie,
then the chain cur -> next[cur]) on the library's own host-made tables reproduces, byte for byte, the blocks the
reference's OutStream::tunstall_compress wrote (tests/golden/tunstall_enc_kat.npz, made by make_tunstall_enc.py)."""
import os
import struct
import subprocess

import numpy as np
import pytest

import corto_amd as ca
from conftest import ROOT

GOLDEN = os.path.join(ROOT, "tests", "golden")


def _kat():
    z = np.load(os.path.join(GOLDEN, "tunstall_enc_kat.npz"))
    n = int(z["count"])
    return [z["input_%02d" % i] for i in range(n)], [z["block_%02d" % i] for i in range(n)]


@pytest.fixture(scope="module")
def model(tmp_path_factory):
    from corto_amd import build
    build.build()
    exe = str(tmp_path_factory.mktemp("enc") / "enc_window_model")
    libdir = os.path.dirname(ca.LIB_PATH)
    subprocess.check_call([build.hipcc(), "-O1", "-std=c++17", "-x", "hip", "--offload-arch=gfx950", os.path.join(ROOT, "tests", "cpp", "enc_window_model.cpp"),
                           "-o", exe, "-L", libdir, "-lcorto_hip", "-Wl,-rpath," + libdir])
    return exe


def _run(model, streams):
    inp = struct.pack("<I", len(streams)) + b"".join(struct.pack("<I", len(s)) + np.asarray(s, dtype=np.uint8).tobytes() for s in streams)
    r = subprocess.run([model], input=inp, capture_output=True)
    assert r.returncode == 0
    return r.stdout


def test_window_parse_model_reproduces_the_reference_blocks(model):
    streams, blocks = _kat()
    assert _run(model, streams) == b"".join(b.tobytes() for b in blocks)


def test_window_parse_model_round_trips_through_the_oracle(model):
    """random streams of every flavour: the model's block decodes back to the input with the C oracle's Tunstall decoder"""
    from oracle import oracle as oc
    rng = np.random.default_rng(77)
    streams = []
    for k in range(40):
        n = int(rng.integers(1, 6000))
        nsym = int(rng.integers(1, 40))
        p = rng.dirichlet(np.full(nsym, 0.3 if k % 2 else 2.0))
        streams.append(rng.choice(np.arange(nsym, dtype=np.uint8) * 3, n, p=p))
    out = _run(model, streams)
    o = 0
    for s in streams:
        ns = out[o]
        probs = np.frombuffer(out, dtype=np.uint8, count=2 * ns, offset=o + 1).reshape(-1, 2)
        size, csize = struct.unpack_from("<II", out, o + 1 + 2 * ns)
        payload = np.frombuffer(out, dtype=np.uint8, count=csize, offset=o + 9 + 2 * ns)
        assert size == len(s)
        dec = oc.tunstall_decompress(probs, payload, size) if ns > 1 else np.full(size, probs[0, 0], dtype=np.uint8)
        assert np.array_equal(dec, s)
        o += 9 + 2 * ns + csize
    assert o == len(out)


def test_std_sort_model_matches_libstdcxx(tmp_path):
    """csrc/std_sort_model.h (what k_enc_tables runs on the device to order equal probabilities) against std::sort itself:
    80 000 tie-heavy arrays, incl. the heapsort branch (tests/cpp/sort_model_test.cpp)"""
    import subprocess
    exe = str(tmp_path / "sort_model_test")
    subprocess.check_call(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "cpp", "sort_model_test.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "mismatches 0" in out.stdout, out.stdout + out.stderr
