"""GPU (-m gpu): the HIP path, called through the C ABI, against (a) the golden outputs the reference produced,
(b) the C oracle on the same inputs, (c) size-independent properties at BASELINE.json's full sizes.
Bit-exact everywhere (integer/byte/index work; floats are compared as raw bytes - tolerance 0)."""
import hashlib
import os

import numpy as np
import pytest

import corto_amd as ca
from conftest import ALL_CASES, CLOUD_CASES, MESH_CASES, GOLDEN, aligned, load_golden
from oracle import oracle as oc

pytestmark = pytest.mark.gpu


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def ctx():
    c = ca.Context(0)
    yield c
    c.close()


def run_batch(ctx, blobs, **kw):
    kw.setdefault("fill", 0)      # untouched outputs (e.g. cloud normals in BORDER mode) stay as the caller left them
    b = ca.Batch(ctx, blobs)
    b.allocate_outputs(**kw)
    b.decode()
    st = b.sync()
    assert (st == 0).all(), st
    return b


def assert_same(got, exp, keys, tag=""):
    for k in keys:
        if k in exp:
            assert k in got, (tag, k)
            a, e = got[k], exp[k]
            assert a.dtype == e.dtype and a.shape == e.shape, (tag, k, a.dtype, e.dtype, a.shape, e.shape)
            if a.tobytes() != e.tobytes():
                bad = np.argwhere(a.reshape(len(a), -1) != e.reshape(len(e), -1))
                raise AssertionError("%s %s: %d mismatching entries, first %s got %s expected %s" % (
                    tag, k, len(bad), bad[0], a.reshape(len(a), -1)[bad[0][0]], e.reshape(len(e), -1)[bad[0][0]]))


KEYS = ("position", "normal", "color", "uv", "radius", "index")


@pytest.mark.parametrize("name", ALL_CASES)
def test_golden_single_blob(ctx, name):
    g = load_golden(name)
    cc = int(g["color_components"])
    b = run_batch(ctx, [g["crt"]], color_components=cc if "color" in g else None)
    got = b.host_outputs(0)
    if "index" in g:                                   # stage-level: CLERS symbols + prediction triples
        cl = b.debug_read(0, "clers", len(g["_clers"]) + 16)
        assert np.array_equal(cl, g["_clers"])
        pr = b.debug_read(0, "prediction", g["_prediction"].size * 4).view(np.uint32).reshape(-1, 3)
        assert np.array_equal(pr[1:], g["_prediction"][1:])
    assert_same(got, g, KEYS, name)
    o = oc.decode(g["crt"], color_components=cc)
    assert_same(got, o, KEYS, name + "/oracle")


@pytest.mark.parametrize("name", ALL_CASES)
def test_golden_int16_normals_u16_index(ctx, name):
    g = load_golden(name)
    cc = int(g["color_components"])
    b = run_batch(ctx, [g["crt"]], normal_format=ca.FMT_INT16, index16=True, color_components=cc if "color" in g else None)
    got = b.host_outputs(0)
    if "normal_i16" in g:
        assert got["normal"].tobytes() == g["normal_i16"].tobytes()
    if "index_u16_sha256" in g:
        assert sha(got["index"]) == g["index_u16_sha256"].tobytes().decode()


def test_heterogeneous_batch(ctx):
    """all fixtures (meshes, clouds, entropy NONE, 3/4-component colours) in ONE batch / one set of launches"""
    gs = [load_golden(n) for n in ALL_CASES]
    b = run_batch(ctx, [g["crt"] for g in gs])
    for i, g in enumerate(gs):
        assert_same(b.host_outputs(i), g, KEYS, ALL_CASES[i])


def test_bit_unpack_one_wave_per_stream_and_chunked(monkeypatch):
    """K-BIT has two kernels: the log streams of a small bit block (<= 16 384 logs) take one wave each, lanes interleaved over the
    vertices, the bits in front of a stream re-added from the earlier streams' logs (k_unpack_wave); larger blocks go in chunks of
    1 024 with look-back (k_unpack_extract; $CORTO_UNPACK_CHUNKED=1 sends everything there).  Same bytes from both, and the
    oracle's: ragged sizes around the 64-lane round and the 512-log block, 1..4 fields per log, u8 and int32 outputs, a cloud."""
    from corto_amd import synth
    meshes = [synth.bumpy_sphere(nu, nv, seed=nu) for nu, nv in ((3, 2), (7, 3), (8, 7), (9, 7), (21, 3), (32, 15), (32, 16), (33, 16), (64, 32), (70, 60), (127, 120))]
    meshes += [synth.holey_disc(12, seed=2), synth.strip(130, seed=3), synth.torus(20, 9, seed=4), synth.point_cloud(40, 13, seed=5), synth.point_cloud(150, 120, seed=6)]
    blobs = []
    for i, m in enumerate(meshes):
        kw = dict(position_bits=(10, 14, 18)[i % 3], uv_bits=12, normal_bits=10, normal_prediction=(ca.BORDER, ca.ESTIMATED, ca.DIFF)[i % 3])
        blobs.append(ca.encode(m, **kw))
    refs = [oc.decode(b, color_components=4) for b in blobs]
    for chunked in ("0", "1"):
        monkeypatch.setenv("CORTO_UNPACK_CHUNKED", chunked)
        c = ca.Context(0)
        b = run_batch(c, blobs, color_components=4)
        for i, r in enumerate(refs):
            assert_same(b.host_outputs(i), r, KEYS, "mesh %d chunked=%s" % (i, chunked))
        b.close(); c.close()


@pytest.mark.parametrize("env", [{"CORTO_DELTA_WIDE": "1"}, {"CORTO_UNPACK_CHUNKED": "1"}, {"CORTO_DELTA_WIDE": "1", "CORTO_TUN_SHARE": "2"}, {"CORTO_DELTA_ROUNDS": "1"},
                                 {"CORTO_DELTA_ROUNDS": "1", "CORTO_DELTA_WIDE": "1"}, {"CORTO_VALUES_I32": "1"}],
                         ids=lambda e: "+".join("%s=%s" % kv for kv in e.items()))
def test_context_settings_are_bit_exact(monkeypatch, env):
    """csrc/debug_config.h: the settings a context reads select other kernel paths for the same bytes (32-bit records in K-DELTA's LDS -
    what a context learns from values beyond int16 -, the chunked K-BIT, one dictionary per stream, K-BIT handing 32-bit values on where it would
    hand int16 ones) - every fixture and the 16 C4 blobs
    through each, on a two-stream and on a single-stream context.  (The experiment switches of rounds 2-3 were removed with their kernels.)"""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    z = np.load(os.path.join(GOLDEN, "c4_blobs16.npz"))
    gs = [load_golden(n) for n in ALL_CASES]
    blobs = [g["crt"] for g in gs] + [aligned(z["crt_%02d" % s]) for s in range(16)]
    for single in (False, True):
        c = ca.Context(0)                                  # (the switches are read when a context is made)
        if single:
            c.set_single_stream(True)
        b = run_batch(c, blobs)
        for i, g in enumerate(gs):
            assert_same(b.host_outputs(i), g, KEYS, "%s %s single=%s" % (ALL_CASES[i], env, single))
        for s_ in range(16):
            got = b.host_outputs(len(gs) + s_)
            for k in ("position", "normal", "color", "uv", "index"):
                assert sha(got[k]) == z["%s_sha256_%02d" % (k, s_)].tobytes().decode(), (s_, k, env, single)
        b.close(); c.close()


def test_host_blobs_upload_is_not_waited_for_and_packed_pinned_blobs_go_up_in_place():
    """crthip_batch_create / _reset with host blobs (round 3): the blobs are gathered into the context's pinned image and go up in one
    copy that nobody waits for - so a second batch made on the same context before the first is decoded must not disturb the first's
    bytes (the image is reused only once the upload is through) - and with crthip_ctx_set_packed_host_blobs a batch whose blobs are
    views of ONE pinned buffer (corto_amd.pinned_host_arena) is uploaded straight from it; blobs that are not laid out that way take
    the gathering path whatever the switch says.  Every output against the golden fixtures."""
    names = list(ALL_CASES)
    gs = [load_golden(n) for n in names]
    blobs = [g["crt"] for g in gs]
    rev = blobs[::-1]
    c = ca.Context(0)
    a = ca.Batch(c, blobs)                                   # two uploads queued on one context, nothing decoded yet
    b = ca.Batch(c, rev)
    for bt in (a, b):
        bt.allocate_outputs(fill=0)
    a.decode(); assert (a.sync() == 0).all()
    b.decode(); assert (b.sync() == 0).all()
    for i, g in enumerate(gs):
        assert_same(a.host_outputs(i), g, KEYS, "first batch " + names[i])
        assert_same(b.host_outputs(len(gs) - 1 - i), g, KEYS, "second batch " + names[i])
    a.close(); b.close()
    pin, views = ca.pinned_host_arena(blobs)
    c.set_packed_host_blobs(True)
    for round_ in range(3):                                  # in place, three times over (reset re-uses the batch object)
        bt = ca.Batch(c, views) if round_ == 0 else bt
        if round_:
            bt.reset(views)
        bt.allocate_outputs(fill=0)
        bt.decode(); assert (bt.sync() == 0).all()
        for i, g in enumerate(gs):
            assert_same(bt.host_outputs(i), g, KEYS, "packed " + names[i])
    bt.close()
    sc = ca.Batch(c, rev)                                    # scattered blobs with the switch on: gathered as before
    sc.allocate_outputs(fill=0)
    sc.decode(); assert (sc.sync() == 0).all()
    for i, g in enumerate(gs):
        assert_same(sc.host_outputs(len(gs) - 1 - i), g, KEYS, "scattered " + names[i])
    sc.close(); c.close()


@pytest.mark.parametrize("nu,nv", [(127, 63), (128, 63), (128, 64), (129, 64), (180, 90), (181, 90)])
def test_bit_unpack_kernel_choice_around_the_threshold(ctx, nu, nv):
    """K-BIT picks its kernel per bit block (UNPACK_WAVE_MAX_LOGS = 16 384 logs: one wave per stream below, chunks with look-back above):
    meshes whose uv block (two logs a vertex) and position block (one) sit just under, on and just over it - one blob then has blocks of
    both kinds - against the oracle"""
    from corto_amd import synth
    m = synth.bumpy_sphere(nu, nv, seed=nu + nv)
    blob = ca.encode(m, position_bits=14, uv_bits=12, normal_bits=10, normal_prediction=ca.BORDER)
    r = oc.decode(blob, color_components=4)
    assert 8000 < r["nvert"] < 17000, r["nvert"]
    b = run_batch(ctx, [blob], color_components=4)
    assert_same(b.host_outputs(0), r, KEYS, "%dx%d (%d vertices)" % (nu, nv, r["nvert"]))
    b.close()


def test_pool_takes_packed_pinned_items_in_place(c5_blobs):
    """crthip_pool_set_packed_host_blobs: items whose blobs are views of one pinned buffer in arena layout are uploaded straight from it
    (no device arena, no gathering); every context's last outputs against the oracle, poison behind the arrays"""
    items = [c5_blobs[100:148], c5_blobs[700:748]]
    pins = [ca.pinned_host_arena(it) for it in items]
    pool = ca.Pool([0], threads=2, depth=2)
    pool.set_packed_host_blobs(True)
    rep, _ = pool.run([v for _, v in pins], steps=40, warmup=4, arenas=None)
    assert rep.failed_blobs == 0 and rep.poisoned_lanes == pool.lanes
    for lane in range(pool.lanes):
        it, _slot = pool.lane_item(lane)
        for i in (0, 17, 47):
            ref = oc.decode(items[it][i])
            for k, dt, w in (("position", np.float32, 3), ("uv", np.float32, 2), ("normal", np.float32, 3), ("index", np.uint32, 3)):
                cnt = (ref["nface"] if k == "index" else ref["nvert"]) * w
                assert pool.lane_read(lane, i, k, dt, cnt).tobytes() == ref[k].tobytes(), (lane, i, k)
        assert (pool.lane_read(lane, 0, "#tail", np.uint8, 256) == 0xA5).all()
    pool.set_packed_host_blobs(False)
    rep, _ = pool.run(items, steps=16, warmup=2, arenas=None)      # ... and the same pool back on scattered blobs
    assert rep.failed_blobs == 0
    pool.close()


def test_pool_outputs_to_host_is_the_secondary_region(c5_blobs):
    """crthip_pool_set_outputs_to_host (SURVEY 8d's secondary region): every step ends with a D2H copy of its outputs into the lane's pinned block,
    behind its kernels; what the copies delivered is the oracle's bytes (the mirror is poisoned before every lane's last step), and switching it
    off again leaves the outputs in HBM as before"""
    items = [c5_blobs[300:340], c5_blobs[900:940]]
    pool = ca.Pool([0], threads=2, depth=2)
    pool.set_outputs_to_host(True)
    dts = {"position": (np.float32, 3), "normal": (np.float32, 3), "color": (np.uint8, 4), "uv": (np.float32, 2), "index": (np.uint32, 3)}
    for to_host in (True, False, True):
        pool.set_outputs_to_host(to_host)
        rep, _ = pool.run(items, steps=24, warmup=4, arenas=None)
        assert rep.failed_blobs == 0 and rep.poisoned_lanes == pool.lanes
        for lane in range(pool.lanes):
            it, _slot = pool.lane_item(lane)
            for i in (0, 13, 39):
                ref = oc.decode(items[it][i])
                for k, (dt, w) in dts.items():
                    cnt = (ref["nface"] if k == "index" else ref["nvert"]) * w
                    assert pool.lane_read(lane, i, k, dt, cnt).tobytes() == ref[k].tobytes(), (to_host, lane, i, k)
            assert (pool.lane_read(lane, 0, "#tail", np.uint8, 256) == 0xA5).all()
    pool.close()


def test_pool_render_layouts(c5_blobs):
    """crthip_pool_set_render_layouts: every lane's normals as int16 (upstream's INT16 output) and its index as uint16 (Decoder::setIndex(uint16_t *)),
    with and without the D2H mirror; a blob of 65 536+ vertices keeps its uint32 index; switching the layouts off again gives the float / uint32 bytes"""
    from corto_amd import synth
    big = ca.aligned_blob(ca.encode(synth.bumpy_sphere(300, 230, seed=3), normal_prediction=ca.BORDER))      # 69 300 vertices: no uint16 index
    items = [c5_blobs[100:124] + [big]]
    pool = ca.Pool([0], threads=2, depth=2)
    for render, to_host in ((True, False), (True, True), (False, True)):
        pool.set_render_layouts(render); pool.set_outputs_to_host(to_host)
        rep, _ = pool.run(items, steps=12, warmup=2, arenas=None)
        assert rep.failed_blobs == 0 and rep.poisoned_lanes == pool.lanes
        for lane in range(pool.lanes):
            for i in (0, 11, 24):
                nv = ca.probe(items[0][i]).nvert
                u16 = render and nv < 65536
                ref = oc.decode(items[0][i], normal_format=oc.FMT_INT16 if render else oc.FMT_FLOAT, index16=u16)
                dts = {"position": (np.float32, 3), "normal": (np.int16 if render else np.float32, 3), "color": (np.uint8, 4), "uv": (np.float32, 2),
                       "index": (np.uint16 if u16 else np.uint32, 3)}
                for k, (dt, w) in dts.items():
                    cnt = (ref["nface"] if k == "index" else ref["nvert"]) * w
                    assert pool.lane_read(lane, i, k, dt, cnt).tobytes() == ref[k].tobytes(), (render, to_host, lane, i, k)
    pool.close()


def test_streams_with_the_same_table_share_one_dictionary(monkeypatch):
    """a Tunstall dictionary is a function of the probability table alone (src/tunstall.cpp:125-256), so a batch builds each DISTINCT
    table once and every stream that carries it decodes from that dictionary (k_tun_tables + k_tun_stream_grouped); $CORTO_TUN_SHARE=0
    builds one per stream (k_tun_stream).  Same bytes either way: every fixture twice + the 16 C4 blobs, in one batch."""
    z = np.load(os.path.join(GOLDEN, "c4_blobs16.npz"))
    gs = [load_golden(n) for n in ALL_CASES]
    blobs = [g["crt"] for g in gs] * 2 + [aligned(z["crt_%02d" % s]) for s in range(16)]
    seen = {}
    for share in ("1", "0", "2", None):
        if share is None:
            monkeypatch.delenv("CORTO_TUN_SHARE", raising=False)
        else:
            monkeypatch.setenv("CORTO_TUN_SHARE", share)
        c = ca.Context(0)                                  # (the switch is read when a context is made)
        b = run_batch(c, blobs)
        for i, g in enumerate(gs + gs):
            assert_same(b.host_outputs(i), g, KEYS, "%s share=%s" % (ALL_CASES[i % len(gs)], share))
        for s_ in range(16):
            got = b.host_outputs(2 * len(gs) + s_)
            for k in ("position", "normal", "color", "uv", "index"):
                assert sha(got[k]) == z["%s_sha256_%02d" % (k, s_)].tobytes().decode(), (s_, k, share)
        st = b.stats()
        seen[share] = (st.tunstall_dictionaries, st.tunstall_streams)
        b.close(); c.close()
    assert seen["0"][0] == seen["0"][1] and seen["2"][0] == seen["2"][1], seen      # one dictionary per stream (one kernel / two kernels)
    assert seen["1"][0] < seen["1"][1] // 2, seen                   # every fixture is there twice
    assert seen[None][0] <= seen["0"][0], seen


def test_vertex_counts_around_the_bitmap_words(monkeypatch):
    """K-DELTA's fallback walk keeps its fired flags and stretch starts as bitmaps (one dword per 32 vertices, written 64 vertices a round):
    vertex counts on and around the word boundaries, tiny meshes, a long thin strip (2 049 vertices: 683 stretches) - with 16-bit and with
    32-bit records in LDS ($CORTO_DELTA_WIDE=1), and on a single-stream context (the LDS-lean normals path)"""
    from corto_amd import synth
    meshes = [synth.bumpy_sphere(w, h, seed=w * h) for w, h in ((3, 2), (4, 3), (9, 6), (8, 7), (13, 4), (16, 7), (43, 2), (31, 32), (64, 31), (683, 2))]
    assert [m.nvert for m in meshes] == [9, 16, 63, 64, 65, 128, 129, 1023, 2048, 2049]
    blobs = [ca.encode(m, normal_prediction=ca.ESTIMATED if k % 2 else ca.BORDER) for k, m in enumerate(meshes)]
    refs = [oc.decode(b) for b in blobs]
    for walk, single in (("1", False), ("0", False), ("1", True), ("0", True)):
        monkeypatch.setenv("CORTO_DELTA_WIDE", walk)
        c = ca.Context(0)
        if single:
            c.set_single_stream(True)
        b = run_batch(c, blobs)
        for i, r in enumerate(refs):
            assert_same(b.host_outputs(i), r, KEYS, "nvert %d wide=%s single=%s" % (meshes[i].nvert, walk, single))
        b.close(); c.close()


def test_irregular_connectivity_batch(ctx):
    """48 lat-long grids with every quad's diagonal flipped at random (valences 4-8, no two CLERS streams alike): short (VERTEX LEFT) runs,
    short scan blocks, the walk taking over for some; on a two-stream and on a single-stream (LDS-lean) context"""
    from corto_amd import synth
    meshes = [synth.bumpy_sphere_flipped(64, 32, seed=s) for s in range(40)] + [synth.bumpy_sphere_flipped(24, 12, seed=100 + s, flip=0.3 + 0.1 * s) for s in range(8)]
    blobs = [ca.encode(m, position_bits=14, uv_bits=12, normal_bits=10, normal_prediction=ca.BORDER if k % 2 else ca.ESTIMATED) for k, m in enumerate(meshes)]
    refs = [oc.decode(b) for b in blobs]
    for single in (False, True):
        c = ca.Context(0)
        if single:
            c.set_single_stream(True)
        b = run_batch(c, blobs)
        for i, r in enumerate(refs):
            assert_same(b.host_outputs(i), r, KEYS, "flipped %d single=%s" % (i, single))
        b.close(); c.close()
    # VERTEX / LEFT in any order is the automaton's mix step (one symbol a lane): meshes whose streams outgrow the LDS symbol window
    # (the step runs up against the window's end and the slides), flip rates from one quad in fifty to all, u32 and u16 indices
    big = [synth.bumpy_sphere_flipped(128, 64, seed=7), synth.bumpy_sphere_flipped(200, 100, seed=8, flip=0.1), synth.bumpy_sphere_flipped(90, 45, seed=9, flip=1.0)]
    big += [synth.bumpy_sphere_flipped(64, 32, seed=200 + s, flip=f) for s, f in enumerate((0.02, 0.05, 0.2, 0.8, 0.98))]
    blobs = [ca.encode(m, position_bits=14, uv_bits=12, normal_bits=10, normal_prediction=ca.BORDER) for m in big]
    for u16 in (False, True):
        b = run_batch(ctx, blobs, index16=u16, normal_format=ca.FMT_INT16 if u16 else ca.FMT_FLOAT, color_components=4)
        for i, blob in enumerate(blobs):
            r = oc.decode(blob, index16=u16, normal_format=oc.FMT_INT16 if u16 else oc.FMT_FLOAT, color_components=4)
            assert_same(b.host_outputs(i), r, KEYS, "big flipped %d u16=%s" % (i, u16))
        assert b.stats().topology_fallbacks == 0
        b.close()


def test_chain_end_runs_stay_in_lds(ctx):
    """BOUNDARY / DELAY runs are the automaton's chain-end step (the next live gates of the ring each moved to the pool by its own lane,
    links patched through forwarding marks): meshes full of them - discs with holes, ribbons, several groups, open grids - decode byte
    for byte AND without the HBM redo (a wrong front that only runs out of slots falls back and still decodes right: the counter is the test)"""
    from corto_amd import synth
    meshes = [synth.holey_disc(40, seed=s) for s in range(12)] + [synth.holey_disc(24, seed=50 + s, hole_frac=0.05 + 0.03 * s) for s in range(6)]
    meshes += [synth.strip(200 + 37 * s, seed=s) for s in range(4)] + [synth.bumpy_sphere(64, 32, seed=s) for s in range(4)]
    blobs = [ca.encode(m, position_bits=14, uv_bits=12, normal_bits=10, normal_prediction=ca.BORDER if k % 2 else ca.ESTIMATED) for k, m in enumerate(meshes)]
    blobs += [load_golden(n)["crt"] for n in ("group_props", "two_groups", "multi_component", "holey_disc")]
    c = ca.Context(0)
    run_batch(c, blobs).close()                    # (a fresh context may plan too few slots for a disc full of holes: the first batch teaches it)
    for u16 in (False, True):
        b = run_batch(c, blobs, index16=u16, color_components=4)
        for i, blob in enumerate(blobs):
            r = oc.decode(blob, index16=u16, color_components=4)
            assert_same(b.host_outputs(i), r, KEYS, "chain ends %d u16=%s" % (i, u16))
        assert b.stats().topology_fallbacks == 0
        b.close()
    c.close()


def test_small_closed_meshes_whose_runs_reach_the_current_edge(ctx):
    """A regular run that closes the previous layer's slots right up to the current edge's own `next` reads the link a VERTEX in front of it
    has just written: closed spheres of a few dozen vertices do that at every ring (a grid of thousands never does).  Round 4's lead lane
    (the lone VERTEX riding along with the run step) wrote that link with the step's other writes, after the lanes had read it - the test
    suite was green, tools/stress_topology.py was not; these are its two blobs and the family around them.  Also small tori, closed
    meshes with flipped diagonals and every second one cut into groups."""
    from corto_amd import synth
    meshes = [synth.closed_sphere(nu, nv, seed=nu * 31 + nv) for nu in range(6, 41, 2) for nv in (4, 5, 7, 10, 13, 19)]
    meshes += [synth.closed_sphere(15, 10, seed=1), synth.closed_sphere(9, 7, seed=2)]                     # (270 and 108 faces: the stress run's)
    meshes += [synth.torus(nu, nv, seed=nu + nv) for nu in (6, 9, 14, 23) for nv in (4, 6, 11)]
    meshes += [synth.bumpy_sphere_flipped(nu, nv, seed=nu, flip=f) for nu in (8, 13, 21) for nv in (4, 9) for f in (0.05, 0.5)]
    for k, m in enumerate(meshes):
        if k % 2 and m.nface > 24:
            m.groups = [m.nface // 3, m.nface // 2 + 1, m.nface]
    blobs = [ca.encode(m, position_bits=10 + k % 7, uv_bits=12, normal_bits=10, normal_prediction=[ca.BORDER, ca.ESTIMATED, ca.DIFF][k % 3]) for k, m in enumerate(meshes)]
    c = ca.Context(0)
    run_batch(c, blobs).close()
    for u16 in (False, True):
        b = run_batch(c, blobs, index16=u16, color_components=4)
        for i, blob in enumerate(blobs):
            r = oc.decode(blob, index16=u16, color_components=4)
            assert_same(b.host_outputs(i), r, KEYS, "small closed mesh %d (%d faces) u16=%s" % (i, r["nface"], u16))
        assert b.stats().topology_fallbacks == 0
        b.close()
    c.close()


@pytest.mark.parametrize("seed", [5, 11, 23])
def test_random_mesh_stress_run(ctx, seed):
    """tools/stress_topology.py as a test: 21 rounds of 24 random meshes (1 008 decodes a seed) of every synthetic family - the lattices
    (sizes, flip rates, hole fractions, random group cuts, shuffles, merges; seed 5: the run that caught round 4's lead-lane bug with
    everything else green) and round 5's non-lattice ones (icospheres, Delaunay discs with holes, cones with an apex of valence up to 260,
    decimated spheres, confetti of 1..12-face components, shuffled merges of them) - byte for byte against the oracle, u16 and u32 indices,
    twice (the second pass with the slots the first one taught the context)"""
    import subprocess, sys as _sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([_sys.executable, os.path.join(root, "tools", "stress_topology.py"), "21", str(seed)], capture_output=True, text=True, timeout=900)
    tail = [l for l in out.stdout.splitlines() if l.startswith("decodes")]
    assert out.returncode == 0 and tail, out.stdout[-2000:] + out.stderr[-2000:]
    assert " mismatching arrays 0 " in tail[-1], "\n".join(l for l in out.stdout.splitlines() if "MISMATCH" in l)[:4000]
    assert int(tail[-1].split()[1]) >= 1000, tail[-1]
    assert "persistent fallback" not in out.stdout, "\n".join(l for l in out.stdout.splitlines() if "persistent" in l)[:4000]


def test_device_sqrtf_is_the_reference_norm_for_every_float(tmp_path):
    """k_normal.hip's norm3() takes the f32 square root (17 issue slots) where upstream's Point3::norm() says (float)sqrt((double)s)
    (34, half of them f64): the same bits for EVERY non-negative float if the device routine is correctly rounded - which
    tests/cpp/sqrt_equiv.hip checks exhaustively (2^31 patterns, a few milliseconds), built with the library's own flags"""
    import subprocess
    from conftest import ROOT
    from corto_amd import build as cb
    exe = str(tmp_path / "sqrt_equiv")
    flags = [f for f in cb.FLAGS if f not in ("-fPIC",)]
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + [os.path.join(ROOT, "tests", "cpp", "sqrt_equiv.hip"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and " mismatches 0 " in out.stdout, out.stdout + out.stderr


def test_single_stream_context_decodes_the_same(ctx):
    """crthip_ctx_set_single_stream: everything on one HIP stream (what crthip_pool gives its contexts once their streams would outnumber
    the hardware queues) - same bytes as the two-stream schedule"""
    gs = [load_golden(n) for n in ALL_CASES]
    c = ca.Context(0)
    c.set_single_stream(True)
    for _ in range(2):
        b = run_batch(c, [g["crt"] for g in gs])
        for i, g in enumerate(gs):
            assert_same(b.host_outputs(i), g, KEYS, ALL_CASES[i])
        b.close()
    c.close()


def test_rgb_expands_to_rgba(ctx):
    g = load_golden("nrm_estimated_rgb")
    b = run_batch(ctx, [g["crt"]], color_components=4)
    got = b.host_outputs(0)
    o = oc.decode(g["crt"], color_components=4)
    assert_same(got, o, KEYS, "rgb->rgba")
    assert (got["color"][:, 3] == 248).all()           # (uchar)(255*8), SURVEY a14


def test_unbound_attributes_are_skipped(ctx):
    g = load_golden("c4_unit")
    b = run_batch(ctx, [g["crt"]], only={"position", "index"})
    got = b.host_outputs(0)
    assert set(k for k in got if k not in ("nvert", "nface")) == {"position", "index"}
    assert_same(got, g, ("position", "index"), "only position")


def test_c4_blobs_digests_and_replicated_batch(ctx):
    z = np.load(os.path.join(GOLDEN, "c4_blobs16.npz"))
    blobs = [aligned(z["crt_%02d" % s]) for s in range(16)]
    b = run_batch(ctx, blobs * 4)                       # 64 blobs, one launch set
    for i in range(64):
        got = b.host_outputs(i)
        for k in ("position", "normal", "color", "uv", "index"):
            assert sha(got[k]) == z["%s_sha256_%02d" % (k, i % 16)].tobytes().decode(), (i, k)


def test_mid_mesh_digests(ctx):
    g = load_golden("mid34k_digest")
    b = run_batch(ctx, [g["crt"]])
    got = b.host_outputs(0)
    for k in ("position", "normal", "color", "uv", "index"):
        assert sha(got[k]) == g[k + "_sha256"].tobytes().decode(), k


def test_resident_arena_and_redecode(ctx):
    """inputs already resident in HBM (device_arena) + decoding the same batch twice gives the same bytes"""
    gs = [load_golden(n) for n in ("c4_unit", "cloud_diff", "torus")]
    blobs = [g["crt"] for g in gs]
    arena = ca.upload_arena(blobs)
    b = ca.Batch(ctx, blobs, device_arena=arena)
    b.allocate_outputs(fill=0xAB)
    for _ in range(2):
        b.decode(); assert (b.sync() == 0).all()
        for i, g in enumerate(gs):
            assert_same(b.host_outputs(i), g, KEYS, "arena")


def test_decoder_facade_host_buffers(ctx):
    """crt::Decoder-shaped one-blob API with host buffers (crthip_decode_host)"""
    g = load_golden("c4_unit")
    d = ca.Decoder(g["crt"])
    assert d.nvert == 2112 and d.nface == 4096 and d.hasAttr("uv") and not d.hasAttr("radius")
    pos = np.zeros((d.nvert, 3), np.float32); nrm = np.zeros((d.nvert, 3), np.float32)
    col = np.zeros((d.nvert, 4), np.uint8); uv = np.zeros((d.nvert, 2), np.float32); idx = np.zeros((d.nface, 3), np.uint32)
    assert d.setPositions(pos) and d.setNormals(nrm) and d.setColors(col, 4) and d.setUvs(uv)
    assert not d.setAttribute("radius", pos, ca.FMT_FLOAT)
    d.setIndex(idx)
    d.decode()
    assert_same(dict(position=pos, normal=nrm, color=col, uv=uv, index=idx), g, KEYS, "facade")


def test_cpp_decoder_dropin(ctx, tmp_path):
    """the C++ crt::Decoder facade, compiled against include/corto/decoder.h exactly as a libcorto user would"""
    import subprocess
    from conftest import ROOT
    exe = str(tmp_path / "facade_decode")
    subprocess.check_call(["g++", "-O1", "-std=c++11", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "facade_decode.cpp"),
                           "-L", os.path.dirname(ca.LIB_PATH), "-lcorto_hip", "-Wl,-rpath," + os.path.dirname(ca.LIB_PATH), "-o", exe])
    for name in ("c4_unit", "two_groups", "radius_attr", "group_props"):
        g = load_golden(name)
        src, dst = str(tmp_path / (name + ".crt")), str(tmp_path / (name + ".bin"))
        g["crt"].tofile(src)
        out = subprocess.run([exe, src, dst], capture_output=True, text=True)
        assert out.returncode == 0, out.stderr
        if "groups_ref" in g:                       # Decoder::index.groups[g].end / .properties == what the reference Decoder reported
            got = [l[len("group "):] for l in out.stdout.splitlines() if l.startswith("group ")]
            assert got == g["groups_ref"].tobytes().decode().split("\n"), name
        cc = int(g["color_components"])
        exp = oc.decode(g["crt"], color_components=4)
        blob = np.fromfile(dst, dtype=np.uint8)
        want = np.concatenate([exp[k].reshape(-1).view(np.uint8) for k in ("position", "normal", "color", "uv", "index")])
        assert blob.tobytes() == want.tobytes(), name
    bad = str(tmp_path / "bad.crt")
    np.zeros(64, dtype=np.uint8).tofile(bad)
    out = subprocess.run([exe, bad, str(tmp_path / "x")], capture_output=True, text=True)
    assert out.returncode == 1 and "Not a crt file." in out.stderr


def test_cpp_custom_codec_object(ctx, tmp_path):
    """Decoder::setAttribute(name, buffer, VertexAttribute *) (src/decoder.cpp:104-114): the device decodes the attribute's stream (CRTHIP_BIND_STREAM_VALUES),
    the caller's object runs deltaDecode / postDelta / dequantize on the host with upstream's prediction triples.  A codec that restates GenericAttr<int>
    gives the built-in codec's bytes; one with a twist gives its own; estimated normals over a custom position throw as upstream's do"""
    import subprocess
    from conftest import ROOT
    exe = str(tmp_path / "facade_custom_codec")
    subprocess.check_call(["g++", "-O1", "-std=c++11", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "facade_custom_codec.cpp"),
                           "-L", os.path.dirname(ca.LIB_PATH), "-lcorto_hip", "-Wl,-rpath," + os.path.dirname(ca.LIB_PATH), "-o", exe])
    for name, attr in (("nrm_diff", "position"), ("c4_unit", "uv"), ("torus", "uv"), ("radius_attr", "radius"), ("nonmanifold_fins", "position"), ("two_groups", "uv"),
                       ("cloud_diff", "position"), ("fields32", "uv")):
        g = load_golden(name)
        src, dst = str(tmp_path / (name + ".crt")), str(tmp_path / (name + ".bin"))
        g["crt"].tofile(src)
        out = subprocess.run([exe, src, attr, dst], capture_output=True, text=True)
        assert out.returncode == 0, (name, out.stderr)
        exp = oc.decode(g["crt"])
        e = exp[attr].reshape(-1)
        got = np.fromfile(dst, dtype=np.float32)
        n = e.size
        assert got[:n].tobytes() == e.tobytes(), (name, "built-in")
        assert got[n:2 * n].tobytes() == e.tobytes(), (name, "the codec object that restates GenericAttr<int>")
        assert got[2 * n:3 * n].tobytes() == (np.float32(100.0) - e).astype(np.float32).tobytes(), (name, "the codec object with a twist")
        if attr != "position":
            assert got[3 * n:].tobytes() == exp["position"].tobytes(), (name, "positions beside a custom attribute")
    g = load_golden("c4_unit")                                   # BORDER normals read the integer positions: not there under a custom position codec
    src = str(tmp_path / "c4.crt"); g["crt"].tofile(src)
    out = subprocess.run([exe, src, "position", str(tmp_path / "x.bin"), "normals"], capture_output=True, text=True)
    assert out.returncode == 1 and "Use DIFF normal strategy instead" in out.stderr, (out.returncode, out.stderr)
    # the C ABI underneath: stream values are int32, packed, generic attributes only
    b = ca.Batch(ctx, [g["crt"]])
    outs = b.allocate_outputs()[0]
    names = [a["name"] for a in b.infos[0].attrs()]

    def binding(name, fmt, flags):
        bs = [ca.AttrBinding() for _ in names]
        k = names.index(name)
        bs[k].buffer = outs[name][0].data_ptr(); bs[k].format = fmt; bs[k].reserved = flags
        return bs
    for bs, code in ((binding("normal", ca.FMT_FLOAT, 1), -7),        # CRTHIP_E_FORMAT: normals / colours have codecs of their own
                     (binding("uv", ca.FMT_FLOAT, 1), -8),            # CRTHIP_E_ARGUMENT: stream values are INT32
                     (binding("uv", ca.FMT_INT32, 2), -8)):           # unknown flag bits
        with pytest.raises(ca.CortoError) as ei:
            b.bind(0, bs)
        assert ei.value.code == code
    b.bind(0, binding("uv", ca.FMT_INT32, 1))
    b.decode(); assert (b.sync() == 0).all()
    pred = np.zeros((b.infos[0].nvert, 3), dtype=np.uint32)
    import ctypes as C
    L = ca.lib()
    L.crthip_batch_read_prediction.restype = C.c_int64
    L.crthip_batch_read_prediction.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t]
    assert L.crthip_batch_read_prediction(b.handle, 0, pred.ctypes.data, pred.nbytes) == pred.nbytes
    tr = oc.decode(g["crt"], trace=True)
    assert pred.tobytes() == tr["_prediction"].tobytes()
    import torch
    assert outs["uv"][0].view(torch.int32).cpu().numpy().tobytes() == tr["_raw_uv"].astype(np.int32).tobytes()      # the stream's values, before delta inversion
    b.close()


@pytest.mark.timeout(300)
def test_random_corpus_of_mesh_kinds(ctx):
    """the CLERS automaton's ISA paths (runs of (VERTEX LEFT) pairs, BOUNDARY / DELAY / SPLIT, ring and DELAY-stack pops) and the delta
    scans against the oracle on ~200 meshes of every kind the generator has - grids, tori, closed spheres, discs with holes, ribbons,
    merged components, shuffled faces and vertices, several groups - in batches, u32 and u16 indices, f32 and i16 normals"""
    from corto_amd import synth as S
    rng = np.random.default_rng(99)
    meshes = []
    for k in range(40):
        a, b2 = int(rng.integers(5, 70)), int(rng.integers(4, 40))
        meshes += [S.bumpy_sphere(a, b2, seed=k), S.torus(max(a, 6), max(b2, 5), seed=k), S.closed_sphere(max(a // 2, 4), max(b2 // 2, 3), seed=k),
                   S.holey_disc(max(a // 2, 6), seed=k, hole_frac=0.03 + 0.01 * (k % 12))]
        if k % 4 == 0:
            meshes.append(S.strip(20 + 17 * k, seed=k))
            meshes.append(S.merge([S.closed_sphere(6 + k % 5, 4, seed=k), S.holey_disc(8 + k % 7, seed=k, color_components=4), S.torus(7, 5, seed=k)]))
        if k % 3 == 0:
            meshes[-1] = S.shuffled(meshes[-1], seed=k)
        if k % 5 == 0:
            g = meshes[-2]; g.groups = sorted(set([g.nface // 3, g.nface // 2, g.nface]))
    blobs = [ca.encode(m, normal_prediction=i % 3, position_bits=10 + i % 9) for i, m in enumerate(meshes)]
    refs = [oc.decode(b, color_components=4) for b in blobs]
    for lo in range(0, len(blobs), 64):
        part = blobs[lo:lo + 64]
        b = run_batch(ctx, part, color_components=4)
        for i in range(len(part)):
            assert_same(b.host_outputs(i), refs[lo + i], KEYS, "corpus mesh %d" % (lo + i))
        b16 = run_batch(ctx, part, normal_format=ca.FMT_INT16, index16=True, color_components=4)
        for i in range(len(part)):
            r16 = oc.decode(part[i], normal_format=oc.FMT_INT16, color_components=4, index16=True)
            assert_same(b16.host_outputs(i), r16, KEYS, "corpus mesh %d (i16/u16)" % (lo + i))


def test_interleaved_vertex_buffers(ctx):
    """every fixture decoded into ONE interleaved vertex buffer per blob (crthip_attr_binding.stride: position f32x3 |
    normal i16x3 + pad | uv f32x2 | colour u8x4 | radius f32) + u16 indices, all in one batch; de-interleaved it is the reference's
    output.  Bytes between the records' fields keep the fill value: nothing is written outside the attributes."""
    names = list(ALL_CASES)
    gs = [load_golden(n) for n in names]
    b = ca.Batch(ctx, [g["crt"] for g in gs])
    b.allocate_interleaved(fill=0)
    b.decode()
    assert (b.sync() == 0).all()
    for i, (g, name) in enumerate(zip(gs, names)):
        got = b.host_outputs(i)
        want = oc.decode(g["crt"], normal_format=oc.FMT_INT16, color_components=4, index16=True)
        assert_same(got, want, KEYS, "interleaved i16 " + name)
        if "normal_i16" in g:
            assert got["normal"].tobytes() == g["normal_i16"].tobytes(), name        # the reference's own int16 normals
        if "index_u16_sha256" in g:
            assert sha(got["index"]) == g["index_u16_sha256"].tobytes().decode(), name
    # f32 normals + u32 indices interleaved as well
    b.allocate_interleaved(normal_format=ca.FMT_FLOAT, index16=False, fill=0)
    b.decode(); assert (b.sync() == 0).all()
    for i, (g, name) in enumerate(zip(gs, names)):
        assert_same(b.host_outputs(i), oc.decode(g["crt"], color_components=4), KEYS, "interleaved f32 " + name)
    # the 2 padding bytes behind an i16 normal are nobody's: they keep what the caller had there
    metas = b.allocate_interleaved(fill=0xA5)
    b.decode(); assert (b.sync() == 0).all()
    i = names.index("c4_unit")
    vb, rec, layout, ib, i16 = metas[i]
    nv = b.infos[i].nvert
    raw = b._keep[0][vb:vb + nv * rec].cpu().numpy().reshape(nv, rec)
    o = layout["normal"][0]
    assert rec == 32 and (raw[:, o + 6:o + 8] == 0xA5).all() and not (raw[:, :o + 6] == 0xA5).all()
    # a stride that cannot hold the element, or breaks its alignment, is refused
    info = b.infos[0]
    bad = (ca.AttrBinding * info.nattr)()
    bad[0].buffer = b._keep[0].data_ptr(); bad[0].format = ca.FMT_FLOAT; bad[0].stride = 6
    assert ca.lib().crthip_batch_bind(b.handle, 0, bad, None, ca.FMT_UINT32) == -8
    b.close()


@pytest.fixture(scope="module")
def c5_blobs():
    """BASELINE config C5: 2 048 distinct C4-unit blobs (seeds 0 .. 2047), made by the repo's byte-identical writer"""
    from corto_amd import synth
    return [ca.encode(synth.bumpy_sphere(64, 32, seed=i), position_bits=14, uv_bits=12, normal_bits=10, normal_prediction=ca.BORDER)
            for i in range(2048)]


@pytest.mark.timeout(300)
def test_c5_every_shard_on_one_gpu(ctx, c5_blobs):
    """config C5 without eight GPUs: the 2 048-blob list is cut with shard.balanced_ranges(.., 8) exactly as bench.py cuts it, and
    every one of the eight shards is decoded here (one batch each), a sample of every shard compared with the oracle"""
    from corto_amd import shard
    ranges = shard.balanced_ranges([4096 + 2112] * len(c5_blobs), 8)
    assert [hi - lo for lo, hi in ranges] == [256] * 8 and ranges[0][0] == 0 and ranges[-1][1] == 2048
    b = None
    for r, (lo, hi) in enumerate(ranges):
        part = c5_blobs[lo:hi]
        if b is None:
            b = ca.Batch(ctx, part)
        else:
            b.reset(part)
        b.allocate_outputs()
        b.decode()
        st = b.sync()
        assert (st == 0).all(), (r, st)
        assert b.stats().total_nface == 256 * 4096
        for i in range(r, 256, 41):
            assert_same(b.host_outputs(i), oc.decode(part[i]), KEYS, "C5 shard %d blob %d" % (r, lo + i))
    b.close()


@pytest.mark.timeout(300)
def test_pool_shared_queue_two_devices(c5_blobs):
    """crthip_pool: two pool devices (both on this box's one GPU), 2 threads x 2 batches in flight each, four C5 shards as
    work items on ONE queue; every context's last outputs against the oracle; inputs resident in HBM, then uploaded per step"""
    items = [c5_blobs[256 * g: 256 * g + 64] for g in range(4)]          # 64 blobs of each of four shards: enough to overlap
    pool = ca.Pool([0, 0], threads=2, depth=2)
    assert pool.lanes == 8
    arenas = [[ca.upload_arena(it, 0), ca.upload_arena(it, 0)] for it in items]
    dts = {"position": (np.float32, 3), "normal": (np.float32, 3), "color": (np.uint8, 4), "uv": (np.float32, 2), "index": (np.uint32, 3)}
    for ar in (arenas, None):
        rep, stamps = pool.run(items, steps=40, warmup=8, arenas=ar)
        assert rep.steps == 40 and rep.failed_blobs == 0 and rep.first_error == 0
        assert rep.devices_used == 2 and sum(list(rep.steps_per_device)[:2]) == 40
        assert rep.triangles == sum(64 * 4096 for _ in range(40)) and rep.elapsed_s > 0
        assert len(stamps) == 40 and (np.diff(stamps) >= 0).all() and abs(stamps[-1] - rep.elapsed_s) < 1e-9
        seen = set()
        for lane in range(pool.lanes):
            it, slot = pool.lane_item(lane)
            assert 0 <= it < 4 and slot == lane // 4
            seen.add(it)
            for i in (lane, 63 - lane):
                ref = oc.decode(items[it][i])
                for k, (dt, w) in dts.items():
                    got = pool.lane_read(lane, i, k, dt, (ref["nface"] if k == "index" else ref["nvert"]) * w)
                    assert got.tobytes() == ref[k].tobytes(), (lane, it, i, k)
        assert len(seen) >= 2
    # a blob that cannot be decoded is counted, not fatal
    bad = aligned(items[0][1].copy())
    probs = int(ca.probe(bad).body_offset) + 9 + 4 + 1
    bad[probs:probs + 2] = (7, 255)
    rep, _ = pool.run([[items[0][0], bad]], steps=4, warmup=0)
    assert rep.failed_blobs >= 4 and rep.first_error == -5
    pool.close()


@pytest.mark.timeout(300)
def test_pool_of_sixteen_contexts_on_one_gpu(c5_blobs):
    """the shape bench.py runs: 4 host threads x 4 batches in flight on one GPU - more streams than hardware queues, so the pool gives
    every context ONE stream and the LDS-lean kernel layouts (crthip_ctx_set_single_stream), and its workers refill whichever batch
    finishes first; every context's last outputs against the oracle"""
    items = [c5_blobs[256 * g + 16: 256 * g + 16 + 96] for g in range(3)]
    pool = ca.Pool([0], threads=4, depth=4)
    assert pool.lanes == 16
    arenas = [[ca.upload_arena(it, 0)] for it in items]
    dts = {"position": (np.float32, 3), "normal": (np.float32, 3), "color": (np.uint8, 4), "uv": (np.float32, 2), "index": (np.uint32, 3)}
    rep, stamps = pool.run(items, steps=96, warmup=16, arenas=arenas)
    assert rep.steps == 96 and rep.failed_blobs == 0 and rep.first_error == 0 and rep.devices_used == 1
    assert rep.triangles == 96 * 96 * 4096 and len(stamps) == 96
    for lane in range(pool.lanes):
        it, slot = pool.lane_item(lane)
        assert 0 <= it < 3 and slot == 0
        for i in (lane, 95 - lane, 40 + lane):
            ref = oc.decode(items[it][i])
            for k, (dt, w) in dts.items():
                got = pool.lane_read(lane, i, k, dt, (ref["nface"] if k == "index" else ref["nvert"]) * w)
                assert got.tobytes() == ref[k].tobytes(), (lane, it, i, k)
    pool.close()


def test_cpp_decoder_threads(ctx, tmp_path):
    """SURVEY §8b Threading: distinct crt::Decoder objects on 4 and on 16 threads at once (tests/cpp/facade_threads.cpp), every decode
    repeated and compared with its first, thread 0's outputs compared with the oracle"""
    import subprocess
    from conftest import ROOT
    exe = str(tmp_path / "facade_threads")
    subprocess.check_call(["g++", "-O1", "-std=c++11", "-pthread", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "facade_threads.cpp"),
                           "-L", os.path.dirname(ca.LIB_PATH), "-lcorto_hip", "-Wl,-rpath," + os.path.dirname(ca.LIB_PATH), "-o", exe])
    names = ("c4_unit", "two_groups", "torus", "holey_disc", "nrm_estimated_rgb", "cloud_diff")
    files = []
    for name in names:
        g = load_golden(name)
        src = str(tmp_path / (name + ".crt")); g["crt"].tofile(src); files.append(src)
    out = subprocess.run([exe, "4", "6", str(tmp_path / "out"), *files], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr + out.stdout
    for i, name in enumerate(names):
        g = load_golden(name)
        exp = oc.decode(g["crt"], color_components=4)
        keys = [k for k in ("position", "normal", "color", "uv", "index") if k in exp]
        want = np.concatenate([exp[k].reshape(-1).view(np.uint8) for k in keys])
        blob = np.fromfile(str(tmp_path / ("out%d.bin" % i)), dtype=np.uint8)
        assert blob.tobytes() == want.tobytes(), name
    # two pool contexts only: the other two threads wait their turn
    out = subprocess.run([exe, "4", "2", "-", *files], capture_output=True, text=True, env=dict(os.environ, CORTO_HIP_CONTEXTS="2"))
    assert out.returncode == 0, out.stderr + out.stdout
    # sixteen threads: their decode() calls are coalesced into batches (the facade's combiner: one leader decodes what has queued up, every
    # caller copies its own outputs); every decode of every thread equals its first and thread 0's; then without the gather window, and
    # with one context for all sixteen
    for env in ({}, {"CORTO_HIP_COMBINE_US": "0"}, {"CORTO_HIP_CONTEXTS": "1"}):
        out = subprocess.run([exe, "16", "3", str(tmp_path / "w"), *files], capture_output=True, text=True, env=dict(os.environ, **env))
        assert out.returncode == 0, (env, out.stderr + out.stdout)
        for i in range(len(names)):
            assert (tmp_path / ("w%d.bin" % i)).read_bytes() == (tmp_path / ("out%d.bin" % i)).read_bytes(), (env, names[i])
    # 72 threads - more callers queued behind the two leaders than one batch takes (COMBINE_MAX 64): the thread that wakes
    # as the next leader must decode ITS OWN blob whatever its place in the queue (it used to return "ok" with its buffers untouched)
    for env in ({}, {"CORTO_HIP_LEADERS": "1"}):
        out = subprocess.run([exe, "72", "3", str(tmp_path / "x"), *files], capture_output=True, text=True, env=dict(os.environ, **env))
        assert out.returncode == 0, (env, out.stderr + out.stdout)
        for i in range(len(names)):
            assert (tmp_path / ("x%d.bin" % i)).read_bytes() == (tmp_path / ("out%d.bin" % i)).read_bytes(), (env, names[i])
    # a binding the device path refuses fails ITS caller (on every decode), not the callers co-batched with it
    out = subprocess.run([exe, "16", "3", str(tmp_path / "y"), *files], capture_output=True, text=True, env=dict(os.environ, FACADE_BAD_BIND_THREAD="5"))
    assert out.returncode == 0, out.stderr + out.stdout
    for i in range(len(names)):
        assert (tmp_path / ("y%d.bin" % i)).read_bytes() == (tmp_path / ("out%d.bin" % i)).read_bytes(), names[i]
    # a blob that cannot be decoded fails alone: its thread gets upstream's exception, the threads decoding beside it their meshes
    bad = load_golden("c4_unit")["crt"].copy()
    probs = int(ca.probe(aligned(bad)).body_offset) + 9 + 4 + 1
    bad[probs:probs + 2] = (7, 255)
    badf = str(tmp_path / "bad.crt"); bad.tofile(badf)
    out = subprocess.run([exe, "8", "2", "-", badf], capture_output=True, text=True)
    assert out.returncode == 1 and "Decoding topology failed" in out.stderr, out.stderr + out.stdout


def test_generic_attribute_output_formats(tmp_path):
    """Decoder::setAttribute(name, buffer, format) with the integer formats and DOUBLE (src/decoder.cpp:96-102, GenericAttr::dequantize
    include/corto/vertex_attribute.h:195-228): the device path leaves in the caller's nvert*N*8-byte buffer what the compiled reference left
    (tests/golden/generic_formats.npz) - every format, through crthip_decode_host (the facade's path) one attribute at a time, and with every
    attribute of a blob bound at once (positions as INT32 under estimated normals: the normals still read the integers)"""
    z = np.load(os.path.join(GOLDEN, "generic_formats.npz"))
    for name in z["cases"].tobytes().decode().split(","):
        blob = aligned(z["crt_" + name])
        info = ca.probe(blob)
        for key in [k for k in z.files if k.startswith(name + ".")]:
            _, attr, fmt = key.split(".")
            want = z[key]
            buf = np.full(len(want), 0xCD, dtype=np.uint8)
            d = ca.Decoder(blob)
            assert d.setAttribute(attr, buf, int(fmt))
            idx = np.zeros((max(info.nface, 1), 3), dtype=np.uint32)
            if info.nface:
                d.setIndex(idx)
            d.decode()
            n4 = len(want) // 2
            used = len(want) if int(fmt) == ca.FMT_DOUBLE else n4            # (the facade copies back the bytes the format uses: the int32 array, or the doubles)
            assert buf[:used].tobytes() == want[:used].tobytes(), key
            assert (buf[used:] == 0xCD).all(), key
    # positions as INT32 beside float normals (estimated from the integer positions) and uvs as UINT16: one decode
    blob = aligned(z["crt_sphere_q0p75"])
    info = ca.probe(blob)
    ref = oc.decode(blob)
    pos = np.full(info.nvert * 3 * 4, 0xCD, dtype=np.uint8)
    nrm = np.zeros((info.nvert, 3), dtype=np.float32)
    idx = np.zeros((info.nface, 3), dtype=np.uint32)
    d = ca.Decoder(blob)
    d.setAttribute("position", pos, ca.FMT_INT32); d.setNormals(nrm); d.setIndex(idx)
    d.decode()
    assert pos.tobytes() == z["sphere_q0p75.position.%d" % ca.FMT_INT32][:len(pos)].tobytes()
    assert nrm.tobytes() == ref["normal"].tobytes() and idx.tobytes() == ref["index"].tobytes()


def test_status_survives_other_batches(ctx):
    """decode(A); decode(B) on one context; sync(A) must still report A's per-blob failures (they used to be lost when
    another call synchronised the stream first)"""
    g = load_golden("c4_unit")
    good = aligned(g["crt"])
    bad = aligned(g["crt"].copy())
    probs = int(ca.probe(bad).body_offset) + 9 + 4 + 1     # groups (u32 n, u32 end, u8 nprops), max_front, u8 nsym
    bad[probs:probs + 2] = (7, 255)                          # the likeliest CLERS symbol becomes the invalid symbol 7: topology must fail
    a = ca.Batch(ctx, [good, bad]); a.allocate_outputs()
    b = ca.Batch(ctx, [good]); b.allocate_outputs()
    a.decode()
    b.decode()                                     # implicit sync of A inside the library
    sb = b.sync()
    sa = a.sync(raise_on_error=False)
    assert (sb == 0).all()
    assert sa[0] == 0 and sa[1] == -5, sa
    ref = oc.decode(good)
    assert_same(a.host_outputs(0), ref, KEYS, "good blob of batch A")
    a.close(); b.close()


def test_unity_veneer_decode_mesh(ctx):
    """CreateDecoder / DecodeMesh / DestroyDecoder (include/corto/corto_codec.h = upstream src/corto_codec.h:41-43) driven the
    way unity/CortoMeshLoader.cs does: arrays sized from info, one DecodeMesh call"""
    import ctypes as C
    from corto_amd import build
    V = C.CDLL(build.VENEER)
    V.CreateDecoder.restype = C.c_void_p
    V.CreateDecoder.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
    V.DestroyDecoder.argtypes = [C.c_void_p]
    V.DecodeMesh.restype = C.c_int
    V.DecodeMesh.argtypes = [C.c_void_p] * 6
    for name in ("c4_unit", "torus", "two_groups"):
        g = load_golden(name)
        blob = aligned(g["crt"])
        info = np.zeros(2, dtype=np.float32)
        d = V.CreateDecoder(len(blob), blob.ctypes.data, info.ctypes.data)
        assert d
        nface, nvert = int(info[0]), int(info[1])
        pos = np.zeros((nvert, 3), np.float32); nrm = np.zeros((nvert, 3), np.float32); uv = np.zeros((nvert, 2), np.float32)
        col = np.zeros((nvert, 4), np.float32); idx = np.zeros((nface, 3), np.int32)
        assert V.DecodeMesh(d, pos.ctypes.data, idx.ctypes.data, nrm.ctypes.data, col.ctypes.data, uv.ctypes.data) == nface
        V.DestroyDecoder(d)
        exp = oc.decode(g["crt"], color_components=4)
        assert pos.tobytes() == exp["position"].tobytes() and idx.tobytes() == exp["index"].tobytes(), name
        if "normal" in exp: assert nrm.tobytes() == exp["normal"].tobytes(), name
        if "uv" in exp: assert uv.tobytes() == exp["uv"].tobytes(), name
        if "color" in exp: assert col.tobytes() == (exp["color"].astype(np.float32) / np.float32(255.0)).tobytes(), name


def test_cli_round_trip_ply_identical_to_the_reference_cli(ctx, tmp_path):
    """corto_hip in.ply -o out.crt -P out.ply (tools/corto_hip_cli.cpp: this repo's encoder, decode on the GPU through the
    crt::Decoder facade) against the reference's own CLI (oracle/_ref/corto_ref_cli) on the same PLY and options: same .crt,
    same decoded .ply, byte for byte"""
    from cli_common import REF_CLI, our_cli, run, write_ply
    from corto_amd import synth
    if not os.path.exists(REF_CLI):
        pytest.skip("oracle/_ref/corto_ref_cli did not travel")
    cases = [(synth.bumpy_sphere(32, 16, seed=21), dict(), ["-v", "13"]),
             (synth.torus(24, 12, seed=22), dict(binary=False), ["-N", "estimated"]),
             (synth.holey_disc(20, seed=23, color_components=4), dict(uv_names=("s", "t")), ["-v", "12", "-N", "delta", "-n", "11"]),
             (synth.bumpy_sphere(20, 10, seed=24), dict(faces=False), ["-v", "14", "-N", "delta"])]
    for k, (m, kw, opts) in enumerate(cases):
        d = tmp_path / ("case%d" % k)
        d.mkdir()
        write_ply(str(d / "in.ply"), m, **kw)
        run(REF_CLI, ["in.ply", "-o", "ref.crt", "-P", "ref.ply"] + opts, str(d))
        run(our_cli(), ["in.ply", "-o", "ours.crt", "-P", "ours.ply"] + opts, str(d))
        assert (d / "ref.crt").read_bytes() == (d / "ours.crt").read_bytes(), k
        assert (d / "ref.ply").read_bytes() == (d / "ours.ply").read_bytes(), k


def test_topology_failure_is_reported_per_blob(ctx):
    g = load_golden("holey_disc"); ok = load_golden("torus")
    bad = g["crt"].copy()
    h = oc.parse_header(bad)
    # corrupt the split/vertex-id bit block region heavily: flip bytes in the CLERS payload
    start = h["body_offset"] + 40
    bad[start:start + 200] ^= 0x5A
    b = ca.Batch(ctx, [aligned(bad), ok["crt"]])
    b.allocate_outputs()
    b.decode()
    st = b.sync(raise_on_error=False)
    assert st[1] == 0
    assert_same(b.host_outputs(1), ok, KEYS, "good blob next to a corrupt one")
    # the corrupt blob either fails topology or decodes garbage without faulting; it must not take the batch down
    assert st[0] in (0, -5)


@pytest.mark.timeout(180)
def test_corrupt_blobs_never_fault_or_disturb_neighbours(ctx):
    """byte-flipped, burst-corrupted, garbage-tailed and zero-windowed copies of every fixture, decoded in ONE batch next to
    intact blobs: whatever the host walk cannot reject either fails per blob (CRTHIP_E_TOPOLOGY) or decodes garbage - no
    fault, no hang, and the intact neighbours stay bit-exact (the reference itself reads out of bounds on such input)"""
    rng = np.random.default_rng(7)
    names = list(MESH_CASES) + list(CLOUD_CASES)
    good = load_golden("c4_unit")
    blobs, intact = [], []
    for i in range(96):
        g = load_golden(names[i % len(names)])
        b = g["crt"].copy()
        body = oc.parse_header(b)["body_offset"]
        mode = i % 4
        if mode == 0:
            for p in rng.integers(body, len(b), 6):
                b[p] ^= rng.integers(1, 256)
        elif mode == 1:
            p = int(rng.integers(body, max(body + 1, len(b) - 64))); b[p:p + 48] ^= 0xA5
        elif mode == 2:
            p = int(rng.integers(body, len(b))); b[p:] = rng.integers(0, 256, len(b) - p, dtype=np.uint8)
        else:
            p = int(rng.integers(body, max(body + 1, len(b) - 200))); b[p:p + 160] = 0
        blobs.append(aligned(b)); intact.append(False)
        if i % 8 == 7:
            blobs.append(aligned(good["crt"])); intact.append(True)
    keep = []
    for i, b in enumerate(blobs):                     # what the bounds-checked host walk proves truncated never reaches the device
        try:
            ca.Batch(ctx, [b]).close(); keep.append(i)
        except ca.CortoError:
            assert not intact[i]
    assert len(keep) > len(blobs) // 2
    bt = ca.Batch(ctx, [blobs[i] for i in keep])
    bt.allocate_outputs(fill=0)
    bt.decode()
    st = bt.sync(raise_on_error=False)
    assert set(np.unique(st)) <= {0, -5}
    ref = oc.decode(good["crt"])
    for j, i in enumerate(keep):
        if intact[i]:
            assert st[j] == 0
            assert_same(bt.host_outputs(j), ref, KEYS, "intact blob %d among corrupt ones" % j)


def test_unsupported_format_fails_loudly(ctx):
    g = load_golden("c4_unit")
    b = ca.Batch(ctx, [g["crt"]])
    info = b.infos[0].attrs()
    binds = [ca.AttrBinding() for _ in info]
    # what the device path refuses: a colour as FLOAT (broken upstream, color_attribute.cpp:96-110), a normal as anything but FLOAT / INT16
    for name, fmt in (("color", ca.FMT_FLOAT), ("normal", ca.FMT_UINT8), ("normal", ca.FMT_DOUBLE)):
        binds = [ca.AttrBinding() for _ in info]
        k = [a["name"] for a in info].index(name)
        binds[k].buffer = 4096; binds[k].format = fmt
        with pytest.raises(ca.CortoError, match="Format not supported"):
            b.bind(0, binds)
    # a generic attribute takes every format (test_generic_attribute_output_formats) - packed only, and aligned for its type
    binds = [ca.AttrBinding() for _ in info]
    k = [a["name"] for a in info].index("position")
    binds[k].buffer = 4096; binds[k].format = ca.FMT_INT16; binds[k].stride = 32
    with pytest.raises(ca.CortoError, match="Invalid argument"):
        b.bind(0, binds)
    binds[k].stride = 0; binds[k].buffer = 4100; binds[k].format = ca.FMT_DOUBLE
    with pytest.raises(ca.CortoError, match="Invalid argument"):
        b.bind(0, binds)


# ---------------------------------------------------------------------------------------------------
# Tunstall stage in isolation
def _kat():
    z = np.load(os.path.join(GOLDEN, "tunstall_kat.npz"))
    return {k: z[k] for k in z.files}


def _run_blocks(ctx, blocks, sizes):
    import torch
    offs, total = [], 0
    for b in blocks:
        offs.append(total); total += (len(b) + 15) & ~15
    host = np.zeros(total + 16, dtype=np.uint8)
    for b, o in zip(blocks, offs):
        host[o:o + len(b)] = b
    oo, ot = [], 0
    for s in sizes:
        oo.append(ot); ot += (s + 15) & ~15
    dblk = torch.from_numpy(host).cuda()
    dout = torch.full((ot + 16,), 0xEE, dtype=torch.uint8, device="cuda")     # (tunstall_decode_blocks waits for this fill: the library's
    times = ca.tunstall_decode_blocks(ctx, host, dblk, offs, dout, oo)         #  streams are non-blocking, include/corto_hip.h "Device buffers")
    out = dout.cpu().numpy()
    return [out[o:o + s] for o, s in zip(oo, sizes)], times


def test_tunstall_tables_kat_on_device(ctx):
    """every dictionary of the reference KAT: decode the payload 0,1,...,255 -> concatenation of all 256 words"""
    k = _kat()
    blocks, sizes, expect = [], [], []
    for i in range(int(k["count"])):
        pr, idx, ln, tab = k["probs_%02d" % i], k["index_%02d" % i].astype(int), k["length_%02d" % i].astype(int), k["table_%02d" % i]
        words = np.concatenate([tab[idx[c]:idx[c] + ln[c]] for c in range(256)])
        hdr = bytes([len(pr)]) + pr.tobytes() + int(len(words)).to_bytes(4, "little") + (256).to_bytes(4, "little")
        blocks.append(np.frombuffer(hdr + bytes(range(256)), dtype=np.uint8)); sizes.append(len(words)); expect.append(words)
    outs, _ = _run_blocks(ctx, blocks, sizes)
    for i, (o, e) in enumerate(zip(outs, expect)):
        assert np.array_equal(o, e), i


def test_tunstall_random_dictionaries_on_device(ctx):
    """800 random probability tables - 2 to 255 symbols, flat / skewed / one dominant symbol (the low-entropy seed) / zero tails /
    ties - built on the device and read back through the payload 0..255, against the oracle's tables (which are pinned to the
    reference's createDecodingTables2 on 1 500 random tables, tests/test_oracle_vs_reference.py)"""
    rng = np.random.default_rng(77)
    blocks, sizes, expect = [], [], []
    for t in range(800):
        n = int(rng.integers(2, 256)) if t % 5 == 0 else int(rng.integers(2, 24)) if t % 5 < 3 else int(rng.integers(24, 130))
        kind = t % 6
        if kind == 0:
            p = np.sort(rng.integers(0, 256, n))[::-1]
        elif kind == 1:
            p = np.sort((255 * rng.dirichlet(np.ones(n) * 0.3)).astype(int))[::-1]
        elif kind == 2:
            p = np.array([max(250 - n, 1)] + list(np.sort(rng.integers(0, 4, n - 1))[::-1]))
        elif kind == 3:
            p = np.sort((255 * rng.dirichlet(np.ones(n) * 4)).astype(int))[::-1]
        elif kind == 4:
            p = np.full(n, max(255 // n, 1))                                      # all ties
        else:
            p = np.sort(rng.integers(0, 3, n))[::-1]                              # mostly zeros
        pr = np.stack([rng.permutation(256)[:n], np.clip(p, 0, 255)], 1).astype(np.uint8)
        idx, ln, tab = oc.tunstall_tables(pr)
        idx, ln = np.asarray(idx).astype(int), np.asarray(ln).astype(int)
        words = np.concatenate([np.asarray(tab)[idx[c]:idx[c] + ln[c]] for c in range(256)]) if ln.sum() else np.zeros(0, np.uint8)
        if (ln > 0).sum() < 256:                       # the reference's own word buffer would overflow (its assert at src/tunstall.cpp:229; all-zero
            continue                                   # probabilities, which its encoder never writes): no reference result to agree with
        hdr = bytes([len(pr)]) + pr.tobytes() + int(len(words)).to_bytes(4, "little") + (256).to_bytes(4, "little")
        blocks.append(np.frombuffer(hdr + bytes(range(256)), dtype=np.uint8)); sizes.append(len(words)); expect.append(words)
    outs, _ = _run_blocks(ctx, blocks, sizes)
    bad = [i for i, (o, e) in enumerate(zip(outs, expect)) if not np.array_equal(o, e)]
    assert not bad, (len(bad), bad[:5], [int(blocks[i][0]) for i in bad[:5]])


def test_tunstall_streams_kat_on_device(ctx):
    k = _kat()
    blocks = [k["stream_block_%d" % i] for i in range(8)]
    syms = [k["stream_symbols_%d" % i] for i in range(8)]
    outs, _ = _run_blocks(ctx, blocks, [len(s) for s in syms])
    for i, (o, e) in enumerate(zip(outs, syms)):
        assert np.array_equal(o, e), i


def test_tunstall_long_streams_multi_chunk(ctx):
    """streams far longer than one workgroup's chunk: any byte string is a valid codeword stream, so random
    payloads + the oracle's dictionary give the expected output without needing an encoder"""
    rng = np.random.default_rng(5)
    k = _kat()
    blocks, sizes, expect = [], [], []
    for i, ncode in ((3, 100_000), (20, 300_001), (0, 70_000), (40, 16_385)):
        pr = k["probs_%02d" % i]
        idx, ln, tab = oc.tunstall_tables(pr)
        payload = rng.integers(0, 256, ncode).astype(np.uint8)
        size = int(ln[payload].sum())
        if i == 20:
            size -= 1                                   # clip the last word by one byte
        hdr = bytes([len(pr)]) + pr.tobytes() + size.to_bytes(4, "little") + ncode.to_bytes(4, "little")
        blocks.append(np.frombuffer(hdr + payload.tobytes(), dtype=np.uint8)); sizes.append(size)
        expect.append(oc.tunstall_decompress(pr, payload, size))
    outs, times = _run_blocks(ctx, blocks, sizes)
    for i, (o, e) in enumerate(zip(outs, expect)):
        assert np.array_equal(o, e), i
    assert "tunstall_chunk_sums" in times or not times  # multi-chunk path taken when profiling is on


@pytest.mark.timeout(120)
def test_tunstall_long_streams_on_four_contexts_at_once(ctx):
    """the single-pass decode finds a chunk's output offset by look-back over its predecessors' state words; with several launches in
    flight a predecessor may not have started (each XCD dispatches on its own), so nobody may WAIT for one (kernels_common.h:
    chain_lookback recomputes instead).  Four contexts decode long streams concurrently, four rounds each: exact, and no stall."""
    import threading, time
    rng = np.random.default_rng(8)
    k = _kat()
    jobs = []
    for t in range(4):
        blocks, sizes, expect = [], [], []
        for i in (3 + t, 20, 9, 37, 13, 25):
            pr = k["probs_%02d" % i]
            idx, ln, tab = oc.tunstall_tables(pr)
            payload = rng.integers(0, 256, 400_000 + 1000 * t).astype(np.uint8)
            size = int(np.asarray(ln)[payload].sum())
            hdr = bytes([len(pr)]) + pr.tobytes() + size.to_bytes(4, "little") + len(payload).to_bytes(4, "little")
            blocks.append(np.frombuffer(hdr + payload.tobytes(), dtype=np.uint8)); sizes.append(size)
            expect.append(oc.tunstall_decompress(pr, payload, size))
        jobs.append((ca.Context(0), blocks, sizes, expect))
    errors = []
    def work(job):
        try:
            c, blocks, sizes, expect = job
            for _ in range(4):
                outs, _ = _run_blocks(c, blocks, sizes)
                for i, (o, e) in enumerate(zip(outs, expect)):
                    assert np.array_equal(o, e), i
        except BaseException as e:
            errors.append(e)
    t0 = time.perf_counter()
    ths = [threading.Thread(target=work, args=(j,)) for j in jobs]
    for th in ths: th.start()
    for th in ths: th.join()
    assert not errors, errors[0]
    assert time.perf_counter() - t0 < 60
    for j in jobs: j[0].close()


def test_tunstall_long_stream_pipelines():
    """both ways the long-stream path finds a chunk's output offset (k_tunstall.hip): quarter sums + a scan per stream (streams of more
    than 256 chunks) or the decode waves adding up the sums in front of them (fewer); one decode launch for all word-width classes.
    A two-symbol dictionary with words up to 255 bytes (hundreds of small chunks), a flat one, a short stream beside them."""
    env = {}
    c = ca.Context(0)
    rng = np.random.default_rng(23)
    k = _kat()
    blocks, sizes, expect = [], [], []
    for pr, ncode in ((np.array([[0, 240], [1, 15]], dtype=np.uint8), 1_200_000), (k["probs_20"], 700_001), (k["probs_09"], 5_000), (k["probs_37"], 90_000)):
        idx, ln, tab = oc.tunstall_tables(pr)
        payload = rng.integers(0, 256, ncode).astype(np.uint8)
        size = int(np.asarray(ln)[payload].sum())
        hdr = bytes([len(pr)]) + pr.tobytes() + size.to_bytes(4, "little") + ncode.to_bytes(4, "little")
        blocks.append(np.frombuffer(hdr + payload.tobytes(), dtype=np.uint8)); sizes.append(size)
        expect.append(oc.tunstall_decompress(pr, payload, size))
    for _ in range(2):
        outs, _t = _run_blocks(c, blocks, sizes)
        for i, (o, e) in enumerate(zip(outs, expect)):
            assert np.array_equal(o, e), (i, env)
    outs, _t = _run_blocks(c, blocks[2:], sizes[2:])     # few chunks per stream: no scan launch
    for i, (o, e) in enumerate(zip(outs, expect[2:])):
        assert np.array_equal(o, e), (i, env, "short")
    c.close()


def test_tunstall_long_streams_every_step_geometry(ctx):
    """the staged decode sizes a wave's step (8/4/2/1 codewords per lane) and the chunk from the stream's mean word
    length: low-entropy dictionaries (two symbols, words up to 255 bytes) down to flat ones, multi-chunk, clipped ends"""
    rng = np.random.default_rng(11)
    k = _kat()
    dicts = [k["probs_09"], k["probs_25"], k["probs_37"], k["probs_13"], k["probs_08"],
             np.array([[0, 235], [1, 20]], dtype=np.uint8), np.array([[0, 240], [1, 15]], dtype=np.uint8),
             np.array([[7, 250], [9, 5]], dtype=np.uint8), np.array([[3, 254], [200, 1]], dtype=np.uint8)]
    blocks, sizes, expect = [], [], []
    for i, pr in enumerate(dicts):
        idx, ln, tab = oc.tunstall_tables(pr)
        ncode = 20_000 + 4_099*i
        payload = rng.integers(0, 256, ncode).astype(np.uint8)
        size = int(np.asarray(ln)[payload].sum()) - (i % 3)           # clip the last word by 0..2 bytes
        hdr = bytes([len(pr)]) + pr.tobytes() + size.to_bytes(4, "little") + ncode.to_bytes(4, "little")
        blocks.append(np.frombuffer(hdr + payload.tobytes(), dtype=np.uint8)); sizes.append(size)
        expect.append(oc.tunstall_decompress(pr, payload, size))
    outs, _ = _run_blocks(ctx, blocks, sizes)
    for i, (o, e) in enumerate(zip(outs, expect)):
        assert np.array_equal(o, e), (i, int(np.argmax(o != e[:len(o)])) if len(o) == len(e) else (len(o), len(e)))


# ---------------------------------------------------------------------------------------------------
# GPU encoder stage: crthip_tunstall_encode_blocks = OutStream::tunstall_compress for a batch of streams
def _enc_kat():
    z = np.load(os.path.join(GOLDEN, "tunstall_enc_kat.npz"))
    n = int(z["count"])
    return [z["input_%02d" % i] for i in range(n)], [z["block_%02d" % i] for i in range(n)]


def test_tunstall_encode_blocks_kat(ctx):
    """every block byte-identical to what the reference wrote for the same symbols (logs-like, CLERS-like, low-entropy runs,
    64 and 200 distinct symbols (the 200-symbol trie is walked in L2, not LDS), one symbol, streams ending inside a word)"""
    streams, blocks = _enc_kat()
    got, times = ca.tunstall_encode_blocks(ctx, streams, with_times=True)
    assert "enc_hist" in times and "enc_tun_parse" in times
    for i, (g, e) in enumerate(zip(got, blocks)):
        assert g.tobytes() == e.tobytes(), (i, len(g), len(e))
    # empty batch, empty stream
    assert ca.tunstall_encode_blocks(ctx, []) == []
    assert ca.tunstall_encode_blocks(ctx, [np.zeros(0, np.uint8)])[0].tobytes() == bytes(9)


def test_tunstall_encode_blocks_round_trip_and_reference(ctx):
    """hundreds of random streams in one call, one of them several windows' restaging long: the device decoder returns the
    input; where oracle/_ref travelled, the reference's own block is the same bytes"""
    from oracle import refcodec as rc
    rng = np.random.default_rng(123)
    streams = []
    for k in range(300):
        n = int(rng.integers(1, 5000)) if k else 300_001
        nsym = int(rng.integers(1, 48))
        p = rng.dirichlet(np.full(nsym, 0.25 if k % 3 else 3.0))
        streams.append(rng.choice((np.arange(nsym) * 5 % 256).astype(np.uint8), n, p=p))
    blocks = ca.tunstall_encode_blocks(ctx, streams)
    outs, _ = _run_blocks(ctx, blocks, [len(s) for s in streams])
    for i, (o, s) in enumerate(zip(outs, streams)):
        assert np.array_equal(o, s), i
    if rc.available():
        for i in range(0, 300, 7):
            assert blocks[i].tobytes() == rc.tunstall_compress_block(streams[i]).tobytes(), i


def test_tunstall_encoder_tables_made_on_the_device(ctx):
    """probabilities, dictionary and trie of every stream come from k_enc_tables / k_enc_trie.  The streams here are
    made of symbols with EQUAL counts (17 .. 40 of them, more than std::sort's insertion-sort threshold), so the block depends on
    the order libstdc++'s introsort leaves equal probabilities in (csrc/std_sort_model.h): every block against the reference's."""
    from oracle import refcodec as rc
    rng = np.random.default_rng(77)
    streams = []
    for k in range(120):
        nsym = 2 + k % 39
        reps = [int(rng.integers(1, 4)) * (3 if j % 5 == 0 else 1) for j in range(nsym)] if k % 2 else [7] * nsym
        sym = np.repeat((np.arange(nsym) * 11 % 251).astype(np.uint8), np.array(reps) * 9)
        streams.append(rng.permutation(sym))
    blocks, times = ca.tunstall_encode_blocks(ctx, streams, with_times=True)
    assert "enc_tables" in times and "enc_trie" in times
    assert times["enc_trie"]["launches"] == 0, "every one of these tries fits the device builder"      # (= streams the host had to make tables for)
    outs, _ = _run_blocks(ctx, blocks, [len(s) for s in streams])
    for i, (o, s) in enumerate(zip(outs, streams)):
        assert np.array_equal(o, s), i
    if rc.available():
        for i, s in enumerate(streams):
            assert blocks[i].tobytes() == rc.tunstall_compress_block(s).tobytes(), i
    # a 200-symbol alphabet: its trie is left to the host routine, same bytes
    big = rng.integers(0, 200, 30000).astype(np.uint8)
    b2, t2 = ca.tunstall_encode_blocks(ctx, [big, streams[3]], with_times=True)
    assert t2["enc_trie"]["launches"] == 1
    if rc.available():
        assert b2[0].tobytes() == rc.tunstall_compress_block(big).tobytes() and b2[1].tobytes() == blocks[3].tobytes()


def _model_bits(fields):
    """MSB-first bit writer (src/bitstream.cpp:86-101): fields = iterable of (value, nbits) -> uint32 words"""
    acc, nb, words = 0, 0, []
    for v, n in fields:
        acc = (acc << n) | (int(v) & ((1 << n) - 1)); nb += n
        while nb >= 32:
            words.append((acc >> (nb - 32)) & 0xFFFFFFFF); nb -= 32; acc &= (1 << nb) - 1
    if nb:
        words.append((acc << (32 - nb)) & 0xFFFFFFFF)
    return np.array(words, dtype=np.uint32)


def _model_array(a):
    """OutStream::encodeArray<int> (include/corto/cstream.h:143-164): one width per element -> (words, [logs])"""
    def needed(x):
        x = int(x)
        if x == 0: return 0
        if x == -1: return 1
        if x < 0: x = -x - 1
        return 1 + x.bit_length()
    logs = np.array([max(needed(x) for x in row) for row in a], dtype=np.uint8)
    fields = [(int(x) + (1 << (int(d) - 1)), int(d)) for row, d in zip(a, logs) if d for x in row]
    return _model_bits(fields), [logs]


def _model_values(a):
    """OutStream::encodeValues (include/corto/cstream.h:115-141): component-major, one width per value, sign folded"""
    logs, fields = [], []
    for c in range(a.shape[1]):
        lg = np.zeros(len(a), dtype=np.uint8)
        for i, x in enumerate(a[:, c]):
            x = int(x)
            if x == 0: continue
            r = abs(x).bit_length()
            lg[i] = r
            fields.append((x if x > 0 else -x - (1 << (r - 1)), r))
        logs.append(lg)
    return _model_bits(fields), logs


def test_encode_values_stage_against_the_cstream_model(ctx):
    """crthip_encode_values: bit widths + bit packing on the device; the words equal a direct restatement of
    encodeArray / encodeValues, the blocks equal the (reference-pinned) Tunstall stage run on the model's width arrays"""
    rng = np.random.default_rng(31)
    streams, models = [], []
    for k, (n, N, scale) in enumerate(((2112, 3, 40), (1000, 2, 3), (257, 1, 2000), (513, 4, 1), (1, 3, 5), (300, 16, 9), (4000, 3, 100000))):
        a = np.rint(rng.normal(0, scale, (n, N))).astype(np.int32)
        if k == 0:
            a[5] = (-1, 0, -1); a[6] = 0; a[7] = (2**31 - 1, 0, 0); a[8] = (-2**31 + 1, 1, -2)
        streams.append((ca.ENC_ARRAY, a)); models.append(_model_array(a))
        streams.append((ca.ENC_VALUES_I32, a)); models.append(_model_values(a))
    c8 = np.rint(rng.normal(0, 6, (2112, 4))).clip(-128, 127).astype(np.int8)
    streams.append((ca.ENC_VALUES_I8, c8)); models.append(_model_values(c8.astype(np.int32)))
    sym = rng.integers(0, 7, 4319).astype(np.uint8)
    streams.append((ca.ENC_SYMBOLS, sym)); models.append((None, [sym]))
    for entropy in (1, 0):
        got, times = ca.encode_values(ctx, streams, entropy=entropy, with_times=True)
        assert "enc_pack" in times
        flat = [lg for _, logs in models for lg in logs]
        blocks = ca.tunstall_encode_blocks(ctx, flat) if entropy else [np.concatenate([np.array([len(x)], dtype="<u4").view(np.uint8), x]) for x in flat]
        bi = 0
        for i, (g, (words, logs)) in enumerate(zip(got, models)):
            exp = b""
            if words is not None:
                exp += np.array([len(words)], dtype="<u4").tobytes() + words.astype("<u4").tobytes()
            for _ in logs:
                exp += blocks[bi].tobytes(); bi += 1
            assert g.tobytes() == exp, (entropy, i, len(g), len(exp))


def test_encoder_stage_argument_errors(ctx):
    """bad descriptors are refused with an error code and a message, nothing is launched"""
    a = np.zeros((4, 3), np.int32)
    for kind, arr in ((7, a), (ca.ENC_ARRAY, np.zeros((4, 17), np.int32))):
        with pytest.raises(ca.CortoError):
            ca.encode_values(ctx, [(kind, arr)])
    with pytest.raises(ca.CortoError):
        ca.encode_values(ctx, [(ca.ENC_ARRAY, a)], entropy=5)
    # still usable afterwards
    assert len(ca.encode_values(ctx, [(ca.ENC_ARRAY, a)])[0]) > 4


def test_gpu_encoder_blobs_identical_to_the_reference_made_fixtures(ctx):
    """crthip_encode_gpu (value coding + entropy coder on the device, topology and container on the host) writes the same
    bytes as the reference encoder did for every golden case, as the host encoder on the C4 units, and for entropy NONE"""
    import sys
    sys.path.insert(0, GOLDEN)
    from cases import cases
    from corto_amd import synth
    for name, mesh, kw in cases():
        g = load_golden(name)
        mine = ca.encode(mesh, ctx=ctx, **kw)
        assert len(mine) == len(g["crt"]) and mine.tobytes() == g["crt"].tobytes(), name
    for seed in (0, 7):
        m = synth.bumpy_sphere(64, 32, seed=seed)
        assert ca.encode(m, normal_prediction=ca.BORDER, ctx=ctx).tobytes() == ca.encode(m, normal_prediction=ca.BORDER).tobytes()
    cloud = synth.point_cloud(60, 30, seed=3)
    assert ca.encode(cloud, normal_prediction=ca.DIFF, ctx=ctx).tobytes() == ca.encode(cloud, normal_prediction=ca.DIFF).tobytes()
    m = synth.bumpy_sphere(260, 130, seed=34)
    assert ca.encode(m, normal_prediction=ca.BORDER, ctx=ctx).tobytes() == load_golden("mid34k_digest")["crt"].tobytes()


# ---------------------------------------------------------------------------------------------------
# BASELINE.json full sizes: inputs synthesised on the box by the repo's own encoder (byte-identical to the reference's,
# tests/test_encoder_cpu.py), outputs checked against the C oracle and, when oracle/_ref travelled, the reference itself.
def _maybe_ref_decode(blob):
    from oracle import refcodec as rc
    return rc.decode(blob) if rc.available() else None


@pytest.mark.parametrize("pred", [0, 1, 2])
def test_config2_128k_mesh(ctx, pred):
    from corto_amd import synth
    m = synth.bumpy_sphere(512, 250, seed=2)           # 128 512 verts / 256 000 tris
    blob = ca.encode(m, position_bits=14, uv_bits=12, normal_bits=10, normal_prediction=pred)
    b = run_batch(ctx, [blob])
    got = b.host_outputs(0)
    assert got["nvert"] == 128512 and got["nface"] == 256000
    assert_same(got, oc.decode(blob), KEYS, "C2 pred %d" % pred)
    r = _maybe_ref_decode(blob)
    if r is not None:
        assert_same(got, r, KEYS, "C2 pred %d vs reference" % pred)
    # properties: every index valid, every triangle non-degenerate, normals unit length
    idx = got["index"].astype(np.int64)
    assert idx.max() == got["nvert"] - 1 and (idx[:, 0] != idx[:, 1]).all() and (idx[:, 1] != idx[:, 2]).all()
    assert np.abs(np.linalg.norm(got["normal"].astype(np.float64), axis=1) - 1).max() < 1e-5


def test_config3_167k_point_cloud(ctx):
    from corto_amd import synth
    m = synth.point_cloud(578, 289, seed=3)            # 167 042 points
    blob = ca.encode(m, position_bits=14, normal_bits=10, normal_prediction=0)
    b = run_batch(ctx, [blob])
    got = b.host_outputs(0)
    assert got["nvert"] == 167042 and got["nface"] == 0
    assert_same(got, oc.decode(blob), KEYS, "C3")
    r = _maybe_ref_decode(blob)
    if r is not None:
        assert_same(got, r, KEYS, "C3 vs reference")


def test_random_corpus_one_batch(ctx):
    """the oracle-vs-reference corpus (tests/test_oracle_vs_reference.py: spheres, holey discs with shuffled vertex order, tori,
    closed spheres, a merge of components with a 4-component colour) plus clouds, ribbons and two-group meshes, every blob with its
    own quantisation / normal prediction / entropy, decoded in ONE batch: each equals the oracle byte for byte (f32 and i16/u16)"""
    from corto_amd import synth as S
    rng = np.random.default_rng(2027)
    meshes = []
    for seed in range(6):
        meshes += [S.bumpy_sphere(8 + 7 * seed, 4 + 5 * seed, seed), S.shuffled(S.holey_disc(6 + 6 * seed, seed, hole_frac=0.05 + 0.05 * seed), seed),
                   S.torus(6 + 5 * seed, 4 + 3 * seed, seed), S.closed_sphere(5 + 4 * seed, 3 + 3 * seed, seed)]
    meshes += [S.merge([S.closed_sphere(9, 5, 1), S.closed_sphere(7, 4, 2), S.torus(8, 5, 3), S.holey_disc(9, 4, color_components=4)]),
               S.strip(37, seed=4), S.point_cloud(17, 9, 1), S.point_cloud(64, 40, 2), S.bumpy_sphere(20, 10, 5, color_components=3)]
    blobs = []
    for k, m in enumerate(meshes):
        cloud = m.nface == 0
        pred = (ca.DIFF if cloud else (ca.DIFF, ca.ESTIMATED, ca.BORDER)[k % 3])
        blobs.append(ca.encode(m, position_bits=int(rng.choice([9, 12, 14, 18])), normal_bits=int(rng.choice([8, 10, 12])), uv_bits=int(rng.choice([8, 12])),
                               normal_prediction=pred, entropy=0 if k % 7 == 3 else 1))
    for kw, okw in ((dict(color_components=4), dict(color_components=4)),
                    (dict(color_components=4, normal_format=ca.FMT_INT16, index16=True), dict(color_components=4, normal_format=oc.FMT_INT16, index16=True))):
        b = run_batch(ctx, blobs, **kw)
        for i, blob in enumerate(blobs):
            assert_same(b.host_outputs(i), oc.decode(blob, **okw), KEYS, "corpus blob %d %s" % (i, sorted(kw)))
        b.close()


def test_topology_lds_slot_overflow_redone_on_hbm_front():
    """the LDS automaton holds the LIVE front: a ring of 8*sqrt(nface) queued edges and a pool as large for surviving ones.  A
    torus' queue and a ribbon's boundary outgrow that; those blobs are redone on the HBM front - same results, reported in the
    stats - in one batch with blobs that fit.  The context learns from it: the next decode is planned with as many edge slots as the redone blobs
    report they would have needed (twice here; round 4 went up four-fold whatever was missing) and keeps all of them in LDS."""
    from corto_amd import synth
    ctx = ca.Context(0)                      # its own context: the feedback is per context
    meshes = [synth.strip(400, seed=3), synth.bumpy_sphere(24, 12, seed=5), synth.torus(100, 50, seed=4), synth.holey_disc(40, seed=2, color_components=4)]
    blobs = [ca.encode(m, normal_prediction=ca.BORDER) for m in meshes]
    for u16, scale, fallbacks in ((False, 1, 2), (True, None, 0), (False, None, 0)):
        b = run_batch(ctx, blobs, index16=u16)
        for i in range(len(blobs)):
            exp = oc.decode(blobs[i])
            if u16:
                exp["index"] = exp["index"].astype(np.uint16)
            assert_same(b.host_outputs(i), exp, KEYS, "blob %d u16=%s" % (i, u16))
        # first pass: the torus (queue of 3 800) and the holey disc (each hole adds boundary the header does not show); the ribbon's 800 boundary
        # edges are in the header (2V - F, kernels.h: topo_boundary_estimate) and its pool is planned for them from the start (round 5)
        # (later passes: ring and pool each by the factor the redone blobs reported - the torus' queue four-fold, the disc's pool 2-3 x; `topology_scale` is the larger)
        assert b.stats().topology_fallbacks == fallbacks and (b.stats().topology_scale == scale if scale else 2 <= b.stats().topology_scale <= 4), (b.stats().topology_scale, b.stats().topology_fallbacks)
        b.close()
    ctx.close()


def test_large_and_small_meshes_in_one_batch(ctx):
    """a 66K-triangle mesh (its automaton gets a 64 KB front, launched apart from the small ones), a 15K-triangle one and 4K-triangle
    blobs in one batch; the large one's symbol window (8K symbols) is slid ~8 times"""
    from corto_amd import synth
    meshes = [synth.bumpy_sphere(64, 32, seed=31), synth.bumpy_sphere(260, 128, seed=32), synth.torus(48, 24, seed=33),
              synth.bumpy_sphere(124, 62, seed=34), synth.bumpy_sphere(64, 32, seed=35)]
    blobs = [ca.encode(m, normal_prediction=p) for m, p in zip(meshes, (ca.BORDER, ca.ESTIMATED, ca.DIFF, ca.BORDER, ca.ESTIMATED))]
    b = run_batch(ctx, blobs)
    for i in range(len(blobs)):
        assert_same(b.host_outputs(i), oc.decode(blobs[i]), KEYS, "blob %d" % i)
    assert b.stats().topology_fallbacks == 0


def test_wide_context_on_meshes_beyond_the_lds_records(monkeypatch):
    """with 32-bit records (a context that met values beyond int16, or $CORTO_DELTA_WIDE=1) K-DELTA keeps meshes of up to 32 767 vertices in LDS;
    33K-66K vertices (16-bit ids would still do) and positions quantised to 18 bits take the stretch walk over HBM (k_delta_mesh).  A performance trade-off,
    not a correctness one: the same bytes as the oracle on both sides of the limit, narrow and wide, rounds forced or not"""
    from corto_amd import synth
    meshes = [synth.bumpy_sphere(260, 128, seed=41), synth.bumpy_sphere(250, 130, seed=42), synth.delaunay_disc(2310, seed=43, holes=5),
              synth.decimated(synth.icosphere(4, seed=44), keep=0.7, seed=44), synth.bumpy_sphere(64, 32, seed=45)]
    blobs = [ca.encode(m, position_bits=18 if i % 2 == 0 else 14, normal_prediction=(ca.BORDER, ca.ESTIMATED, ca.DIFF)[i % 3]) for i, m in enumerate(meshes)]
    refs = [oc.decode(bl) for bl in blobs]
    for env in ({}, {"CORTO_DELTA_WIDE": "1"}, {"CORTO_DELTA_WIDE": "1", "CORTO_DELTA_ROUNDS": "1"}):
        for k in ("CORTO_DELTA_WIDE", "CORTO_DELTA_ROUNDS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        c = ca.Context(0)
        for _ in range(2):                                   # (the second decode: with what the first taught the context)
            b = run_batch(c, blobs)
            for i, r in enumerate(refs):
                assert_same(b.host_outputs(i), r, KEYS, "blob %d env %s" % (i, env))
            b.close()
        c.close()


@pytest.mark.parametrize("env", [{}, {"CORTO_DELTA_WALK": "1"}])
def test_big_meshes_through_the_delta_tiles(monkeypatch, env):
    """Attributes too big for K-DELTA's LDS records go through k_delta_tiles (tiles of 1 024 vertices out of an LDS ring of the last 4 096 values;
    $CORTO_DELTA_WALK=1: rounds 1-5's stretch walk over L2): a 45K-vertex mesh with SHUFFLED vertex ids (parents anywhere: behind the ring, inside the
    tile, one back - every pass pattern), flipped diagonals, a torus, 18-bit positions, rgb colours, several groups - on a lone context (the tiles run on
    the second stream BESIDE the automaton and wait for its progress word) and on a single-stream one (behind it); plus one of each in ONE batch"""
    from corto_amd import synth
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    meshes = [synth.shuffled(synth.bumpy_sphere_flipped(300, 150, seed=5, flip=0.3), seed=9),
              synth.bumpy_sphere_flipped(260, 140, seed=6, color_components=3),
              synth.torus(220, 160, seed=7),
              synth.shuffled(synth.delaunay_disc(40000, seed=8, holes=9), seed=3, faces=False)]
    meshes[1].groups = [5000, 5001, 40000, meshes[1].nface]
    kws = [dict(normal_prediction=ca.BORDER), dict(normal_prediction=ca.ESTIMATED, position_bits=18), dict(normal_prediction=ca.DIFF), dict(normal_prediction=ca.BORDER, position_bits=16)]
    blobs = [ca.encode(m, **kw) for m, kw in zip(meshes, kws)]
    refs = [oc.decode(x, color_components=4) for x in blobs]
    for single in (False, True):
        c = ca.Context(0)
        if single:
            c.set_single_stream(True)
        for i, x in enumerate(blobs):
            b = run_batch(c, [x], color_components=4)
            assert_same(b.host_outputs(0), refs[i], KEYS, "big mesh %d single=%s env %s" % (i, single, env))
            b.close()
        b = run_batch(c, blobs, color_components=4)
        for i, r in enumerate(refs):
            assert_same(b.host_outputs(i), r, KEYS, "big meshes in one batch: %d single=%s env %s" % (i, single, env))
        b.close()
        if not env:
            # eighteen big meshes at once: more tile workgroups (54) than may wait BESIDE the automata (48: they hold 66 KB of LDS each and the automata of
            # big meshes ask for up to 156 KB) - these run behind them, and sixteen (48) beside them
            for n in (18, 16):
                b = run_batch(c, [blobs[1], blobs[2]] * (n // 2), color_components=4)
                for i in range(n):
                    assert_same(b.host_outputs(i), refs[1 + (i & 1)], KEYS, "%d big meshes in one batch: %d single=%s" % (n, i, single))
                b.close()
        c.close()


def test_nonlattice_c4_sized_blobs_against_the_reference_digests(ctx):
    """eight C4-sized blobs with no lattice in them - bench.py's `realistic` Delaunay discs, decimated spheres, an icosphere, a cone of fans - in one batch, twice
    (the second decode with the edge slots the first taught the context), u32 indices: SHA-256 of every output array = what the compiled REFERENCE decoded
    (tests/golden/nonlattice_blobs8.npz), not only what the oracle says"""
    z = np.load(os.path.join(GOLDEN, "nonlattice_blobs8.npz"))
    names = z["names"].tobytes().decode().split(",")
    blobs = [aligned(z["crt_" + n]) for n in names]
    c = ca.Context(0)
    for attempt in range(2):
        b = run_batch(c, blobs, color_components=4)
        for i, n in enumerate(names):
            got = b.host_outputs(i)
            for k in ("position", "normal", "color", "uv", "index"):
                assert hashlib.sha256(np.ascontiguousarray(got[k]).tobytes()).hexdigest() == z["%s_sha256_%s" % (k, n)].tobytes().decode(), (n, k, attempt)
        if attempt:
            assert b.stats().topology_fallbacks == 0
        b.close()
    c.close()


def test_high_valence_vertices(ctx):
    """K-NRM adds a vertex' incident face normals in ascending face id (src/normal_attribute.cpp:40-59): <= 8 faces sort in registers, <= 16 in Batcher's
    network, <= 512 by their wave (ranks by counting), more by walking the faces in order - a cone's apex of valence 9 .. 3 000 takes each of them, with
    ESTIMATED and BORDER normals (float and int16), and the delta stage's round loop meets its fans"""
    from corto_amd import synth
    ks = (9, 12, 16, 17, 33, 64, 65, 128, 300, 512, 513, 700, 3000)
    meshes = [synth.cone_fan(k, 1 + i % 3, seed=k, closed=bool(i & 1), flip=0.3) for i, k in enumerate(ks)]
    blobs = [ca.encode(m, normal_prediction=(ca.ESTIMATED, ca.BORDER)[i % 2]) for i, m in enumerate(meshes)]
    for nf in (oc.FMT_FLOAT, oc.FMT_INT16):
        b = ca.Batch(ctx, blobs)
        b.allocate_outputs(fill=0, normal_format=nf)
        b.decode()
        assert (b.sync() == 0).all()
        for i, bl in enumerate(blobs):
            assert_same(b.host_outputs(i), oc.decode(bl, normal_format=nf), KEYS, "cone of valence %d, normals %s" % (ks[i], "i16" if nf == oc.FMT_INT16 else "f32"))
        b.close()


def test_js_veneer_decode(ctx):
    """newDecoder / set* / decode / deleteDecoder (include/corto/emcorto.h = upstream html/js/emscripten/emcorto.cpp:14-89)
    driven the way corto.em.js does: sizes from nvert/nface, u16 index when nvert < 65536, int16 normals on request"""
    from test_abi_cpu import em_veneer
    E = em_veneer()
    for name, n16, i16 in (("c4_unit", False, True), ("two_groups", True, False), ("cloud_diff", False, False), ("nrm_estimated_rgb", True, True)):
        g = load_golden(name)
        blob = aligned(g["crt"])
        d = E.newDecoder(len(blob), blob.ctypes.data)
        assert d
        nv, nf = E.nvert(d), E.nface(d)
        pos = np.zeros((nv, 3), np.float32); uv = np.zeros((nv, 2), np.float32); col = np.zeros((nv, 4), np.uint8)
        nrm = np.zeros((nv, 3), np.int16 if n16 else np.float32)
        idx = np.zeros((nf, 3), np.uint16 if i16 else np.uint32)
        E.setPositions(d, pos.ctypes.data)
        if E.hasNormal(d): (E.setNormals16 if n16 else E.setNormals32)(d, nrm.ctypes.data)
        if E.hasColor(d): E.setColors(d, col.ctypes.data, 4)
        if E.hasUv(d): E.setUvs(d, uv.ctypes.data)
        if nf: (E.setIndex16 if i16 else E.setIndex32)(d, idx.ctypes.data)
        E.decode(d)
        ng = E.ngroups(d)
        ends = np.zeros(max(ng, 1), dtype=np.int32)
        E.groups(d, ends.ctypes.data)
        assert ng == (2 if name == "two_groups" else 1 if nf else 0), (name, ng)
        if ng: assert ends[ng - 1] == nf and (np.diff(ends[:ng]) > 0).all()
        E.deleteDecoder(d)
        exp = oc.decode(g["crt"], color_components=4, normal_format=ca.FMT_INT16 if n16 else ca.FMT_FLOAT, index16=i16)
        assert pos.tobytes() == exp["position"].tobytes(), name
        if nf: assert idx.tobytes() == exp["index"].tobytes(), name
        if "normal" in exp: assert nrm.tobytes() == exp["normal"].tobytes(), name
        if "uv" in exp: assert uv.tobytes() == exp["uv"].tobytes(), name
        if "color" in exp: assert col.tobytes() == exp["color"].tobytes(), name


def test_batch_reset_reuses_the_object_for_other_blobs(ctx):
    """crthip_batch_reset = destroy + create on the same object (a serving loop's per-batch call): different blobs, different
    count, attributes in a different order of presence, a point cloud after meshes - nothing of the previous plan may leak"""
    sets = [["c4_unit", "torus", "two_groups"], ["cloud_border", "radius_attr"], ["entropy_none"], ["holey_disc", "c4_unit", "closed_sphere", "multi_component"]]
    b = None
    for names in sets + sets[:1]:
        gs = [load_golden(nm) for nm in names]
        blobs = [aligned(g["crt"]) for g in gs]
        if b is None:
            b = ca.Batch(ctx, blobs)
        else:
            b.reset(blobs)
        b.allocate_outputs(fill=0, color_components=4)
        b.decode()
        st = b.sync()
        assert (st == 0).all()
        for i, g in enumerate(gs):
            got = b.host_outputs(i)
            exp = oc.decode(g["crt"], color_components=4)
            assert_same(got, exp, KEYS, "%s after reset" % names[i])


def test_config4_256_distinct_blobs(ctx):
    """256 distinct 4K-tri blobs in one batch: every blob equals the oracle; decoded positions equal the quantised inputs"""
    from corto_amd import synth
    meshes = [synth.bumpy_sphere(64, 32, seed=1000 + s) for s in range(256)]
    blobs = [ca.encode(m, normal_prediction=ca.BORDER) for m in meshes]
    b = run_batch(ctx, blobs)
    st = b.stats()
    assert st.total_nface == 256 * 4096 and st.total_nvert == 256 * 2112
    assert st.topology_fallbacks == 0
    import hashlib
    for i in range(256):                      # every blob, every array, by SHA-256 of the raw bytes against the oracle's
        got, exp = b.host_outputs(i), oc.decode(blobs[i])
        for k in ("position", "normal", "color", "uv", "index"):
            assert hashlib.sha256(np.ascontiguousarray(got[k]).tobytes()).digest() == hashlib.sha256(np.ascontiguousarray(exp[k]).tobytes()).digest(), "C4 blob %d %s" % (i, k)
    # round trip: the multiset of decoded positions is the multiset of quantised input positions
    got = b.host_outputs(77)
    q = [a["q"] for a in b.infos[77].attrs() if a["name"] == "position"][0]
    want = np.sort(np.trunc(meshes[77].position / np.float32(q)).astype(np.int64).view([("", np.int64)] * 3), axis=0)
    have = np.sort(np.rint(got["position"] / np.float32(q)).astype(np.int64).view([("", np.int64)] * 3), axis=0)
    assert np.array_equal(want, have)


def test_misaligned_float_and_int16_buffers_are_refused(ctx):
    """a packed generic buffer doubles as int32 workspace and leaves K-DELTA as floats through dword / 16-byte stores: a FLOAT binding
    that is not 4-byte aligned (INT16 normals: 2) is refused at bind time - never decoded into something else"""
    import ctypes as C
    import torch
    g = load_golden("c4_unit")
    blob = aligned(g["crt"])
    L = ca.lib()
    b = ca.Batch(ctx, [blob])
    b.allocate_outputs()
    info = b.infos[0]
    attrs = info.attrs()
    buf = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda")
    for which, fmt, off, want in (("position", ca.FMT_FLOAT, 2, -8), ("uv", ca.FMT_FLOAT, 1, -8), ("normal", ca.FMT_FLOAT, 2, -8), ("normal", ca.FMT_INT16, 1, -8),
                                  ("position", ca.FMT_FLOAT, 4, 0), ("normal", ca.FMT_INT16, 2, 0), ("color", ca.FMT_UINT8, 3, 0)):
        binds = (ca.AttrBinding * len(attrs))()
        for k, a in enumerate(attrs):
            binds[k].buffer = buf.data_ptr() + 65536 * k + (off if a["name"] == which else 0)
            binds[k].format = fmt if a["name"] == which else (ca.FMT_UINT8 if a["name"] == "color" else ca.FMT_FLOAT)
            binds[k].out_components = 4 if a["name"] == "color" else 0
        rc = L.crthip_batch_bind(b.handle, 0, binds, None, ca.FMT_UINT32)
        assert rc == want, (which, fmt, off, rc)
    b.close()


def test_decode_host_without_bindings_and_with_a_stride(ctx):
    """crthip_decode_host: attrs == NULL binds nothing (an index-only decode, like a Decoder nobody called set*() on); host buffers are
    packed, so a stride is refused rather than silently ignored"""
    import ctypes as C
    g = load_golden("c4_unit")
    blob = aligned(g["crt"])
    L = ca.lib()
    idx = np.zeros((len(g["index"]), 3), np.uint32)
    assert L.crthip_decode_host(ctx.handle, blob.ctypes.data, len(blob), None, idx.ctypes.data, ca.FMT_UINT32) == 0
    assert idx.tobytes() == g["index"].tobytes()
    attrs = ca.probe(blob).attrs()
    binds = (ca.AttrBinding * len(attrs))()
    pos = np.zeros((len(g["position"]), 4), np.float32)
    for k, a in enumerate(attrs):
        if a["name"] == "position":
            binds[k].buffer = pos.ctypes.data; binds[k].format = ca.FMT_FLOAT; binds[k].stride = 16
    assert L.crthip_decode_host(ctx.handle, blob.ctypes.data, len(blob), binds, None, ca.FMT_UINT32) == -8
    assert not pos.any()
    for k, a in enumerate(attrs):
        binds[k].stride = 0
    pos3 = np.zeros((len(g["position"]), 3), np.float32)
    for k, a in enumerate(attrs):
        if a["name"] == "position":
            binds[k].buffer = pos3.ctypes.data
    assert L.crthip_decode_host(ctx.handle, blob.ctypes.data, len(blob), binds, None, ca.FMT_UINT32) == 0
    assert pos3.tobytes() == g["position"].tobytes()


def test_js_veneer_reports_a_failed_decode(ctx):
    """upstream's decode() throws across the wasm boundary; here it returns, and lastError() (this repo's one addition to the eighteen
    symbols) says what upstream would have thrown - a failed decode is not mistaken for a successful one"""
    from test_abi_cpu import em_veneer
    E = em_veneer()
    g = load_golden("c4_unit")
    good = aligned(g["crt"])
    bad = aligned(g["crt"].copy())
    probs = int(ca.probe(bad).body_offset) + 9 + 4 + 1            # the CLERS stream's probability table: symbols no automaton accepts
    bad[probs:probs + 2] = (7, 255)
    for blob, want in ((bad, -5), (good, 0), (bad, -5)):
        d = E.newDecoder(len(blob), blob.ctypes.data)
        assert d and E.lastError(d) == 0
        nv, nf = E.nvert(d), E.nface(d)
        pos = np.zeros((nv, 3), np.float32); idx = np.zeros((nf, 3), np.uint32)
        E.setPositions(d, pos.ctypes.data); E.setIndex32(d, idx.ctypes.data)
        E.decode(d)
        assert E.lastError(d) == want, want
        if want == 0:
            assert pos.tobytes() == g["position"].tobytes() and idx.tobytes() == g["index"].tobytes()
            E.decode(d)
            assert E.lastError(d) == 0
        E.deleteDecoder(d)


def test_pool_poisons_outputs_before_the_last_steps(c5_blobs):
    """what crthip_pool_lane_read returns after a run was written by the run's last steps: the pool fills every context's output block with
    0xA5 (on the context's own stream) before each step of the last round, and says how many contexts ended that way; a byte between two
    output arrays - never written by a decode - still holds the poison, the arrays hold the oracle's bytes"""
    items = [c5_blobs[300:332], c5_blobs[900:932]]
    pool = ca.Pool([0], threads=2, depth=2)
    arenas = [[ca.upload_arena(it, 0)] for it in items]
    rep, _ = pool.run(items, steps=24, warmup=4, arenas=arenas)
    assert rep.failed_blobs == 0 and rep.poisoned_lanes == pool.lanes == 4
    assert rep.host_us_per_step > 0
    for lane in range(pool.lanes):
        it, _slot = pool.lane_item(lane)
        ref = oc.decode(items[it][5])
        got = pool.lane_read(lane, 5, "position", np.float32, ref["nvert"] * 3)
        assert got.tobytes() == ref["position"].tobytes()
        assert (pool.lane_read(lane, 0, "#tail", np.uint8, 256) == 0xA5).all()
    assert isinstance(pool.warning, str)
    pool.close()


@pytest.mark.timeout(600)
def test_bench_eight_pool_devices_on_one_gpu():
    """`python bench.py --gpus 8` end to end without eight GPUs ($BENCH_SHARE_GPU=1: eight pool devices on this box's one GPU): the
    single-process launch form really runs eight devices - config C5's eight shards, each resident on its home device only, one
    ticket queue - and every one of them decodes.  The line is the driver's contract: one JSON object, n_gpus 8."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BENCH_SHARE_GPU="1")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "5", "--no-cpu", "--no-tunstall-scaled",
                          "--no-other-configs", "--sustain", "0.3", "--host-threads", "1", "--depth", "2"], env=env, cwd=root, capture_output=True, text=True, timeout=560)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 8 and j["shared_gpu"] is True and j["bit_exact"] is True
    assert len(j["steps_per_device"]) == 8 and all(x > 0 for x in j["steps_per_device"]), j["steps_per_device"]
    assert j["poisoned_lanes"] == 16 and j["config"]["launch"] == "single-process-queue"
    assert j["host_us_per_step_per_thread"] > 0 and j["sustained"]["seconds"] >= 0.3
    assert len(j["per_gpu_mtri_per_s"]) == 8 and j["scaling_report"]["one_gpu_alone_mtri_per_s"] > 0 and j["scaling_report"]["efficiency_vs_1gpu"] > 0


@pytest.mark.timeout(600)
def test_bench_two_ranks_under_torch_distributed_run():
    """the driver's launch form for N > 1 - `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` - with two ranks sharing this
    box's one GPU ($BENCH_SHARE_GPU=1: gloo carries the barrier and the reductions): rank 0 prints ONE JSON line with n_gpus 2, both ranks'
    rates in per_gpu_mtri_per_s and the scaling report SURVEY 8e asks for"""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BENCH_SHARE_GPU="1")
    for attempt in range(2):                        # (the free port is found, released and then used: another process may take it in between - once more then)
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]
        out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                              os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5",
                              "--sustain", "0.2", "--host-threads", "2", "--depth", "3"], env=env, cwd=root, capture_output=True, text=True, timeout=280)
        if out.returncode == 0:
            break
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["bit_exact"] is True and j["config"]["launch"] == "process-per-gpu"
    assert len(j["per_gpu_mtri_per_s"]) == 2 and all(x > 0 for x in j["per_gpu_mtri_per_s"]), j["per_gpu_mtri_per_s"]
    sr = j["scaling_report"]
    assert sr["one_gpu_alone_mtri_per_s"] > 0 and 0 < sr["efficiency_vs_1gpu"] < 3.0 and sr["host_us_per_step_per_thread"] > 0, sr
    assert "cpu_baseline" not in j and "n1_only_legs" in j        # (the single-GPU characterisations belong to the N = 1 line)


def test_delta_values_beyond_int16_are_redone_and_the_context_learns():
    """K-DELTA (k_delta.hip) keeps an attribute's values in LDS as int16 relative to vertex 0 - checked, not assumed: positions quantised
    to 17 / 20 bits span more than that, the wave redoes them on the 32-bit values in HBM (bit-exact, counted in stats.delta_redone), and
    the context plans its NEXT batches with 32-bit values in LDS (stats.delta_wide).  A mesh far from the origin (large absolute
    coordinates, 14-bit span) does NOT overflow: the values are relative.  Blobs of both kinds share one batch."""
    from corto_amd import synth
    c = ca.Context(0)
    far = synth.bumpy_sphere(40, 20, seed=5)
    far.position = far.position + np.float32(900.0)                      # quantised coordinates around 900 / q: far beyond int16, span 2^14
    meshes = [(synth.bumpy_sphere(48, 24, seed=1), 14), (synth.bumpy_sphere(24, 12, seed=2), 17), (far, 14), (synth.torus(24, 12, seed=3), 20),
              (synth.bumpy_sphere_flipped(32, 16, seed=4), 15), (synth.holey_disc(20, seed=6), 16)]
    blobs = [ca.encode(m, position_bits=bits, uv_bits=12, normal_bits=10, normal_prediction=ca.BORDER if k % 2 else ca.DIFF) for k, (m, bits) in enumerate(meshes)]
    refs = [oc.decode(b_, color_components=4) for b_ in blobs]
    b = run_batch(c, blobs, color_components=4)
    st = b.stats()
    assert st.delta_wide == 0 and 2 <= st.delta_redone <= 3, (st.delta_wide, st.delta_redone)     # 17 and 20 bits for sure; 16 bits by its extent
    for i, r in enumerate(refs):
        assert_same(b.host_outputs(i), r, KEYS, "blob %d, %d bits, narrow plan" % (i, meshes[i][1]))
    b.decode(); b.sync()                                                     # the same batch again: planned wide now, nothing redone
    st = b.stats()
    assert st.delta_wide == 1 and st.delta_redone == 0
    for i, r in enumerate(refs):
        assert_same(b.host_outputs(i), r, KEYS, "blob %d, %d bits, wide plan" % (i, meshes[i][1]))
    b.close()
    # a context that only ever sees 14-bit meshes stays narrow
    b2 = run_batch(ca.Context(0), blobs[:1] + blobs[2:3], color_components=4)
    assert b2.stats().delta_wide == 0 and b2.stats().delta_redone == 0
    b2.close(); c.close()


def test_bit_unpack_hands_int16_values_on_where_the_tables_prove_them(monkeypatch):
    """K-BIT (k_unpack_wave) writes an attribute's raw deltas as int16 when the probability table of its log stream holds no width above 16 bits (decodeArray,
    cstream.h:337-357: v in [-2^(d-1), 2^(d-1)); the per-component decodeValues folds the sign the other way: 15) and the reader is k_delta_lds16 / k_normal_blob
    (plan_jobs.cpp; crthip_batch_stats.int16_streams) - half the bytes of that trip through HBM.  A finely tessellated sphere at 18 bits has small deltas and a large
    extent: int16 in, relative values beyond int16 - K-DELTA widens the halfwords in place and redoes the attribute on 32-bit values.  Streams with wider
    fields (the 31-bit fixture) and the wide plan of a context that has learnt stay 32-bit.  Same bytes as the oracle and as $CORTO_VALUES_I32=1."""
    from corto_amd import synth
    fine = ca.encode(synth.bumpy_sphere(120, 60, seed=3), position_bits=18, uv_bits=12, normal_bits=10, normal_prediction=ca.BORDER)
    c4 = ca.encode(synth.bumpy_sphere(64, 32, seed=1), position_bits=14, uv_bits=12, normal_bits=10, normal_prediction=ca.ESTIMATED)
    f31 = load_golden("fields31")["crt"]
    blobs = [fine, c4, f31]
    refs = [oc.decode(b_, color_components=4) for b_ in blobs]
    c = ca.Context(0)
    b = run_batch(c, blobs[:1], color_components=4)
    st = b.stats()
    assert st.int16_streams >= 2 and st.delta_redone == 1 and st.delta_wide == 0, (st.int16_streams, st.delta_redone)      # positions + uv + normal corrections as halfwords; the positions redone
    assert_same(b.host_outputs(0), refs[0], KEYS, "fine sphere, 18 bits: int16 deltas widened in place")
    b.close()
    b = run_batch(c, blobs, color_components=4)                             # the context has learnt: 32-bit records, 32-bit values (the corrections of the normals stay halfwords)
    assert b.stats().delta_wide == 1
    for i, r in enumerate(refs):
        assert_same(b.host_outputs(i), r, KEYS, "blob %d, wide plan" % i)
    b.close(); c.close()
    c = ca.Context(0)
    b = run_batch(c, blobs[1:], color_components=4)
    n16 = b.stats().int16_streams
    assert n16 >= 3                                                          # the C4-like blob's positions, uv and corrections; not the 31-bit fields
    for i, r in enumerate(refs[1:]):
        assert_same(b.host_outputs(i), r, KEYS, "blob %d, narrow plan" % (i + 1))
    b.close(); c.close()
    monkeypatch.setenv("CORTO_VALUES_I32", "1")
    c = ca.Context(0)
    b = run_batch(c, blobs, color_components=4)
    assert b.stats().int16_streams == 0
    for i, r in enumerate(refs):
        assert_same(b.host_outputs(i), r, KEYS, "blob %d, 32-bit values" % i)
    b.close(); c.close()
