"""The host model of k_topology_lds's wave-wide steps (tools/topo_run_model.py: run step with its lead lane, mix step with its one RIGHT,
chain-end step, on the ring / pool / lazy-edge structure) against the oracle's faces and prediction triples: the formulations the ISA
implements, pinned on the CPU.  Round 4 wrote a rule that was wrong on small closed fronts into ISA because the model had only been run
on a dozen meshes; this runs it on a seeded random family (a reduced `python tools/topo_run_model.py wide`)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import topo_run_model as tm


def test_model_decodes_a_random_family_like_the_oracle():
    tm.wide(n=72, seed=11)


def test_model_takes_the_steps_it_is_meant_to_take():
    """the regular C4 blob: ~100 run steps, most of them with a lead VERTEX; the irregular one: mix steps carry it, with RIGHTs among them"""
    import corto_amd as ca
    from corto_amd import synth
    from oracle import oracle as oc
    def stats(mesh):
        blob = ca.aligned_blob(ca.encode(mesh)); r = oc.decode(blob, trace=True)
        m = tm.Model(r["_clers"], r["nvert"], r["nface"], ca.probe_groups(blob), ref_faces=r["index"]); m.run()
        return m.stats, len(r["_clers"])
    s, n = stats(synth.bumpy_sphere(64, 32, seed=3))
    assert 2 * s["run_pairs"] > 0.8 * n and s["leads"] > 50 and s["serial"] < 0.1 * n, s
    s, n = stats(synth.bumpy_sphere_flipped(64, 32, seed=1))
    assert s["mix_symbols"] > 0.75 * n and s.get("mix_rights", 0) > 30 and s["serial"] < 0.12 * n, s
