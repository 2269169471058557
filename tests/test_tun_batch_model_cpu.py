"""K-TAB's batched growth (csrc/tun_tables.h) restated on the host (tools/tun_batch_model.py): the batches leave the entry arrays the one-at-a-time
loop leaves (crt::Tunstall::createDecodingTables2, src/tunstall.cpp:207-241), and the words spelled from them are the oracle's tables - on the
reference-made KATs and on random tables (sorted, low-entropy, flat, unsorted)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import tun_batch_model as tm  # noqa: E402


def test_batches_equal_the_serial_growth_and_the_oracle(capsys):
    tm.main(400)
    assert "all equal" in capsys.readouterr().out


def test_batches_are_few():
    import numpy as np
    for probs, most in (([[1, 204], [2, 51]], 30), ([[1, 128], [2, 127]], 14), ([[1, 100], [2, 80], [3, 50], [4, 25]], 10)):
        st = []
        tm.batched(np.array(probs, dtype=np.uint8), st)
        assert len(st) <= most and sum(st) == (255 - len(probs) - len(probs)) // (len(probs) - 1) + 1, (probs, st)
