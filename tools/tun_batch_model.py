"""Host model of K-TAB's BATCHED growth (round 6): the Tunstall dictionary's expansion order is a merge of n non-increasing FIFOs (one per last
symbol), so every unexpanded entry whose probability exceeds the largest child the current best head can get - c_max = (M * Pmax) >> 16 - is popped
before anything created from now on: a whole set S of pops is known at once, in the order (probability desc, row asc, FIFO order), and the k-th of
them creates entries end + k*n + r.  serial() restates crt::Tunstall::createDecodingTables2 (src/tunstall.cpp:125-256) entry by entry; batched()
does the same growth in batches the way the wave does (a window of J = 64 // n entries per row, conservative cuts); both must leave the SAME entry
arrays, and the words spelled from them must be the oracle's tables.   python tools/tun_batch_model.py [ntables]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np

CAP = 768


def seed(probs):
    n = len(probs)
    sym = [int(s) for s, _ in probs]; P = [int(p) << 8 for _, p in probs]
    prob = [0] * CAP; parent = [0] * CAP; length = [0] * CAP; last = [0] * CAP
    p0, p1 = P[0], P[1]
    count, run = 2, (p0 * p0) >> 16
    max_count = 255 // (n - 1)
    while run > p1 and count < max_count:
        run = (run * p0) >> 16; count += 1
    head = list(range(n))
    if count >= 16:
        pw = [0, p0]
        for c in range(2, count + 1):
            pw.append((pw[-1] * p0) >> 16)
        for col in range(count):
            for row in range(1, n):
                e = row + col * n
                prob[e] = P[row] if col == 0 else (pw[col] * P[row]) >> 16
                length[e] = col + 1; last[e] = sym[row]; parent[e] = -1
        first = (count - 1) * n
        prob[first] = pw[count]; length[first] = count; last[first] = sym[0]; parent[first] = -1
        head[0] = first
        nwords = 1 + count * (n - 1); end = count * n
    else:
        for i in range(n):
            prob[i] = P[i]; length[i] = 1; last[i] = sym[i]; parent[i] = -1
        nwords = n; end = n
    return dict(n=n, sym=sym, P=P, prob=prob, parent=parent, length=length, last=last, head=head, nwords=nwords, end=end, seed_end=end, count=count)


def expand_one(S, best, m=None, pop=True):
    n = S["n"]; par = S["head"][best]
    pp = S["prob"][par] if par < CAP else 0
    pl = S["length"][par] if par < CAP else 0
    m = n if m is None else m
    for r in range(m):
        e = S["end"] + r
        S["prob"][e] = (pp * S["P"][r]) >> 16; S["parent"][e] = par; S["length"][e] = pl + 1; S["last"][e] = S["sym"][r]
    S["end"] += m
    if pop:                                      # (the expansion that fills the dictionary keeps its parent even when all n children fit, tunstall.cpp:234-239)
        S["head"][best] = par + n


def best_head(S):
    best, mx = 0, 0
    for i in range(S["n"]):
        h = S["head"][i]
        p = S["prob"][h] if h < S["end"] else 0
        if p > mx:
            best, mx = i, p
    return best, mx


def finish(S):
    """the expansions left, one at a time (the reference's loop), and the last partial one"""
    n = S["n"]
    while S["nwords"] < 256:
        best, _ = best_head(S)
        full = S["nwords"] + n > 255
        expand_one(S, best, 256 - S["nwords"] if full else n, pop=not full)
        S["nwords"] += n - 1
    return S


def serial(probs):
    return finish(seed(probs))


def batched(probs, stats=None):
    S = seed(probs)
    n = S["n"]
    J = 64 // n
    Pmax = max(S["P"])
    whole = (255 - n - S["nwords"]) // (n - 1) + 1 if S["nwords"] + n <= 255 else 0
    # every row's FIFO is non-increasing - what makes a set of pops knowable at once - when the table is sorted (every stream upstream's encoder writes:
    # tunstall.cpp:108-118 sorts); a foreign table that is not keeps the one-at-a-time loop
    ordered = all(S["P"][i] >= S["P"][i + 1] for i in range(n - 1))
    while whole > 0 and n <= 32 and ordered:
        for r in range(n):                           # (the model checks the claim itself)
            h = S["head"][r]
            assert all(S["prob"][e] >= S["prob"][e + n] for e in range(h, S["end"] - n, n)), "row not sorted"
        # the window: J entries of every row from its head on
        cand = []
        for r in range(n):
            for j in range(J):
                e = S["head"][r] + j * n
                if e < S["end"]:
                    cand.append((S["prob"][e], r, j, e))
        M = max((c[0] for c in cand), default=0)
        if M == 0:
            break                                    # (all heads zero / empty: the serial loop's row-0 rule)
        cmax = (M * Pmax) >> 16
        sel = [c for c in cand if c[0] > cmax]
        # a row whose whole window is selected may hold more behind it - all of it later in the order than the window's last entry: what is safe is what
        # comes no later than the EARLIEST such last entry (keys: probability desc, row asc, place in the row asc)
        key = lambda c: (-c[0], c[1], c[2])
        cut = None
        for r in range(n):
            rows = [c for c in sel if c[1] == r]
            if len(rows) == J and S["head"][r] + J * n < S["end"]:
                cut = key(rows[-1]) if cut is None else min(cut, key(rows[-1]))
        if cut is not None:
            sel = [c for c in sel if key(c) <= cut]
        if not sel:
            break
        sel.sort(key=key)
        sel = sel[:min(J, whole)]
        base = S["end"]
        for k, (p, r, j, e) in enumerate(sel):
            for q in range(n):
                c = base + k * n + q
                S["prob"][c] = (p * S["P"][q]) >> 16; S["parent"][c] = e; S["length"][c] = S["length"][e] + 1; S["last"][c] = S["sym"][q]
        for (p, r, j, e) in sel:
            S["head"][r] += n
        S["end"] += len(sel) * n; S["nwords"] += len(sel) * (n - 1); whole -= len(sel)
        if stats is not None:
            stats.append(len(sel))
    return finish(S)


def words(S):
    """the 256 surviving words in creation order, spelled out (tunstall.cpp:243-253)"""
    n = S["n"]; out = []
    A = S["sym"][0]
    for e in range(S["end"]):
        if S["head"][e % n] > e:
            continue
        w = []; cur = e
        while S["parent"][cur] >= 0:
            w.append(S["last"][cur]); cur = S["parent"][cur]
        w.append(S["last"][cur])
        w += [A] * (S["length"][cur] - 1)
        out.append(bytes(reversed(w)))
        if len(out) == 256:
            break
    return out


def check(probs, stats=None):
    from oracle import oracle as oc
    a, b = serial(probs), batched(probs, stats)
    for k in ("prob", "parent", "length", "last", "head", "end"):
        assert a[k] == b[k], (k, probs.tolist())
    # (a table whose expansions would overrun upstream's 8 192-byte buffer - tunstall.cpp:229 asserts - is out of contract: entry arrays only)
    if b["seed_end"] + sum(b["length"][e] for e in range(b["seed_end"], b["end"])) > 8192:
        return
    idx, ln, tab = oc.tunstall_tables(probs)
    w = words(b)
    ref = [bytes(tab[idx[i]:idx[i] + ln[i]]) for i in range(256)]
    assert w == ref[:len(w)] and len(w) == 256, probs.tolist()


def main(nt=1500):
    z = np.load(os.path.join(ROOT, "tests", "golden", "tunstall_kat.npz"))
    sizes = []
    for i in range(int(z["count"])):
        pr = z["probs_%02d" % i]
        if len(pr) <= 64:
            st = []; check(pr, st); sizes.append((len(pr), len(st), sum(st)))
    rng = np.random.default_rng(7)
    for t in range(nt):
        n = int(rng.integers(2, 65)) if t % 4 == 0 else int(rng.integers(2, 12))
        kind = t % 6
        if kind == 0: p = np.sort(rng.integers(0, 256, n))[::-1]
        elif kind == 1: p = np.sort((255 * rng.dirichlet(np.ones(n) * 0.3)).astype(int))[::-1]
        elif kind == 2: p = np.array([max(254 - n, 1)] + [1] * (n - 1))
        elif kind == 3: p = np.sort((255 * rng.dirichlet(np.ones(n) * 5)).astype(int))[::-1]
        elif kind == 4: p = np.array([250] + list(np.sort(rng.integers(0, 5, n - 1))[::-1]))
        else: p = rng.integers(0, 256, n)                      # unsorted (a foreign stream)
        probs = np.stack([rng.permutation(256)[:n], np.clip(p, 0, 255)], 1).astype(np.uint8)
        st = []; check(probs, st); sizes.append((n, len(st), sum(st)))
    by = {}
    for n, nb, ne in sizes:
        a = by.setdefault(min(n, 9), [0, 0, 0]); a[0] += 1; a[1] += nb; a[2] += ne
    print("tables", len(sizes), "all equal to the serial model and the oracle")
    for n in sorted(by):
        c, nb, ne = by[n]
        print("  n %s%d: %d tables, %.1f batches and %.1f batched expansions a table (serial: %.0f)" % (">=" if n == 9 else "", n, c, nb / c, ne / c, (256 - n) / max(n - 1, 1)))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 1500)
