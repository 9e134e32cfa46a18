"""Design study for k_prim (round 3): how many serial ROUNDS does a batched Prim need?

A round takes the frontier points at the current minimum level m (reach == m) that fall into the 64-index
window of the lowest one, in index order f1 < f2 < ..., and accepts the longest prefix for which the library's
Prim (oracle/hdbscan_ref.prim_mst: lowest index wins) provably picks exactly these points next:
after adding f1..fi the next pick is f(i+1) iff no point's reach dropped below m and no point outside the
frontier with an index below f(i+1) reached level m.
Prints rounds per map and checks the emitted (u, v, w) sequence against the oracle."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import hdbscan_ref as H

INF = 1 << 40


def batched_prim(X, core, window=True, cap=64):
    n = len(X)
    X = X.astype(np.int64)
    reach = np.full(n, INF, np.int64)
    intree = np.zeros(n, bool)
    u, v, w = [], [], []
    cur = 0
    intree[0] = True
    def relax(t):
        d = (X[:, 0] - X[t, 0]) ** 2 + (X[:, 1] - X[t, 1]) ** 2
        return np.maximum(np.maximum(d, core), core[t])
    r0 = relax(0); reach = np.where(intree, INF, np.minimum(reach, r0))
    rounds = 0
    sizes = []
    while len(u) < n - 1:
        rounds += 1
        m = reach.min()
        F = np.flatnonzero(reach == m)
        f1 = F[0]
        if window:
            c = F[F < (f1 // 64 + 1) * 64]
        else:
            c = F[:cap]
        acc = 0
        pend = INF                       # lowest index that entered level m during this round
        for i, f in enumerate(c):
            u.append(cur); v.append(f); w.append(m); cur = f
            intree[f] = True
            reach[f] = INF
            acc += 1
            nr = relax(f)
            nr[intree] = INF
            better = nr < reach
            drop = np.any(better & (nr < m))
            newm = np.flatnonzero(better & (nr == m))
            reach = np.minimum(reach, nr)
            if drop:
                break
            if len(newm):
                pend = min(pend, newm[0])
            if i + 1 < len(c) and pend < c[i + 1]:
                break
        sizes.append(acc)
    return np.array(u), np.array(v), np.array(w), rounds, sizes


def main():
    z = np.load(os.path.join(os.path.dirname(__file__), '..', '..', 'tests', 'golden', 'hdbscan_tieorder.npz'))
    tot = {}
    for idx in list(range(0, 63, 4)) + list(range(63, 126, 8)):
        best = bool(z['set_%d' % idx])
        bits = np.unpackbits(z['map_%d' % idx])
        hw = (35, 62) if best else (140, 250)
        occ = bits[:hw[0] * hw[1]].reshape(hw)
        X = np.argwhere(occ)
        n = len(X)
        k = H.effective_min_samples(n, 5 if best else 26, 3 if best else None)
        core = H.core_distances(X, k)
        ou, ov, ow = H.prim_mst(X, core)
        line = '%3d %s N=%4d' % (idx, 'best' if best else 'dflt', n)
        for name, kw in (('win64', dict(window=True)), ('first64', dict(window=False))):
            u, v, w, rounds, sizes = batched_prim(X, core, **kw)
            assert np.array_equal(u, ou) and np.array_equal(v, ov) and np.array_equal(w, ow)
            line += ' | %s rounds %4d (%.1f/round, max %d)' % (name, rounds, (n - 1) / rounds, max(sizes))
        # level changes in the sequence
        line += ' | level changes %d, distinct %d' % (np.count_nonzero(np.diff(ow)), len(np.unique(ow)))
        print(line, flush=True)


if __name__ == '__main__':
    main()
