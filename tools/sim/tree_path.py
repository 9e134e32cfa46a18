"""Executable specification of the data-parallel hierarchy kernel (k_tree, round 3).

The library's Prim records every edge as (last node added, new node, weight) (oracle/hdbscan_ref.prim_mst), so the
"MST" that single linkage sees is a PATH through the points in Prim order: edge k joins positions k and k + 1.
Single linkage of a path merges ADJACENT INTERVALS in the order of the edges' sorted positions (rank), hence
  * when edge k is processed its two sides are the intervals reaching out to the nearest edges of greater rank:
    left size k - PGE(k), right size NGE(k) - k  (all nearest greater values: data parallel);
  * the dendrogram is the Cartesian tree of the rank array: parent(k) = the lower-ranked of PGE(k), NGE(k);
  * an edge is a small union / the birth of a condensed cluster / a true split / an absorption by its side sizes alone;
  * the condensed cluster on top of a big side follows the chain of "big child" pointers down to the first birth or
    split (pointer jumping); a point falls out at the first ancestor of its leaf that is not a small union.
Everything except the float64 stability sums (kept in the library's row order) is then a per-edge or per-point formula.
Checked here against oracle/hdbscan_ref (single_linkage -> condense_tree -> select_and_label)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import hdbscan_ref as H

SMALL, BIRTH, SPLIT, ABS_A, ABS_B = range(5)      # ABS_A: the a (left) side is the big one, the right side falls out


def labels_path(u, v, w, order, mcs):
    n = len(w) + 1
    E = n - 1
    seq = np.concatenate([[u[0]], v])
    assert np.array_equal(u[1:], v[:-1])
    rho = np.empty(E, np.int64); rho[order] = np.arange(E)
    PGE = np.full(E, -1); NGE = np.full(E, E)
    st = []
    for k in range(E):
        while st and rho[st[-1]] < rho[k]:
            NGE[st.pop()] = k
        PGE[k] = st[-1] if st else -1
        st.append(k)
    sa = np.arange(E) - PGE                      # nodes on the a side (positions PGE+1 .. k)
    sb = NGE - np.arange(E)                      # nodes on the b side (positions k+1 .. NGE)
    parent = np.full(E, -1)
    for k in range(E):
        l, r = PGE[k], NGE[k]
        if l >= 0 and (r >= E or rho[l] < rho[r]): parent[k] = l
        elif r < E: parent[k] = r
    lch = np.full(E, -1); rch = np.full(E, -1)
    for k in range(E):
        if parent[k] >= 0:
            if k < parent[k]: lch[parent[k]] = k
            else: rch[parent[k]] = k
    tot = sa + sb
    cls = np.where(tot < mcs, SMALL, np.where((sa < mcs) & (sb < mcs), BIRTH, np.where((sa >= mcs) & (sb >= mcs), SPLIT,
                   np.where(sa >= mcs, ABS_A, ABS_B))))
    isc = (cls == BIRTH) | (cls == SPLIT)
    cid = np.full(E, -1)
    cid[order[isc[order]]] = np.arange(isc.sum())       # creation order = rank order
    nc = int(isc.sum())
    big = np.where(cls == ABS_A, lch, np.where(cls == ABS_B, rch, np.arange(E)))
    def top(k):                                          # device: pointer jumping
        while not isc[k]:
            k = big[k]
        return cid[k]
    evc = np.full(E, -1); evs = np.zeros(E, np.int64)    # indexed by edge k
    ctp = np.full(nc, -1); cleft = np.full(nc, -1); cright = np.full(nc, -1)
    cbirthw = np.zeros(nc, np.int64); cminw = np.zeros(nc, np.int64); cspa = np.zeros(nc, np.int64); cspb = np.zeros(nc, np.int64)
    for k in range(E):
        if cls[k] == BIRTH:
            evc[k] = cid[k]; evs[k] = tot[k]; cminw[cid[k]] = w[k]
        elif cls[k] == SPLIT:
            p = cid[k]; evc[k] = p; evs[k] = 0; cminw[p] = w[k]
            l, r = top(lch[k]), top(rch[k])
            cleft[p], cright[p] = l, r
            ctp[l] = ctp[r] = p
            cbirthw[l] = cbirthw[r] = w[k]
            cspa[p], cspb[p] = sa[k], sb[k]
        elif cls[k] == ABS_A:
            evc[k] = top(lch[k]); evs[k] = sb[k]
        elif cls[k] == ABS_B:
            evc[k] = top(rch[k]); evs[k] = sa[k]
    # stabilities in the library's row order (hdb_tree.h accumulate): sorted positions descending
    cacc = np.zeros(nc)
    for s in range(E - 1, -1, -1):
        k = order[s]; c = evc[k]
        if c < 0: continue
        lam = 1.0 / float(w[k]); birth = 1.0 / float(cbirthw[c]) if cbirthw[c] else 0.0
        acc = cacc[c]
        if evs[k] == 0:
            acc += (lam - birth) * float(cspa[c]); acc += (lam - birth) * float(cspb[c])
        else:
            term = (lam - birth) * 1.0
            for _ in range(evs[k]): acc += term
        cacc[c] = acc
    csel = np.zeros(nc, bool); crep = np.full(nc, -2)
    for c in range(nc):
        stab = cacc[c]; sub = 0.0
        if cleft[c] >= 0: sub = cacc[cleft[c]] + cacc[cright[c]]
        if sub > stab: csel[c] = False; stab = sub
        else: csel[c] = True
        cacc[c] = stab
    nsel = 0
    for c in range(nc - 1, -1, -1):
        p = ctp[c]
        if p >= 0 and crep[p] >= 0: csel[c] = False; crep[c] = crep[p]
        elif csel[c]: crep[c] = c; nsel += 1
        else: crep[c] = -2
    labels = np.full(n, -1)
    root = nc - 1
    for i in range(n):
        cand = [e for e in (i - 1, i) if 0 <= e < E]
        e = min(cand, key=lambda x: rho[x])
        while cls[e] == SMALL:
            e = parent[e]
        assert cls[e] != SPLIT
        c0 = evc[e]
        rep = crep[c0]
        if rep == -2: continue
        if rep == root and nsel == 1:
            labels[seq[i]] = rep if w[e] <= cminw[root] else -1
        else:
            labels[seq[i]] = rep
    return labels


def same_partition(a, b):
    if not np.array_equal(a < 0, b < 0): return False
    m = {}
    for x, y in zip(a, b):
        if x < 0: continue
        if m.setdefault(x, y) != y: return False
    return len(set(m.values())) == len(m)


def check(X, mcs, ms, tag):
    n = len(X)
    k = H.effective_min_samples(n, mcs, ms)
    core = H.core_distances(X, k)
    u, v, w = H.prim_mst(X, core)
    order = H.edge_order(w)
    left, right, weight, csize = H.single_linkage(u, v, w, order)
    ref = H.select_and_label(H.condense_tree(left, right, weight, csize, mcs), n)
    got = labels_path(u, v, w, order, mcs)
    ok = same_partition(ref, got)
    print('%-14s N=%5d clusters %d  %s' % (tag, n, len(set(ref[ref >= 0])), 'same labels' if ok else 'DIFFERENT'), flush=True)
    assert ok, tag


def main():
    z = np.load(os.path.join(os.path.dirname(__file__), '..', '..', 'tests', 'golden', 'hdbscan_tieorder.npz'))
    for idx in range(0, 63, 7):
        occ = np.unpackbits(z['map_%d' % idx])[:35000].reshape(140, 250)
        check(np.argwhere(occ), 26, None, 'golden %d' % idx)
    rng = np.random.RandomState(1)
    for t in range(8):
        yy, xx = np.mgrid[0:35, 0:62]
        occ = (rng.rand(35, 62) < 0.03)
        for _ in range(rng.randint(1, 4)):
            occ |= ((yy - rng.uniform(5, 30)) ** 2 + (xx - rng.uniform(5, 55)) ** 2 < rng.uniform(10, 90))
        check(np.argwhere(occ), 5, 3, 'small %d' % t)
        check(np.argwhere(occ), 2, 1, 'small %d mcs2' % t)
    for t in range(3):
        occ = rng.rand(60, 80) < (0.05, 0.2, 0.6)[t]
        check(np.argwhere(occ), 26, None, 'noise %d' % t)
        check(np.argwhere(occ), 40, 10, 'noise %d b' % t)


if __name__ == '__main__':
    main()
