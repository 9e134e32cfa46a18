"""Executable specification of k_prim_lvl (round 3): the library's Prim (oracle/hdbscan_ref.prim_mst) emitted in
ROUNDS of up to 64 nodes, with disc-local probes instead of all-points updates.

State: level m (the minimum reach of the points outside the tree), F = every point outside the tree whose reach is
m, R[j] = exact reach against the first `done` tree nodes (caught up only when the level has to RISE).
Round at level m (m <= RING2): candidates f1 < f2 < ... = the first 64 members of F.  Candidate i probes the grid
disc d2 <= m around itself: a point outside the tree with mr = max(d2, core_j, core_i) < m is a DROP (after adding
f_i the level falls), one with mr == m outside F is an ENTRANT.  Accepted prefix: f1..fa where a is the first i
with a drop, or whose entrants so far (prefix minimum) precede f(i+1).  Commit: entrants of accepted candidates
join F; after a drop F = the points of the last candidate's disc with the smallest mr.  F empty -> RISE: catch R
up with the nodes added since, m = min R, F = {R == m}.  m > RING2: one node per round, then a rise.
Checked against the oracle on the golden maps and on adversarial point sets; prints the round statistics."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import hdbscan_ref as H

INF = 1 << 40
NBMAX_TEST = int(os.environ.get('NBMAX', 256))
RING_R = 20
RING2 = RING_R * RING_R


def ring_table():
    o = [(dr * dr + dc * dc, dr, dc) for dr in range(-RING_R, RING_R + 1) for dc in range(-RING_R, RING_R + 1)
         if 0 < dr * dr + dc * dc <= RING2]
    o.sort()
    return np.array(o, np.int64)


RING = ring_table()


def prim_levels(X, core, hw, cap=64, stats=None, nbmax=256):
    n = len(X)
    h, w = hw
    X = X.astype(np.int64)
    core = core.astype(np.int64)
    grid = np.full((h, w), -1, np.int64)
    grid[X[:, 0], X[:, 1]] = np.arange(n)
    intree = np.zeros(n, bool)
    R = np.full(n, INF, np.int64)
    seq = [0]
    intree[0] = True
    done = 0
    wts = []
    F = np.zeros(n, bool)
    m = None
    st = dict(rounds=0, rises=0, drops=0, slow=0, probes=0)

    # pruned rise (what the kernel does): the nodes added since the last rise form batches of <= 64 with a bounding box
    # and a smallest core distance; a (batch, chunk of 64 points) block is relaxed only when its lower bound
    # max(box distance^2, smallest cores) does not exceed UB0 = the minimum of R before the rise; blocks stay pending
    nch = (n + 63) // 64
    cbox = [(X[c * 64:c * 64 + 64, 0].min(), X[c * 64:c * 64 + 64, 0].max(), X[c * 64:c * 64 + 64, 1].min(),
             X[c * 64:c * 64 + 64, 1].max(), core[c * 64:c * 64 + 64].min()) for c in range(nch)]
    state = {'swept': 0}    # every pending block has a lower bound above this (the limit of the last sweep over all of them)
    batches = []            # (start, len, rmin, rmax, cmin, cmax, coremin)
    nodes = []              # the tree nodes as the batches see them: Prim order, or (SORTNEW=k) the nodes of a rise with >= k batches in index order
    processed = []          # per batch: set of chunks
    NBMAX = nbmax

    def relax_block(b, c):
        s0, ln = batches[b][0], batches[b][1]
        sl = slice(c * 64, min(n, c * 64 + 64))
        for t in nodes[s0:s0 + ln]:
            d = (X[sl, 0] - X[t, 0]) ** 2 + (X[sl, 1] - X[t, 1]) ** 2
            R[sl] = np.minimum(R[sl], np.maximum(np.maximum(d, core[sl]), core[t]))
        R[intree] = INF                  # (the kernel: tree members carry an infinite core distance)
        st['blocks'] = st.get('blocks', 0) + 1

    def rise():
        nonlocal done, m
        st['rises'] += 1
        R[intree] = INF
        new = list(range(done, len(seq), 64))
        if len(batches) + len(new) > NBMAX:          # table full: settle every pending block, start afresh
            for b in range(len(batches)):
                for c in range(nch):
                    if c not in processed[b] and not intree[c * 64:c * 64 + 64].all():
                        relax_block(b, c)
            batches.clear(); processed.clear(); state['swept'] = 0
            st['flushes'] = st.get('flushes', 0) + 1
        seg = seq[done:]
        sortnew = int(os.environ.get('SORTNEW', 0))
        if sortnew and len(new) >= sortnew:
            seg = sorted(seg)                    # index order = raster order: batches become compact strips, like the chunks
        nodes[done:] = seg
        for s0 in new:
            t = np.array(nodes[s0:min(len(seq), s0 + 64)])
            batches.append((s0, len(t), X[t, 0].min(), X[t, 0].max(), X[t, 1].min(), X[t, 1].max(), core[t].min()))
            processed.append(set())
        done = len(seq)

        dyn = {}
        if os.environ.get('DYNBOX', '0') == '1':       # experiment: the box of a chunk's points that are still outside the tree
            for c in range(nch):
                sl = slice(c * 64, min(n, c * 64 + 64))
                live = ~intree[sl]
                if live.any():
                    Xc = X[sl][live]
                    dyn[c] = (Xc[:, 0].min(), Xc[:, 0].max(), Xc[:, 1].min(), Xc[:, 1].max(), core[sl][live].min())

        def lower(b, c):
            r0, r1, c0, c1, cm = dyn.get(c, cbox[c])
            s0, ln, br0, br1, bc0, bc1, bm = batches[b]
            dr = max(0, br0 - r1, r0 - br1); dc = max(0, bc0 - c1, c0 - bc1)
            return max(dr * dr + dc * dc, bm, cm)

        def sweep(limit, nearest_only=False):
            for c in range(nch):
                if intree[c * 64:c * 64 + 64].all():
                    continue
                todo = [(lower(b, c), b) for b in range(len(batches)) if c not in processed[b]]
                todo = [x for x in todo if x[0] <= limit]
                if nearest_only and todo:
                    todo = [min(todo)]
                for _, b in todo:
                    relax_block(b, c); processed[b].add(c)

        single = len(new) == 1 and batches[-1][1] == 1
        if single:                                    # one new node (the start, a jump beyond the ring table): relaxed against
            b = len(batches) - 1                      # every chunk right away, so the bound below is tight
            for c in range(nch):
                if not intree[c * 64:c * 64 + 64].all():
                    relax_block(b, c); processed[b].add(c)
        ub0 = R.min()
        if single:
            if ub0 > state['swept']:                  # else: every pending block's lower bound is above ub0 already
                sweep(ub0); state['swept'] = ub0
        else:
            sweep(min(ub0, 64)); state['swept'] = min(ub0, 64)
            if ub0 > 64:                              # a far jump: tighten the bound before the wide sweep
                sweep(R.min(), nearest_only=True)
                ub2 = R.min()
                sweep(ub2); state['swept'] = max(state['swept'], ub2)
        R[intree] = INF
        m = R.min()
        F[:] = R == m

    def probe(f, m):
        k = np.searchsorted(RING[:, 0], m, side='right')
        rr = X[f, 0] + RING[:k, 1]
        cc = X[f, 1] + RING[:k, 2]
        ok = (rr >= 0) & (rr < h) & (cc >= 0) & (cc < w)
        j = grid[rr[ok], cc[ok]]
        d2 = RING[:k, 0][ok]
        keep = j >= 0
        j, d2 = j[keep], d2[keep]
        keep = ~intree[j]
        j, d2 = j[keep], d2[keep]
        st['probes'] += k
        return j, np.maximum(np.maximum(d2, core[j]), core[f])

    rise()
    while len(seq) < n:
        st['rounds'] += 1
        if m > RING2:
            f = int(np.flatnonzero(F)[0])
            seq.append(f); wts.append(m); intree[f] = True
            st['slow'] += 1
            rise()
            continue
        c = np.flatnonzero(F)[:cap]
        res = [probe(int(f), m) for f in c]          # all against the state at the start of the round
        pend = INF
        a = len(c)
        dropped = False
        for i, f in enumerate(c):
            j, mr = res[i]
            # candidates accepted earlier in this round are in the tree by now: the device sees them as members of F
            # (not entrants) and their mr >= m (else an earlier drop) -- nothing to exclude explicitly
            if np.any(mr < m):
                a = i + 1; dropped = True
                break
            ent = j[(mr == m) & ~F[j]]
            if len(ent):
                pend = min(pend, ent.min())
            if i + 1 < len(c) and pend < c[i + 1]:
                a = i + 1
                break
        for i in range(a):
            f = int(c[i])
            seq.append(f); wts.append(m); intree[f] = True; F[f] = False
        if dropped:
            st['drops'] += 1
            j, mr = res[a - 1]
            keep = ~intree[j]
            j, mr = j[keep], mr[keep]
            m = mr.min()
            F[:] = False
            F[j[mr == m]] = True
        else:
            for i in range(a):
                j, mr = res[i]
                e = j[(mr == m) & ~intree[j]]
                F[e] = True
            if not F.any():
                rise()
    seq = np.array(seq)
    if stats is not None:
        stats.update(st)
    return seq[:-1], seq[1:], np.array(wts)


def check(X, hw, mcs, ms, tag):
    n = len(X)
    if n < 3:
        return
    k = H.effective_min_samples(n, mcs, ms)
    core = H.core_distances(X, k)
    ou, ov, ow = H.prim_mst(X, core)
    st = {}
    u, v, w = prim_levels(X, core, hw, stats=st, nbmax=NBMAX_TEST)
    assert np.array_equal(u, ou) and np.array_equal(v, ov) and np.array_equal(w, ow), tag
    nb = (n + 63) // 64
    print('%-18s N=%5d rounds %4d (%.1f nodes/round) rises %3d drops %3d slow %3d probes/node %.0f | blocks relaxed %d of %d (all pairs), flushes %d' % (
        tag, n, st['rounds'], (n - 1) / st['rounds'], st['rises'], st['drops'], st['slow'], st['probes'] / n, st.get('blocks', 0), nb * nb, st.get('flushes', 0)), flush=True)


def main():
    z = np.load(os.path.join(os.path.dirname(__file__), '..', '..', 'tests', 'golden', 'hdbscan_tieorder.npz'))
    for idx in range(0, 63, 3):
        occ = np.unpackbits(z['map_%d' % idx])[:35000].reshape(140, 250)
        check(np.argwhere(occ), (140, 250), 26, None, 'golden %d' % idx)
    rng = np.random.RandomState(0)
    for t in range(6):                                    # sparse noise: slow steps, many levels
        occ = rng.rand(140, 250) < (0.002, 0.01, 0.05, 0.2, 0.5, 0.9)[t]
        check(np.argwhere(occ), (140, 250), 26, None, 'noise %d' % t)
        check(np.argwhere(occ), (140, 250), 5, 3, 'noise %d best' % t)
    for t in range(4):                                    # blobs + noise at 35x62 (best settings size)
        yy, xx = np.mgrid[0:35, 0:62]
        occ = ((yy - rng.uniform(5, 30)) ** 2 + (xx - rng.uniform(5, 55)) ** 2 < rng.uniform(20, 120)) | (rng.rand(35, 62) < 0.02)
        check(np.argwhere(occ), (35, 62), 5, 3, 'small %d' % t)
    occ = np.zeros((140, 250), bool); occ[70, :] = True; occ[:, 125] = True       # lines
    check(np.argwhere(occ), (140, 250), 26, None, 'cross')
    occ = np.zeros((140, 250), bool); occ[::3, ::3] = True                        # lattice
    check(np.argwhere(occ), (140, 250), 26, None, 'lattice')


if __name__ == '__main__':
    main()
