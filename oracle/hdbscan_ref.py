"""Oracle: HDBSCAN* as executed by the reference's cluster filter (K10).

TEST INFRASTRUCTURE (see oracle/__init__.py).

Reference call sites: smartVidCrop.py:2340-2348 (ctor: metric='sqeuclidean',
min_cluster_size=CP['hdbscan_min'], min_samples=CP['hdbscan_min_samples'],
cluster_selection_method='eom', allow_single_cluster=True) and
smartVidCrop.py:1099 (fit_predict on X[N,2] = (row, col) of the non-zero pixels in
raster order).

The arithmetic lives in the third-party package ``hdbscan==0.8.26``
(README.md:87), which is NOT vendored under /root/reference and not installed
here.  Its published algorithm for a metric that neither KDTree nor BallTree
supports ('sqeuclidean') is the dense "generic" path, restated below:

  1. D[i,j] = squared Euclidean distance (exact integers on the pixel grid)
  2. k = min(N-1, min_samples or min_cluster_size) (1 if that is 0);
     core[i] = k-th smallest of column i of D, the point itself counted at index 0
  3. mutual reachability M[i,j] = max(core[i], core[j], D[i,j])
  4. Prim from node 0 over the not-yet-added points kept in ascending index order;
     first minimum wins; the edge is recorded as (last added node, new node, weight)
  5. edges sorted by weight with np.argsort's default kind: an UNSTABLE introsort whose order of equal
     keys (nearly all of them on a pixel grid) is restated in oracle/npsort_ref.py (`order='numpy'`:
     the scalar routine every CPU runs under the reference's pinned numpy 1.19) and followed by the
     device path (k_sort, csrc/svc_tail.hip).  `order='stable'` (Prim order kept among equal weights,
     this build's round-1 convention) is kept only to measure what the tie order does downstream:
     tests/golden/hdbscan_tieorder.npz -- other kept clusters on a third of real maps, crop windows up
     to 17 px apart, so it is not an admissible substitute
  6. single linkage by union-find -> condensed tree (min_cluster_size) -> stability
     -> excess-of-mass selection with the root allowed -> labels
     (a lone selected root labels only the points whose lambda >= the root's max lambda)

Parity status: UNPINNED against hdbscan 0.8.26 itself (no reference test or golden
vector holds HDBSCAN outputs, and the package is unavailable offline).  Pinned instead
against scikit-learn 1.7.2's port of the same code (sklearn.cluster.HDBSCAN with
min_samples+1, metric='sqeuclidean') run with numpy's scalar argsort (the reference-era
sort, tools/make_golden_hdbscan.py): labels on 12 seeded point sets
(tests/golden/hdbscan_sklearn.npz) and on 126 thresholded saliency maps of the benchmark
workload (tests/golden/hdbscan_tieorder.npz) are reproduced bit for bit on EVERY case.
"""
import numpy as np

from . import npsort_ref

INF = np.iinfo(np.int64).max
DEFAULT_ORDER = 'numpy'         # the library's order (see step 5 above); the device path follows it


def edge_order(w, order=None):
    """Permutation that sorts the MST edge weights: 'stable' or 'numpy' (restated default argsort), or an
    explicit permutation (array)."""
    order = DEFAULT_ORDER if order is None else order
    if isinstance(order, str):
        if order == 'stable':
            return np.argsort(w, kind='stable')
        if order == 'numpy':
            return np.asarray(npsort_ref.argsort(np.asarray(w, np.int64).tolist()), np.int64)
        raise ValueError(order)
    return np.asarray(order, np.int64)


def effective_min_samples(n, min_cluster_size, min_samples):
    k = min_cluster_size if min_samples is None else min_samples
    k = min(n - 1, k)
    return 1 if k == 0 else k


def core_distances(X, k, chunk=1024):
    """k-th smallest squared distance per point, self included at index 0."""
    X = np.asarray(X, np.int64)
    n = X.shape[0]
    core = np.empty(n, np.int64)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        d = ((X[s:e, None, 0] - X[None, :, 0]) ** 2 + (X[s:e, None, 1] - X[None, :, 1]) ** 2)
        core[s:e] = np.partition(d, k, axis=1)[:, k]
    return core


def prim_mst(X, core):
    """Library-flavoured Prim.  -> (u[N-1], v[N-1], w[N-1]) int64."""
    X = np.asarray(X, np.int64)
    n = X.shape[0]
    u = np.empty(n - 1, np.int64)
    v = np.empty(n - 1, np.int64)
    w = np.empty(n - 1, np.int64)
    labels = np.arange(n, dtype=np.int64)
    reach = np.full(n, INF, np.int64)
    cur = 0
    for i in range(n - 1):
        keep = labels != cur
        labels = labels[keep]
        left = reach[keep]
        d = (X[labels, 0] - X[cur, 0]) ** 2 + (X[labels, 1] - X[cur, 1]) ** 2
        right = np.maximum(np.maximum(d, core[labels]), core[cur])
        reach = np.where(left < right, left, right)
        j = int(np.argmin(reach))
        u[i], v[i], w[i] = cur, labels[j], reach[j]
        cur = int(labels[j])
    return u, v, w


def single_linkage(u, v, w, order=None):
    """Union-find over 2N-1 ids.  -> left[N-1], right[N-1], weight[N-1], size[N-1]."""
    n = len(u) + 1
    order = edge_order(w, order)
    u, v, w = u[order], v[order], w[order]
    parent = np.full(2 * n - 1, -1, np.int64)
    size = np.ones(2 * n - 1, np.int64)
    left = np.empty(n - 1, np.int64)
    right = np.empty(n - 1, np.int64)
    csize = np.empty(n - 1, np.int64)
    nxt = n

    def find(x):
        r = x
        while parent[r] != -1:
            r = parent[r]
        while parent[x] != -1 and parent[x] != r:
            parent[x], x = r, parent[x]
        return r

    for i in range(n - 1):
        a, b = find(int(u[i])), find(int(v[i]))
        left[i], right[i] = a, b
        csize[i] = size[a] + size[b]
        size[nxt] = csize[i]
        parent[a] = parent[b] = nxt
        nxt += 1
    return left, right, w, csize


def _bfs(left, right, n, root):
    out, queue = [], [root]
    while queue:
        out.extend(queue)
        nxt = []
        for x in queue:
            if x >= n:
                nxt.append(int(left[x - n]))
                nxt.append(int(right[x - n]))
        queue = nxt
    return out


def condense_tree(left, right, weight, csize, mcs):
    """-> list of rows (parent, child, lambda, child_size), in the library's row order."""
    n = len(left) + 1
    root = 2 * (n - 1)
    relabel = {root: n}
    next_label = n + 1
    rows = []
    ignore = np.zeros(2 * n - 1, bool)
    for node in _bfs(left, right, n, root):
        if node < n or ignore[node]:
            continue
        l, r = int(left[node - n]), int(right[node - n])
        d = float(weight[node - n])
        lam = 1.0 / d if d > 0.0 else np.inf
        lc = int(csize[l - n]) if l >= n else 1
        rc = int(csize[r - n]) if r >= n else 1
        if lc >= mcs and rc >= mcs:
            relabel[l] = next_label
            next_label += 1
            rows.append((relabel[node], relabel[l], lam, lc))
            relabel[r] = next_label
            next_label += 1
            rows.append((relabel[node], relabel[r], lam, rc))
        elif lc < mcs and rc < mcs:
            for side in (l, r):
                for sub in _bfs(left, right, n, side):
                    if sub < n:
                        rows.append((relabel[node], sub, lam, 1))
                    ignore[sub] = True
        elif lc < mcs:
            relabel[r] = relabel[node]
            for sub in _bfs(left, right, n, l):
                if sub < n:
                    rows.append((relabel[node], sub, lam, 1))
                ignore[sub] = True
        else:
            relabel[l] = relabel[node]
            for sub in _bfs(left, right, n, r):
                if sub < n:
                    rows.append((relabel[node], sub, lam, 1))
                ignore[sub] = True
    return rows


def select_and_label(rows, n):
    """Stability, EOM (root allowed), labelling.  -> labels[n] int64, -1 = noise."""
    root = n
    births = {root: 0.0}
    for p, c, lam, s in rows:
        births[c] = lam
    births[root] = 0.0
    clusters = sorted({p for p, _, _, _ in rows})
    stab = {c: 0.0 for c in clusters}
    for p, c, lam, s in rows:
        stab[p] += (lam - births[p]) * s
    children = {c: [] for c in clusters}
    for p, c, lam, s in rows:
        if s > 1:
            children[p].append(c)
    is_cluster = {c: True for c in clusters}
    for node in sorted(clusters, reverse=True):
        sub = float(np.sum([stab[c] for c in children[node]]))
        if sub > stab[node]:
            is_cluster[node] = False
            stab[node] = sub
        else:
            stack = list(children[node])
            while stack:
                c = stack.pop()
                is_cluster[c] = False
                stack.extend(children[c])
    selected = sorted(c for c in clusters if is_cluster[c])
    label_of = {c: i for i, c in enumerate(selected)}
    # union every non-selected child into its parent, then read off each point's cluster
    top = {}

    def resolve(c):
        path = []
        while c in top and top[c] != c:
            path.append(c)
            c = top[c]
        for p in path:
            top[p] = c
        return c

    point_parent = np.full(n, -1, np.int64)
    point_lambda = np.zeros(n, np.float64)
    parent_of = {}
    for p, c, lam, s in rows:
        if s == 1:
            point_parent[c] = p
            point_lambda[c] = lam
        else:
            parent_of[c] = p
    for c in clusters:
        top[c] = c
    for c in sorted(clusters):
        if c != root and c not in label_of:
            top[c] = resolve(parent_of[c])
    labels = np.full(n, -1, np.int64)
    root_max_lambda = max((lam for p, c, lam, s in rows if p == root), default=0.0)
    single_root = (len(selected) == 1 and selected[0] == root)
    for i in range(n):
        c = resolve(int(point_parent[i]))
        if c != root:
            labels[i] = label_of[c]
        elif single_root and point_lambda[i] >= root_max_lambda:
            labels[i] = label_of[root]
    return labels


def hdbscan_labels(X, min_cluster_size, min_samples=None, return_tree=False, order=None):
    """fit_predict of the reference's clusterer on integer points X[N,2]."""
    X = np.asarray(X, np.int64)
    n = X.shape[0]
    k = effective_min_samples(n, min_cluster_size, min_samples)
    core = core_distances(X, k)
    u, v, w = prim_mst(X, core)
    left, right, weight, csize = single_linkage(u, v, w, order)
    rows = condense_tree(left, right, weight, csize, min_cluster_size)
    labels = select_and_label(rows, n)
    if return_tree:
        return labels, dict(core=core, mst=(u, v, w), rows=rows)
    return labels
