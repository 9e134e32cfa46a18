"""Oracle: numpy's default argsort for float64 keys -- the scalar introsort `aquicksort_double` -- restated.

TEST INFRASTRUCTURE (see oracle/__init__.py).

Why it is on the path: hdbscan 0.8.26 (third party, not vendored, not installed; call site smartVidCrop.py:1099,
ctor :2340-2348) orders the MST edges with `np.argsort(min_spanning_tree.T[2])` before building the single-linkage
tree (hdbscan_.py `_hdbscan_generic`; scikit-learn's port does the same in `_process_mst`).  The default kind is an
UNSTABLE introsort, and on a pixel grid nearly all mutual-reachability weights tie, so the order of equal keys --
a property of numpy's sort, not of HDBSCAN -- decides which components merge first, hence the condensed tree and,
on about a third of real maps, the kept cluster (tests/golden/hdbscan_tieorder.npz).  The reference pins
scipy 1.5.1 / scikit-learn 0.24.1 / torch 1.7.1 (README.md:83-92), i.e. numpy 1.19.x, which predates the SIMD
argsort of numpy >= 1.25 (x86-simd-sort on AVX-512 / AVX2 machines): its argsort is this scalar routine on every CPU.

Algorithm (numpy/core/src/npysort/quicksort.c.src, heapsort.c.src; restated from the published source and pinned
bit for bit against numpy 1.26.4 and 2.2.6 run with their SIMD dispatch disabled, NPY_DISABLE_CPU_FEATURES, on 600
arrays: tools/make_golden_hdbscan.py; permutations committed in tests/golden/npsort_golden.npz):
  * explicit stack, depth limit 2*floor(log2(n)); a range POPPED from the stack below the limit is heap-sorted
  * ranges of more than 16 elements: median of three (first, middle, last, three conditional swaps), pivot parked at
    pr-1, Hoare scan that stops on EQUAL keys on both sides, pivot swapped to its place, the LARGER side pushed
    (the right one when the left is strictly smaller, else the left), the other continued without a depth check
  * ranges of at most 16 elements: insertion sort (stable within the range)
Only `<` on the keys is used, so float64 weights that hold exact integers behave like the integers."""

SMALL_QUICKSORT = 15          # the partition loop runs while pr - pl > 15, i.e. for ranges of 17 or more elements ... as measured:
                              # numpy's scalar argsort partitions a 17-element range and insertion-sorts a 16-element one


def _msb(n):
    d = 0
    n >>= 1
    while n:
        d += 1
        n >>= 1
    return d


def aheapsort(v, a, lo, n):
    """numpy's aheapsort on a[lo:lo+n] (indices into v), 1-based sift-down exactly as written there."""
    def A(i):
        return a[lo + i - 1]

    def S(i, x):
        a[lo + i - 1] = x

    l = n >> 1
    while l > 0:
        tmp = A(l)
        i, j = l, l << 1
        while j <= n:
            if j < n and v[A(j)] < v[A(j + 1)]:
                j += 1
            if v[tmp] < v[A(j)]:
                S(i, A(j))
                i = j
                j += j
            else:
                break
        S(i, tmp)
        l -= 1
    while n > 1:
        tmp = A(n)
        S(n, A(1))
        n -= 1
        i, j = 1, 2
        while j <= n:
            if j < n and v[A(j)] < v[A(j + 1)]:
                j += 1
            if v[tmp] < v[A(j)]:
                S(i, A(j))
                i = j
                j += j
            else:
                break
        S(i, tmp)


def argsort(keys, stats=None):
    """np.argsort(keys) (default kind) for a 1-D sequence of numbers, scalar code path.  -> list of indices."""
    v = [x for x in keys]
    num = len(v)
    a = list(range(num))
    if num < 2:
        return a
    pl, pr = 0, num - 1
    stack = []
    cdepth = _msb(num) * 2
    while True:
        if cdepth < 0:
            if stats is not None:
                stats['heapsorts'] = stats.get('heapsorts', 0) + 1
            aheapsort(v, a, pl, pr - pl + 1)
        else:
            while pr - pl > SMALL_QUICKSORT:
                pm = pl + ((pr - pl) >> 1)
                if v[a[pm]] < v[a[pl]]:
                    a[pm], a[pl] = a[pl], a[pm]
                if v[a[pr]] < v[a[pm]]:
                    a[pr], a[pm] = a[pm], a[pr]
                if v[a[pm]] < v[a[pl]]:
                    a[pm], a[pl] = a[pl], a[pm]
                vp = v[a[pm]]
                pi, pj = pl, pr - 1
                a[pm], a[pj] = a[pj], a[pm]
                while True:
                    pi += 1
                    while v[a[pi]] < vp:
                        pi += 1
                    pj -= 1
                    while vp < v[a[pj]]:
                        pj -= 1
                    if pi >= pj:
                        break
                    a[pi], a[pj] = a[pj], a[pi]
                pk = pr - 1
                a[pi], a[pk] = a[pk], a[pi]
                cdepth -= 1
                if pi - pl < pr - pi:
                    stack.append((pi + 1, pr, cdepth))
                    pr = pi - 1
                else:
                    stack.append((pl, pi - 1, cdepth))
                    pl = pi + 1
            for pi in range(pl + 1, pr + 1):
                vi = a[pi]
                vp = v[vi]
                pj, pk = pi, pi - 1
                while pj > pl and vp < v[a[pk]]:
                    a[pj] = a[pk]
                    pj -= 1
                    pk -= 1
                a[pj] = vi
        if not stack:
            break
        pl, pr, cdepth = stack.pop()
    return a
