// hdb_tree.h — HDBSCAN* hierarchy stage of the cluster filter (K10, second half).
//
// Input: the N-1 edges of the mutual-reachability MST in the library's Prim order,
// sorted by weight the way the library sorts them: numpy's default argsort, an unstable
// introsort whose order of equal weights is restated in oracle/npsort_ref.py and emulated
// on the device by k_sort.  (Any order of equal weights is processed correctly; the order
// decides which components merge first, so it has to be the library's.)  Output: for every point the condensed cluster that
// absorbed it, the weight at which that happened, and the excess-of-mass selection —
// enough to label points exactly like hdbscan's generic path
// (call site smartVidCrop.py:1099; algorithm: SURVEY.md Appendix A steps 6-10).
//
// The reference library builds the full single-linkage dendrogram, condenses it
// top-down (BFS) and then sums stabilities.  This file does the same work in ONE
// bottom-up pass over the sorted edges: a component that first reaches
// min_cluster_size opens a condensed cluster; a smaller component merging into it is
// a batch of points "falling out" of that cluster at lambda = 1/w; two large
// components merging is a true split that opens their parent.  Every such event is
// logged per edge; stabilities are then summed from the log in REVERSE edge order,
// one (lambda - birth) * size term per condensed-tree row, which is exactly the order
// and the float64 arithmetic of the library's row loop (rows of a cluster appear
// top-down, i.e. by increasing lambda; a batch of s points is s rows of size 1), so
// even exact ties in its `subtree > own` comparisons resolve identically.
//
// Cluster numbering: the library numbers condensed clusters in BFS order of the
// dendrogram; the caller only needs that order to break ties between equally
// weighted clusters ("first arg-max wins", smartVidCrop.py:1114), so it is
// evaluated lazily by cluster_before() from the dendrogram parent links.
//
// Plain C++ (no HIP types) so the same source is unit-tested on the CPU
// (tests/native/tree_harness.cpp) and inlined into the device kernel.
#pragma once
#include <stdint.h>

#ifdef __HIPCC__
#include <hip/hip_runtime.h>
#define SVC_HD __host__ __device__ __forceinline__
#else
#define SVC_HD inline
#endif

namespace hdb {

static const uint32_t NONE16 = 0xFFFFu;
static const uint32_t NO_LANE = 0xFFFFFFFFu;     // absw[] of an un-absorbed root: no edge of the current batch contains it
static const int32_t ROOT_NOISE = -2;

struct Edge {          // 8 bytes
    uint16_t a, b;     // a = node already in the tree (library: current_node), b = new node
    uint32_t w;        // mutual reachability (squared distance), >= 1
};

// Per-point arrays (capacity n) and per-cluster arrays (capacity max_clusters(n, mcs)).
struct Tree {
    // per point
    uint16_t *sp;      // small-component union-find parent
    uint16_t *ssz;     // small-component size (valid at roots)
    uint16_t *absc;    // cluster that absorbed this small component (valid at roots), NONE16 = not yet
    uint32_t *absw;    // weight at absorption (valid at roots)
    uint32_t *sdn;     // dendrogram node id of a live small component (valid at roots)
    // per edge: the condensed-tree rows the merge produced
    uint16_t *evc;     // cluster the rows belong to (NONE16 = none)
    uint16_t *evs;     // points falling out (size-1 rows); 0 marks the split that created cluster evc
    // dendrogram: parent link of every node (2n-1 entries): parent << 1 | is_right
    uint32_t *dparent;
    // per cluster
    uint16_t *cup;     // union-find parent towards the current top cluster of the component
    int32_t *ctp;      // tree parent (-1 = root so far)
    int32_t *cleft, *cright;
    uint32_t *cbirthw; // weight at which the cluster is born top-down (0 = root: lambda 0)
    uint32_t *cminw;   // smallest weight among the cluster's rows (largest lambda)
    uint32_t *csize;   // points in the component while this cluster is its top
    uint32_t *cdn;     // dendrogram node of the component while this cluster is its top
    uint32_t *csplit;  // dendrogram node at which the cluster's children were created
    uint32_t *cspa, *cspb;  // sizes of the two child clusters of a split-created cluster (its two cluster rows)
    double *cacc;      // stability (after select(): propagated subtree stability)
    uint8_t *csel;     // selected by excess of mass
    int32_t *crep;     // nearest selected ancestor-or-self, ROOT_NOISE if none
    int32_t nclusters;
    int32_t n;
    int32_t cap_clusters;   // capacity of the per-cluster arrays
};

SVC_HD int max_clusters(int n, int mcs) {
    if (mcs < 1) mcs = 1;
    return 2 * (n / mcs) + 2;
}

SVC_HD uint32_t find_small(Tree &t, uint32_t x) {
    while (t.sp[x] != x) {
        uint16_t g = t.sp[t.sp[x]];
        t.sp[x] = g;              // path halving
        x = g;
    }
    return x;
}

SVC_HD uint32_t find_small_ro(const Tree &t, uint32_t x) {
    while (t.sp[x] != x) x = t.sp[x];
    return x;
}

SVC_HD uint32_t find_top(Tree &t, uint32_t c) {
    while (t.cup[c] != c) {
        uint16_t g = t.cup[t.cup[c]];
        t.cup[c] = g;
        c = g;
    }
    return c;
}

SVC_HD int32_t new_cluster(Tree &t, uint32_t w, uint32_t size, uint32_t node, int32_t l, int32_t r) {
    int32_t c = t.nclusters++;
    t.cup[c] = (uint16_t)c;
    t.ctp[c] = -1;
    t.cleft[c] = l;
    t.cright[c] = r;
    t.cbirthw[c] = 0;
    t.cminw[c] = w;
    t.csize[c] = size;
    t.cdn[c] = node;
    t.csplit[c] = node;
    t.cspa[c] = 0;
    t.cspb[c] = 0;
    t.cacc[c] = 0.0;
    t.csel[c] = 0;
    t.crep[c] = ROOT_NOISE;
    return c;
}

SVC_HD void init_points(Tree &t, int lo, int hi) {      // callable by many threads on disjoint ranges
    for (int i = lo; i < hi; ++i) {
        t.sp[i] = (uint16_t)i;
        t.ssz[i] = 1;
        t.absc[i] = (uint16_t)NONE16;
        t.absw[i] = NO_LANE;       // doubles as the per-batch "first edge containing this root" table (build_batched)
        t.sdn[i] = (uint32_t)i;
    }
}

// Roots / tops of both endpoints of one edge.  c* = top condensed cluster of the component, or
// NONE16 while the component is still smaller than min_cluster_size.
struct Resolved {
    uint32_t ra, rb;
    uint32_t ca, cb;
    uint32_t sa, sb;      // resolve_ro only: size / dendrogram node of a side that is still a small component
    uint32_t na, nb;
};

// read-only resolution (no path compression): safe to run for many edges in parallel
SVC_HD Resolved resolve_ro(const Tree &t, const Edge &e) {
    Resolved r;
    r.ra = find_small_ro(t, e.a);
    r.rb = find_small_ro(t, e.b);
    uint32_t c = t.absc[r.ra];
    if (c != NONE16) { while (t.cup[c] != c) c = t.cup[c]; }
    r.ca = c;
    c = t.absc[r.rb];
    if (c != NONE16) { while (t.cup[c] != c) c = t.cup[c]; }
    r.cb = c;
    r.sa = t.ssz[r.ra]; r.na = t.sdn[r.ra];
    r.sb = t.ssz[r.rb]; r.nb = t.sdn[r.rb];
    return r;
}

SVC_HD Resolved resolve(Tree &t, const Edge &e) {
    Resolved r;
    r.ra = find_small(t, e.a);
    r.rb = find_small(t, e.b);
    r.ca = t.absc[r.ra] != NONE16 ? find_top(t, t.absc[r.ra]) : NONE16;
    r.cb = t.absc[r.rb] != NONE16 ? find_top(t, t.absc[r.rb]) : NONE16;
    r.sa = r.sb = r.na = r.nb = 0;
    return r;
}

// Apply edge i given its (current) roots and tops.  Returns false if the cluster tables are full.
SVC_HD bool merge(Tree &t, int i, int n, int mcs, uint32_t w, const Resolved &q) {
    const uint32_t ra = q.ra, rb = q.rb, ca = q.ca, cb = q.cb;
    const bool abig = ca != NONE16, bbig = cb != NONE16;
    uint32_t sa, sb, na, nb;
    if (abig) { sa = t.csize[ca]; na = t.cdn[ca]; }
    else      { sa = t.ssz[ra]; na = t.sdn[ra]; }
    if (bbig) { sb = t.csize[cb]; nb = t.cdn[cb]; }
    else      { sb = t.ssz[rb]; nb = t.sdn[rb]; }
    const uint32_t node = (uint32_t)(n + i);
    t.dparent[na] = node << 1;                     // the a side is the left child of the new node
    t.dparent[nb] = (node << 1) | 1u;
    t.evc[i] = (uint16_t)NONE16;
    t.evs[i] = 0;
    if (!abig && !bbig) {
        const uint32_t s = sa + sb;
        if ((int)s < mcs) {
            uint32_t big = ra, small = rb;
            if (sb > sa) { big = rb; small = ra; }
            t.sp[small] = (uint16_t)big;
            t.ssz[big] = (uint16_t)s;
            t.sdn[big] = node;
        } else {                                   // a condensed cluster is born bottom-up
            if (t.nclusters >= t.cap_clusters) return false;
            int32_t c = new_cluster(t, w, s, node, -1, -1);
            t.evc[i] = (uint16_t)c; t.evs[i] = (uint16_t)s;
            t.absc[ra] = (uint16_t)c; t.absw[ra] = w;
            t.absc[rb] = (uint16_t)c; t.absw[rb] = w;
        }
    } else if (abig && bbig) {                     // true split: both sides >= mcs
        if (t.nclusters >= t.cap_clusters) return false;
        int32_t p = new_cluster(t, w, sa + sb, node, (int32_t)ca, (int32_t)cb);
        t.evc[i] = (uint16_t)p; t.evs[i] = 0;
        t.cspa[p] = sa; t.cspb[p] = sb;
        t.cup[ca] = (uint16_t)p; t.cup[cb] = (uint16_t)p;
        t.ctp[ca] = p; t.ctp[cb] = p;
        t.cbirthw[ca] = w; t.cbirthw[cb] = w;
    } else {                                       // small side falls out of the big side's cluster
        uint32_t c = abig ? ca : cb, r = abig ? rb : ra, s = abig ? sb : sa;
        t.absc[r] = (uint16_t)c; t.absw[r] = w;
        t.evc[i] = (uint16_t)c; t.evs[i] = (uint16_t)s;
        t.csize[c] += s;
        t.cdn[c] = node;
    }
    return true;
}

// The sequential pass.  edges must be sorted by w (ties in the library's order, see the header).
// Returns false if the per-cluster arrays (cap_clusters) are too small.
SVC_HD bool build(Tree &t, const Edge *edges, int n, int mcs) {
    t.n = n;
    t.nclusters = 0;
    for (int i = 0; i < n - 1; ++i) {
        const Edge e = edges[i];
        const Resolved q = resolve(t, e);
        if (!merge(t, i, n, mcs, e.w, q)) return false;
    }
    t.dparent[2 * n - 2] = 0xFFFFFFFFu;               // root of the dendrogram
    return true;
}

// The same pass the way the device runs it (svc_tail.hip: build_wave), kept here in plain C++ with
// the lanes as array slots so the CPU harness checks the batching logic against the oracle.
//
// The edges arrive in numpy's argsort order (oracle/npsort_ref.py: an unstable introsort, so equal weights --
// nearly all of them on a pixel grid -- come in a scrambled order, not in Prim order).  Nearly every edge of a
// real map then ATTACHES A FRESH SMALL COMPONENT (usually one pixel) to something: to an existing big cluster
// (it falls out of that cluster: an absorption), or to another small component while the sum stays below
// min_cluster_size (a union).  Such edges commute with their neighbours in the sorted list as long as the fresh
// components are different, so a batch of up to B edges is resolved up front (device: one lane per edge,
// read-only) and the longest prefix made of them is applied at once.  Per side of an edge, at the start of the batch:
//   B  big: belongs to a condensed cluster
//   F  fresh: a small root that no EARLIER edge of the batch contains
//   L  linked: a small root that an earlier edge k of the batch contains -- at this edge's turn the root is part
//      of whatever edge k made (k is the first edge containing it, so it was fresh there)
// and per edge:
//   F+F  starts a small tree (a union; a cluster birth, which ends the prefix, if the sizes reach min_cluster_size)
//   F+B  starts / continues the absorption chain of cluster B
//   F+L  attaches its fresh side to the tree of edge k: the chain of a cluster (an absorption) or a small tree
//        (a union, as long as the tree's running size stays below min_cluster_size; the edge at which it
//        reaches it is a cluster birth and ends the prefix)
//   anything else (two existing things meet: a true split, a batch-made tree reaching a cluster, ...) ends the prefix.
// The trees are found by pointer jumping along the k links; what the edges of one tree share -- the running
// dendrogram node, the size, the union-find root -- follows from the lane numbers of its edges.  The edge that ends
// the prefix is resolved afresh and applied alone by merge(); the next batch starts behind it.
template <int B>
inline bool build_batched(Tree &t, const Edge *edges, int n, int mcs) {
    t.n = n;
    t.nclusters = 0;
    Resolved pre[B];
    int i0 = 0;
    while (i0 < n - 1) {
#ifdef HDB_COUNT_BATCHES
        ++HDB_COUNT_BATCHES;
#endif
        const int m = (n - 1 - i0) < B ? (n - 1 - i0) : B;
        for (int j = 0; j < m; ++j) pre[j] = resolve_ro(t, edges[i0 + j]);
        // first edge of the batch containing each small root: one scatter-min of lane numbers per root (the table
        // lives in absw[], which is meaningless for un-absorbed roots and is restored to "none" below)
        for (int j = 0; j < m; ++j) {
            if (pre[j].ca == NONE16 && (uint32_t)j < t.absw[pre[j].ra]) t.absw[pre[j].ra] = (uint32_t)j;
            if (pre[j].cb == NONE16 && (uint32_t)j < t.absw[pre[j].rb]) t.absw[pre[j].rb] = (uint32_t)j;
        }
        uint32_t fa[B], fb[B];
        for (int j = 0; j < m; ++j) {
            fa[j] = pre[j].ca == NONE16 ? t.absw[pre[j].ra] : (uint32_t)j;
            fb[j] = pre[j].cb == NONE16 ? t.absw[pre[j].rb] : (uint32_t)j;
        }
        for (int j = 0; j < m; ++j) {                  // restore the table
            if (pre[j].ca == NONE16) t.absw[pre[j].ra] = NO_LANE;
            if (pre[j].cb == NONE16) t.absw[pre[j].rb] = NO_LANE;
        }
        // classify, link, find the trees
        int par[B], root[B];                           // lane of the edge this one attaches to (itself: starts a tree)
        uint32_t clus[B], add[B];                      // cluster of a chain root (NONE16: a small tree); size this edge brings in
        bool bad[B];                                   // cannot be part of a prefix
        for (int j = 0; j < m; ++j) {
            const bool aB = pre[j].ca != NONE16, bB = pre[j].cb != NONE16;
            const bool aF = !aB && fa[j] == (uint32_t)j, bF = !bB && fb[j] == (uint32_t)j;
            par[j] = j; clus[j] = NONE16; add[j] = 0; bad[j] = false;
            if (aF && bF) add[j] = pre[j].sa + pre[j].sb;
            else if (aF && bB) { clus[j] = pre[j].cb; add[j] = pre[j].sa; }
            else if (aB && bF) { clus[j] = pre[j].ca; add[j] = pre[j].sb; }
            else if (aF && !bB) { par[j] = (int)fb[j]; add[j] = pre[j].sa; }      // F + L
            else if (bF && !aB) { par[j] = (int)fa[j]; add[j] = pre[j].sb; }      // L + F
            else bad[j] = true;
        }
        uint32_t tot[B];
        for (int j = 0; j < m; ++j) {                  // links point to earlier lanes: one ascending pass (device: pointer jumping)
            root[j] = par[j] == j ? j : root[par[j]];
            if (bad[par[j]] || (par[j] != j && bad[root[j]])) bad[j] = true;   // (only matters behind the end of the prefix)
            tot[j] = 0;
        }
        for (int j = 0; j < m; ++j) if (!bad[j]) tot[root[j]] += add[j];
        int P = m;
        for (int j = 0; j < m; ++j) {
            // a small tree gives birth to a cluster at the edge where its running size reaches min_cluster_size
            // (device: a wave-wide prefix sum over the lanes of every tree whose batch total reaches it -- rarely any)
            bool birth = false;
            if (!bad[j] && clus[root[j]] == NONE16 && (int)tot[root[j]] >= mcs) {
                uint32_t run = 0;
                for (int k = 0; k <= j; ++k) if (!bad[k] && root[k] == root[j]) run += add[k];
                birth = (int)run >= mcs;
            }
            if (bad[j] || birth) {
#ifdef HDB_COUNT_STOP
                HDB_COUNT_STOP(bad[j] ? 0 : 1);
#endif
                P = j;
                break;
            }
        }
        // apply lanes 0..P-1 (independent per lane on the device, given the per-tree lane sets)
        for (int j = 0; j < P; ++j) {
            const Resolved &q = pre[j];
            const uint32_t node = (uint32_t)(n + i0 + j);
            const int r0 = root[j];
            const uint32_t c = clus[r0];
            int prev = -1;                             // previous edge of the same small tree / of the same cluster's chain
            for (int k = 0; k < j; ++k)                // (several chains of one batch may feed the same cluster)
                if (c != NONE16 ? clus[root[k]] == c : root[k] == r0) prev = k;
            const bool aB = q.ca != NONE16, bB = q.cb != NONE16;
            const bool aF = !aB && fa[j] == (uint32_t)j;   // (the other side: bF = !bB && fb[j] == j, implied where it matters)
            (void)bB;
            if (c != NONE16) {                         // absorption into cluster c: the fresh side falls out of it
                const bool abig = !aF;                 // the a side is the cluster's side
                const uint32_t r = abig ? q.rb : q.ra, s = abig ? q.sb : q.sa, ns = abig ? q.nb : q.na;
                const uint32_t pn = prev >= 0 ? (uint32_t)(n + i0 + prev) : t.cdn[c];
                t.dparent[abig ? pn : ns] = node << 1;
                t.dparent[abig ? ns : pn] = (node << 1) | 1u;
                t.absc[r] = (uint16_t)c; t.absw[r] = edges[i0 + j].w;
                t.evc[i0 + j] = (uint16_t)c; t.evs[i0 + j] = (uint16_t)s;
            } else {                                   // union inside a small tree; the tree's root is the starter's a side
                const uint32_t R = pre[r0].ra;
                t.evc[i0 + j] = (uint16_t)NONE16; t.evs[i0 + j] = 0;
                if (r0 == j) {
                    t.dparent[q.na] = node << 1;
                    t.dparent[q.nb] = (node << 1) | 1u;
                    t.sp[q.rb] = (uint16_t)R;
                } else {
                    const uint32_t pn = (uint32_t)(n + i0 + prev);
                    const uint32_t rf = aF ? q.ra : q.rb, nf = aF ? q.na : q.nb;
                    t.dparent[aF ? nf : pn] = node << 1;           // a side = left child
                    t.dparent[aF ? pn : nf] = (node << 1) | 1u;
                    t.sp[rf] = (uint16_t)R;
                }
            }
        }
        for (int j = 0; j < P; ++j) {                  // per tree, after all reads of cdn above: totals and last nodes
            const int r0 = root[j];
            const uint32_t c = clus[r0];
            if (c != NONE16) {
                t.csize[c] += add[j];
                t.cdn[c] = (uint32_t)(n + i0 + j);
            } else {
                const uint32_t R = pre[r0].ra;
                if (r0 == j) t.ssz[R] = 0;
            }
        }
        for (int j = 0; j < P; ++j) {
            const int r0 = root[j];
            if (clus[r0] != NONE16) continue;
            const uint32_t R = pre[r0].ra;
            t.ssz[R] = (uint16_t)(t.ssz[R] + add[j]);
            t.sdn[R] = (uint32_t)(n + i0 + j);
        }
        i0 += P;
        if (P < m) {                                   // the roots may have moved (unions): resolve afresh
            if (!merge(t, i0, n, mcs, edges[i0].w, resolve(t, edges[i0]))) return false;
            ++i0;
        }
    }
    t.dparent[2 * n - 2] = 0xFFFFFFFFu;
    return true;
}

// Stabilities: the library's row loop, rows in its condensed-tree order (see header).
// (svc_tail.hip: accumulate_wave is the same loop with the per-row terms computed 64 rows at a time.)
SVC_HD void accumulate(Tree &t, const Edge *edges) {
    uint32_t w_prev = 0, c_prev = NONE16;
    double lam = 0.0, birth = 0.0;
    for (int i = t.n - 2; i >= 0; --i) {
        const uint16_t c = t.evc[i];
        if (c == NONE16) continue;
        const uint32_t w = edges[i].w;
        if (w != w_prev) { lam = 1.0 / (double)w; w_prev = w; }      // weights and clusters repeat in runs:
        if (c != c_prev) {                                            // divide once per run, not per row
            birth = t.cbirthw[c] ? 1.0 / (double)t.cbirthw[c] : 0.0;
            c_prev = c;
        }
        double acc = t.cacc[c];
        if (t.evs[i] == 0) {                            // the two cluster rows of a split, left first
            acc += (lam - birth) * (double)t.cspa[c];
            acc += (lam - birth) * (double)t.cspb[c];
        } else {
            const double term = (lam - birth) * 1.0;
            for (uint32_t k = 0; k < t.evs[i]; ++k) acc += term;
        }
        t.cacc[c] = acc;
    }
}

// Excess-of-mass selection (root allowed) and nearest-selected-ancestor map, from the stabilities.
// Returns the number of selected clusters.
SVC_HD int choose(Tree &t) {
    const int nc = t.nclusters;
    for (int c = 0; c < nc; ++c) {                      // creation order: children before parents
        double stab = t.cacc[c];
        double sub = 0.0;
        if (t.cleft[c] >= 0) sub = t.cacc[t.cleft[c]] + t.cacc[t.cright[c]];
        if (sub > stab) {
            t.csel[c] = 0;
            stab = sub;
        } else {
            t.csel[c] = 1;
        }
        t.cacc[c] = stab;
    }
    int nsel = 0;
    for (int c = nc - 1; c >= 0; --c) {                 // parents before children
        const int32_t p = t.ctp[c];
        if (p >= 0 && t.crep[p] >= 0) {                 // an ancestor is selected: it wins
            t.csel[c] = 0;
            t.crep[c] = t.crep[p];
        } else if (t.csel[c]) {
            t.crep[c] = c;
            ++nsel;
        } else {
            t.crep[c] = ROOT_NOISE;
        }
    }
    return nsel;
}

SVC_HD int select(Tree &t, const Edge *edges) {
    accumulate(t, edges);
    return choose(t);
}

// Label of point p as a cluster index (creation order), -1 = noise.
SVC_HD int32_t point_cluster(const Tree &t, uint32_t p, int nsel) {
    const uint32_t r = find_small_ro(t, p);
    const uint16_t c0 = t.absc[r];
    if (c0 == NONE16) return -1;
    const int32_t rep = t.crep[c0];
    if (rep == ROOT_NOISE) return -1;
    const int root = t.nclusters - 1;
    if (rep == root && nsel == 1) {
        // a lone selected root keeps only the points whose lambda >= the root's max lambda
        return t.absw[r] <= t.cminw[root] ? rep : -1;
    }
    return rep;
}

SVC_HD int dendro_depth(const Tree &t, uint32_t node) {
    int d = 0;
    while (t.dparent[node] != 0xFFFFFFFFu) { node = t.dparent[node] >> 1; ++d; }
    return d;
}

// True if condensed cluster c1 gets a smaller id than c2 in the library's numbering
// (BFS order of the dendrogram; children of one split: left first).
SVC_HD bool cluster_before(const Tree &t, int c1, int c2) {
    if (c1 == c2) return false;
    const int p1 = t.ctp[c1], p2 = t.ctp[c2];
    if (p1 < 0) return true;                            // root has the smallest id
    if (p2 < 0) return false;
    if (p1 == p2) return t.cleft[p1] == c1;
    uint32_t s1 = t.csplit[p1], s2 = t.csplit[p2];
    int d1 = dendro_depth(t, s1), d2 = dendro_depth(t, s2);
    if (d1 != d2) return d1 < d2;
    uint32_t side1 = 0, side2 = 0;
    while (s1 != s2) {
        side1 = t.dparent[s1] & 1u; side2 = t.dparent[s2] & 1u;
        s1 = t.dparent[s1] >> 1; s2 = t.dparent[s2] >> 1;
    }
    return side1 < side2;
}

}  // namespace hdb
