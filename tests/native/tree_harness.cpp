// CPU harness for retargetvid_amd/csrc/hdb_tree.h (the hierarchy stage the HIP kernel
// inlines).  Built by tests/ only (g++), never loaded by the product.
#include <algorithm>
#include <vector>
static int g_batches = 0;                      // batches (= serial steps of the device's one-wavefront pass) of the last batched run
#define HDB_COUNT_BATCHES g_batches
#include "../../retargetvid_amd/csrc/hdb_tree.h"

extern "C" int tree_last_batches(void) { return g_batches; }

static int run(const uint16_t *a, const uint16_t *b, const uint32_t *w, int n, int mcs, int32_t *labels, int batched);

extern "C" int tree_labels(const uint16_t *a, const uint16_t *b, const uint32_t *w, int n, int mcs,
                           int32_t *labels) {
    return run(a, b, w, n, mcs, labels, 0);
}

// the batched form of the pass (what the GPU runs): batch of 64 edges resolved up front
extern "C" int tree_labels_batched(const uint16_t *a, const uint16_t *b, const uint32_t *w, int n, int mcs,
                                   int32_t *labels) {
    g_batches = 0;
    return run(a, b, w, n, mcs, labels, 1);
}

static int run(const uint16_t *a, const uint16_t *b, const uint32_t *w, int n, int mcs, int32_t *labels, int batched) {
    using namespace hdb;
    std::vector<Edge> edges(n - 1);
    for (int i = 0; i < n - 1; ++i) edges[i] = Edge{a[i], b[i], w[i]};
    const int mc = max_clusters(n, mcs);
    std::vector<uint16_t> sp(n), ssz(n), absc(n), cup(mc), evc(n), evs(n);
    std::vector<uint32_t> absw(n), sdn(n), dparent(2 * n), cbirthw(mc), cminw(mc), csize(mc), cdn(mc), csplit(mc), cspa(mc), cspb(mc);
    std::vector<int32_t> ctp(mc), cleft(mc), cright(mc), crep(mc);
    std::vector<double> cacc(mc);
    std::vector<uint8_t> csel(mc);
    Tree t{sp.data(), ssz.data(), absc.data(), absw.data(), sdn.data(), evc.data(), evs.data(),
           dparent.data(), cup.data(), ctp.data(), cleft.data(), cright.data(), cbirthw.data(), cminw.data(),
           csize.data(), cdn.data(), csplit.data(), cspa.data(), cspb.data(), cacc.data(), csel.data(), crep.data(),
           0, n, mc};
    init_points(t, 0, n);
    if (!(batched ? build_batched<64>(t, edges.data(), n, mcs) : build(t, edges.data(), n, mcs))) return -1;
    const int nsel = select(t, edges.data());
    std::vector<int> sel;
    for (int c = 0; c < t.nclusters; ++c) if (t.crep[c] == c) sel.push_back(c);
    std::sort(sel.begin(), sel.end(), [&](int x, int y) { return cluster_before(t, x, y); });
    std::vector<int> lab(t.nclusters, -1);
    for (size_t i = 0; i < sel.size(); ++i) lab[sel[i]] = (int)i;
    for (int p = 0; p < n; ++p) {
        int c = point_cluster(t, p, nsel);
        labels[p] = c < 0 ? -1 : lab[c];
    }
    return nsel;
}
