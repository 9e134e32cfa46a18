"""The executable specifications of the two round-3 tail kernels (tools/sim/) against the oracle, on CPU:

  tools/sim/prim_levels.py   k_prim_lvl: the library's Prim emitted in rounds of up to 64 nodes, block-pruned rises
  tools/sim/tree_path.py     k_tree_par: the hierarchy from nearest-greater ranks on the Prim path

Both must reproduce oracle/hdbscan_ref (prim_mst; single_linkage -> condense_tree -> select_and_label) exactly.  The device
kernels are checked against the same oracle by tests/test_gpu_parity.py; these tests keep the specifications honest."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools', 'sim'))
import prim_levels as PL        # noqa: E402
import tree_path as TP          # noqa: E402
from oracle import hdbscan_ref as H      # noqa: E402


def _cases():
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'hdbscan_tieorder.npz'))
    out = [(np.argwhere(np.unpackbits(z['map_%d' % i])[:35000].reshape(140, 250)), (140, 250), 26, None) for i in (55, 60)]
    rng = np.random.RandomState(5)
    yy, xx = np.mgrid[0:35, 0:62]
    for t in range(3):
        occ = (rng.rand(35, 62) < 0.03) | ((yy - rng.uniform(5, 30)) ** 2 + (xx - rng.uniform(5, 55)) ** 2 < rng.uniform(20, 90))
        out.append((np.argwhere(occ), (35, 62), 5, 3))
    out.append((np.argwhere(rng.rand(60, 80) < 0.02), (60, 80), 26, None))          # sparse: every step a jump
    line = np.zeros((40, 90), bool); line[20, :] = True; line[:, 45] = True
    out.append((np.argwhere(line), (40, 90), 5, 3))
    return out


def test_prim_in_rounds_equals_the_oracles_prim():
    for X, hw, mcs, ms in _cases():
        k = H.effective_min_samples(len(X), mcs, ms)
        core = H.core_distances(X, k)
        ou, ov, ow = H.prim_mst(X, core)
        for nbmax in (256, 4):                                         # 4: the batch table runs full (flush path)
            st = {}
            u, v, w = PL.prim_levels(X, core, hw, stats=st, nbmax=nbmax)
            assert np.array_equal(u, ou) and np.array_equal(v, ov) and np.array_equal(w, ow)
            assert st['rounds'] <= len(X) - 1


def test_path_hierarchy_equals_the_oracles_labels():
    for X, hw, mcs, ms in _cases():
        n = len(X)
        core = H.core_distances(X, H.effective_min_samples(n, mcs, ms))
        u, v, w = H.prim_mst(X, core)
        order = H.edge_order(w)
        left, right, weight, csize = H.single_linkage(u, v, w, order)
        ref = H.select_and_label(H.condense_tree(left, right, weight, csize, mcs), n)
        assert TP.same_partition(ref, TP.labels_path(u, v, w, order, mcs))


def test_no_half_swapping_packed_f32_instruction_in_the_library():
    """Build check (hipcc -S on this box, ~35 s): round 5 found a product LOST in a `v_pk_mul_f32 x2 ; v_pk_add_f32 op_sel:[0,1]
    op_sel_hi:[1,0]` sequence of the smoothing kernel when bf16-MFMA workgroups shared its CU (profiles/r05_mx_reproducibility.txt).
    The trigger is not fully understood, so the form is kept out of the library: no kernel may contain a packed f32 instruction that
    routes the halves of an operand crosswise, and the two smoothing kernels none at all (tools/packed_f32_census.py; kernels whose C
    code the compiler packs that way carry SVC_NO_PK).  The detector itself is checked on a two-line kernel that compiles to the form."""
    import subprocess
    import tempfile
    tool = os.path.join(ROOT, 'tools', 'packed_f32_census.py')
    r = subprocess.run([sys.executable, tool], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert 'svc_net.hip' in r.stdout and 'svc_shot.hip' in r.stdout and 'FAIL' not in r.stdout
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, 'swap.hip')
        with open(src, 'w') as f:
            f.write('#include <hip/hip_runtime.h>\n__global__ void k(const float2* a, const float2* b, float2* c) { int i = threadIdx.x; '
                    'float2 x = a[i], y = b[i], z; z.x = x.x * y.x + x.y; z.y = x.y * y.y + x.x; c[i] = z; }\n')
        r = subprocess.run([sys.executable, tool, src], capture_output=True, text=True)
        assert r.returncode == 1 and 'half-swapping' in r.stdout, r.stdout + r.stderr
