"""160 point sets through the hierarchy harness (serial and batched forms) against the oracle.
  python tools/tree_cases.py [path/to/libtree_harness.so]
With an AddressSanitizer/UBSan build of the harness (tools/asan_tree.sh) this is the sanitizer run of the code the
device shares with the host (retargetvid_amd/csrc/hdb_tree.h)."""
import numpy as np, ctypes, sys
sys.path.insert(0,'/root/repo')
from oracle import hdbscan_ref as H
lib = ctypes.CDLL(sys.argv[1] if len(sys.argv) > 1 else '/root/repo/oracle/_build/libtree_harness.so'); vp=ctypes.c_void_p
bad=0; tot=0
for seed in range(80):
    r = np.random.RandomState(seed)
    hw=(r.randint(30,141), r.randint(40,251))
    m = r.rand(*hw) < r.choice([0.02,0.08,0.25,0.6])
    for _ in range(r.randint(0,5)):
        cy,cx,ry,rx = r.randint(5,hw[0]-5), r.randint(5,hw[1]-5), r.randint(3,14), r.randint(3,20)
        ys,xs=np.mgrid[0:hw[0],0:hw[1]]; m |= (((ys-cy)/ry)**2+((xs-cx)/rx)**2)<1
    X=np.argwhere(m)
    if len(X) > 6000: X = X[:6000]
    mcs, ms = [(26,None),(5,3),(10,None),(15,4),(3,2),(40,10)][seed%6]
    n=len(X)
    if n<=mcs+1: continue
    lab,tr=H.hdbscan_labels(X,mcs,ms,return_tree=True); u,v,w=tr['mst']; o=np.argsort(w,kind='stable')
    a,b,ww=u[o].astype(np.uint16),v[o].astype(np.uint16),w[o].astype(np.uint32)
    for fn in (lib.tree_labels, lib.tree_labels_batched):
        out=np.zeros(n,np.int32); fn(a.ctypes.data_as(vp),b.ctypes.data_as(vp),ww.ctypes.data_as(vp),n,mcs,out.ctypes.data_as(vp))
        tot+=1
        if not np.array_equal(out,lab): bad+=1; print('MISMATCH', seed, n, mcs, (out!=lab).sum())
print('cases', tot, 'bad', bad)
