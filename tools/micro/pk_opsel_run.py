"""Stand-alone run of the packed-instruction probes (tools/micro/pk_opsel_victim.hip) beside synthetic aggressors: the probes on one
stream, an aggressor kernel of one instruction kind on two others, no library involved.   python tools/micro/pk_opsel_run.py  (GPU box)"""
import ctypes, os, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
vl = ctypes.CDLL(os.path.join(here, 'libpkvictim.so'))
vl.victim_launch.argtypes = [ctypes.c_void_p]
vl.aggressor_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
assert vl.victim_init() == 0
vs = torch.cuda.Stream()
ags = [torch.cuda.Stream() for _ in range(2)]
names = ['v_pk_mov_b32 op_sel:[1,0]', 'v_perm_b32', 'v_perm_b32 + v_mfma_f32_32x32x16_bf16', 'ds_read_b128 + v_perm_b32 + the MFMA']
prev = [0] * 32
for kind in range(4):
    for rep in range(40):
        for a in ags:
            vl.aggressor_launch(ctypes.c_void_p(a.cuda_stream), kind, 2048, 4000)
        for _ in range(3):
            vl.victim_launch(ctypes.c_void_p(vs.cuda_stream))
    torch.cuda.synchronize()
    rep_ = (ctypes.c_uint64 * 32)()
    vl.victim_report(rep_)
    now = [int(v) for v in rep_]
    d = [n - p for n, p in zip(now, prev)]
    prev = now
    print('aggressor %d (%s): probe disagreements per test x quarter: %s' % (kind, names[kind], [d[4 * t:4 * t + 4] for t in range(6)]))
