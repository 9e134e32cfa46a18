"""Run-to-run reproducibility of the NETWORK with several passes sharing the chip: N engines on N streams push the same 32 frames
ITERS times; every pass's maps (and, on a mismatch, the taps of the differing frame) are compared with a single-stream reference.
python tools/soak_network_concurrent.py [N] [ITERS]    (GPU box; SVC_MX / SVC_MX_MASK select the matrix pipe)
GEOM=HxW: maps of another geometry (187x250 = 4:3 sources, 250x140 = portrait; default 140x250) -- the taps of a differing frame are
then not printed (their shapes belong to the 16:9 network input), the count of differing passes is."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from retargetvid_amd import ops, synth, scheduler
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4
ITERS = int(sys.argv[2]) if len(sys.argv) > 2 else 300
GH, GW = [int(v) for v in os.environ.get('GEOM', '140x250').split('x')]
fr = torch.from_numpy(synth.blob_frames(32, GH, GW, seed=0)).cuda()
TAPS = [('feat4x', ops.TAP_FEAT4X, (32, 52, 64)), ('feat2x', ops.TAP_FEAT2X, (16, 26, 160)), ('feat1x', ops.TAP_FEAT1X, (8, 13, 1296)),
        ('postcnn', ops.TAP_POSTCNN, (8, 13, 256)), ('dec', ops.TAP_DEC, (32, 52, 64)), ('pre', ops.TAP_PRE, (140, 250))]
engs = [ops.Engine(seed=0) for _ in range(N)]
# BURN=kind[:blocks[:iters]]: a register-only MFMA burner (tools/micro/bf16_burner.hip; kind 0 bf16, 1 f16, 2 f32) on two extra streams beside every pass
BURN = os.environ.get('BURN')
if BURN:
    import ctypes
    bl = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'micro', 'libbf16_burner.so'))
    bl.burn_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    bk = [int(v) for v in BURN.split(':')] + [512, 400][len(BURN.split(':')) - 1:]
    bstreams = [torch.cuda.Stream() for _ in range(2)]
sts = scheduler.lane_streams(torch.device('cuda', 0), N)
ref = engs[0].saliency(fr).clone()
reft = {f: [engs[0].tap(w, f, sh) for _, w, sh in TAPS] for f in range(32)} if (GH, GW) == (140, 250) else None
outs = [torch.empty_like(ref) for _ in range(N)]
# VICTIM=k: beside every pass, k launches of a stand-in for the smoothing kernel (tools/micro/victim.hip: pure function of block and
# thread) on a stream of its own; the words that differ from its lone reference run are counted
VICTIM = int(os.environ.get('VICTIM', '0'))
if VICTIM:
    import ctypes
    vl = ctypes.CDLL(os.environ.get('VICTIM_LIB') or os.path.join(os.path.dirname(os.path.abspath(__file__)), 'micro', 'libvictim.so'))   # VICTIM_LIB: tools/micro/libpkvictim.so = the packed-instruction probes
    vl.victim_launch.argtypes = [ctypes.c_void_p]
    vl.victim_diffs.restype = ctypes.c_uint64
    assert vl.victim_init() == 0
    vstream = torch.cuda.Stream()
# SHOT=k: beside every pass, TransNet V1 (k windows of 100 frames, cells on the split-bf16 pipe whatever SVC_MX says) on a stream of its own:
# a second kind of bf16-MFMA co-runner (no v_perm_b32 in its loop, operands loaded pre-split)
SHOT = int(os.environ.get('SHOT', '0'))
if SHOT:
    from retargetvid_amd import transnetv1_handler as Hd, weights as W_
    os.environ['SVC_SHOT_MX'] = 'bf16x6'
    snet = Hd.ShotTransNet(Hd.ShotTransNetParams(), weights=W_.make_transnet_state_dict(0))
    sfr = torch.from_numpy(np.random.RandomState(1).randint(0, 256, (SHOT, 100, 27, 48, 3)).astype(np.uint8)).cuda()
    sstream = torch.cuda.Stream()
    sref = snet.predict_raw_device(sfr).clone()
    torch.cuda.synchronize()
    sbad = 0
bad = 0
for it in range(ITERS):
    if SHOT:
        with torch.cuda.stream(sstream):
            sout = snet.predict_raw_device(sfr)
    if VICTIM:
        for _ in range(VICTIM):
            vl.victim_launch(ctypes.c_void_p(vstream.cuda_stream))
    if BURN:
        for bs in bstreams:
            bl.burn_launch(ctypes.c_void_p(bs.cuda_stream), bk[1], bk[2], bk[0])
    for i in range(N):
        with torch.cuda.stream(sts[i]):
            engs[i].saliency(fr, out=outs[i])
    torch.cuda.synchronize()
    if SHOT and not torch.equal(sout, sref):
        sbad += 1
    for i in range(N):
        if not torch.equal(outs[i], ref):
            bad += 1
            if reft is None:
                print('iter %d engine %d: %d pixels differ' % (it, i, int((outs[i] != ref).sum())), flush=True)
                continue
            d = (outs[i] != ref)
            frames = torch.nonzero(d.flatten(1).any(1)).flatten().tolist()
            msg = []
            for f in frames[:2]:
                t = [engs[i].tap(w, f, sh) for _, w, sh in TAPS]
                msg.append('frame %d: %d px; taps differing: %s' % (f, int(d[f].sum()), [(n, int((a != b).sum()), '%.1e' % float(np.abs(a - b).max()))
                                                                                          for (n, _, _), a, b in zip(TAPS, t, reft[f]) if not np.array_equal(a, b)]))
            print('iter %d engine %d: %s' % (it, i, '; '.join(msg)), flush=True)
            if bad <= 4:
                f = frames[0]
                got = engs[i].tap(ops.TAP_PRE, f, (140, 250)).reshape(-1)
                want = reft[f][5].reshape(-1)
                idx = np.flatnonzero(got != want)
                print('   pre: flat indices %s (mod 16 of the first: %d; byte offset in the buffer mod 128: %d)' % (idx.tolist(), idx[0] % 16, ((f * 35000 + idx[0]) * 4) % 128))
                print('   got  %s' % np.array2string(got[idx], precision=4, max_line_width=250))
                print('   want %s' % np.array2string(want[idx], precision=4, max_line_width=250))
                # is the wrong data the reference's data of another place (a shifted / misplaced store)?
                for k in range(len(idx)):
                    w = np.flatnonzero(want == got[idx[k]])
                    if len(w): print('   got[%d] equals want at flat index %s' % (idx[k], w[:4].tolist()))
print('%d passes, %d with a map that differs from the reference' % (ITERS * N, bad))
if SHOT:
    print('TransNet beside them: %d calls of %d windows, %d with probabilities that differ from its lone run' % (ITERS, SHOT, sbad))
if VICTIM:
    print('stand-in victim: %d launches, %d words differ from its lone run' % (VICTIM * ITERS, vl.victim_diffs()))
    if hasattr(vl, 'victim_report'):
        rep = (ctypes.c_uint64 * 32)()
        vl.victim_report(rep)
        for t, name in enumerate(('pk_mul x2 + pk_add op_sel swap', 'pk_add op_sel swap alone', 'pk_mul x2 + scalar adds', 'pk_mul x2 + pk_add, no swap', 'test 0 with operands from LDS, 256 threads', 'test 0 with operands from LDS, 1024 threads')):
            print('   test %d (%s): disagreements with the scalar instructions, by quarter of the wavefront (lanes 0-15 .. 48-63): %s' % (t, name, [int(rep[t * 4 + q]) for q in range(4)]))
if hasattr(engs[0].lib, 'svc_debug_sd_log'):              # the -DSD_DEBUG build: what the smoothing kernel's self-check saw
    import ctypes
    cnt = (ctypes.c_uint * 4)()
    rec = np.zeros((64, 16), np.float32)
    assert engs[0].lib.svc_debug_sd_log(cnt, rec.ctypes.data_as(ctypes.c_void_p)) == 0
    print('smoothing kernel self-check: %d pixels checked, %d with memory != second evaluation' % (cnt[3], cnt[0]))
    print('   frame oy ox tid | memory  second  third (variant 3: first pass t1[x1]) | tile a00 a01 a10 a11 | lx1 ly1 | block cu simd (variant 3: first pass t1[x0], its t1[x1] index, the index now)')
    for r_ in rec[:min(int(cnt[0]), 64)]:
        print('   %3d %3d %3d %3d | %.6f %.6f %.6g | %.5f %.5f %.5f %.5f | %.4f %.4f | %.6g %d %d' % (r_[0], r_[1], r_[2], r_[3], r_[4], r_[5], r_[6], r_[7], r_[8], r_[9],
                                                                                                  r_[10], r_[11], r_[12], r_[13], r_[14], r_[15]))
if os.environ.get('SVC_SD_POISON'):
    print('LDS canary words the smoothing kernel found changed, per engine: %s' % [e.threshold_census()['maps'] for e in engs])
