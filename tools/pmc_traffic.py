"""HBM traffic per kernel, per kernel class and per STEP from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

  python tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json> [--steps N] [--note TEXT]

N = the number of bench steps both passes ran (warm-up + profiling + timed steps all launch the same kernels;
pass the total, see tools/refresh_profiles.sh) so that the table can be quoted per step of 32 frames.

Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes: both counters are in
KB; they count the L2's memory-side requests, Infinity-Cache hits included, so they are an UPPER bound on HBM
bytes for tensors that stay in the 256 MiB cache between producer and consumer.  On gfx950 FETCH_SIZE reports
half the bytes of a wide (16 B per lane) coalesced read: the read side is doubled ONLY for kernels whose loads
are of that kind (WIDE below); byte- and dword-granular kernels are uncalibrated and left as counted.  WRITE_SIZE
is exact for 16-B-per-lane stores.  Separate passes because FETCH_SIZE (3 TCC slots) and WRITE_SIZE (2) do not
fit one pass.

Classes are the library's own (include/svc.h SVC_K_*, the ProfScope around each launch in csrc/): first match
in CLASS wins, so the longer prefixes come first (k_dwpw is a `pw` kernel, not `dw`)."""
import argparse
import collections
import csv
import glob
import json

CLASS = [('k_dwpw', 'pw'), ('k_pw', 'pw'), ('k_irb', 'pw'), ('k_dw', 'dw'), ('k_front', 'stem'), ('k_stem', 'stem'), ('k_lanczos', 'lanczos'),
         ('k_cv_resize', 'resize'), ('k_smooth', 'smooth'), ('k_quant', 'smooth'),
         ('k_subsample', 'resample'), ('k_upsample', 'resample'), ('k_gauss', 'resample'), ('k_adapt', 'resample'),
         ('k_tail_front', 'prim'), ('k_tail_back', 'finish'), ('k_centre_argmax', 'tail_misc'), ('k_prim_lvl', 'prim'), ('k_prim_pt', 'prim'), ('k_prim_big', 'prim'), ('k_prim', 'prim'), ('k_tree_par', 'finish'), ('k_sort', 'finish'), ('k_tree', 'finish'), ('k_finish', 'finish'), ('k_core', 'core'),
         ('k_compact', 'compact'), ('k_threshold', 'threshold'),
         ('k_blend', 'tail_misc'), ('k_map_resize', 'tail_misc'), ('k_centre', 'tail_misc'), ('k_iou', 'tail_misc')]
# classes whose global loads are 16 B per lane (float4): the gfx950 FETCH_SIZE correction (x2) applies
WIDE = {'pw', 'dw', 'stem', 'resample'}


def classify(name):
    return next((c for p, c in CLASS if name.startswith(p)), 'other')


def load(d, counter):
    per = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] != counter:
                continue
            name = r['Kernel_Name'].replace('void ', '').split('(')[0]
            per[name][0] += 1
            per[name][1] += float(r['Counter_Value'])
    return per


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('fetch_dir')
    ap.add_argument('write_dir')
    ap.add_argument('out')
    ap.add_argument('--steps', type=int, default=0, help='steps of 32 frames both passes ran (0: per-step figures omitted)')
    ap.add_argument('--note', default='')
    a = ap.parse_args()
    fetch, write = load(a.fetch_dir, 'FETCH_SIZE'), load(a.write_dir, 'WRITE_SIZE')
    kernels, classes = {}, collections.defaultdict(lambda: dict(launches=0, fetch_raw_KB=0.0, write_KB=0.0, hbm_bytes=0.0))
    for name in sorted(set(fetch) | set(write)):
        nf, kf = fetch.get(name, [0, 0.0])
        nw, kw = write.get(name, [0, 0.0])
        cls = classify(name)
        corr = 2.0 if cls in WIDE else 1.0
        n = max(nf, nw)
        byt = (corr * kf + kw) * 1024.0
        kernels[name] = dict(cls=cls, launches=n, fetch_raw_KB_per_launch=round(kf / max(nf, 1), 1),
                             write_KB_per_launch=round(kw / max(nw, 1), 1), fetch_correction=corr,
                             hbm_bytes_per_launch=round(byt / max(n, 1)))
        c = classes[cls]
        c['launches'] += n
        c['fetch_raw_KB'] += kf
        c['write_KB'] += kw
        c['hbm_bytes'] += byt
    out_cls = {}
    for cls, c in sorted(classes.items()):
        n = max(c['launches'], 1)
        row = dict(launches=c['launches'], fetch_correction=2.0 if cls in WIDE else 1.0,
                   calibrated=cls in WIDE, fetch_raw_KB_per_launch=round(c['fetch_raw_KB'] / n, 1),
                   write_KB_per_launch=round(c['write_KB'] / n, 1), hbm_bytes_per_launch=round(c['hbm_bytes'] / n))
        if a.steps:
            row.update(launches_per_step=round(c['launches'] / a.steps, 2), hbm_bytes_per_step=round(c['hbm_bytes'] / a.steps))
        out_cls[cls] = row
    doc = dict(note='rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; fetch doubled (gfx950 correction) only for the '
                    'classes whose loads are 16 B per lane, other classes as counted (uncalibrated); memory-side requests '
                    'include Infinity-Cache hits. ' + a.note,
               steps=a.steps, classes=out_cls, kernels=kernels)
    if a.steps:
        net = ('lanczos', 'stem', 'pw', 'dw', 'resample', 'smooth')
        doc['network_hbm_bytes_per_step'] = round(sum(classes[c]['hbm_bytes'] for c in net if c in classes) / a.steps)
        doc['conv_stack_hbm_bytes_per_step'] = round(sum(classes[c]['hbm_bytes'] for c in ('stem', 'pw', 'dw', 'resample') if c in classes) / a.steps)
    json.dump(doc, open(a.out, 'w'), indent=1)
    print(json.dumps(dict(classes=out_cls, **{k: v for k, v in doc.items() if k.endswith('per_step')}), indent=1))


if __name__ == '__main__':
    main()
