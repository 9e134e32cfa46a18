"""HBM traffic per launch and kernel class from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

  python tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json>

Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md prescribes: both counters are in
KB; on gfx950 FETCH_SIZE reports half the bytes of a wide (16 B/lane) coalesced read, so the read
side is doubled; WRITE_SIZE is exact for 16-B-per-lane stores.  Separate passes because FETCH_SIZE
(3 TCC slots) and WRITE_SIZE (2) do not fit one pass."""
import collections, csv, glob, json, sys

CLASS = [('k_pw', 'pw'), ('k_irb', 'pw'), ('k_dw', 'dw'), ('k_stem', 'stem'), ('k_lanczos', 'lanczos'), ('k_cv_resize', 'resize'),
         ('k_smooth', 'smooth'), ('k_quant', 'smooth'), ('k_prim', 'prim'), ('k_finish', 'finish'), ('k_core', 'core'),
         ('k_compact', 'compact'), ('k_threshold', 'threshold')]


def load(d, counter):
    tot = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] != counter:
                continue
            name = r['Kernel_Name'].replace('void ', '')
            cls = next((c for p, c in CLASS if name.startswith(p)), 'resample')
            tot[cls][0] += 1
            tot[cls][1] += float(r['Counter_Value'])
    return tot


fetch, write = load(sys.argv[1], 'FETCH_SIZE'), load(sys.argv[2], 'WRITE_SIZE')
out = {}
for cls in sorted(set(fetch) | set(write)):
    nf, kf = fetch.get(cls, [0, 0.0])
    nw, kw = write.get(cls, [0, 0.0])
    n = max(nf, nw, 1)
    out[cls] = dict(launches=n, fetch_raw_KB_per_launch=round(kf / max(nf, 1), 1), write_KB_per_launch=round(kw / max(nw, 1), 1),
                    hbm_bytes_per_launch=round((2.0 * kf / max(nf, 1) + kw / max(nw, 1)) * 1024))
json.dump(dict(note='rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, fetch doubled per the gfx950 correction; '
                    'bench.py --steps 3 --warmup 1 --pipeline 1 --cpu-sample 0', classes=out), open(sys.argv[3], 'w'), indent=1)
print(json.dumps(out, indent=1))
