"""RetargetVid evaluator counterpart (reference: retargetvid_eval.py:1-286).

Scores every run (sub-directory or zip member directory) of a results folder against
the six annotators: per-frame IoU -> per-video mean -> per-annotator mean ->
worst / best / mean x100 per aspect ratio, and writes ``eval_current.txt`` in the
reference's column format (retargetvid_eval.py:226-283).  The reference script runs
unchanged on this package's output files; this module exists so the same numbers can
be produced on a box without the reference tree, with the per-frame IoU (K16,
retargetvid_eval.py:10-27 == smartVidCrop.py:927-944) evaluated by the HIP kernel
behind ``svc_iou_i32`` (include/svc.h).

Differences from the reference script, none of which change the numbers:
annotation zips are read in place instead of being extracted next to the script
(:45-63); box pairs are batched into two int32 arrays before scoring.  A run file with
fewer rows than the annotation is scored over the rows it has, with a warning, like the
reference's print-and-break at :163-178 (a file with no rows at all raises, as
statistics.mean does there); 'cuts_extra:' / 'no_extra_cuts:' info lines are parsed (:208-218).
"""
import io
import math
import os
import warnings
import zipfile

import numpy as np

VID_INDS = list(range(1, 101)) + list(range(601, 701))        # retargetvid_eval.py:68
ARS = ('1-3', '3-1')


def _read_boxes(text):
    rows = [l.split(',') for l in text.splitlines() if l.strip()]
    return np.array([[int(c[0]), int(c[1]), int(c[2]), int(c[3])] for c in rows], np.int32).reshape(-1, 4)


def load_annotations(folder):
    """folder holds annotator_{1..6}.zip or extracted annotator_{1..6}/ directories.
    -> list over annotators of {ar: {vid: int32[n,4]}}."""
    annots = []
    for a in range(1, 7):
        name = 'annotator_%d' % a
        d, z = os.path.join(folder, name), os.path.join(folder, name + '.zip')
        cur = {ar: {} for ar in ARS}
        if os.path.isdir(d):
            for ar in ARS:
                for v in VID_INDS:
                    with open(os.path.join(d, '%03d_%s.txt' % (v, ar))) as fp:
                        cur[ar][v] = _read_boxes(fp.read())
        elif os.path.isfile(z):
            with zipfile.ZipFile(z) as zf:
                for ar in ARS:
                    for v in VID_INDS:
                        cur[ar][v] = _read_boxes(zf.read('%s/%03d_%s.txt' % (name, v, ar)).decode())
        else:
            raise FileNotFoundError('%s: neither directory nor zip found' % d)
        annots.append(cur)
    return annots


def frame_counts(folder):
    """{vid: number of frames} from the first annotator's 1-3 files alone (retargetvid_eval.py:99-101), without parsing
    the 2 400 annotation files."""
    name = 'annotator_1'
    d, z = os.path.join(folder, name), os.path.join(folder, name + '.zip')
    out = {}
    if os.path.isdir(d):
        for v in VID_INDS:
            with open(os.path.join(d, '%03d_1-3.txt' % v)) as fp:
                out[v] = sum(1 for l in fp if l.strip())
    elif os.path.isfile(z):
        with zipfile.ZipFile(z) as zf:
            for v in VID_INDS:
                out[v] = sum(1 for l in zf.read('%s/%03d_1-3.txt' % (name, v)).decode().splitlines() if l.strip())
    else:
        raise FileNotFoundError('%s: neither directory nor zip found' % d)
    return out


def list_runs(results):
    """results: a directory of run sub-directories, or a zip whose top-level entries are runs."""
    if os.path.isdir(results):
        return sorted(f.name for f in os.scandir(results) if f.is_dir())
    with zipfile.ZipFile(results) as zf:
        return sorted({n.split('/')[0] for n in zf.namelist() if '/' in n})


def load_run(results, run):
    """-> ({ar: {vid: int32[n,4]}}, {ar: {vid: info_text}}, missing_file_count)."""
    boxes = {ar: {} for ar in ARS}
    infos = {ar: {} for ar in ARS}
    missing = 0
    zf = None if os.path.isdir(results) else zipfile.ZipFile(results)

    def read(rel):
        if zf is None:
            p = os.path.join(results, rel)
            if not os.path.isfile(p):
                return None
            with open(p) as fp:
                return fp.read()
        try:
            return zf.read(rel).decode()
        except KeyError:
            return None

    for ar in ARS:
        for v in VID_INDS:
            t = read('%s/%03d_%s.txt' % (run, v, ar))
            if t is None:
                missing += 1
                continue
            boxes[ar][v] = _read_boxes(t)
            t = read('%s/%03d_%s_info.txt' % (run, v, ar))
            if t is not None:
                infos[ar][v] = t
    if zf is not None:
        zf.close()
    return boxes, infos, missing


def pair_boxes(annots, boxes):
    """Lay out every (aspect ratio, annotator, video, frame) pair as two int32[M,4]
    arrays, negatives clamped to 0 (retargetvid_eval.py:181-190).
    -> gt[M,4], method[M,4], index[(ar, user, vid, start, count)]."""
    gts, mts, index = [], [], []
    pos = 0
    for ar in ARS:
        for v in VID_INDS:
            if v not in boxes[ar]:
                continue
            n = len(annots[0]['1-3'][v])                   # frame_counts, retargetvid_eval.py:99-101
            m = boxes[ar][v]
            if len(m) < n:                                 # reference: "could not find annotation!" + break, mean over what exists
                if len(m) == 0:
                    raise ValueError('run has no rows for video %03d_%s' % (v, ar))
                warnings.warn('run has %d rows for video %03d_%s, annotations have %d: scoring the first %d frames'
                              % (len(m), v, ar, n, len(m)))
                n = len(m)
            for user in range(6):
                gts.append(np.maximum(annots[user][ar][v][:n], 0))
                mts.append(np.maximum(m[:n], 0))
                index.append((ar, user, v, pos, n))
                pos += n
    if not gts:
        return np.zeros((0, 4), np.int32), np.zeros((0, 4), np.int32), index
    return (np.ascontiguousarray(np.concatenate(gts), np.int32),
            np.ascontiguousarray(np.concatenate(mts), np.int32), index)


def aggregate(ious, index):
    """per-frame IoU (float64[M]) -> {ar: (worst, best, mean)} in percent."""
    per_user = {ar: [[] for _ in range(6)] for ar in ARS}
    for ar, user, v, start, n in index:
        per_user[ar][user].append(math.fsum(ious[start:start + n]) / n)
    out = {}
    for ar in ARS:
        if not per_user[ar][0]:
            continue
        users = [math.fsum(x) / len(x) for x in per_user[ar]]
        out[ar] = (min(users) * 100, max(users) * 100, math.fsum(users) / len(users) * 100)
    return out


def parse_info_stats(infos):
    """'%'-bearing lines of *_info.txt -> {ar: {key: [values]}} (retargetvid_eval.py:196-207)."""
    stats = {ar: {} for ar in ARS}
    for ar in ARS:
        for v, text in infos[ar].items():
            for k in text.splitlines():
                if '%' in k:
                    key = k.split(':')[0].strip().lower()
                    stats[ar].setdefault(key, []).append(float(k.split(',')[1].replace('%', '').strip()))
                else:
                    for name in ('cuts_clust', 'cuts_extra', 'no_extra_cuts'):      # first match wins, like the elif chain
                        if name + ':' in k:
                            stats[ar].setdefault(name, []).append(int(k.split(':')[1].strip()))
                            break
    return stats


def format_report(rows):
    """rows: [(run, {ar: (worst,best,mean)}, stats, missing)] -> text of eval_current.txt."""
    head = ('Method', 'Worst', 'Best', 'Mean', 'ttm', 'tta', 'tcm', 'tca', 'ccm', 'cca', 'ecm', 'eca',
            'Worst', 'Best', 'Mean', 'ttm', 'tta', 'tcm', 'tca', 'ccm', 'cca', 'ecm', 'eca', 'mf')
    lines = [('%-36s,' + ','.join(['%-6s'] * 23)) % head]

    def mm(d, key):
        if key in d and d[key]:
            return max(d[key]), math.fsum(d[key]) / len(d[key])
        return -1, -1

    for run, scores, stats, missing in rows:
        name = run.replace('_', ',')
        if 'mt=1.0_rf=' not in name:                       # retargetvid_eval.py:229-231 (a no-op after the first replace, kept verbatim)
            name = name.replace('_mt=1.0', '_mt=1.0_rf=1')
        s = '%-36s,' % name.replace('_', ',')
        for ar in ARS:
            if ar not in scores:
                continue
            w, b, m = scores[ar]
            vals = (w, b, m) + mm(stats[ar], 't_total') + mm(stats[ar], 't__clustering') + \
                mm(stats[ar], 'cuts_clust') + mm(stats[ar], 'cuts_extra')
            s += ''.join('%05.3f,' % x for x in vals)
        s += '%d' % missing
        lines.append(s)
    return '\n'.join(lines) + '\n'


def evaluate(results, annotations, out_path='eval_current.txt'):
    """Score every run under ``results``.  Per-frame IoU runs on the GPU through
    ``svc_iou_i32`` (retargetvid_amd.ops.iou_boxes), which raises if the HIP library or a
    GPU is missing — there is no CPU fallback."""
    from . import ops
    iou_fn = ops.iou_boxes
    annots = load_annotations(annotations)
    rows = []
    for run in list_runs(results):
        boxes, infos, missing = load_run(results, run)
        gt, mt, index = pair_boxes(annots, boxes)
        ious = np.asarray(iou_fn(gt, mt), np.float64)
        rows.append((run, aggregate(ious, index), parse_info_stats(infos), missing))
    text = format_report(rows)
    if out_path:
        with open(out_path, 'w') as fp:
            fp.write(text)
    return rows, text
