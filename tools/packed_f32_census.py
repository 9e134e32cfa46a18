"""Build-time check of the packed-f32 instruction forms in the library's kernels (CPU box: hipcc -S, no GPU).

Round 5 located a lost product in k_smooth_down's bilinear stage: `v_pk_mul_f32 x2 ; v_pk_add_f32 ... op_sel:[0,1] op_sel_hi:[1,0]`
(a packed add that SWAPS halves) dropped the high half of the second multiply in lanes 48-63 when a bf16-MFMA workgroup of
another stream shared the CU (profiles/r05_mx_reproducibility.txt).  The stage is scalar code since; the trigger is not fully
understood, so this tool keeps the pattern out of the build:

  * per kernel: v_pk_{mul,add,fma}_f32 counts, and how many of them carry a NON-DEFAULT op_sel / op_sel_hi (half-swapping or
    half-broadcasting forms -- the default, op_sel:[0,0] op_sel_hi:[1,1], is never printed);
  * exit code 1 if a kernel named in MUST_BE_SCALAR contains any v_pk_*_f32, or if ANY kernel contains a HALF-SWAPPING packed
    mul / add / fma (op_sel and op_sel_hi that route the two halves of one source crosswise; kernels whose C code the compiler
    packs that way carry SVC_NO_PK, svc_internal.h).  `--no-packed` compiles with -target-feature -packed-fp32-ops (the whole
    library without packed f32: 1 - 3 % slower, measured in round 6) and expects zero everywhere.

usage: python tools/packed_f32_census.py [--no-packed] [--all] [file.hip ...]      (default: the three library sources)
tests/test_kernel_specs.py runs it on the library sources (about 40 s of hipcc)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'retargetvid_amd', 'csrc')
DEFAULT = [os.path.join(CSRC, f) for f in ('svc_net.hip', 'svc_tail.hip', 'svc_shot.hip')]
MUST_BE_SCALAR = ('k_smooth_down', 'k_smooth_down_mfma')
FLAGS = ['-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-ffp-contract=off', '-S', '--cuda-device-only', '-Wno-pass-failed']
NO_PACKED = ['-Xclang', '-target-feature', '-Xclang', '-packed-fp32-ops']

PK = re.compile(r'^\s*(v_pk_(?:mul|add|fma)_f32)\b(.*)$')
SEL = re.compile(r'op_sel:\[([01,]+)\]')
SELHI = re.compile(r'op_sel_hi:\[([01,]+)\]')


def swaps_halves(rest):
    """True when some source operand feeds its HIGH half to the low result and its LOW half to the high result
    (op_sel bit 1, op_sel_hi bit 0 for the same operand): the form of the round-5 fault."""
    sel = SEL.search(rest)
    hi = SELHI.search(rest)
    s = [int(x) for x in sel.group(1).split(',')] if sel else []
    h = [int(x) for x in hi.group(1).split(',')] if hi else []
    n = max(len(s), len(h))
    s += [0] * (n - len(s))
    h += [1] * (n - len(h))
    return any(a == 1 and b == 0 for a, b in zip(s, h))


def census(path, no_packed=False):
    asm = subprocess.run(['/opt/rocm/bin/hipcc'] + FLAGS + (NO_PACKED if no_packed else []) + ['-o', '-', path],
                         capture_output=True, text=True, check=True).stdout
    rows, cur = {}, None
    for line in asm.splitlines():
        m = re.match(r'^(_Z\w+|k_\w+):', line)
        if m:
            cur = m.group(1)
            continue
        if line.startswith('\t.end_amdhsa_kernel') or line.startswith('.Lfunc_end'):
            cur = None if line.startswith('.Lfunc_end') else cur
            continue
        m = PK.match(line)
        if m and cur:
            r = rows.setdefault(cur, dict(mul=0, add=0, fma=0, modified=0, swapping=0))
            r[m.group(1)[5:8]] += 1
            if 'op_sel' in m.group(2):
                r['modified'] += 1
                if swaps_halves(m.group(2)):
                    r['swapping'] += 1
    return rows


def demangle(names):
    out = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout.splitlines()
    return {n: re.sub(r'^void ', '', d.split('(')[0]) for n, d in zip(names, out)}


def main(argv):
    no_packed = '--no-packed' in argv
    show_all = '--all' in argv
    files = [a for a in argv if not a.startswith('--')] or DEFAULT
    bad = []
    for f in files:
        rows = census(f, no_packed)
        names = demangle(list(rows))
        print('%s: %d kernels with packed f32 instructions%s' % (os.path.basename(f), len(rows), ' (built with -packed-fp32-ops)' if no_packed else ''))
        for k, r in sorted(rows.items(), key=lambda kv: -kv[1]['swapping'] * 100000 - kv[1]['modified']):
            d = names[k]
            if show_all or r['modified']:
                print('  %-72s mul %4d add %4d fma %4d | op_sel-modified %4d  half-swapping %3d' % (d[:72], r['mul'], r['add'], r['fma'], r['modified'], r['swapping']))
            base = d.split('<')[0]
            if no_packed or base in MUST_BE_SCALAR:
                bad.append('%s: %s contains packed f32 instructions' % (os.path.basename(f), d))
            elif r['swapping']:
                bad.append('%s: %s contains %d half-swapping packed f32 instructions' % (os.path.basename(f), d, r['swapping']))
    for b in bad:
        print('FAIL', b)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main(sys.argv[1:]))
