"""Print VGPR / spill / LDS / occupancy per kernel (hipcc -Rpass-analysis=kernel-resource-usage)."""
import re, subprocess, sys
for src in sys.argv[1:]:
    out = subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-ffp-contract=off', '-c', src,
                          '-o', '/dev/null', '-Rpass-analysis=kernel-resource-usage'], capture_output=True, text=True).stderr
    cur = None
    rows = {}
    for line in out.splitlines():
        m = re.search(r'remark: \s*(.+?): (.+?) \[-Rpass', line)
        if not m:
            m = re.search(r'remark: \s*Function Name: (\S+)', line)
            continue
        k, v = m.group(1).strip(), m.group(2).strip()
        if k == 'Function Name':
            cur = v; rows[cur] = {}
        elif cur:
            rows[cur][k] = v
    for name, r in rows.items():
        dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
        print('%-64s vgpr %-4s agpr %-3s spill %-3s sgpr %-4s lds %-6s occ %s' % (
            re.sub(r'^void ', '', dem.split('(')[0])[:64], r.get('VGPRs'), r.get('AGPRs'), r.get('VGPRs Spill'), r.get('SGPRs'),
            r.get('LDS Size [bytes/block]'), r.get('Occupancy [waves/SIMD]')))
