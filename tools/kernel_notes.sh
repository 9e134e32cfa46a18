#!/bin/bash
# Private segment / spill table of every kernel of the three device code objects (compiled from the sources with the
# Makefile's flags, device side only): tools/kernel_notes.sh > profiles/rNN_kernel_resources.txt
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
for src in svc_tail svc_net svc_shot; do
  echo "# $src.hip: hipcc -Rpass-analysis=kernel-resource-usage (tools/kernel_resources.py)"
  python3 $R/tools/kernel_resources.py $R/retargetvid_amd/csrc/$src.hip
  echo
  echo "# $src.hip: .private_segment_fixed_size / .vgpr_count / .vgpr_spill_count of the gfx950 code object (clang-offload-bundler --unbundle, llvm-readelf --notes)"
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function -Wno-pass-failed --cuda-device-only -c $R/retargetvid_amd/csrc/$src.hip -o $T/$src.co 2>/dev/null
  /opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/$src.co --output=$T/$src.elf
  /opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/$src.elf > $T/$src.notes
  python3 - $T/$src.notes <<'PY'
import re, sys, subprocess
t = open(sys.argv[1]).read()
for blk in t.split('- .agpr_count')[1:]:
    name = re.search(r'\.name:\s+(\S+)', blk).group(1)
    dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip().split('(')[0]
    g = lambda k: re.search(r'\.%s:\s+(\d+)' % k, blk).group(1)
    print('%-58s private_segment %5s  vgpr %4s  spills %5s  lds(static) %6s' % (dem[:58], g('private_segment_fixed_size'), g('vgpr_count'), g('vgpr_spill_count'), g('group_segment_fixed_size')))
PY
  echo
done
rm -rf $T
