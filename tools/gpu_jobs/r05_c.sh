#!/bin/bash
run() { echo "== $*"; env "$@" python tools/soak_network_concurrent.py 4 600 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-200; }
run SVC_MX_MASK=1 SVC_SD_POISON=2
run SVC_MX_MASK=1 SVC_SD_POISON=3
