#!/bin/bash
for v in 4 8 16 32; do SVC_LZ_ROWS=$v python tools/gpu_check.py 2>&1 | grep -E "class lanczos|ALL OK|FAIL" | tr '\n' ' '; echo " lz_rows=$v"; done
for v in 4 7 14 28; do SVC_SD_ROWS=$v python tools/gpu_check.py 2>&1 | grep -E "class smooth|ALL OK|FAIL" | tr '\n' ' '; echo " sd_rows=$v"; done
