#!/bin/bash
# AddressSanitizer + UBSan over hdb_tree.h (the hierarchy code shared by the host harness and the device kernel),
# on the CPU build only (GPU sanitizers are not available on the pool).
set -e
cd "$(dirname "$0")/.."
g++ -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -shared -fPIC -o /tmp/libtree_asan.so tests/native/tree_harness.cpp
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python tools/tree_cases.py /tmp/libtree_asan.so | tail -2
