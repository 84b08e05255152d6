#!/bin/bash
# build and run scripts/tailfuse_lab.hip on the GPU box: plain kernel time, then the per-phase timeline
F="--offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -Iinclude"
/opt/rocm/bin/hipcc $F -DTAIL_LAB_PLAIN scripts/tailfuse_lab.hip -o /tmp/tail_lab_plain 2>&1 | grep -E "error" ; /tmp/tail_lab_plain ${1:-4096} ${2:-416}
/opt/rocm/bin/hipcc $F scripts/tailfuse_lab.hip -o /tmp/tail_lab 2>&1 | grep -E "error"; /tmp/tail_lab ${1:-4096} ${2:-416} | tail -28
