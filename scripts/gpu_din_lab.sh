#!/bin/bash
# build and run scripts/din_lab.hip on the GPU box: plain kernel times, then the per-phase timeline
F="--offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -Iinclude"
/opt/rocm/bin/hipcc $F -DDIN_LAB_PLAIN scripts/din_lab.hip -o /tmp/din_lab_plain 2>/dev/null && /tmp/din_lab_plain ${1:-50}
echo "ragged lengths U{0..T}:"; /tmp/din_lab_plain ${1:-50} 1
/opt/rocm/bin/hipcc $F scripts/din_lab.hip -o /tmp/din_lab 2>/dev/null && /tmp/din_lab ${1:-50} | tail -16
