#!/bin/bash
# run pytest on the GPU box; full log + the FAILURES section on its own (the strict-parity table drowns it otherwise)
# usage: scripts/gpu_pytest.sh <tag> <pytest args...>
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
T=$1; shift
timeout ${PYTEST_TIMEOUT:-1500} python -m pytest "$@" -q --tb=short > gpurun_out/${T}_pytest_full.log 2>&1
awk '/^=+ FAILURES =+/{f=1} /^# strict /{f=0} f' gpurun_out/${T}_pytest_full.log | cut -c1-400 | head -400 > gpurun_out/${T}_failures.log
rm -f gpurun_out/${T}_pytest_full.log.gz; tail -c 3000000 gpurun_out/${T}_pytest_full.log | gzip > gpurun_out/${T}_pytest_full.log.gz; 
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/${T}_pytest_full.log | tail -40
rm -f gpurun_out/${T}_pytest_full.log
