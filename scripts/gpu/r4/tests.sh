#!/bin/bash
# the GPU suite only.  Usage: bash scripts/gpu/r4/tests.sh <tag> [pytest args]
set -o pipefail
TAG=${1:-r4t}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 1000 python -m pytest tests -m gpu -q --durations=15 -s "$@" > $OUT/pytest.log 2>&1
echo "pytest exit $?" | tee -a $OUT/pytest.log
grep -E "passed|failed|FAILED|ERROR" $OUT/pytest.log | tail -30
