#!/bin/bash
# GPU box: the hazard-covered build -- the whole GPU suite, then A/B against the build before the covers (C2 and C5) on one board.
set -o pipefail
export TMPDIR=/tmp
T=${1:-r4j}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q --durations=6 > gpurun_out/${T}_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -12 gpurun_out/${T}_pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 tools/ab_libs.sh c2,c5 4 prehz new 2>&1 | tee gpurun_out/${T}_ab_hazard_covers.log
