#!/bin/bash
# round 3, call 14: full GPU suite + smoke on the final build
set -x
O=gpurun_out/c14; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --tb=short > $O/gpu_tests.log 2>&1; echo "rc tests $?"; tail -3 $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "rc smoke $?"; tail -2 $O/smoke.log
