#!/bin/bash
# round 5, GPU call 15: the whole GPU suite + smoke on the current tree
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5c15; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --durations=12 > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -25 $O/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "rc=$?" >> $O/smoke.log; tail -4 $O/smoke.log
