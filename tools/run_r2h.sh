set -x
O=gpurun_out/r2h; mkdir -p $O
python -m pytest tests -m gpu -q 2>&1 | tail -40 > $O/tests_all.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
tail -5 $O/tests_all.log; tail -3 $O/smoke.log
