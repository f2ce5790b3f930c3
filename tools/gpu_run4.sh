cd $GRAFT_REPO_ROOT
O=gpurun_out/r2f; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "knn or config4" --durations=5 > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log; tail -12 $O/pytest.log
timeout -k 10 900 python tools/config4_timing.py > $O/c4_timing.log 2>&1; echo "c4 exit $?"; cat $O/c4_timing.log
