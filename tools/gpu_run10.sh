cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_ch1; mkdir -p $O
timeout -k 10 400 python tests/fuzz_parity.py 300 991 > $O/fuzz.log 2>&1; echo "fuzz exit $?"; tail -1 $O/fuzz.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -2 $O/pytest.log
bash tools/gpu_ab.sh r2_ch1 sc1 ch1
