# fused row walk: fuzz, gpu suite, A/B against the previous kernel
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_fused; mkdir -p $O
timeout -k 10 400 python tests/fuzz_parity.py 300 > $O/fuzz.log 2>&1; echo "fuzz exit $?"; tail -2 $O/fuzz.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest.log
bash tools/gpu_ab.sh r2_fused old fused fused_w5
