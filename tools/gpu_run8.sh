cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_fused2; mkdir -p $O
for V in fused3 fused4; do
NIMRUD_HIP_LIBRARY=$GRAFT_REPO_ROOT/build_abl/lib_$V.so timeout -k 10 300 python tests/fuzz_parity.py 120 777 > $O/fuzz_$V.log 2>&1; echo "fuzz $V exit $?"; tail -1 $O/fuzz_$V.log
done
bash tools/gpu_ab.sh r2_fused2 old fused3 fused3_occ3 fused4 fused4_w5
