cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_ms; mkdir -p $O
for rep in 1 2; do for V in bnds ms256k ms32k ms4k; do for W in c1_uniform_100k c2_scene_1m; do
  NIMRUD_HIP_LIBRARY=$GRAFT_REPO_ROOT/build_abl/lib_$V.so timeout -k 10 200 python bench.py --workload $W --steps 50 --warmup 10 --cpu-sample 0 > $O/${W}_${V}_$rep.json 2> $O/${W}_${V}_$rep.err
  python -c "
import json;d=json.loads(open('$O/${W}_${V}_$rep.json').read().strip().splitlines()[-1]);print('$V',$rep,'$W','ms %.4f'%d['ms_per_step'],{k:round(v,4) for k,v in d['stage_ms_per_step'].items()})"
done; done; done
NIMRUD_HIP_LIBRARY=$GRAFT_REPO_ROOT/build_abl/lib_ms32k.so timeout -k 10 300 python tests/fuzz_parity.py 150 31337 > $O/fuzz.log 2>&1; echo "fuzz exit $?"; tail -1 $O/fuzz.log
