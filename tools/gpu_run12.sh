cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_bnds; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -2 $O/pytest.log
for rep in 1 2; do for V in head bnds; do for W in c1_uniform_100k c2_scene_1m c3_scene_10m; do
  NIMRUD_HIP_LIBRARY=$GRAFT_REPO_ROOT/build_abl/lib_$V.so timeout -k 10 200 python bench.py --workload $W --steps 30 --warmup 5 --cpu-sample 0 > $O/${W}_${V}_$rep.json 2> $O/${W}_${V}_$rep.err
  python -c "
import json;d=json.loads(open('$O/${W}_${V}_$rep.json').read().strip().splitlines()[-1]);print('$V',$rep,'$W','ms %.4f'%d['ms_per_step'],{k:round(v,4) for k,v in d['stage_ms_per_step'].items()})"
done; done; done
