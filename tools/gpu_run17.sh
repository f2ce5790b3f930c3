cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_tune4; mkdir -p $O
for rep in 1 2; do for V in cur it_1024_13 it_1024_14 it_1024_15; do for W in c3_scene_10m; do
  NIMRUD_HIP_LIBRARY=$GRAFT_REPO_ROOT/build_abl/lib_$V.so timeout -k 10 200 python bench.py --workload $W --steps 20 --warmup 5 --cpu-sample 0 > $O/${W}_${V}_$rep.json 2> $O/${W}_${V}_$rep.err
  python -c "
import json;d=json.loads(open('$O/${W}_${V}_$rep.json').read().strip().splitlines()[-1]);print('$V',$rep,'ms %.4f'%d['ms_per_step'],{k[:5]:round(v,4) for k,v in d['stage_ms_per_step'].items()})"
done; done; done
