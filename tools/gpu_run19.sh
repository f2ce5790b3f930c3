cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_runin; mkdir -p $O
for rep in 1 2 3; do for R in 0 0.4 1.5; do
  NIMRUD_BENCH_RUN_IN_S=$R timeout -k 10 200 python bench.py --cpu-sample 0 > $O/c3_${R}_$rep.json 2> $O/c3_${R}_$rep.err
  python -c "
import json;d=json.loads(open('$O/c3_${R}_$rep.json').read().strip().splitlines()[-1]);print('run-in $R',$rep,'steps',d['steps'],d['warmup'],'ms %.4f'%d['ms_per_step'],{k[:5]:round(v,4) for k,v in d['stage_ms_per_step'].items()})"
done; done
