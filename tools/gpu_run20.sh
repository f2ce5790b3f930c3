cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_rank; mkdir -p $O
NIMRUD_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 2 --points 4000000 --steps 3 --warmup 1 > $O/rehearsal.json 2> $O/rehearsal.err; echo "rehearsal exit $?"; tail -c 300 $O/rehearsal.json
for rep in 1 2; do for V in cur rank_basic rank_basic_memoize; do
  [ -f build_abl/lib_$V.so ] || continue
  NIMRUD_HIP_LIBRARY=$GRAFT_REPO_ROOT/build_abl/lib_$V.so timeout -k 10 200 python bench.py --steps 20 --warmup 5 --cpu-sample 0 > $O/c3_${V}_$rep.json 2> $O/c3_${V}_$rep.err
  python -c "
import json;d=json.loads(open('$O/c3_${V}_$rep.json').read().strip().splitlines()[-1]);print('$V',$rep,'ms %.4f'%d['ms_per_step'],{k[:5]:round(v,4) for k,v in d['stage_ms_per_step'].items()})"
done; done
