cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_ftop; mkdir -p $O
for V in ftop2 ftop3; do
NIMRUD_HIP_LIBRARY=$GRAFT_REPO_ROOT/build_abl/lib_$V.so timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "forest or config5 or c5 or classif" > $O/pytest_$V.log 2>&1; echo "pytest $V exit $?"; tail -2 $O/pytest_$V.log
done
for rep in 1 2; do for V in cur ftop2 ftop3; do
  NIMRUD_HIP_LIBRARY=$GRAFT_REPO_ROOT/build_abl/lib_$V.so timeout -k 10 200 python bench.py --workload c5_scene_10m_rf --steps 20 --warmup 5 --cpu-sample 0 > $O/c5_${V}_$rep.json 2> $O/c5_${V}_$rep.err
  python -c "
import json;d=json.loads(open('$O/c5_${V}_$rep.json').read().strip().splitlines()[-1]);print('$V',$rep,'ms %.4f'%d['ms_per_step'],{k[:5]:round(v,4) for k,v in d['stage_ms_per_step'].items()}, 'forest %.3f'%d['forest']['ms_per_step'])"
done; done
