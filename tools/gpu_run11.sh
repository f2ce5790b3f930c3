cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_bnd; mkdir -p $O
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "argument_checks" > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -2 $O/pytest.log
for rep in 1 2 3; do for V in ch1 bnd4; do
  NIMRUD_HIP_LIBRARY=$GRAFT_REPO_ROOT/build_abl/lib_$V.so timeout -k 10 200 python bench.py --steps 20 --warmup 5 --cpu-sample 0 > $O/c3_${V}_$rep.json 2> $O/c3_${V}_$rep.err
  python -c "
import json;d=json.loads(open('$O/c3_${V}_$rep.json').read().strip().splitlines()[-1]);print('$V',$rep,'ms %.3f'%d['ms_per_step'],d['stage_ms_per_step'])"
done; done
bash tools/collect_profiles_r2.sh r2_final3 > $O/collect.log 2>&1; tail -3 $O/collect.log
