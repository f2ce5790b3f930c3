cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_sort; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -2 $O/pytest.log
timeout -k 10 400 python tests/fuzz_parity.py 300 8675309 > $O/fuzz.log 2>&1; echo "fuzz exit $?"; tail -1 $O/fuzz.log
for W in c1_uniform_100k c2_scene_1m c3_scene_10m c5_scene_10m_rf; do
  timeout -k 10 200 python bench.py --workload $W --steps 20 --warmup 5 --cpu-sample 0 > $O/$W.json 2> $O/$W.err
  python -c "
import json;d=json.loads(open('$O/$W.json').read().strip().splitlines()[-1]);print('$W','ms %.4f'%d['ms_per_step'],'value %.4g'%d['value'],{k[:5]:round(v,4) for k,v in d['stage_ms_per_step'].items()})"
done
timeout -k 10 400 python tools/config4_timing.py 50000000 > $O/c4.log 2>&1; echo "c4 exit $?"; tail -4 $O/c4.log
