# search kernel with a workgroup per (scale, batch) for small clouds against the scale-walking wave: bash tools/gpu_r3_split.sh
cd $GRAFT_REPO_ROOT
for P in 300000 1250000 2500000; do for V in nosplit split3m nosplit split3m; do NIMRUD_HIP_LIBRARY=$PWD/build_abl/lib_$V.so python bench.py --points $P --steps 60 --warmup 8 --cpu-sample 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$P $V', round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['stage_ms_per_step'].items()})"; done; done
NIMRUD_HIP_LIBRARY=$PWD/build_abl/lib_split3m.so timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "golden or config2 or oracle or ladder" 2>&1 | tail -2
