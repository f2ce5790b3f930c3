# round 3: index clear on a side stream (nm_set_overlap 1) against the sequential form, same box, same library
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
line() {
  python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    st = d["stage_ms_per_step"]
    print("%-22s ms/step %.4f  search %.4f index %.4f order %.4f" % (
        sys.argv[2], d["ms_per_step"], st["search_feature_kernel"], st["index_build"], st["cell_keys_and_sort"]))
except Exception as e:
    print(sys.argv[2], "ERR", e)
PY
}
for rep in 1 2; do
for V in 0 1; do
  timeout -k 10 200 python bench.py --overlap $V --steps 30 --warmup 3 --cpu-sample 0 > $O/c3_ov${V}_$rep.json 2> $O/c3_ov${V}_$rep.err || echo "c3 $V failed"
  line $O/c3_ov${V}_$rep.json "overlap=$V rep$rep c3"
done
done
for V in 0 1; do
  timeout -k 10 200 python bench.py --overlap $V --workload c1_uniform_100k --steps 200 --warmup 20 --cpu-sample 0 > $O/c1_ov$V.json 2> $O/c1_ov$V.err || echo "c1 $V failed"
  line $O/c1_ov$V.json "overlap=$V c1"
  timeout -k 10 200 python bench.py --overlap $V --workload c2_scene_1m --steps 100 --warmup 10 --cpu-sample 0 > $O/c2_ov$V.json 2> $O/c2_ov$V.err || echo "c2 $V failed"
  line $O/c2_ov$V.json "overlap=$V c2"
  timeout -k 10 200 python bench.py --overlap $V --points 1250000 --steps 100 --warmup 10 --cpu-sample 0 > $O/p1250k_ov$V.json 2> $O/p1250k_ov$V.err || echo "p1250k $V failed"
  line $O/p1250k_ov$V.json "overlap=$V c3@1.25M"
done
