set -e
O=gpurun_out/r3b4ab
mkdir -p $O
for cfg in "0 0" "1 0" "0 1" "1 1"; do
set -- $cfg
for b in 4 16; do
DTP_GRAPH_MAIN_FIRST=$1 DTP_GRAPH_DEFER_JOIN=$2 timeout -k 10 120 python bench.py --batch-per-gpu $b --steps 300 --warmup 20 > $O/tmp.json 2>> $O/err.log
python - $1 $2 $b <<'PY'
import json, sys
d = json.load(open("gpurun_out/r3b4ab/tmp.json")); print("main_first", sys.argv[1], "defer_join", sys.argv[2], "b", sys.argv[3], round(d["value"], 1), "img/s", round(d["ms_per_step"], 3), "ms", d["hip_graph"], d["final_loss"])
PY
done
done
