set -e
O=gpurun_out/r3mid
mkdir -p $O
for g in off on off on; do
timeout -k 10 300 python bench.py --hip-graph $g --steps 20 --warmup 5 > $O/tmp.json 2>> $O/err.log
python - $g <<'PY'
import json, sys
d = json.load(open("gpurun_out/r3mid/tmp.json")); print("b1024 graph", sys.argv[1], round(d["value"], 1), "img/s", round(d["ms_per_step"], 3), "ms", d["hip_graph"], d["max_memory_reserved_gb"])
PY
done
