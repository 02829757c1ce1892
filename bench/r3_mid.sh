set -e
O=gpurun_out/r3mid
mkdir -p $O
for b in 32 64 128 256; do
for g in off on off on; do
timeout -k 10 200 python bench.py --batch-per-gpu $b --hip-graph $g --steps 100 --warmup 15 > $O/tmp.json 2>> $O/err.log
python - $b $g <<'PY'
import json, sys
d = json.load(open("gpurun_out/r3mid/tmp.json")); print("b", sys.argv[1], "graph", sys.argv[2], round(d["value"], 1), "img/s", round(d["ms_per_step"], 3), "ms", d["hip_graph"])
PY
done
done
