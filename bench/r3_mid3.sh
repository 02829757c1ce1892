# mid-batch benches on the current tree (default --hip-graph auto)
O=gpurun_out/r3mid3
mkdir -p $O
for b in 32 64 128 256 512; do
timeout -k 10 200 python bench.py --batch-per-gpu $b --steps 60 --warmup 10 > $O/b$b.json 2>> $O/err.log || exit 1
done
timeout -k 10 200 python bench.py --steps 20 --warmup 5 > $O/b1024.json 2>> $O/err.log || exit 1
timeout -k 10 300 python bench.py --batch-norm --fp8-classifier --steps 20 --warmup 5 > $O/cfg5.json 2>> $O/err.log || exit 1
python - <<'PY'
import json
for b in ("b32", "b64", "b128", "b256", "b512", "b1024", "cfg5"):
    d = json.load(open("gpurun_out/r3mid3/%s.json" % b)); print(b, round(d["value"], 1), "img/s", round(d["ms_per_step"], 3), "ms graph", d.get("hip_graph"))
PY
