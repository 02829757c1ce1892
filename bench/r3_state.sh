# GPU suite + default bench + small-batch benches on the current tree
mkdir -p gpurun_out/r3state
O=gpurun_out/r3state
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/gpu_suite.log 2>&1
rc=$?
tail -8 $O/gpu_suite.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python bench.py --steps 20 --warmup 5 > $O/b1024.json 2> $O/err.log || exit 1
timeout -k 10 120 python bench.py --batch-per-gpu 4 --steps 300 --warmup 20 > $O/b4.json 2>> $O/err.log || exit 1
timeout -k 10 120 python bench.py --batch-per-gpu 16 --steps 300 --warmup 20 > $O/b16.json 2>> $O/err.log || exit 1
python - <<'PY'
import json
for b in ("1024", "4", "16"):
    d = json.load(open("gpurun_out/r3state/b%s.json" % b)); print("b" + b, round(d["value"], 1), "img/s", round(d["ms_per_step"], 3), "ms graph", d["hip_graph"], "loss", d["final_loss"], "first5/last5", d["ms_first5"], d["ms_last5"])
PY
