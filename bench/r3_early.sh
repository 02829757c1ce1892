O=gpurun_out/early
timeout -k 10 600 python -u -m pytest tests/test_gpu_model.py -m gpu -x -q --timeout 300 --timeout-method thread -k "hip_graph" > gpurun_out/early_tests.log 2>&1 || { tail -20 gpurun_out/early_tests.log; exit 1; }
mkdir -p $O
for i in 1 2; do
for m in 0 100 50; do
  DTP_GRAPH_EARLY_OPT_M=$m timeout -k 10 120 python bench.py --batch-per-gpu 4 --steps 300 --warmup 20 > $O/b4_${m}_$i.json 2>> $O/err.log || exit 1
  DTP_GRAPH_EARLY_OPT_M=$m timeout -k 10 120 python bench.py --batch-per-gpu 16 --steps 300 --warmup 20 > $O/b16_${m}_$i.json 2>> $O/err.log || exit 1
done
done
python - <<'PY'
import json
for i in "12":
    for m in ("0", "100", "50"):
        for b in ("b4", "b16"):
            d = json.load(open("gpurun_out/early/%s_%s_%s.json" % (b, m, i))); print(b, "early_M", m, round(d["value"], 1), "img/s", round(d["ms_per_step"], 4), "ms loss", d["final_loss"])
PY
