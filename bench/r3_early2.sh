O=gpurun_out/early2
mkdir -p $O
for i in 1 2; do
for c in 512 256 128; do
  DTP_OPT_GRID_CAP=$c DTP_GRAPH_EARLY_OPT_M=100 timeout -k 10 120 python bench.py --batch-per-gpu 4 --steps 300 --warmup 20 > $O/b4_${c}_$i.json 2>> $O/err.log || exit 1
  DTP_OPT_GRID_CAP=$c DTP_GRAPH_EARLY_OPT_M=100 timeout -k 10 120 python bench.py --batch-per-gpu 16 --steps 300 --warmup 20 > $O/b16_${c}_$i.json 2>> $O/err.log || exit 1
done
done
python - <<'PY'
import json
for i in "12":
    for m in ("512", "256", "128"):
        for b in ("b4", "b16"):
            d = json.load(open("gpurun_out/early2/%s_%s_%s.json" % (b, m, i))); print(b, "grid cap", m, round(d["value"], 1), "img/s", round(d["ms_per_step"], 4), "ms")
PY
