set -e
O=gpurun_out/r3soak
mkdir -p $O
timeout -k 10 500 python bench/determinism_probe.py --trials 60 --steps 3 > $O/90_determinism_probe_final.log 2>&1
tail -2 $O/90_determinism_probe_final.log
timeout -k 10 300 python bench.py --steps 300 --warmup 10 > $O/soak300.json
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --data host-u8 > $O/soak_hostu8.json
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --batch-norm --fp8-classifier > $O/soak_bn.json
python - <<'PY'
import json
for f in ("soak300", "soak_hostu8", "soak_bn"):
    d = json.load(open("gpurun_out/r3soak/%s.json" % f)); print(f, round(d["value"], 1), round(d["ms_per_step"], 3), d["ms_first5"], d["ms_last5"], d["final_loss"])
PY
