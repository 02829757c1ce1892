# generic same-box A/B of one environment switch on the default bench: bash bench/r3_ab.sh VAR [extra bench args]
V=$1; shift
mkdir -p gpurun_out/ab
O=gpurun_out/ab
timeout -k 10 600 python -u -m pytest tests/test_gpu_model.py -m gpu -x -q --timeout 300 --timeout-method thread -k "forward_backward or layerwise or fused_bias or hip_graph_step or unpool or accumulation" > $O/tests.log 2>&1
rc=$?
tail -4 $O/tests.log
[ $rc -eq 0 ] || exit $rc
for i in 1 2 3; do
  env $V=0 timeout -k 10 200 python bench.py --steps 20 --warmup 5 "$@" > $O/off_$i.json 2>> $O/err.log || exit 1
  env $V=1 timeout -k 10 200 python bench.py --steps 20 --warmup 5 "$@" > $O/on_$i.json 2>> $O/err.log || exit 1
done
python - $V <<'PY'
import json, sys
for n in ("off_1", "on_1", "off_2", "on_2", "off_3", "on_3"):
    d = json.load(open("gpurun_out/ab/%s.json" % n)); print(sys.argv[1], n, round(d["value"], 1), "img/s", round(d["ms_per_step"], 3), "ms loss", d["final_loss"])
PY
