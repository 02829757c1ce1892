# un-pooling data gradient: tests, then same-box A/B of the default bench
mkdir -p gpurun_out/unpool
O=gpurun_out/unpool
timeout -k 10 600 python -u -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -x -q --timeout 300 --timeout-method thread -k "unpool or bias_grad or relu_bitmask or forward_backward or hip_graph_step or maxpool" > $O/tests.log 2>&1
rc=$?
tail -15 $O/tests.log
[ $rc -eq 0 ] || exit $rc
for i in 1 2; do
  DTP_FUSE_UNPOOL=0 timeout -k 10 200 python bench.py --steps 20 --warmup 5 > $O/off_$i.json 2>> $O/err.log || exit 1
  DTP_FUSE_UNPOOL=1 timeout -k 10 200 python bench.py --steps 20 --warmup 5 > $O/on_$i.json 2>> $O/err.log || exit 1
done
python - <<'PY'
import json
for n in ("off_1", "on_1", "off_2", "on_2"):
    d = json.load(open("gpurun_out/unpool/%s.json" % n)); print(n, round(d["value"], 1), "img/s", round(d["ms_per_step"], 3), "ms loss", d["final_loss"])
PY
