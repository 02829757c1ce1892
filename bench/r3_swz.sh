set -e
O=gpurun_out/r3swz
mkdir -p $O
timeout -k 10 400 python -u -m pytest tests/test_gpu_kernels.py -x -q --timeout 120 --timeout-method thread -k "shared_activation or fused_conv_relu_pool or conv3x3" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
bash bench/ab_bench.sh bench/base_libdtp_hip.so 3
cp gpurun_out/ab_bench.jsonl $O/ab_bench.jsonl
timeout -k 10 400 python3 bench/bench_kernels.py --batch 256 --no-stock --iters 40 > $O/per_layer_new.jsonl 2> $O/per_layer.err
DTP_HIP_LIB=bench/base_libdtp_hip.so timeout -k 10 400 python3 bench/bench_kernels.py --batch 256 --no-stock --iters 40 > $O/per_layer_base.jsonl 2>> $O/per_layer.err
python - <<'PY'
import json
a=[json.loads(l) for l in open("gpurun_out/r3swz/per_layer_base.jsonl") if "fprop_tf" in l]
b=[json.loads(l) for l in open("gpurun_out/r3swz/per_layer_new.jsonl") if "fprop_tf" in l]
for x,y in zip(a,b):
    print(x["shape"], "fprop %.0f -> %.0f (%+.1f%%)  dgrad %.0f -> %.0f (%+.1f%%)" % (x["fprop_tf"], y["fprop_tf"], 100*(y["fprop_tf"]/x["fprop_tf"]-1), x["dgrad_tf"], y["dgrad_tf"], 100*(y["dgrad_tf"]/x["dgrad_tf"]-1)))
PY
