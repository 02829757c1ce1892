set -e
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_kernels.py -x -q --timeout 300 --timeout-method thread -k "fused_conv_relu_pool or shared_activation" > gpurun_out/r3_pool_tests.log 2>&1 || { tail -40 gpurun_out/r3_pool_tests.log; exit 1; }
tail -3 gpurun_out/r3_pool_tests.log
for i in 1 2; do
DTP_FUSE_IGEMM_POOL=0 timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r3_pool_bench_off_$i.json 2>> gpurun_out/r3_pool_bench.err
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r3_pool_bench_on_$i.json 2>> gpurun_out/r3_pool_bench.err
done
python - <<'PY'
import json
for n in ("off_1","on_1","off_2","on_2"):
    d=json.load(open("gpurun_out/r3_pool_bench_%s.json"%n)); print(n, round(d["value"],1), round(d["ms_per_step"],2), d["final_loss"], d["max_memory_reserved_gb"])
PY
