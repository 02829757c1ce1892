set -e
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_kernels.py -x -q --timeout 300 --timeout-method thread -k "shared_activation or 8phase or fused_bias_grad or bitmask" > gpurun_out/r3_sa2_tests.log 2>&1 || { tail -40 gpurun_out/r3_sa2_tests.log; exit 1; }
tail -3 gpurun_out/r3_sa2_tests.log
DTP_IGEMM_SHARED_A_N128=0 timeout -k 10 240 python bench/bench_kernels.py --batch 256 --no-stock --iters 8 --out gpurun_out/r3_sa2_iso_off.json > gpurun_out/r3_sa2_iso_off.log 2>&1
timeout -k 10 240 python bench/bench_kernels.py --batch 256 --no-stock --iters 8 --out gpurun_out/r3_sa2_iso_on.json > gpurun_out/r3_sa2_iso_on.log 2>&1
for i in 1 2; do
DTP_IGEMM_SHARED_A_N128=0 timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r3_sa2_bench_off_$i.json 2>> gpurun_out/r3_sa2_bench.err
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r3_sa2_bench_on_$i.json 2>> gpurun_out/r3_sa2_bench.err
done
python - <<'PY'
import json
for n in ("off_1","on_1","off_2","on_2"):
    d=json.load(open("gpurun_out/r3_sa2_bench_%s.json"%n)); print(n, round(d["value"],1), round(d["ms_per_step"],2), d["final_loss"])
a=json.load(open('gpurun_out/r3_sa2_iso_off.json'))['rows']; d=json.load(open('gpurun_out/r3_sa2_iso_on.json'))['rows']
for x,y in zip(a,d):
    if 'fprop_tf' in x: print(x['shape'], 'fprop %.0f -> %.0f'%(x['fprop_tf'],y['fprop_tf']), 'dgrad %.0f -> %.0f'%(x['dgrad_tf'],y['dgrad_tf']))
PY
