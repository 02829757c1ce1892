# Round 3: cfg 18 (one staged activation image per tap row) — numerics, isolated A/B, full-step A/B.
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_kernels.py -x -q --timeout 300 --timeout-method thread -k "shared_activation or 8phase or fused_bias_grad or bitmask or statistics_from" > gpurun_out/r3_sa_tests.log 2>&1 || { tail -40 gpurun_out/r3_sa_tests.log; exit 1; }
tail -3 gpurun_out/r3_sa_tests.log
DTP_IGEMM_SHARED_A=0 timeout -k 10 240 python bench/bench_kernels.py --batch 256 --no-stock --iters 8 --out gpurun_out/r3_sa_iso_off.json > gpurun_out/r3_sa_iso_off.log 2>&1
timeout -k 10 240 python bench/bench_kernels.py --batch 256 --no-stock --iters 8 --out gpurun_out/r3_sa_iso_on.json > gpurun_out/r3_sa_iso_on.log 2>&1
for i in 1 2; do
DTP_IGEMM_SHARED_A=0 timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r3_sa_bench_off_$i.json 2>> gpurun_out/r3_sa_bench.err
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r3_sa_bench_on_$i.json 2>> gpurun_out/r3_sa_bench.err
done
python - <<'PY'
import json
for n in ("off_1","on_1","off_2","on_2"):
    d=json.load(open("gpurun_out/r3_sa_bench_%s.json"%n)); print(n, round(d["value"],1), round(d["ms_per_step"],2), d["ms_first5"], d["ms_last5"], d["final_loss"])
PY
