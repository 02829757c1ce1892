# Round 3: upper bound of "one staged activation image per tap row" (diagnostic build skips the dx=+-1 A loads).
set -e
mkdir -p gpurun_out
timeout -k 10 240 python bench/bench_kernels.py --batch 256 --no-stock --iters 8 --out gpurun_out/r3_iso_base.json > gpurun_out/r3_iso_base.log 2>&1
DTP_HIP_LIB=distributed_training_pytorch_amd/lib/libdtp_hip_diag.so timeout -k 10 240 python bench/bench_kernels.py --batch 256 --no-stock --iters 8 --out gpurun_out/r3_iso_diag.json > gpurun_out/r3_iso_diag.log 2>&1
timeout -k 10 240 python bench/bench_kernels.py --batch 256 --no-stock --iters 8 --out gpurun_out/r3_iso_base2.json > gpurun_out/r3_iso_base2.log 2>&1
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r3_bench_base.json 2> gpurun_out/r3_bench_base.err
DTP_HIP_LIB=distributed_training_pytorch_amd/lib/libdtp_hip_diag.so timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r3_bench_diag.json 2> gpurun_out/r3_bench_diag.err
cat gpurun_out/r3_bench_base.json gpurun_out/r3_bench_diag.json
