set -e
mkdir -p gpurun_out/r3opt
for cap in 256 512 1024 4096; do
DTP_OPT_GRID_CAP=$cap timeout -k 10 120 python bench/bench_optim.py >> gpurun_out/r3opt/optim.jsonl
done
cat gpurun_out/r3opt/optim.jsonl
