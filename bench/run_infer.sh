set -e
export PYTHONPATH=bench:.
O=gpurun_out/infer.jsonl; : > $O
for b in 64 256 1024; do
  timeout -k 10 200 python bench/bench_infer.py --batch $b 2>/dev/null | tee -a $O
  timeout -k 10 200 python bench/bench_infer.py --batch $b --batch-norm 2>/dev/null | tee -a $O
done
timeout -k 10 200 python bench/bench_infer.py --batch 256 --uint8 2>/dev/null | tee -a $O
timeout -k 10 200 python bench/bench_infer.py --batch 256 --batch-norm --fp8-classifier 2>/dev/null | tee -a $O
for b in 64 256; do
  timeout -k 10 500 python bench/bench_infer.py --batch $b --stock 2>/dev/null | tee -a $O
  timeout -k 10 500 python bench/bench_infer.py --batch $b --stock --batch-norm 2>/dev/null | tee -a $O
done
