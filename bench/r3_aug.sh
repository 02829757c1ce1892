set -e
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_gpu_kernels.py -x -q --timeout 200 --timeout-method thread -k "device_augmentation" > gpurun_out/r3_aug_tests.log 2>&1 || { tail -40 gpurun_out/r3_aug_tests.log; exit 1; }
tail -2 gpurun_out/r3_aug_tests.log
timeout -k 10 500 python bench/bench_image_folder_cache.py --per-class 2048 --epochs 5 > gpurun_out/r3_aug_bench.log 2>&1 || { tail -40 gpurun_out/r3_aug_bench.log; exit 1; }
tail -2 gpurun_out/r3_aug_bench.log
timeout -k 10 400 python bench/measure_grad_parity.py > gpurun_out/r3_grad_parity.jsonl 2> gpurun_out/r3_grad_parity.err || { tail -20 gpurun_out/r3_grad_parity.err; exit 1; }
cat gpurun_out/r3_grad_parity.jsonl
