mkdir -p gpurun_out
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r3_gpu_suite.log 2>&1
echo "exit $?"
tail -15 gpurun_out/r3_gpu_suite.log
