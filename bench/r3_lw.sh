set -e
timeout -k 10 400 python -u -m pytest tests/test_gpu_model.py -x -q --timeout 300 --timeout-method thread -k "layerwise" > gpurun_out/r3_lw.log 2>&1 || { tail -60 gpurun_out/r3_lw.log; exit 1; }
tail -3 gpurun_out/r3_lw.log
