# round 3 re-entry: GPU suite + default bench + small-batch bench on the restored tree
mkdir -p gpurun_out/r3state
O=gpurun_out/r3state
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/gpu_suite.log 2>&1
rc=$?
tail -8 $O/gpu_suite.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python bench.py --steps 20 --warmup 5 > $O/b1024.json 2> $O/err.log || exit 1
cat $O/b1024.json
timeout -k 10 120 python bench.py --batch-per-gpu 4 --steps 200 --warmup 20 > $O/b4.json 2>> $O/err.log || exit 1
cat $O/b4.json
