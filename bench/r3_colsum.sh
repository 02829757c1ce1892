set -e
timeout -k 10 600 python -u -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -x -q --timeout 300 --timeout-method thread -k "colsum or bias_grad or unpool or forward_backward or layerwise" > gpurun_out/colsum_tests.log 2>&1 || { tail -20 gpurun_out/colsum_tests.log; exit 1; }
tail -2 gpurun_out/colsum_tests.log
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3prof5
mkdir -p $O
rm -rf $O/prof_b1024
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b1024 -- python3 $R/bench.py --steps 3 --warmup 2 > $O/prof_b1024.log 2>&1
cd $R
python3 bench/stats_to_md.py $O/prof_b1024 "colsum pass 1 vectorised" > $O/stats.md
find $O -name "*kernel_trace.csv" -delete
find $O -name "*.db" -delete
grep -i "colsum" $O/stats.md
for i in 1 2 3; do timeout -k 10 200 python bench.py --steps 20 --warmup 5 > $O/bench_$i.json; done
python -c "
import json
for i in '123':
    d=json.load(open('gpurun_out/r3prof5/bench_%s.json'%i)); print(round(d['value'],1), d['ms_per_step'])"
