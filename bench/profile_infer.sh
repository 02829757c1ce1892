set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/infer_prof
rm -rf $O && mkdir -p $O
export PYTHONPATH=$R/bench:$R
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/plain -- python3 $R/bench/bench_infer.py --batch 64 --steps 10 --warmup 3 > $O/plain.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bn -- python3 $R/bench/bench_infer.py --batch 64 --steps 10 --warmup 3 --batch-norm > $O/bn.log 2>&1
cd $R
python3 bench/stats_to_md.py gpurun_out/infer_prof/plain "inference b64 plain" > gpurun_out/infer_prof/plain.md
python3 bench/stats_to_md.py gpurun_out/infer_prof/bn "inference b64 bn" > gpurun_out/infer_prof/bn.md
find gpurun_out/infer_prof -name "*kernel_trace.csv" -delete
