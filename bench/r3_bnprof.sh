set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3bn
mkdir -p $O
rm -rf $O/prof_bn
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bn -- python3 $R/bench.py --batch-norm --fp8-classifier --steps 3 --warmup 2 > $O/prof_bn.log 2>&1
cd $R
python3 bench/stats_to_md.py $O/prof_bn "rocprofv3 --kernel-trace --stats, bench.py --batch-norm --fp8-classifier --steps 3 --warmup 2 at per-GPU batch 1024 (BASELINE.json config 5), round 3 final build" > $O/92_full_step_bn_fp8_b1024_round3.md
MIN_US=250 python3 bench/trace_timeline.py $O/prof_bn > $O/93_step_timeline_bn_round3.md || true
find $O -name "*kernel_trace.csv" -delete
find $O -name "*.db" -delete
head -30 $O/92_full_step_bn_fp8_b1024_round3.md
