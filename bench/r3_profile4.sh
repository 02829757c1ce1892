# last session: full-step kernel stats at b1024 with the un-pooling data gradient
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3prof4
mkdir -p $O
rm -rf $O/prof_b1024
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b1024 -- python3 $R/bench.py --steps 3 --warmup 2 > $O/prof_b1024.log 2>&1
cd $R
python3 bench/stats_to_md.py $O/prof_b1024 "rocprofv3 --kernel-trace --stats, bench.py --steps 3 --warmup 2 at per-GPU batch 1024 (1 MI355X), last session of round 3 (un-pooling data gradient)" > $O/100_full_step_b1024_unpool.md
MIN_US=0 python3 bench/trace_timeline.py $O/prof_b1024 > $O/timeline_all.md
find $O -name "*kernel_trace.csv" -delete
find $O -name "*.db" -delete
head -45 $O/100_full_step_b1024_unpool.md | cut -c1-140
