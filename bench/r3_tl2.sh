# one-step timeline at batch 1024 (eager) from a rocprofv3 kernel trace
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3tl2
mkdir -p $O
rm -rf $O/prof_b1024
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof_b1024 -- python3 $R/bench.py --steps 3 --warmup 2 > $O/prof_b1024.log 2>&1
cd $R
MIN_US=250 python3 bench/trace_timeline.py $O/prof_b1024 > $O/timeline_b1024.md
find $O -name "*kernel_trace.csv" -delete
find $O -name "*.db" -delete
echo timeline done
