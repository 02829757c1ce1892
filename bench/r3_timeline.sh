# round 3: one-step timelines (b1024 eager, b4 under the whole-step hipGraph) from rocprofv3 kernel traces
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3tl
mkdir -p $O
rm -rf $O/prof_b1024 $O/prof_b4g
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof_b1024 -- python3 $R/bench.py --steps 3 --warmup 2 > $O/prof_b1024.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof_b4g -- python3 $R/bench.py --batch-per-gpu 4 --steps 20 --warmup 5 > $O/prof_b4g.log 2>&1
cd $R
MIN_US=250 python3 bench/trace_timeline.py $O/prof_b1024 > $O/71_step_timeline_b1024_round3.md
MIN_US=0 python3 bench/trace_timeline.py $O/prof_b4g > $O/76_step_timeline_b4_hipgraph_round3.md
find $O -name "*kernel_trace.csv" -delete
find $O -name "*.db" -delete
echo timelines done
