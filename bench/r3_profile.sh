set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3prof
mkdir -p $O
rm -rf $O/prof_b1024
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b1024 -- python3 $R/bench.py --steps 3 --warmup 2 > $O/prof_b1024.log 2>&1
cd $R
python3 bench/stats_to_md.py gpurun_out/r3prof/prof_b1024 "rocprofv3 --kernel-trace --stats, bench.py --steps 3 --warmup 2 at per-GPU batch 1024 (1 MI355X), round 3 after cfg 18/19" > gpurun_out/r3prof/70_full_step_b1024.md
python3 bench/trace_timeline.py gpurun_out/r3prof/prof_b1024 > gpurun_out/r3prof/71_step_timeline_b1024.md || true
find gpurun_out/r3prof -name "*kernel_trace.csv" -delete
find gpurun_out/r3prof -name "*.db" -delete
cat gpurun_out/r3prof/70_full_step_b1024.md | head -30
