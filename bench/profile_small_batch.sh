# per-kernel time of the full step at the reference's own per-GPU batch (4) and at 16
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/small
mkdir -p $O
for b in 4 16; do
  rm -rf $O/prof_b$b
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b$b -- python3 $R/bench.py --batch-per-gpu $b --steps 40 --warmup 10 > $O/prof_b$b.log 2>&1
done
cd $R
for b in 4 16; do
  python3 bench/stats_to_md.py gpurun_out/small/prof_b$b "rocprofv3 --kernel-trace --stats, bench.py --batch-per-gpu $b --steps 40 --warmup 10" > gpurun_out/small/50_full_step_b$b.md
  python3 bench/trace_timeline.py gpurun_out/small/prof_b$b > gpurun_out/small/51_timeline_b$b.md || true
done
find gpurun_out/small -name "*kernel_trace.csv" -delete
echo done
