# round 3 (re-entry): full-step kernel stats + timeline at b1024, kernel stats at b4 (hipGraph off so
# that kernels are attributed), per-layer isolated rates at b256
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3prof2
mkdir -p $O
rm -rf $O/prof_b1024 $O/prof_b4
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b1024 -- python3 $R/bench.py --steps 3 --warmup 2 > $O/prof_b1024.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b4 -- python3 $R/bench.py --batch-per-gpu 4 --hip-graph off --steps 40 --warmup 10 > $O/prof_b4.log 2>&1
cd $R
python3 bench/stats_to_md.py $O/prof_b1024 "rocprofv3 --kernel-trace --stats, bench.py --steps 3 --warmup 2 at per-GPU batch 1024 (1 MI355X), round 3 (shared-activation-image igemm cfg 18/19, pool fused into the igemm cfg 20/21)" > $O/70_full_step_b1024_round3.md
python3 bench/trace_timeline.py $O/prof_b1024 > $O/71_step_timeline_b1024_round3.md || true
TOP=60 python3 bench/stats_to_md.py $O/prof_b4 "rocprofv3 --kernel-trace --stats, bench.py --batch-per-gpu 4 --hip-graph off --steps 40 --warmup 10 (50 steps in the trace), round 3" > $O/72_full_step_b4_round3.md
find $O -name "*kernel_trace.csv" -delete
find $O -name "*.db" -delete
timeout -k 10 400 python3 bench/bench_kernels.py --batch 256 --no-stock --iters 40 > $O/73_per_layer_isolated_b256_round3.jsonl 2> $O/per_layer.err
echo profiles done
head -40 $O/70_full_step_b1024_round3.md
