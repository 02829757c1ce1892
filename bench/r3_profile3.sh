# round 3, final build: full-step kernel stats + timeline at b1024, per-layer isolated rates at b256
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3prof3
mkdir -p $O
rm -rf $O/prof_b1024
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b1024 -- python3 $R/bench.py --steps 3 --warmup 2 > $O/prof_b1024.log 2>&1
cd $R
python3 bench/stats_to_md.py $O/prof_b1024 "rocprofv3 --kernel-trace --stats, bench.py --steps 3 --warmup 2 at per-GPU batch 1024 (1 MI355X), round 3 final build (wgrad_tt8, image swizzle, merged phases)" > $O/87_full_step_b1024_final.md
MIN_US=250 python3 bench/trace_timeline.py $O/prof_b1024 > $O/88_step_timeline_b1024_final.md
find $O -name "*kernel_trace.csv" -delete
find $O -name "*.db" -delete
timeout -k 10 400 python3 bench/bench_kernels.py --batch 256 --no-stock --iters 40 > $O/89_per_layer_isolated_b256_final.jsonl 2> $O/per_layer.err
timeout -k 10 200 python bench.py --steps 20 --warmup 5 > $O/bench.json
head -12 $O/88_step_timeline_b1024_final.md
python -c "
import json; d=json.load(open('gpurun_out/r3prof3/bench.json')); print(round(d['value'],1), d['ms_per_step'])"
