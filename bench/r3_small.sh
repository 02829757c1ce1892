set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3small
mkdir -p $O
rm -rf $O/prof_b4
timeout -k 10 200 python3 $R/bench.py --batch-per-gpu 4 --steps 200 --warmup 20 > $O/bench_b4.json 2> $O/bench_b4.err
timeout -k 10 200 python3 $R/bench.py --batch-per-gpu 16 --steps 200 --warmup 20 > $O/bench_b16.json 2>> $O/bench_b4.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b4 -- python3 $R/bench.py --batch-per-gpu 4 --steps 40 --warmup 10 > $O/prof_b4.log 2>&1
cd $R
python3 bench/stats_to_md.py gpurun_out/r3small/prof_b4 "rocprofv3 --kernel-trace --stats, bench.py --batch-per-gpu 4 --steps 40 --warmup 10 (round 3)" > gpurun_out/r3small/b4_stats.md
python3 bench/trace_timeline.py gpurun_out/r3small/prof_b4 > gpurun_out/r3small/b4_timeline.md || true
find gpurun_out/r3small -name "*kernel_trace.csv" -delete
find gpurun_out/r3small -name "*.db" -delete
python3 -c "
import json
for n in ('b4','b16'):
    d=json.load(open('gpurun_out/r3small/bench_%s.json'%n)); print(n, round(d['value'],1), round(d['ms_per_step'],3))
"
head -12 gpurun_out/r3small/b4_timeline.md
