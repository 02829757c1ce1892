# round 3: the remaining evidence runs (batch-4 kernel stats + replay timeline of the final small-batch path,
# fresh-data / host-u8 / config-5 bench lines, inference throughput)
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_b4 $O/prof_b4g
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b4 -- python3 $R/bench.py --batch-per-gpu 4 --hip-graph off --steps 40 --warmup 10 > $O/prof_b4.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof_b4g -- python3 $R/bench.py --batch-per-gpu 4 --steps 20 --warmup 5 > $O/prof_b4g.log 2>&1
cd $R
TOP=60 python3 bench/stats_to_md.py $O/prof_b4 "rocprofv3 --kernel-trace --stats, bench.py --batch-per-gpu 4 --hip-graph off --steps 40 --warmup 10 (50 steps in the trace), round 3 after transpose_multi / bounded-grid optimizer" > $O/78_full_step_b4_final.md
MIN_US=0 python3 bench/trace_timeline.py $O/prof_b4g > $O/78_step_timeline_b4_hipgraph_final.md
find $O -name "*kernel_trace.csv" -delete
find $O -name "*.db" -delete
: > $O/79_bench_variants_round3.jsonl
timeout -k 10 200 python bench.py --steps 20 --warmup 5 >> $O/79_bench_variants_round3.jsonl 2> $O/err.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --fresh-data >> $O/79_bench_variants_round3.jsonl 2>> $O/err.log
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --data host-u8 >> $O/79_bench_variants_round3.jsonl 2>> $O/err.log
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --batch-norm --fp8-classifier >> $O/79_bench_variants_round3.jsonl 2>> $O/err.log
timeout -k 10 200 python bench.py --steps 40 --warmup 10 --batch-per-gpu 256 >> $O/79_bench_variants_round3.jsonl 2>> $O/err.log
timeout -k 10 200 python bench.py --steps 100 --warmup 10 --batch-per-gpu 64 >> $O/79_bench_variants_round3.jsonl 2>> $O/err.log
python - <<'PY'
import json
for l in open("gpurun_out/r3final/79_bench_variants_round3.jsonl"):
    d = json.loads(l); print(d["config"]["model"][:14], d["config"]["global_batch"], d["data"][:40], round(d["value"], 1), round(d["ms_per_step"], 3), d["ms_first5"], d["ms_last5"], d["hip_graph"])
PY
