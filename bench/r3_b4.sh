# batch-4 iteration loop: new tests, bench at b4 (hipGraph auto), one-step timeline under the profiler
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3b4
mkdir -p $O
if [ -n "$TESTS" ]; then
timeout -k 10 300 python -u -m pytest $TESTS -x -q --timeout 120 --timeout-method thread > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
fi
timeout -k 10 120 python bench.py --batch-per-gpu 4 --steps 200 --warmup 20 > $O/b4.json 2> $O/err.log
timeout -k 10 120 python bench.py --batch-per-gpu 16 --steps 200 --warmup 20 > $O/b16.json 2>> $O/err.log
python - <<'PY'
import json
for b in (4, 16):
    d = json.load(open("gpurun_out/r3b4/b%d.json" % b)); print("b%d" % b, round(d["value"], 1), "img/s", round(d["ms_per_step"], 3), "ms graph", d["hip_graph"], "loss", d["final_loss"])
PY
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_b4g
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof_b4g -- python3 $R/bench.py --batch-per-gpu 4 --steps 20 --warmup 5 > $O/prof_b4g.log 2>&1
cd $R
MIN_US=0 python3 bench/trace_timeline.py $O/prof_b4g > $O/timeline_b4.md
find $O -name "*kernel_trace.csv" -delete
find $O -name "*.db" -delete
head -6 $O/timeline_b4.md
