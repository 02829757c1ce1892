set -e
O=gpurun_out/r3small3
mkdir -p $O
timeout -k 10 300 python -u -m pytest tests/test_gpu_kernels.py -x -q --timeout 120 --timeout-method thread -k "colsum or linear" > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for b in 4 16 64; do
timeout -k 10 200 python bench.py --batch-per-gpu $b --steps 200 --warmup 20 > $O/b${b}.json 2>> $O/err.log
done
timeout -k 10 200 python bench.py --batch-per-gpu 256 --steps 40 --warmup 10 > $O/b256.json 2>> $O/err.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/b1024.json 2>> $O/err.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_b4 -o b4 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --batch-per-gpu 4 --steps 40 --warmup 10 > $GRAFT_REPO_ROOT/$O/prof_b4.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3small3/b*.json")):
    d=json.load(open(f)); print(f.split('/')[-1], round(d["value"],1), round(d["ms_per_step"],3), d.get("hip_graph"), round(d["final_loss"],5))
PY
ls $O/prof_b4 | head
