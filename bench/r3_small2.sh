set -e
O=gpurun_out/r3small
mkdir -p $O
timeout -k 10 300 python -u -m pytest tests/test_gpu_model.py -x -q --timeout 300 --timeout-method thread -k "hip_graph or per_bucket or training_reduces or dropout_rng" > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for b in 4 16; do
timeout -k 10 200 python bench.py --batch-per-gpu $b --steps 200 --warmup 20 --hip-graph off --no-bucket-optimizer > $O/b${b}_eager_plain.json 2>> $O/err.log
timeout -k 10 200 python bench.py --batch-per-gpu $b --steps 200 --warmup 20 --hip-graph off > $O/b${b}_eager_overlap.json 2>> $O/err.log
timeout -k 10 200 python bench.py --batch-per-gpu $b --steps 200 --warmup 20 --hip-graph on --no-bucket-optimizer > $O/b${b}_graph_plain.json 2>> $O/err.log
timeout -k 10 200 python bench.py --batch-per-gpu $b --steps 200 --warmup 20 > $O/b${b}_graph_overlap.json 2>> $O/err.log
done
for i in 1 2; do
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-bucket-optimizer > $O/b1024_plain_$i.json 2>> $O/err.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/b1024_overlap_$i.json 2>> $O/err.log
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3small/b*_*.json")):
    if 'bench_' in f: continue
    d=json.load(open(f)); print(f.split('/')[-1], round(d["value"],1), round(d["ms_per_step"],3), d.get("hip_graph"), round(d["final_loss"],5))
PY
