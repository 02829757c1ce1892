# Same-box A/B of two builds of the kernel library: alternating bench.py runs.
#   bash bench/ab_bench.sh <base.so> [rounds] [bench.py args...]   -> gpurun_out/ab_bench.jsonl
set -e
BASE=$1; ROUNDS=${2:-3}; shift; shift || true
OUT=gpurun_out/ab_bench.jsonl
: > $OUT
for i in $(seq 1 $ROUNDS); do
  DTP_HIP_LIB=$BASE timeout -k 10 300 python bench.py --steps 20 --warmup 5 "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'arm':'base','img_s':round(d['value'],1),'ms':round(d['ms_per_step'],3)}))" >> $OUT
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'arm':'new','img_s':round(d['value'],1),'ms':round(d['ms_per_step'],3)}))" >> $OUT
done
cat $OUT
