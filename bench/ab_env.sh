# Same-box A/B of an environment switch: alternating bench.py runs.
#   bash bench/ab_env.sh VAR A B [rounds] [bench.py args...]
VAR=$1; A=$2; B=$3; R=${4:-3}; shift; shift; shift; shift || true
for i in $(seq 1 $R); do
  for v in $A $B; do
    env $VAR=$v timeout -k 10 300 python bench.py --steps 20 --warmup 5 "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'$VAR':'$v','img_s':round(d['value'],1),'ms':round(d['ms_per_step'],3)}))" || exit 1
  done
done
