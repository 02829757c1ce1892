# generic same-box A/B of one environment variable on the default bench: bash bench/r3_ab.sh VAR A B [extra bench args]
V=$1; A=$2; B=$3; shift 3
mkdir -p gpurun_out/ab
O=gpurun_out/ab
for i in 1 2 3; do
  env $V=$A timeout -k 10 200 python bench.py --steps 20 --warmup 5 "$@" > $O/a_$i.json 2>> $O/err.log || exit 1
  env $V=$B timeout -k 10 200 python bench.py --steps 20 --warmup 5 "$@" > $O/b_$i.json 2>> $O/err.log || exit 1
done
python - $V $A $B <<'PY'
import json, sys
for i in "123":
    for n, val in (("a", sys.argv[2]), ("b", sys.argv[3])):
        d = json.load(open("gpurun_out/ab/%s_%s.json" % (n, i))); print(sys.argv[1] + "=" + val, round(d["value"], 1), "img/s", round(d["ms_per_step"], 3), "ms loss", d["final_loss"])
PY
