set -e
O=gpurun_out/r3wg8
mkdir -p $O
for arm in 0 1 0 1; do
DTP_WGRAD_8PHASE=$arm timeout -k 10 400 python3 bench/bench_kernels.py --batch 256 --no-stock --iters 40 > $O/pl_$arm.jsonl 2> $O/per_layer.err
python - $arm <<'PY'
import json, sys
a=[json.loads(l) for l in open("gpurun_out/r3wg8/pl_%s.jsonl" % sys.argv[1])]
print("8phase", sys.argv[1], " ".join("%s:%.0f" % (x["shape"], x["wgrad_tf"]) for x in a if "wgrad_tf" in x), " ".join("%s:%.3f" % (x["shape"][:14], x["wgrad_ms"]) for x in a if "wgrad_ms" in x))
PY
done
bash bench/ab_env.sh DTP_WGRAD_8PHASE 0 1 3
