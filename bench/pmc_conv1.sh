# LDS bank-conflict share of the conv1 kernels alone (one SQ pass over bench/bench_conv1.py at batch 64)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_conv1
rm -rf $O
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O -- python3 $R/bench/bench_conv1.py --batch 64 --iters 2 > $O.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_conv1/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, m in acc.items():
    if "conv1" in k:
        print(k, {c: int(v) for c, v in m.items()}, "lds_conf%%=%.1f" % (100 * m["SQ_LDS_BANK_CONFLICT"] / max(1, m["SQ_LDS_IDX_ACTIVE"])))
PY
rm -rf $O
