# diagnostic: shared-image igemm WITHOUT the zero-region select (wrong math at the w borders): how much of the
# remaining LDS conflict share and time is that select?
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3diag
mkdir -p $O
rm -rf $O/pmc_a
DTP_HIP_LIB=$R/bench/diag_libdtp_hip.so PMC_BATCH=256 timeout -s KILL 200 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/pmc_a -- python3 $R/bench/pmc_kernels.py > $O/pmc_a.log 2>&1
cd $R
python3 bench/summarize_pmc.py $O/pmc_a > $O/pmc_diag.md
rm -rf $O/pmc_a
grep "igemm_nt\|kernel |" $O/pmc_diag.md
timeout -k 10 400 python3 bench/bench_kernels.py --batch 256 --no-stock --iters 40 > $O/per_layer_new.jsonl 2> $O/per_layer.err
DTP_HIP_LIB=bench/diag_libdtp_hip.so timeout -k 10 400 python3 bench/bench_kernels.py --batch 256 --no-stock --iters 40 > $O/per_layer_diag.jsonl 2>> $O/per_layer.err
python - <<'PY'
import json
a=[json.loads(l) for l in open("gpurun_out/r3diag/per_layer_new.jsonl") if "fprop_tf" in l]
b=[json.loads(l) for l in open("gpurun_out/r3diag/per_layer_diag.jsonl") if "fprop_tf" in l]
for x,y in zip(a,b):
    print(x["shape"], "fprop %.0f -> diag %.0f (%+.1f%%)  dgrad %.0f -> %.0f (%+.1f%%)" % (x["fprop_tf"], y["fprop_tf"], 100*(y["fprop_tf"]/x["fprop_tf"]-1), x["dgrad_tf"], y["dgrad_tf"], 100*(y["dgrad_tf"]/x["dgrad_tf"]-1)))
PY
