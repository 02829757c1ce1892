# round 3: hardware counters per kernel at batch 256 (three --pmc passes, never mixed with API traces) and a
# determinism probe of the final build
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3pmc
mkdir -p $O
rm -rf $O/pmc_a $O/pmc_b $O/pmc_c
PMC_BATCH=256 timeout -s KILL 200 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/pmc_a -- python3 $R/bench/pmc_kernels.py > $O/pmc_a.log 2>&1
PMC_BATCH=256 timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_b -- python3 $R/bench/pmc_kernels.py > $O/pmc_b.log 2>&1
PMC_BATCH=256 timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_c -- python3 $R/bench/pmc_kernels.py > $O/pmc_c.log 2>&1
cd $R
python3 bench/summarize_pmc.py $O/pmc_a $O/pmc_b $O/pmc_c > $O/81_pmc_b256_round3.md
rm -rf $O/pmc_a $O/pmc_b $O/pmc_c
timeout -k 10 500 python bench/determinism_probe.py --trials 60 --steps 3 > $O/82_determinism_probe_round3.log 2>&1
tail -3 $O/82_determinism_probe_round3.log
head -40 $O/81_pmc_b256_round3.md
