# SQ-counter pass only (batch 256) — LDS conflict share / matrix-pipe share per kernel
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3pmc
mkdir -p $O
rm -rf $O/pmc_a
PMC_BATCH=256 timeout -s KILL 200 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/pmc_a -- python3 $R/bench/pmc_kernels.py > $O/pmc_a.log 2>&1
cd $R
python3 bench/summarize_pmc.py $O/pmc_a > $O/pmc_a_only.md
rm -rf $O/pmc_a
grep "igemm_nt\|kernel |" $O/pmc_a_only.md
