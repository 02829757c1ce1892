# PMC pass A (SQ counters) at batch 256: at batch 64 the 256x256-tile kernels run 1.5-3 rounds of tiles
# (a partial last round costs up to 25 % of the kernel), which understates the matrix-pipe share
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2b
mkdir -p $O
rm -rf $O/pmc_a256 $O/pmc_b256
PMC_BATCH=256 timeout -s KILL 200 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/pmc_a256 -- python3 $R/bench/pmc_kernels.py > $O/pmc_a256.log 2>&1
PMC_BATCH=256 timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_b256 -- python3 $R/bench/pmc_kernels.py > $O/pmc_b256.log 2>&1
cd $R
python3 bench/summarize_pmc.py gpurun_out/r2b/pmc_a256 gpurun_out/r2b/pmc_b256 > gpurun_out/r2b/45_pmc_b256.md
rm -rf gpurun_out/r2b/pmc_a256 gpurun_out/r2b/pmc_b256
echo done
