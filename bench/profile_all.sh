set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof4 -- python3 $R/bench.py --batch-per-gpu 256 --steps 5 --warmup 2 > $R/gpurun_out/prof4.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $R/gpurun_out/pmc4a -- python3 $R/bench/pmc_kernels.py > $R/gpurun_out/pmc4a.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc4b -- python3 $R/bench/pmc_kernels.py > $R/gpurun_out/pmc4b.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $R/gpurun_out/pmc4c -- python3 $R/bench/pmc_kernels.py > $R/gpurun_out/pmc4c.log 2>&1
echo profiles done
