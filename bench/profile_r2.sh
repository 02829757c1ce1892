# round-2 profiles: full step at b1024 (plain and BatchNorm variants) + the three PMC passes
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r2
rm -rf $R/gpurun_out/r2/prof_b1024 $R/gpurun_out/r2/prof_bn $R/gpurun_out/r2/pmc_a $R/gpurun_out/r2/pmc_b $R/gpurun_out/r2/pmc_c
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2/prof_b1024 -- python3 $R/bench.py --steps 3 --warmup 2 > $R/gpurun_out/r2/prof_b1024.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2/prof_bn -- python3 $R/bench.py --batch-norm --fp8-classifier --steps 3 --warmup 2 > $R/gpurun_out/r2/prof_bn.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $R/gpurun_out/r2/pmc_a -- python3 $R/bench/pmc_kernels.py > $R/gpurun_out/r2/pmc_a.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/r2/pmc_b -- python3 $R/bench/pmc_kernels.py > $R/gpurun_out/r2/pmc_b.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $R/gpurun_out/r2/pmc_c -- python3 $R/bench/pmc_kernels.py > $R/gpurun_out/r2/pmc_c.log 2>&1
cd $R
python3 bench/stats_to_md.py gpurun_out/r2/prof_b1024 "rocprofv3 --kernel-trace --stats, bench.py --steps 3 --warmup 2 at per-GPU batch 1024 (1 MI355X), round 2" > gpurun_out/r2/23_full_step_b1024.md
python3 bench/stats_to_md.py gpurun_out/r2/prof_bn "rocprofv3 --kernel-trace --stats, bench.py --batch-norm --fp8-classifier --steps 3 --warmup 2 at per-GPU batch 1024 (BASELINE.json config 5)" > gpurun_out/r2/24_full_step_bn_fp8_b1024.md
python3 bench/summarize_pmc.py gpurun_out/r2/pmc_a gpurun_out/r2/pmc_b gpurun_out/r2/pmc_c > gpurun_out/r2/25_pmc_b64.md
# keep the merged-back payload small
find gpurun_out/r2/prof_b1024 gpurun_out/r2/prof_bn -name "*kernel_trace.csv" -delete
rm -rf gpurun_out/r2/pmc_a gpurun_out/r2/pmc_b gpurun_out/r2/pmc_c
echo profiles done
