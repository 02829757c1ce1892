# round-2 (second half) profiles: full step at b1024 (plain and BN+fp8), step timeline, the three PMC passes
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2b
mkdir -p $O
rm -rf $O/prof_b1024 $O/prof_bn $O/pmc_a $O/pmc_b $O/pmc_c
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b1024 -- python3 $R/bench.py --steps 3 --warmup 2 > $O/prof_b1024.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bn -- python3 $R/bench.py --batch-norm --fp8-classifier --steps 3 --warmup 2 > $O/prof_bn.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/pmc_a -- python3 $R/bench/pmc_kernels.py > $O/pmc_a.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_b -- python3 $R/bench/pmc_kernels.py > $O/pmc_b.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_c -- python3 $R/bench/pmc_kernels.py > $O/pmc_c.log 2>&1
cd $R
python3 bench/stats_to_md.py gpurun_out/r2b/prof_b1024 "rocprofv3 --kernel-trace --stats, bench.py --steps 3 --warmup 2 at per-GPU batch 1024 (1 MI355X), end of round 2" > gpurun_out/r2b/41_full_step_b1024.md
python3 bench/stats_to_md.py gpurun_out/r2b/prof_bn "rocprofv3 --kernel-trace --stats, bench.py --batch-norm --fp8-classifier --steps 3 --warmup 2 at per-GPU batch 1024 (BASELINE.json config 5), end of round 2" > gpurun_out/r2b/42_full_step_bn_fp8_b1024.md
python3 bench/trace_timeline.py gpurun_out/r2b/prof_b1024 > gpurun_out/r2b/43_step_timeline_b1024.md || true
python3 bench/summarize_pmc.py gpurun_out/r2b/pmc_a gpurun_out/r2b/pmc_b gpurun_out/r2b/pmc_c > gpurun_out/r2b/44_pmc_b64.md
find gpurun_out/r2b -name "*kernel_trace.csv" -delete
rm -rf gpurun_out/r2b/pmc_a gpurun_out/r2b/pmc_b gpurun_out/r2b/pmc_c
echo profiles done
