set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for o in 0 1; do
  rm -rf $R/gpurun_out/pmc_k$o
  DTP_IGEMM_KORDER=$o timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_k$o -- python3 $R/bench/pmc_kernels.py > $R/gpurun_out/pmc_k$o.log 2>&1
  rm -rf $R/gpurun_out/pmc_m$o
  DTP_IGEMM_KORDER=$o timeout -s KILL 120 rocprofv3 --pmc TCC_MISS_sum WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_m$o -- python3 $R/bench/pmc_kernels.py > $R/gpurun_out/pmc_m$o.log 2>&1
done
echo pmc done
