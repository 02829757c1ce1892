set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_b1024
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_b1024 -- python3 $R/bench.py --steps 3 --warmup 2 > $R/gpurun_out/prof_b1024.log 2>&1
echo profile done
