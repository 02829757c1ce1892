set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof5
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof5 -- python3 $R/bench.py --batch-per-gpu 256 --steps 5 --warmup 2 > $R/gpurun_out/prof5.log 2>&1
echo profile done
