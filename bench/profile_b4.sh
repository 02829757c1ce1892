set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_b4
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_b4 -- python3 $R/bench.py --steps 20 --warmup 5 --batch-per-gpu 4 > $R/gpurun_out/prof_b4.log 2>&1
echo profile done
