# config 5 (VGG16-BN + fp8 classifier) at b1024: kernel trace -> one step as a timeline
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/bn_tl
rm -rf $O && mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --batch-norm --fp8-classifier --steps 3 --warmup 2 > $O/prof.log 2>&1
cd $R
python3 bench/trace_timeline.py gpurun_out/bn_tl/prof > gpurun_out/bn_tl/timeline.md
find gpurun_out/bn_tl -name "*kernel_trace.csv" -delete
echo done
