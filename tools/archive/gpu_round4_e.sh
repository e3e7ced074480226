ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof4e; rm -rf $OUT; mkdir -p $OUT
export MSDP_NO_GRAPH=1
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/theta -- python3 $ROOT/tools/theta_probe.py 32 > $OUT/theta.log 2>&1
cd $ROOT
f=$(find $OUT/theta -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -6 "$f" | cut -c1-200
