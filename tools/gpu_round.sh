set -u
export TMPDIR=/tmp
REPO=$PWD
mkdir -p gpurun_out/gq
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/gq -o t -f csv -- python $REPO/bench.py --mode forward --steps 3 --warmup 1 --no-cpu-baseline > $REPO/gpurun_out/gq/log.txt 2>&1
grep -E "gemm_pre|split_rows|pk_kernel|lstm_persistent" $REPO/gpurun_out/gq/t_kernel_stats.csv | cut -d, -f1-4 | cut -c1-160
find $REPO/gpurun_out/gq -type f ! -name "*stats.csv" -delete
