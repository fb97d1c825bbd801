# Vendor-GEMM yardstick:  gpurun -- 'bash tools/gemm_yardstick.sh <tag>'
# -> gpurun_out/<tag>/gemm_yardstick.txt (default BLAS), gemm_yardstick_hipblaslt.txt (TORCH_BLAS_PREFER_HIPBLASLT=1) and the
#    rocprofv3 kernel stats of the default run (vendor kernel symbols = their tile configurations).
TAG=${1:-r05_yard}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
python3 $R/tools/gemm_yardstick.py > $O/gemm_yardstick.txt 2>$O/gemm_yardstick.err
export TORCH_BLAS_PREFER_HIPBLASLT=1
python3 $R/tools/gemm_yardstick.py > $O/gemm_yardstick_hipblaslt.txt 2>$O/gemm_yardstick_hipblaslt.err
unset TORCH_BLAS_PREFER_HIPBLASLT
export ROUNDS=2 ITERS=10
rm -rf /tmp/y1; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/y1 -- python3 $R/tools/gemm_yardstick.py > /dev/null 2>&1
cp $(find /tmp/y1 -name "*kernel_stats.csv" | head -1) $O/gemm_yardstick_kernel_stats.csv
python3 $R/tools/trace_summary.py $(find /tmp/y1 -name "*kernel_trace.csv" | head -1) 1 60 > $O/gemm_yardstick_trace_summary.txt 2>&1
tail -5 $O/gemm_yardstick.err; cat $O/gemm_yardstick.txt
