set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r01g; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
python3 $R/bench.py --steps 60 --warmup 6 > $O/bench_eval_graph_bf16.json 2>$O/err1.txt
python3 $R/bench.py --steps 30 --warmup 5 --in-flight 1 --no-cpu-baseline > $O/bench_eval_graph_bf16_one_in_flight.json 2>/dev/null
python3 $R/bench.py --steps 20 --warmup 5 --dtype f32 --no-cpu-baseline > $O/bench_eval_graph_f32.json 2>/dev/null
python3 $R/bench.py --workload train --steps 20 --warmup 3 > $O/bench_train_bf16.json 2>/dev/null
python3 $R/bench.py --workload retrieval --steps 3 --warmup 1 > $O/bench_retrieval_1gpu.json 2>/dev/null
rm -rf /tmp/p1; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --launch eager --in-flight 1 > $O/bench_eval_under_rocprof.json 2>/dev/null
cp $(find /tmp/p1 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_eval_eager_bf16.csv; python3 $R/tools/trace_summary.py $(find /tmp/p1 -name "*kernel_trace.csv" | head -1) 17 40 > $O/eval_trace_summary.txt
rm -rf /tmp/p2; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p2 -- python3 $R/bench.py --workload train --steps 10 --warmup 3 > $O/bench_train_under_rocprof.json 2>/dev/null
cp $(find /tmp/p2 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_train_bf16.csv; python3 $R/tools/trace_summary.py $(find /tmp/p2 -name "*kernel_trace.csv" | head -1) 15 50 > $O/train_trace_summary.txt
rm -rf /tmp/p3; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p3 -- python3 $R/bench.py --workload retrieval --steps 2 --warmup 1 > $O/bench_retrieval_under_rocprof.json 2>/dev/null
cp $(find /tmp/p3 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_retrieval_bf16.csv
rm -rf /tmp/p4; rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/p4 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --launch eager --in-flight 1 > /dev/null 2>&1
rm -rf /tmp/p5; rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/p5 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --launch eager --in-flight 1 > /dev/null 2>&1
python3 $R/tools/pmc_summary.py $(find /tmp/p4 -name "*counter_collection.csv" | head -1) $(find /tmp/p5 -name "*counter_collection.csv" | head -1) $O/pmc_summary.json
rm -rf /tmp/p6; rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/p6 -- python3 $R/tools/xpool_only.py 8192 256 > /dev/null 2>&1
rm -rf /tmp/p7; rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/p7 -- python3 $R/tools/xpool_only.py 8192 256 > /dev/null 2>&1
python3 $R/tools/pmc_summary.py $(find /tmp/p6 -name "*counter_collection.csv" | head -1) $(find /tmp/p7 -name "*counter_collection.csv" | head -1) $O/pmc_xpool_fused_8192x256.json
ls -la $O
