R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r01f; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/p4; rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/p4 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --launch eager > /dev/null 2>&1
rm -rf /tmp/p5; rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/p5 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --launch eager > /dev/null 2>&1
python3 $R/tools/pmc_summary.py $(find /tmp/p4 -name "*counter_collection.csv" | head -1) $(find /tmp/p5 -name "*counter_collection.csv" | head -1) $O/pmc_summary.json > /dev/null
cp $O/pmc_summary.json $R/profiles/r01_pmc_summary.json
python3 $R/bench.py --steps 30 --warmup 5 > $O/bench_eval_graph_bf16.json 2>/dev/null
rm -rf /tmp/p1; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --launch eager > $O/bench_eval_under_rocprof.json 2>/dev/null
cp $(find /tmp/p1 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_eval_eager_bf16.csv; python3 $R/tools/trace_summary.py $(find /tmp/p1 -name "*kernel_trace.csv" | head -1) 17 40 > $O/eval_trace_summary.txt
python3 -c "
import json
for f in ('bench_eval_graph_bf16','bench_eval_under_rocprof'):
    d=json.loads(open('$O/'+f+'.json').read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d['roofline'])"
grep 'linear_glds_kernel<1, false, 64>' $O/eval_trace_summary.txt | head -3
