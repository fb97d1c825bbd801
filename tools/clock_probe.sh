# samples the GPU's clocks and power while the retrieval leg runs (is the 53 k x 4 k launch clock- or power-limited?)
MADE_BENCH_RETRIEVAL_512=0 python3 ${GRAFT_REPO_ROOT:-.}/bench.py --workload retrieval --no-cpu-baseline --steps 300 > /tmp/bench_clk.json 2>/dev/null &
BP=$!
for i in $(seq 1 40); do rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Power (W)" | sed 's/.*: //' | tr '\n' ' '; echo; sleep 0.7; kill -0 $BP 2>/dev/null || break; done
wait $BP
python -c "import json; d=json.load(open('/tmp/bench_clk.json')); print('ms_per_step', d['ms_per_step'])"
