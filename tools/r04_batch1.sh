# round 4, first GPU batch: co-run determinism probes of the decoder's backward chain (+ bisection builds), SQ counter passes, baseline bench line
O=gpurun_out/r04_a; mkdir -p $O
{
echo "### micro"; timeout 300 python tools/dec_corun_probe.py micro 2000
echo "### step, MADE_RET_SPLIT=0 (tape)"; MADE_RET_SPLIT=0 timeout 300 python tools/dec_corun_probe.py step 400
echo "### step, default (tape)"; timeout 300 python tools/dec_corun_probe.py step 400
echo "### step, MADE_RET_SPLIT=0, hipGraph"; MODE=graph MADE_RET_SPLIT=0 timeout 300 python tools/dec_corun_probe.py step 400
for v in dec_forcezero dec_shfl dec_o1; do
  echo "### step, MADE_RET_SPLIT=0, $v"; MADE_LIB_PATH=$PWD/tools/_ab/$v.so MADE_RET_SPLIT=0 timeout 300 python tools/dec_corun_probe.py step 400
done
} > $O/dec_corun_probe.txt 2>&1
grep -v amdgpu.ids $O/dec_corun_probe.txt
bash tools/pmc_sq_round4.sh r04_a > $O/pmc.log 2>&1
cat $O/sq_counters_attention.txt $O/sq_counters_xpool_fused.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python - <<PY
import json
d = json.load(open("$O/bench_default.json"))
print("bench:", d["value"], d["unit"], d["ms_per_step"], "ms; roofline", d["roofline"]["kernel"], d["roofline"]["frac"], "; retrieval", d["retrieval"]["ms_per_step"], "ms")
PY
