# A/B of the training step over environment knobs on ONE box: bash tools/ab_step.sh <tag> "K1=V1" "K2=V2 K3=V3" ...  (first the default, then each setting, twice over)
export MADE_DEBUG_VARIANTS=1          # (measurement knobs are honoured only under this switch)
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
run() { env $2 timeout 200 python bench.py --workload train --no-cpu-baseline --steps 40 > $O/$1.json 2>$O/$1.err; python - <<PY
import json
try:
    d = json.load(open("$O/$1.json")); print("%-40s %.3f ms (eager %s)" % ("$2" or "default", d["ms_per_step"], d["config"].get("eager_ms_per_step")))
except Exception as e:
    print("%-40s failed: %s" % ("$2", e))
PY
}
for rep in 1 2; do
  run default_$rep ""
  i=0
  for kv in "$@"; do i=$((i+1)); run s${i}_$rep "$kv"; done
done
