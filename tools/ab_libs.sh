# A/B of several builds of libmade_hip.so on ONE box: bash tools/ab_libs.sh lib1.so lib2.so ...  (each is copied over mgsv_amd/libmade_hip.so
# of the box's snapshot in turn, two rounds; prints the taped and the eager step)
cp mgsv_amd/libmade_hip.so /tmp/_orig.so
for rep in 1 2; do
  for v in "$@"; do
    cp $v mgsv_amd/libmade_hip.so
    python bench.py --workload train --no-cpu-baseline --steps 40 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('%-24s' % '$v', d['ms_per_step'], d['config'].get('eager_ms_per_step'), [(k, v2['ms_per_step']) for k, v2 in list(d['kernels'].items())[:3]])"
  done
done
cp /tmp/_orig.so mgsv_amd/libmade_hip.so
