# round 4, second GPU batch: which build of dec_stage_bwd / which co-runner makes its results move
O=gpurun_out/r04_b; mkdir -p $O
{
echo "### tape check"; timeout 300 python tools/tape_check_probe.py
echo "### micro, product build, every co-runner"; timeout 600 python tools/dec_corun_probe.py micro 2000
for v in dec_o1 dec_noslp dec_o2 dec_nopostsched dec_forcezero dec_shfl; do
  echo "### micro, $v"; CORUN="LDS-DMA 64-row" MADE_LIB_PATH=$PWD/tools/_ab/$v.so timeout 300 python tools/dec_corun_probe.py micro 3000
done
for v in dec_noslp dec_o2; do
  echo "### step, MADE_RET_SPLIT=0, $v"; MADE_LIB_PATH=$PWD/tools/_ab/$v.so MADE_RET_SPLIT=0 timeout 300 python tools/dec_corun_probe.py step 400
done
} > $O/dec_corun_probe2.txt 2>&1
grep -v amdgpu.ids $O/dec_corun_probe2.txt
