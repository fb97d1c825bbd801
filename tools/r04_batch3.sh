O=gpurun_out/r04_c; mkdir -p $O
{
for v in "" dec_noslp dec_nopk dec_o1 dec_nopostsched dec_shfl; do
  echo "### micro, ${v:-product build}"
  if [ -z "$v" ]; then CORUN="register-staged" timeout 300 python tools/dec_corun_probe.py micro 2000
  else CORUN="register-staged" MADE_LIB_PATH=$PWD/tools/_ab/$v.so timeout 300 python tools/dec_corun_probe.py micro 2000; fi
done
echo "### step, MADE_RET_SPLIT=0, dec_nopk"; MADE_LIB_PATH=$PWD/tools/_ab/dec_nopk.so MADE_RET_SPLIT=0 timeout 300 python tools/dec_corun_probe.py step 400
} > $O/dec_corun_probe3.txt 2>&1
grep -v amdgpu.ids $O/dec_corun_probe3.txt
