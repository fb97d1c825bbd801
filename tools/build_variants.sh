# Bisection builds of libmade_hip.so for tools/dec_corun_probe.py (MADE_LIB_PATH=tools/_ab/<name>.so): one source file recompiled with other
# flags, the rest of the objects from the product build.  usage: bash tools/build_variants.sh   (in the build container; the .so files travel)
set -e
cd "$(dirname "$0")/.."
make -C mgsv_amd/csrc -j8 > /dev/null
mkdir -p tools/_ab
B=mgsv_amd/csrc/build
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-pass-failed"
others() { ls $B/*.o | grep -v "/$1.o"; }
variant() {   # name, source stem, extra flags
  /opt/rocm/bin/hipcc $FLAGS $3 -c mgsv_amd/csrc/$2.hip -o tools/_ab/$1_$2.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/_ab/$1.so tools/_ab/$1_$2.o $(others $2)
  echo "built tools/_ab/$1.so"
}
variant dec_forcezero decoder "-mllvm -amdgpu-waitcnt-forcezero"
variant dec_shfl decoder "-DMADE_DEBUG_WAVE_SUM_SHFL"
variant dec_o1 decoder "-O1"
variant dec_noslp decoder "-fno-slp-vectorize"
variant dec_o2 decoder "-O2"
variant dec_nopk decoder "-Xclang -target-feature -Xclang -packed-fp32-ops"
variant dec_nopostsched decoder "-mllvm -enable-post-misched=0"
