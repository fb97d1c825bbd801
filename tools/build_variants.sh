# Bisection builds of libmade_hip.so for tools/dec_corun_probe.py (MADE_LIB_PATH=tools/_ab/<name>.so): one source file recompiled with other
# flags, the rest of the objects from the product build.  usage: bash tools/build_variants.sh   (in the build container; the .so files travel)
#   dec_packed       decoder.hip WITH the SLP vectoriser (the product build has -fno-slp-vectorize): v_pk_*_f32 in made_dec_stage_bwd --
#                    the build whose results move beside a register-staged Linear (profiles/r04_c_dec_corun_probe3.txt)
#   dec_packed_nopk  the same with the packed-fp32-ops target feature off: SLP on, no packed instructions -- stable again
#   dec_packed_o1 / dec_packed_forcezero / dec_packed_shfl: -O1 (no SLP: stable), every s_waitcnt forced to zero (still moves), the row sums
#                    through ds_bpermute instead of DPP (still moves)
export MADE_DEBUG_VARIANTS=1          # (measurement knobs are honoured only under this switch)
set -e
cd "$(dirname "$0")/.."
make -C mgsv_amd/csrc -j8 > /dev/null
mkdir -p tools/_ab
B=mgsv_amd/csrc/build
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-pass-failed"
others() { ls $B/*.o | grep -v "/$1.o"; }
variant() {   # name, source stem, extra flags
  /opt/rocm/bin/hipcc $FLAGS $3 -c mgsv_amd/csrc/$2.hip -o tools/_ab/$1_$2.o 2>&1 | grep -v "not a recognized feature" || true
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/_ab/$1.so tools/_ab/$1_$2.o $(others $2)
  echo "built tools/_ab/$1.so"
}
variant dec_packed decoder ""
variant dec_packed_nopk decoder "-Xclang -target-feature -Xclang -packed-fp32-ops"
variant dec_packed_o1 decoder "-O1"
variant dec_packed_forcezero decoder "-mllvm -amdgpu-waitcnt-forcezero"
variant dec_packed_shfl decoder "-DMADE_DEBUG_WAVE_SUM_SHFL"
