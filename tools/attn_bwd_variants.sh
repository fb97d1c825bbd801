# Elimination builds of the single-pass attention backward (tools/_ab/fskip<mask>.so; FUSED_SKIP bits: 1 S / dP products, 2 per-score arithmetic,
# 4 dV / dK products, 8 dQ phase): what each part of a step costs.  bash tools/attn_bwd_variants.sh 1 2 4 8 15   then on the GPU box
# MADE_LIB_PATH=tools/_ab/fskip2.so python tools/attn_bwd_check.py time
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_ab
B=mgsv_amd/csrc/build
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-pass-failed -fno-slp-vectorize"
if [ "$1" = stamps ]; then
  /opt/rocm/bin/hipcc $FLAGS -DFUSED_STAMPS=1 -c mgsv_amd/csrc/attention_bwd_fused.hip -o tools/_ab/fstamps.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/_ab/fstamps.so tools/_ab/fstamps.o $(ls $B/*.o | grep -v "/attention_bwd_fused.o")
  echo "built tools/_ab/fstamps.so (MADE_LIB_PATH=tools/_ab/fstamps.so python tools/attn_bwd_check.py stamps)"; exit 0
fi
if [ "$1" = flags ]; then   # bash tools/attn_bwd_variants.sh flags <name> "<-D flags>"
  /opt/rocm/bin/hipcc $FLAGS $3 -c mgsv_amd/csrc/attention_bwd_fused.hip -o tools/_ab/$2.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/_ab/$2.so tools/_ab/$2.o $(ls $B/*.o | grep -v "/attention_bwd_fused.o")
  echo "built tools/_ab/$2.so"; exit 0
fi
for m in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS -DFUSED_SKIP=$m -c mgsv_amd/csrc/attention_bwd_fused.hip -o tools/_ab/fskip$m.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/_ab/fskip$m.so tools/_ab/fskip$m.o $(ls $B/*.o | grep -v "/attention_bwd_fused.o")
  echo "built tools/_ab/fskip$m.so"
done
