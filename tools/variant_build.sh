# One-off variant of libmade_hip.so: ONE source file recompiled with extra flags, the other objects from the product build.
#   bash tools/variant_build.sh <name> <source stem> "<extra flags>"      -> tools/_ab/<name>.so   (then MADE_LIB_PATH=tools/_ab/<name>.so on the GPU box)
set -e
cd "$(dirname "$0")/.."
make -C mgsv_amd/csrc -j8 > /dev/null
mkdir -p tools/_ab
B=mgsv_amd/csrc/build
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-pass-failed -fno-slp-vectorize"
/opt/rocm/bin/hipcc $FLAGS $3 -c mgsv_amd/csrc/$2.hip -o tools/_ab/$1_$2.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/_ab/$1.so tools/_ab/$1_$2.o $(ls $B/*.o | grep -v "/$2.o")
echo "built tools/_ab/$1.so"
