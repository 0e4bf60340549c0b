# Development aid: builds vqa_amd/libvqa_hip_<name>.so for each "<name>:<extra hipcc flags>" argument, differing from
# the development library (make dev) in gemm.hip only.  Example: bash tools/build_variants.sh d3:-DSP_D=3 onebar:-DSP_ONEBAR
set -e
cd "$(dirname "$0")/.."
make -j8 dev > /dev/null
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Iinclude -Ivqa_amd/csrc -Wall -Wno-unused-function -DVQA_DEV"
OTHERS=$(ls build/dev/*.o | grep -v gemm.o)
for spec in "$@"; do
  name=${spec%%:*}; extra=${spec#*:}
  ( mkdir -p build/var_$name && /opt/rocm/bin/hipcc $FLAGS $extra -c vqa_amd/csrc/gemm.hip -o build/var_$name/gemm.o 2>/dev/null \
    && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o vqa_amd/libvqa_hip_$name.so build/var_$name/gemm.o $OTHERS && echo built $name ) &
done
wait
