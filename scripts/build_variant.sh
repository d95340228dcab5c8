#!/bin/bash
# Builds lz4_amd/liblz4mi_<name>.so with extra compiler flags (A/B runs: LZ4MI_LIB=$PWD/lz4_amd/liblz4mi_<name>.so).
# usage: scripts/build_variant.sh <name> "<flags>"
set -e
cd "$(dirname "$0")/../lz4_amd/csrc"
name=$1; flags=$2
mkdir -p build_$name
SRCS="decode_kernel.hip decode_linked_kernel.hip encode_fast_kernel.hip encode_hc_kernel.hip xxh32_kernel.hip frame_kernels.hip api_block.hip api_frame.hip api_stream.hip"
for f in $SRCS; do
  o=build_$name/${f%.hip}.o
  case $f in
    decode_kernel.hip|decode_linked_kernel.hip|encode_fast_kernel.hip|encode_hc_kernel.hip|xxh32_kernel.hip) hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result $flags -c $f -o $o & ;;
    *) [ build/${f%.hip}.o -nt $f ] && cp build/${f%.hip}.o $o || hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -c $f -o $o & ;;
  esac
done
wait
TORCH_LIB=$(python3 -c "import torch,os;print(os.path.join(os.path.dirname(torch.__file__),'lib'))")
g++ -shared -fPIC -o ../liblz4mi_$name.so build_$name/*.o $TORCH_LIB/libamdhip64.so -Wl,-rpath,/opt/rocm/lib -Wl,--enable-new-dtags
echo built liblz4mi_$name.so
