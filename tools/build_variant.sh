#!/bin/bash
# Build an alternative kernel library with extra compiler flags for A/B experiments:
#   tools/build_variant.sh alt -DNSID_RPAD=16     ->  neuralsampleid_amd/libnsid_hip_alt.so   (use with NSID_LIB=<path>)
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
obj=/tmp/nsid_variant_$name; mkdir -p $obj
for s in gemm wgrad bn knn mr ntxent misc; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -fno-gpu-rdc -I $root/include "$@" -c $root/neuralsampleid_amd/csrc/$s.hip -o $obj/$s.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/neuralsampleid_amd/libnsid_hip_$name.so $obj/*.o
echo built $root/neuralsampleid_amd/libnsid_hip_$name.so
