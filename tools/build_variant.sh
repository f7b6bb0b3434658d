#!/bin/bash
# Build an alternative kernel library with extra compiler flags for A/B experiments:
#   tools/build_variant.sh alt -DNSID_RPAD=16     ->  neuralsampleid_amd/libnsid_hip_alt.so   (use with NSID_LIB=<path>)
#   ONLY="gemm256" tools/build_variant.sh x -DFOO ->  recompiles only the named sources; the other objects come from the main build
#   VARIANT_SRC=dir: a source that exists as dir/<name>.hip is compiled from there (timing experiments edit a COPY, never the product file)
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
obj=/tmp/nsid_variant_$name; mkdir -p $obj
# (the product sources no longer carry result-changing timing switches: a timing experiment edits a copy under tools/variants/)
all="tuning gemm gemm256 wsgemm ffn_fused ffn256_fused mrconv_fused wgrad bn knn mr ntxent misc"
for s in $all; do
  if [ -n "$ONLY" ] && ! echo " $ONLY " | grep -q " $s "; then
    cp $root/neuralsampleid_amd/csrc/_obj/$s.o $obj/$s.o
    continue
  fi
  src=$root/neuralsampleid_amd/csrc/$s.hip
  [ -n "$VARIANT_SRC" ] && [ -f "$VARIANT_SRC/$s.hip" ] && src=$VARIANT_SRC/$s.hip       # an edited copy (tools/variants/...) of one source
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -fno-gpu-rdc -I $root/include -I $root/neuralsampleid_amd/csrc "$@" -c $src -o $obj/$s.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/neuralsampleid_amd/libnsid_hip_$name.so $obj/*.o
echo built $root/neuralsampleid_amd/libnsid_hip_$name.so
