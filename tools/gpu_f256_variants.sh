#!/bin/bash
# times the C = 256 fused FFN in every libnsid_hip_f*.so variant present (tools/build_variant.sh), one box
for lib in neuralsampleid_amd/libnsid_hip_f*.so; do
  echo "== $lib"
  NSID_ALLOW_DIAGNOSIS_LIB=1 NSID_LIB=$PWD/$lib timeout -k 10 120 python tools/ffn256_time.py --fused-only || exit 1
done
