#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE passes) of the eager single-stream step with the deferred weight-gradient phase
name=${1:-r06_pmc}; shift
out=gpurun_out/$name; mkdir -p $out
export TMPDIR=/tmp
common="--steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-other --no-graph --no-overlap"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 bench.py $common "$@" > $out/pmc_fetch.log 2>&1 || { tail -5 $out/pmc_fetch.log; exit 1; }
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 bench.py $common "$@" > $out/pmc_write.log 2>&1 || { tail -5 $out/pmc_write.log; exit 1; }
python3 tools/hbm_traffic.py $out/pmc_fetch $out/pmc_write $out/hbm_traffic.json
rm -rf $out/pmc_fetch $out/pmc_write
python3 - $out/hbm_traffic.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print("step_traffic_GB", d.get("step_traffic_GB"))
for k,v in d["kernels"].items():
    if "wgrad" in k: print(k, v)
PY
