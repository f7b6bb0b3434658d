#!/bin/bash
# the whole GPU suite, then the default bench run exactly as the driver invokes it
out=gpurun_out/r04_full; mkdir -p $out
timeout -k 10 1700 python -m pytest tests -m gpu -x -q > $out/full.log 2>&1; rc=$?
tail -6 $out/full.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 900 python3 bench.py > $out/bench_default.json 2> $out/bench_default.err; rc=$?
tail -c 1500 $out/bench_default.json
exit $rc
