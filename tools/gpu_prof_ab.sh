#!/bin/bash
# kernel-trace of the training step for two host-flag settings (one stream: clean per-kernel durations; two streams: the real step)
name=${1:-prof}; out=gpurun_out/$name; mkdir -p $out
export TMPDIR=/tmp
for cfg in "fused:" "unfused:--flag ops.FUSE_BN_BWD_APPLY=0"; do
  tag=${cfg%%:*}; flags=${cfg#*:}
  rocprofv3 --kernel-trace --output-format csv -d $out/tr_$tag -- python3 bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-roofline --no-other --no-overlap $flags > $out/bench_$tag.json 2> $out/bench_$tag.err || { tail -5 $out/bench_$tag.err; exit 1; }
  python3 tools/trace_by_kernel.py $out/tr_$tag 5 > $out/by_kernel_$tag.txt
  rm -rf $out/tr_$tag
  head -3 $out/by_kernel_$tag.txt
done
