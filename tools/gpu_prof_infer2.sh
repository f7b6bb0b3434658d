#!/bin/bash
# kernel statistics of the extraction bench under given bench flags: tools/gpu_prof_infer2.sh NAME [bench flags...]
name=${1:-pinf}; shift
out=gpurun_out/$name; mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/st -- python3 bench.py --mode infer --clips 20480 --infer-streams 1 --no-cpu-baseline --no-roofline "$@" > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
cp $(find $out/st -name "*kernel_stats.csv" | head -1) $out/kernel_stats_infer.csv
rm -rf $out/st
python3 - $out/kernel_stats_infer.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:22]:
    n=r['Name'].replace('(anonymous namespace)::','')[:86]
    print(f"{n:86s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.1f} us  {100*float(r['TotalDurationNs'])/tot:5.1f}%")
PY
