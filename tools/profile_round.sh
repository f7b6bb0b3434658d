#!/bin/bash
# The measurement set behind profiles/<name>/ (run on the MI355X box from the repo root):
#   tools/profile_round.sh r01d_valu_diet
# PMC passes first (separate runs, no other trace domains), so that the bench line can cite the measured traffic; then the
# kernel statistics, the per-step trace summary and the default bench run itself. Results land in gpurun_out/<name>/.
set -e
name=$1
out=gpurun_out/$name
mkdir -p $out profiles/$name
export TMPDIR=/tmp
common="--no-cpu-baseline --no-roofline"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 $common --no-graph --no-overlap > $out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 bench.py --steps 2 --warmup 1 $common --no-graph --no-overlap > $out/pmc_write.log 2>&1
python3 tools/hbm_traffic.py $out/pmc_fetch $out/pmc_write profiles/$name/hbm_traffic.json
cp profiles/$name/hbm_traffic.json $out/
rm -rf $out/pmc_fetch $out/pmc_write
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 10 --warmup 4 --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/stats.log
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
rm -rf $out/stats
# per-step kernel budget of the bf16 step alone (the run above also times the fp32 arithmetic at its end)
rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python3 bench.py --steps 10 --warmup 4 $common > $out/trace.log 2>&1
python3 tools/trace_summary.py $out/trace 8 > $out/trace_summary.txt
rm -rf $out/trace
python3 bench.py > $out/bench.json 2> $out/bench.log
tail -c 600 $out/bench.json
