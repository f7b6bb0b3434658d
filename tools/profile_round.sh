#!/bin/bash
# The measurement set behind profiles/<name>/ (run on the MI355X box from the repo root):
#   tools/profile_round.sh r02a_baseline [train|infer|deep ...]
# PMC passes first (separate runs, --kernel-trace only beside --pmc), so that the bench line can cite the measured traffic;
# then the kernel statistics (rocprofv3 --kernel-trace --stats), the per-step trace summary and the un-profiled bench run.
# Everything lands in gpurun_out/<name>/ (the only directory gpurun brings back); copy the summaries to profiles/<name>/:
#   mkdir -p profiles/<name> && cp gpurun_out/<name>/{*.json,*.csv,*.txt} profiles/<name>/
set -e
name=$1; shift
modes=${@:-train infer deep}
out=gpurun_out/$name
mkdir -p $out
export TMPDIR=/tmp
common="--no-cpu-baseline --no-roofline --no-other"
for mode in $modes; do
  case $mode in
    train) flags=""; tag="" ;;
    infer) flags="--mode infer --clips 20480 --infer-streams 1"; tag="_infer" ;;
    deep)  flags="--deep"; tag="_deep" ;;
  esac
  eager="--no-graph --no-overlap"; [ $mode = infer ] && eager=""
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch$tag -- python3 bench.py --steps 2 --warmup 1 $common $eager $flags > $out/pmc_fetch$tag.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write$tag -- python3 bench.py --steps 2 --warmup 1 $common $eager $flags > $out/pmc_write$tag.log 2>&1
  python3 tools/hbm_traffic.py $out/pmc_fetch$tag $out/pmc_write$tag $out/hbm_traffic$tag.json
  rm -rf $out/pmc_fetch$tag $out/pmc_write$tag
  # matrix-core busy cycles and LDS bank conflicts per kernel (SQ counters: their own pass)
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_sq$tag -- python3 bench.py --steps 2 --warmup 1 $common $eager $flags > $out/pmc_sq$tag.log 2>&1
  python3 tools/pmc_sq_summary.py $(find $out/pmc_sq$tag -name "*counter_collection.csv" | head -1) $(find $out/pmc_sq$tag -name "*kernel_trace.csv" | head -1) > $out/pmc_sq_summary$tag.txt || true
  rm -rf $out/pmc_sq$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats$tag -- python3 bench.py --steps 10 --warmup 4 $common $flags > $out/bench_under_rocprof$tag.json 2> $out/stats$tag.log
  cp $(find $out/stats$tag -name "*kernel_stats.csv" | head -1) $out/kernel_stats$tag.csv
  if [ $mode != infer ]; then
    python3 tools/trace_summary.py $out/stats$tag 8 > $out/trace_summary$tag.txt || true
  fi
  rm -rf $out/stats$tag
  [ $mode = infer ] && flags="--mode infer"
  python3 bench.py --no-other $flags > $out/bench$tag.json 2> $out/bench$tag.log
  tail -c 400 $out/bench$tag.json; echo
done
