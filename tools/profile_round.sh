#!/bin/bash
# Runs ON THE GPU BOX from the repo root:  bash tools/profile_round.sh [workloads...]
# For every workload: rocprofv3 --kernel-trace --stats of the bench command, then FETCH_SIZE and WRITE_SIZE in their own
# passes (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass; --pmc never together with trace domains
# other than kernel-trace), then one SQ pass.  Raw output under gpurun_out/<round>/prof/<workload>/; tools/summarize_profiles.py
# turns it into the files kept under profiles/.
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$PWD}
WLS=${@:-cfg2 k8192 cfg3a cfg3b cfg4 wide1024}
RND=${VQ_ROUND:-r03}
OUT=$ROOT/gpurun_out/$RND/prof
mkdir -p $OUT
cd /tmp
for wl in $WLS; do
  CMD="python3 $ROOT/bench.py --workload $wl --no-legs --no-cpu-baseline --no-sharded --steps 20 --warmup 5"
  echo "== $wl: kernel trace ($(date +%T))"; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$wl/stats -o $wl -- $CMD > $OUT/$wl.stats.log 2>&1 || { echo "FAILED stats $wl"; tail -5 $OUT/$wl.stats.log; exit 1; }
  grep '^{' $OUT/$wl.stats.log > $OUT/$wl.bench_under_trace.json || true
  echo "== $wl: FETCH_SIZE"; rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/$wl/fetch -o $wl -- $CMD > $OUT/$wl.fetch.log 2>&1 || { echo "FAILED fetch $wl"; tail -5 $OUT/$wl.fetch.log; exit 1; }
  echo "== $wl: WRITE_SIZE"; rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/$wl/write -o $wl -- $CMD > $OUT/$wl.write.log 2>&1 || { echo "FAILED write $wl"; tail -5 $OUT/$wl.write.log; exit 1; }
  echo "== $wl: SQ"; rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/$wl/sq -o $wl -- $CMD > $OUT/$wl.sq.log 2>&1 || { echo "FAILED sq $wl"; tail -5 $OUT/$wl.sq.log; }
done
cd $ROOT
find gpurun_out/$RND/prof -name "*.csv" | head -40
VQ_ROUND=$RND python3 tools/summarize_profiles.py gpurun_out/$RND/prof gpurun_out/$RND/prof_summary $WLS
