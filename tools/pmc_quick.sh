#!/bin/bash
# On the GPU box:  bash tools/pmc_quick.sh NAME "M,K,D[,H[,Q]]" [ENV=VALUE ...]   -> SQ instruction / wait counters of every
# kernel of one quick_bench case, averaged per dispatch, printed and kept under gpurun_out/r3/pmc_NAME.txt
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$PWD}
name=$1; shape=$2; shift 2
for kv in "$@"; do export "$kv"; done
OUT=$ROOT/gpurun_out/r3; mkdir -p $OUT
cd /tmp
rm -rf /tmp/pmcq_$name
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d /tmp/pmcq_$name -o r -- python3 $ROOT/tools/quick_bench.py $shape > /tmp/pmcq_$name.log 2>&1 || { echo "rocprofv3 failed"; tail -5 /tmp/pmcq_$name.log; }
python3 - "$name" "$OUT/pmc_$name.txt" <<'PY'
import csv, glob, collections, sys
name, out = sys.argv[1], sys.argv[2]
f = glob.glob(f"/tmp/pmcq_{name}/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("(")[0][-70:]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
lines = []
for k in acc:
    d = len(n[k])
    lines.append(f"{k}  dispatches {d}")
    for c, v in sorted(acc[k].items()):
        lines.append(f"    {c:28s} {v / d:16.0f} per dispatch")
open(out, "w").write("\n".join(lines) + "\n"); print("\n".join(lines))
PY
