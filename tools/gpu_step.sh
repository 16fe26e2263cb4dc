#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/gpu_step.sh NAME CMD...   -> runs CMD, output to gpurun_out/r2/NAME.log
mkdir -p gpurun_out/r2
name=$1; shift
"$@" > gpurun_out/r2/$name.log 2>&1
rc=$?
echo "== $name rc=$rc"; tail -n ${TAILN:-25} gpurun_out/r2/$name.log
exit $rc
