#!/bin/bash
# usage: exp/trace.sh <name> [lib]   -> gpurun_out/exp_<name>/ kernel trace of one bench step
NAME=$1; LIB=${2:-}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/exp_$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
[ -f /tmp/lba_inputs.pkl ] || python3 $ROOT/bench.py --windows 512 --cache-inputs /tmp/lba_inputs.pkl --prepare-only || exit 1
[ -n "$LIB" ] && export ORBSLAM3_HIP_LIB=$ROOT/$LIB
timeout -k 10 280 rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 $ROOT/bench.py --windows 512 --cache-inputs /tmp/lba_inputs.pkl --workers 1 --streams 1 --steps 1 --warmup 0 --no-orb --no-cpu-baseline --inertial-windows 0 > $OUT/log.txt 2>&1
echo "$NAME rc=$?"
