#!/bin/bash
# Collects the rocprofv3 evidence for one round on the GPU box.  Usage: bash profiles/run_profiles.sh rNN [windows]
# kernel trace + stats in ONE run; every PMC group in its OWN run (never mixed with tracing); inputs are
# generated once without the profiler (the generator forks workers) and re-read from a cache afterwards.
# Pass 1: local BA only (the headline loop).  Pass 2 (trace + one PMC group): the other kernels of the path -- ORB search,
# LocalInertialBA (k_liba), PoseOptimization (k_pose_opt) -- through profiles/other_kernels.py.
set -u
TAG=${1:-r03}
WIN=${2:-512}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py --windows $WIN --cache-inputs /tmp/lba_inputs.pkl --prepare-only || exit 1
BENCH="python3 $ROOT/bench.py --windows $WIN --cache-inputs /tmp/lba_inputs.pkl --workers 1 --streams 1 --steps 1 --warmup 1 --no-orb --no-cpu-baseline --no-sweeps --inertial-windows 0 --e2e-batches 0"
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $BENCH > $OUT/trace.log 2>&1
echo "trace rc=$?"
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAIT_ANY" \
           "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "SQ_INSTS_VALU_MFMA_MOPS_F64"; do
  i=$((i+1))
  timeout -k 10 280 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc$i -o pmc -- $BENCH > $OUT/pmc$i.log 2>&1
  echo "pmc$i ($grp) rc=$?"
done
OTHER="python3 $ROOT/profiles/other_kernels.py"
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/other_trace -o trace -- $OTHER > $OUT/other_trace.log 2>&1
echo "other trace rc=$?"
timeout -k 10 280 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/other_pmc1 -o pmc -- $OTHER > $OUT/other_pmc1.log 2>&1
echo "other pmc rc=$?"
# HBM traffic of k_orb_bruteforce (64 frame pairs per launch) and k_liba (128 windows per launch): uniform launches, one counter per pass
timeout -k 10 280 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/other_pmc2 -o pmc -- $OTHER --uniform > $OUT/other_pmc2.log 2>&1
echo "other FETCH_SIZE rc=$?"
timeout -k 10 280 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/other_pmc3 -o pmc -- $OTHER --uniform > $OUT/other_pmc3.log 2>&1
echo "other WRITE_SIZE rc=$?"
python3 $ROOT/profiles/summarize.py $OUT $TAG $WIN
