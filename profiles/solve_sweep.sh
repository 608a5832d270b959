# resident throughput against the panel width / block size of k_solve (LDS per block decides how many windows share a CU)
cd $GRAFT_REPO_ROOT
python3 bench.py --windows 512 --cache-inputs /tmp/lba_inputs.pkl --prepare-only > /dev/null 2>&1
for cfg in 24x512 12x512 24x256 12x256 6x256; do
  set -- $(echo $cfg | tr "x" " ")
  OSH_LBA_SOLVE_NB=$1 OSH_LBA_SOLVE_THREADS=$2 timeout -k 10 250 python3 bench.py --windows 512 --cache-inputs /tmp/lba_inputs.pkl --steps 3 --warmup 1 --no-orb --no-cpu-baseline --inertial-windows 0 --e2e-batches 0 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('$cfg', 'windows/s %.0f' % d['value'], 'ms/step %.2f' % d['ms_per_step'], 'solve ms/step %.2f' % d['kernels']['solve']['total_ms'])
"
done
