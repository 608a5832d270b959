# resident throughput against the number of solver contexts (HIP streams) the 512 windows are split over
cd $GRAFT_REPO_ROOT
python3 bench.py --windows 512 --cache-inputs /tmp/lba_inputs.pkl --prepare-only > /dev/null 2>&1
for st in 1 2 3 4 6 8; do
  timeout -k 10 250 python3 bench.py --windows 512 --cache-inputs /tmp/lba_inputs.pkl --steps 3 --warmup 1 --no-orb --no-cpu-baseline --inertial-windows 0 --e2e-batches 0 --streams $st 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('streams', d['config']['streams_per_gpu'], 'windows/s %.0f' % d['value'], 'ms/step %.2f' % d['ms_per_step'], 'frac %.3f' % d['roofline']['frac'])
"
done
