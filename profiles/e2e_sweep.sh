cd $GRAFT_REPO_ROOT
python3 bench.py --windows 512 --cache-inputs /tmp/lba_inputs.pkl --prepare-only > /dev/null 2>&1
B="python3 bench.py --windows 512 --cache-inputs /tmp/lba_inputs.pkl --steps 1 --warmup 1 --no-orb --no-cpu-baseline --inertial-windows 0"
for cfg in ${E2E_CFGS:-2x16 4x8 4x8 4x6 4x16 6x4 6x6 8x4}; do
  set -- $(echo $cfg | tr "x" " ")
  ORBSLAM3_HIP_UPLOAD_THREADS=$2 timeout -k 10 250 $B --e2e-contexts $1 --e2e-batches 3 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline())
e = d['end_to_end']
print('contexts', e['contexts'], 'threads', e['upload_threads_per_context'], 'ms/batch %.1f' % e['ms_per_batch'], 'frac %.3f' % e['fraction_of_resident'], 'pack %.1f copy %.1f' % (e['pack_ms_mean'], e['copy_ms_mean']))
"
done
