set -e
B="python bench.py --windows 512 --no-cpu-baseline --no-orb --inertial-windows 0"
P="import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), {k: round(v['total_ms']/max(v['launches'],1),3) for k,v in d['kernels'].items()})"
for i in 1 2; do
echo base; ORBSLAM3_HIP_LIB=exp/lib_base.so $B 2>&1 | tail -1 | python -c "$P"
echo new; $B 2>&1 | tail -1 | python -c "$P"
done
