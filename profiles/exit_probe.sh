cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/exitprobe
for w in none devcount lba orb liba pose torch+lba; do
  timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/ep_$w -o t -- python3 $R/profiles/exit_probe.py $w $R/gpurun_out/exitprobe/maps_$w.txt > $R/gpurun_out/exitprobe/log_$w.txt 2>&1
  echo "$w rc=$?"
done
