#!/bin/bash
# clocks and power while the 10 000-clip launch repeats (diag/ramp.py as the load)
R=$GRAFT_REPO_ROOT; cd $R
python diag/ramp.py 600 > gpurun_out/pw_ramp.txt 2>&1 &
BP=$!
for i in $(seq 1 14); do
  sleep 1
  /opt/rocm/bin/rocm-smi --showpower --showclocks --showtemp --showuse 2>/dev/null | grep -E "Power \(W\)|sclk|Sensor junction|GPU use" | sed 's/.*: //' | tr '\n' ' '
  echo
done
wait $BP
tail -3 gpurun_out/pw_ramp.txt
