#!/bin/bash
# Experiment: board power and clocks while the fused forward sweep loops (is 1.72 GHz a power cap?).
#   bash tools/power_watch.sh   (on the GPU box, through gpurun)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/power; mkdir -p $O
rocm-smi --showpower --showclocks --showmaxpower --showperflevel > $O/idle.txt 2>&1
LOOP=${LOOP:-400} WHICH=${WHICH:-fwd} python3 $R/tools/chain_loop.py > $O/loop.log 2>&1 &
PID=$!
sleep ${SLEEP:-4}
for i in 1 2 3 4 5 6; do
  rocm-smi --showpower --showclocks -t > $O/busy_$i.txt 2>&1
  sleep 0.7
done
wait $PID
tail -5 $O/loop.log
grep -h -i "power\|sclk\|mclk\|fclk\|Temperature" $O/idle.txt | head -12
echo ---- busy
grep -h -i "power\|sclk\|Temperature (Sensor junction)" $O/busy_3.txt $O/busy_5.txt | head -12
