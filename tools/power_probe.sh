#!/bin/bash
# Does the GLR run against the board's power limit?  tools/glr_only.py in a loop (400 runs) with
# rocm-smi sampled beside it: average socket power, power cap, shader clock.
#   tools/power_probe.sh > gpurun_out/power_probe.txt
python3 tools/glr_only.py 600 f16x2 400 > /tmp/glr_loop.txt 2>&1 &
pid=$!
sleep 25
echo "--- rocm-smi while the GLR loops (spatial + spectral, 3681x600x600, f16x2)"
for i in 1 2 3 4 5 6 7 8; do
  rocm-smi --showpower --showmaxpower --showclocks 2>/dev/null | grep -i "power\|sclk\|mclk\|fclk" | tr '\n' ';'
  echo
  sleep 0.5
done
wait $pid
echo "--- the loop"
cat /tmp/glr_loop.txt
echo "--- rocm-smi idle"
sleep 3
rocm-smi --showpower --showmaxpower --showclocks 2>/dev/null | grep -i "power\|sclk\|mclk\|fclk"
