#!/bin/bash
# round 3: the scatter's level moves with the PROCESS (1.56 vs 1.85-1.90 ms). Eight processes in a row on one box, clocks / power /
# temperature read before each: does the level follow anything the driver reports?
mkdir -p gpurun_out/r03
OUT=gpurun_out/r03/scatter_by_process.txt
: > $OUT
for i in 1 2 3 4 5 6 7 8; do
  echo "== process $i" >> $OUT
  rocm-smi --showclocks --showpower --showtemp --showmemuse 2>/dev/null | grep -i "sclk\|mclk\|fclk\|socclk\|Power\|Temperature (Sensor junction)\|Temperature (Sensor memory)\|VRAM%" | sed 's/^/   /' >> $OUT
  timeout 300 tools/gather_scatter_bench -t chunked -l device -e 51200000000 -g 5120000000 -d 128 -c 20 -f scatter -n 1 2>&1 | grep -i "time per call\|kernel" >> $OUT
  timeout 300 tools/gather_scatter_bench -t chunked -l device -e 51200000000 -g 5120000000 -d 128 -c 20 -f gather -n 1 2>&1 | grep -i "time per call" | sed 's/^/   (gather, next process) /' >> $OUT
done
cat $OUT
