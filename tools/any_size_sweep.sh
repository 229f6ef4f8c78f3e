#!/bin/bash
# end-to-end slice-steps/s over grid sizes that are not powers of two (through gpurun): 64 probes, 100 slices (50 above 512)
for n in 501 500 448 360 300 251 200 160 128 100; do
 r=$(python bench.py --grid $n --slices 100 --probes 64 --steps 8 --warmup 4 --no-cpu-baseline --no-tacaw 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print(d['value'], d['roofline']['avg_launch_us'], d['config']['frame_batch'])")
 echo "grid $n x 100 slices: $r"
done
for n in 1500 1200 1023 1000 997 900 768 700 600 540; do
 r=$(python bench.py --grid $n --slices 50 --probes 64 --steps 8 --warmup 4 --no-cpu-baseline --no-tacaw 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print(d['value'], d['roofline']['avg_launch_us'], d['config']['frame_batch'])")
 echo "grid $n x 50 slices: $r"
done
