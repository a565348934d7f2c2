#!/bin/bash
# kernel timeline of a lone Spend proof (rocprofv3 --kernel-trace over tools/lone_sweep.py; durations are right, start times are
# stretched by the profiler's ~10 us per launch): usage tools/lone_trace.sh <out.txt>
export TMPDIR=/tmp
root=$PWD
d=$root/gpurun_out/prof_lone
rm -rf $d; mkdir -p $d
(cd /tmp && LONE_ONLY=spend rocprofv3 --kernel-trace --output-format rocpd -d $d -o run -- python $root/tools/lone_sweep.py > $d/run.log 2>&1)
python tools/lone_timeline.py $(find $d -name "*.db" | head -1) > ${1:-gpurun_out/lone_timeline.txt} 2>&1
tail -5 $d/run.log
