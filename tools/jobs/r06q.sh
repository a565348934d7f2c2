#!/bin/bash
# round 6: do later contexts at 4 slots lose their rate because two MAIN streams share a hardware queue?  contexts A / B / C, stream order 2 (slot by slot) and 1 (mains first)
o=gpurun_out/r06q; mkdir -p $o
for L in tools/_build/ab/libmasp_hip_order2.so tools/_build/ab/libmasp_hip_order1.so; do
  echo "=== $L" | tee -a $o/contexts_main_streams.txt
  MASP_HIP_LIBRARY=$PWD/$L timeout 700 python tools/second_context_stage_probe.py 2>&1 | grep "context\|own streams\|host to host\|isolated" | tee -a $o/contexts_main_streams.txt
done
