#!/bin/bash
# round 6, third GPU call: kernel durations of a lone Spend proof with the bucket-aligned accumulation of its h query (why did MSM h not get
# shorter?) and where a second context of a process loses its end-to-end rate (tools/second_context_stage_probe.py)
o=gpurun_out/r06c; mkdir -p $o
bash tools/lone_trace.sh $o/lone_timeline_aligned.txt > $o/lone_trace.log 2>&1
grep -n "k_msm_accumulate\|k_msm_bucket_gather\|k_msm_bucket_heavy\|k_msm_heavy_join\|k_msm_wsum\|total" $o/lone_timeline_aligned.txt | tail -30
python tools/second_context_stage_probe.py > $o/second_context_stages.txt 2>&1; cat $o/second_context_stages.txt
SCP_KEEP_FIRST=1 MASP_HIP_TREE_SUB=43 python tools/second_context_stage_probe.py > $o/second_context_stages_first_kept_open.txt 2>&1; cat $o/second_context_stages_first_kept_open.txt
