#!/bin/bash
# the intermittent "H2D copy failed" of three threads proving lone proofs with lone_proof_graph=1: how often, which HIP error, and does it need the upload chain?
o=gpurun_out/r06zd; mkdir -p $o
for v in diag nochain; do MASP_HIP_LIBRARY=tools/_build/ab/libmasp_hip_$v.so timeout 600 python tools/lone_graph_stress.py 40 > $o/$v.txt 2>&1; tail -10 $o/$v.txt; done
STRESS_GRAPH=0 MASP_HIP_LIBRARY=tools/_build/ab/libmasp_hip_diag.so timeout 600 python tools/lone_graph_stress.py 40 > $o/diag_no_graph.txt 2>&1; tail -4 $o/diag_no_graph.txt
