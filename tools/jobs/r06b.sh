#!/bin/bash
# round 6, second GPU call: the bucket-aligned accumulation of a lone proof's h query (tests + same-box latency A/B over the lanes per
# bucket and against the chunked form), the additions pass unrolled by two, and b_g2 on a window width of its own
o=gpurun_out/r06b; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_aligned_accumulation.py tests/test_golden_proofs.py tests/test_gpu_lone_and_warm.py -m gpu -x -q > $o/tests.txt 2>&1; tail -3 $o/tests.txt
for rep in 1 2; do
  for L in masp_amd/libmasp_hip.so tools/_build/ab/libmasp_hip_chunked_lone.so tools/_build/ab/libmasp_hip_lpb2.so tools/_build/ab/libmasp_hip_lpb8.so; do
    echo "== $L" >> $o/lone_ab.txt
    MASP_HIP_LIBRARY=$PWD/$L LONE_CHAINS=1 python tools/lone_sweep.py >> $o/lone_ab.txt 2>&1
  done
done
grep "lone ms\|==" $o/lone_ab.txt
bash tools/abn.sh 2 masp_amd/libmasp_hip.so tools/_build/ab/libmasp_hip_unroll2.so > $o/unroll2_ab.txt 2>&1; cat $o/unroll2_ab.txt
for rep in 1 2; do
  for c in "" 13 14; do
    v=$(MASP_HIP_MSM_C_B2=$c MASP_BENCH_E2E=0 MASP_BENCH_LONE=0 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); o=d['other_circuits']; print('%.1f resident %.1f output %.0f convert %.0f' % (d['value'], d['resident']['value'], o['output']['value'], o['convert']['value']))")
    echo "window_bits_b2=${c:-default}: $v" | tee -a $o/b2_window_ab.txt
  done
done
