run() { v=$(env "$@" python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f  h2h %.1f  lat %.2f acc %.1f' % (d['value'], d['host_to_host']['value'], d['single_proof_latency_ms'], d['roofline']['avg_launch_ms']))"); echo "$* : $v"; }
python -m pytest tests/test_gpu_batch_mode.py tests/test_golden_proofs.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
run MASP_HIP_SORT=scatter
run MASP_HIP_SORT=twopass
run MASP_HIP_SORT=scatter
run MASP_HIP_SORT=twopass
PROF_ARGS="--steps 2 --warmup 1 --no-cpu-baseline" PROF_GY=128 bash tools/prof_run.sh sort2 MASP_HIP_SLOTS=1 > /dev/null 2>&1
grep "hist\|offsets\|scatter\|coarse\|partition\|bucketize\|span" gpurun_out/prof_sort2/batch.txt | cut -c1-130
rm -f gpurun_out/prof_sort2/*.db
