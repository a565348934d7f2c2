run() { v=$(env "$@" python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f  h2h %.1f' % (d['value'], d['host_to_host']['value']))"); echo "$* : $v"; }
run A=1
run MASP_HIP_HEAVY_WAVES=1
run MASP_HIP_HEAVY_WAVES=1 MASP_HIP_HEAVY_BLOCKS=64
run MASP_HIP_HEAVY_WAVES=1 MASP_HIP_HEAVY_BLOCKS=128
run MASP_HIP_HEAVY_WAVES=1 MASP_HIP_HEAVY_BLOCKS=256
run MASP_HIP_HEAVY_WAVES=1 MASP_HIP_HEAVY_BLOCKS=128 MASP_HIP_HEAVY_SPAN=8
run MASP_HIP_HEAVY_WAVES=1 MASP_HIP_HEAVY_BLOCKS=128 MASP_HIP_HEAVY_SPAN=16
run MASP_HIP_HEAVY_WAVES=1 MASP_HIP_HEAVY_BLOCKS=256 MASP_HIP_HEAVY_SPAN=12
run MASP_HIP_HEAVY_WAVES=4 MASP_HIP_HEAVY_BLOCKS=128
run A=1
