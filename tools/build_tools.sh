#!/bin/bash
# The measurement binaries behind profiles/ (tools/_build/ is not tracked): instruction-rate and field micro-benchmarks,
# the MFMA experiment, the NTT pass micro-benchmark, the FETCH_SIZE calibration, the code-generation probe.
cd "$(dirname "$0")" && mkdir -p _build
for t in ubench carry_ubench valu_rate_ubench fp28_field_ubench mfma_redc_ubench dfma_mul_ubench fp28_mul_ubench batch_affine_ubench ntt_ubench pmc_calib; do  # (tools/g2_probe.hip: the round-1 code-generation probe, kept for DESIGN section 8; it predates the table-row layout and no longer builds)
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -w $t.hip -o _build/$t && echo "built tools/_build/$t"
done
