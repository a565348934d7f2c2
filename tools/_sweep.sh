#!/bin/bash
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
bash tools/ab.sh masp_amd/libmasp_hip_base.so masp_amd/libmasp_hip.so 3
