#!/bin/bash
set -e
cd "$(dirname "$0")/.."
touch domain-specific-image-compression_amd/csrc/conv_wino.hip
DSIC_EXTRA_FLAGS="-DWINO_STAMP=1" python domain-specific-image-compression_amd/build.py > /dev/null 2>&1
python tools/wino_stamps.py 2>&1 | grep -v amdgpu
