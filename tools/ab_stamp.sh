#!/bin/bash
# cycle stamps of one tile (diagnostic): bash tools/ab_stamp.sh ["extra flags"]; LAYER=s2 for a 5x5/s2 layer
set -e
cd "$(dirname "$0")/.."
# leave the default library behind, whatever happens (build.py also rebuilds when the flag string changes)
trap 'DSIC_EXTRA_FLAGS= python domain-specific-image-compression_amd/build.py > /dev/null 2>&1' EXIT
touch domain-specific-image-compression_amd/csrc/conv_wino.hip
DSIC_EXTRA_FLAGS="-DWINO_STAMP=1 $1" python domain-specific-image-compression_amd/build.py > /dev/null 2>&1
python tools/wino_stamps.py 2>&1 | grep -v amdgpu
