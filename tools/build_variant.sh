#!/bin/bash
# usage: build_variant.sh NAME "extra hipcc flags"  -> scratch/ab/NAME.so (then rebuild the default with build.py)
cd /root/repo
FRIES_EXTRA_HIPCC_FLAGS="$2" python -c "from fries_amd import build; build.build(force=True)" || exit 1
cp fries_amd/libfries_hip.so scratch/ab/$1.so
