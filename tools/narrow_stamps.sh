#!/bin/bash
# usage (GPU box): bash tools/narrow_stamps.sh [rows]
DSG_EXTRA_CXXFLAGS="-DDSG_CYCLE_STAMPS" python3 -c "from diffsg_amd import _lib; _lib.build(force=True)" || exit 1
DSG_EXTRA_CXXFLAGS="-DDSG_CYCLE_STAMPS" python3 tools/narrow_stamps.py "$@"
python3 -c "from diffsg_amd import _lib; _lib.build(force=True)"
