#!/bin/bash
# Measurement build with cycle stamps, the phase tables, then the production build again.
# usage (GPU box): bash tools/cycle_stamps.sh [sample_rows] [train_rows]
DSG_EXTRA_CXXFLAGS="-DDSG_CYCLE_STAMPS" python3 -c "from diffsg_amd import _lib; _lib.build(force=True)" || exit 1
DSG_EXTRA_CXXFLAGS="-DDSG_CYCLE_STAMPS" python3 tools/cycle_stamps.py "$@"   # only a process that names the flags accepts the measurement build (_lib._stale)
python3 -c "from diffsg_amd import _lib; _lib.build(force=True)"
