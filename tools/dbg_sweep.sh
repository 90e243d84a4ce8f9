#!/bin/bash
# Time k_wide128_h with one ingredient removed at a time (results are WRONG with any switch set: measurement only).
# Builds the library with the switches compiled in, runs tools/prof_op.py per switch value, then rebuilds the production binary.
#   on the GPU box:  bash tools/dbg_sweep.sh [op] > gpurun_out/dbg_sweep.txt      (profiles/r02_dbg_sweep.txt)
# Switches (DSG_WIDE_DBG): 1 no barrier, 2 no DMA wait, 16 no DMA issue, 32 second read of x from one hot KiB, 64 first read of
# in0 from one hot KiB + no output store.  (The 4 = no MFMA / 8 = no VALU switches of the committed sweep were removed with the
# code paths they sat in.)
OP=${1:-up.17.res}
DSG_EXTRA_CXXFLAGS="-DDSG_WIDE_DBG_ENABLE=127" python3 -c "from diffsg_amd import _lib; _lib.build(force=True)"
for d in 0 1 2 3 16 19 32 96; do
  echo -n "dbg=$d  "; DSG_WIDE_DBG=$d timeout 120 python3 tools/prof_op.py $OP 65536 30 2>/dev/null | tail -n 1
done
python3 -c "from diffsg_amd import _lib; _lib.build(force=True)"
