export TMPDIR=/tmp
for d in ${DBGS:-0}; do
  DSG_WG_DBG=$d rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/wg$d -o tr -- python3 tools/train_prof.py 5 > /dev/null 2>&1
  python - <<PY
import csv,glob
f=glob.glob("gpurun_out/wg$d/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "k_wgrad" in r["Name"] or "k_colsum" in r["Name"]: print("dbg=$d", r["Name"][:20], r["AverageNs"])
PY
done
