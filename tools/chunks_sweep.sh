export TMPDIR=/tmp
for c in 32 16 8; do
  DSG_CHUNKS=$c rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ch$c -o tr -- python3 tools/train_prof.py 6 ${1:-65536} > /dev/null 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/ch$c/**/*kernel_stats.csv",recursive=True)[0]
t=0
out=[]
for r in csv.DictReader(open(f)):
    if any(k in r["Name"] for k in ("k_wgrad_h","k_reduce_slabs","k_cs_reduce","k_colsum","k_time_wgrad","k_time_dgrad(")):
        t+=float(r["AverageNs"]); out.append((r["Name"][5:22], round(float(r["AverageNs"])/1e3)))
print("chunks=$c total_us", round(t/1e3), out)
PY
done
