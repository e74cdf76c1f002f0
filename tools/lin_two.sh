R=$PWD; cd /tmp; export TMPDIR=/tmp; export SVGPU_BA_LIN_TWO_LAUNCHES=1
for w in global local; do
timeout 200 rocprofv3 --kernel-trace --truncate-kernels -d /tmp/two_$w -o k --output-format csv -- python $R/tools/ba_prof.py $w > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("/tmp/two_$w/**/k_kernel_trace.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith("k_ba_lin") and "fin" not in r["Kernel_Name"]]
d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows]
print("$w", len(d), "grid", rows[0]["Grid_Size_X"] if "Grid_Size_X" in rows[0] else "", rows[1].get("Grid_Size_X",""))
print(" lm  :", " ".join("%.0f"%x for x in d[0::2][:24]))
print(" pose:", " ".join("%.0f"%x for x in d[1::2][:24]))
PY
done
