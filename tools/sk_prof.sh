cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for shp in "1 24 4608" "8 24 4608" "1 24 8704"; do
rocprofv3 --kernel-trace -d gpurun_out/sktrace -o sk --output-format csv -- python tools/attn_streamk_time.py $shp > gpurun_out/sktrace.log 2>&1
python - "$shp" <<PY
import csv,glob,sys,collections
f=glob.glob("gpurun_out/sktrace/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows=[r for r in rows if "attn_w4" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# classify main launches by whether the next kernel is a merge
out=collections.defaultdict(list)
for i,r in enumerate(rows):
    d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    if "merge" in r["Kernel_Name"]: out["merge"].append(d)
    else:
        nxt=rows[i+1]["Kernel_Name"] if i+1<len(rows) else ""
        out["main+sk" if "merge" in nxt else "main whole"].append(d)
print(sys.argv[1], {k:(len(v), round(sorted(v)[len(v)//2],1)) for k,v in out.items()})
PY
rm -rf gpurun_out/sktrace
done
