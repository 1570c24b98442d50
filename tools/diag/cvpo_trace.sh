cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_cvpo
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_cvpo -- python $GRAFT_REPO_ROOT/tools/bench_cvpo.py --no-cpu --updates 400 > /tmp/cvpo.log 2>&1
tail -1 /tmp/cvpo.log | cut -c1-300
python - <<'PY'
import csv, glob, collections
rows=[]
for f in glob.glob('/tmp/prof_cvpo/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ','')[:50], int(r.get('Grid_Size_X',0))//max(int(r.get('Workgroup_Size_X',1)),1)))
rows.sort()
tail=rows[-60:]
prev=None
for s,e,n,g in tail:
    print(f"{n:50s} blocks {g:5d} dur {(e-s)/1e3:7.2f} gap {((s-prev)/1e3 if prev else 0):6.2f}")
    prev=e
PY
