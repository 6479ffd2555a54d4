set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OSLAM_LBA_CALL_STATS=1 OSLAM_LBA_SERVICE_STATS=1 python bench.py --no-extras --no-cpu-baseline > gpurun_out/r05_callstats.json 2> gpurun_out/r05_callstats.err || { tail -20 gpurun_out/r05_callstats.err; exit 1; }
grep "lba call" gpurun_out/r05_callstats.err | tail -2000 > gpurun_out/r05_callstats_tail.txt
grep "lba service" gpurun_out/r05_callstats.err | tail -3
python - <<PY
import re,collections
rows=[tuple(map(int,re.findall(r"\d+",l))) for l in open("gpurun_out/r05_callstats_tail.txt")]
# group into calls: a call ends when active==0
calls=[];cur=[]
for r in rows:
    cur.append(r)
    if r[2]==0: calls.append(cur);cur=[]
import statistics as st
print("calls",len(calls),"mean windows",st.mean(c[0][0] for c in calls),"mean slots",st.mean(c[-1][1] for c in calls))
frac=collections.defaultdict(list)
for c in calls:
    for r in c: frac[r[1]].append(r[2]/r[0])
for k in sorted(frac): print("after",k,"slots: mean active frac %.3f over %d calls"%(st.mean(frac[k]),len(frac[k])))
PY
