"""Idle time between the dispatches of one bench step: python3 profiles/gaps.py <dir of a `rocprofv3 --kernel-trace --output-format csv` run of bench.py>.
Dispatches of the copy stream overlap the compute stream (negative gaps): read the largest gaps, not the sum."""
import csv,sys,glob
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# find last occurrence of k_pool_pack start as step begin; take the last full step
idx=[i for i,r in enumerate(rows) if 'k_pool_pack' in r['Kernel_Name']]
# steps have 2 pack launches; take the step starting at the 4th-from-last pack
s=idx[-4]; e=idx[-2]
step=rows[s:e]
t0=int(step[0]['Start_Timestamp']); t1=int(step[-1]['End_Timestamp'])
busy=sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in step)
print('step span %.3f ms, kernel busy %.3f ms, kernels %d' % ((t1-t0)/1e6, busy/1e6, len(step)))
gaps=[]
for a,b in zip(step,step[1:]):
    g=int(b['Start_Timestamp'])-int(a['End_Timestamp'])
    gaps.append((g,a['Kernel_Name'][:40],b['Kernel_Name'][:40]))
print('sum gaps %.3f ms' % (sum(g for g,_,_ in gaps)/1e6))
for g,a,b in sorted(gaps,reverse=True)[:25]:
    print('%8.1f us  %s -> %s' % (g/1e3,a,b))
import collections
small=sum(g for g,_,_ in gaps if g<20000)
print('gaps < 20us: count %d sum %.3f ms' % (sum(1 for g,_,_ in gaps if g<20000), small/1e6))
