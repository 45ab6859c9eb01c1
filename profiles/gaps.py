"""GPU idle time inside one bench step: python3 profiles/gaps.py <dir of a `rocprofv3 --kernel-trace --output-format csv` run of bench.py>.
The dispatches of all streams are merged into busy intervals (copies on the copy stream overlap the compute stream); what is left of the
step's span is idle time, listed by the dispatches on either side of the longest holes."""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_pool_pack' in r['Kernel_Name']]
no = int(sys.argv[2]) if len(sys.argv) > 2 else None
step = rows[idx[2 * no]:idx[2 * no + 2]] if no is not None else rows[idx[-4]:idx[-2]]      # two pack launches per step: step `no` (0-based), or the last full step
iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:44]) for r in step)
t0, t1 = iv[0][0], max(e for _, e, _ in iv)
busy, holes = 0, []
cur_s, cur_e, last = iv[0][0], iv[0][1], iv[0][2]
for s, e, n in iv[1:]:
    if s > cur_e:
        holes.append((s - cur_e, last, n))
        busy += cur_e - cur_s
        cur_s, cur_e = s, e
    cur_e = max(cur_e, e)
    if e >= cur_e:
        last = n
busy += cur_e - cur_s
print('step span %.3f ms, GPU busy (any stream) %.3f ms, idle %.3f ms in %d holes' % ((t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6, len(holes)))
for g, a, b in sorted(holes, reverse=True)[:16]:
    print('%8.1f us  %s -> %s' % (g / 1e3, a, b))
