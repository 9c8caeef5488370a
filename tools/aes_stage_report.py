"""Reduces a rocprofv3 --kernel-trace CSV of tools/aes_trace.py to the stage picture of one AES block: per bootstrap
launch its width (grid), mapping and duration; achieved bootstraps/s per width bucket; GPU idle time between kernels.
usage: python3 tools/aes_stage_report.py <kernel_trace.csv> [first_launch_to_count]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def width(r):
    name, grid, wg = r["Kernel_Name"], int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r["Grid_Size"]), int(r["Workgroup_Size_X"]) if "Workgroup_Size_X" in r else int(r["Workgroup_Size"])
    wgs = grid // wg
    if "pbs_lat" in name: return wgs, "lat"
    if "pbs_asm" in name or "pbs_kernel<" in name: return wgs * (wg // 64), "%dw" % (wg // 64)   # one bootstrap per wave
    return None, None
# the timed block starts after the warm-up SubBytes (seven stages): everything after the 7th keyswitch
ks_idx = [i for i, r in enumerate(rows) if "keyswitch" in r["Kernel_Name"] and "init" not in r["Kernel_Name"]]
start = ks_idx[6] + 1 if len(ks_idx) > 7 else 0
rows = rows[start:]
t0, t1 = int(rows[0]["Start_Timestamp"]), int(rows[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
print(f"span {(t1 - t0) / 1e6:.1f} ms, kernels {busy / 1e6:.1f} ms, idle between kernels {(t1 - t0 - busy) / 1e6:.1f} ms ({100 * (t1 - t0 - busy) / (t1 - t0):.1f} %)")
buckets = collections.OrderedDict()
edges = [64, 128, 256, 512, 1024, 2048, 4096, 1 << 30]
tot_pbs = tot_ns = 0
for r in rows:
    w, kind = width(r)
    if w is None: continue
    ns = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    b = next(e for e in edges if w <= e)
    d = buckets.setdefault(b, [0, 0, 0, collections.Counter()])
    d[0] += 1; d[1] += w; d[2] += ns; d[3][kind] += 1
    tot_pbs += w; tot_ns += ns
print(f"bootstrap launches: {sum(d[0] for d in buckets.values())}, bootstrap slots {tot_pbs} (grid-rounded), {tot_ns / 1e6:.1f} ms, {tot_pbs / tot_ns * 1e9 / 1e3:.1f} k/s")
for b, d in sorted(buckets.items()):
    print(f"  width <= {b if b < (1 << 30) else 'inf':>5}: {d[0]:3d} launches, {d[1]:7d} bootstraps, {d[2] / 1e6:7.1f} ms, {d[1] / d[2] * 1e9 / 1e3:7.1f} k/s, mappings {dict(d[3])}")
other = collections.Counter()
for r in rows:
    if width(r)[0] is None: other[r["Kernel_Name"].split("(")[0][:40]] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, v in other.items(): print(f"  other: {k}: {v / 1e6:.2f} ms")
