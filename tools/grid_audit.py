"""Aggregate a rocprofv3 kernel trace by (kernel, workgroup count): finds launches that under-fill 256 CUs.
usage: python tools/grid_audit.py <kernel_trace.csv> [max_workgroups]"""
import csv, re, sys, collections
path, cap = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 512
agg = collections.defaultdict(lambda: [0, 0.0])
tot = 0.0
for r in csv.DictReader(open(path)):
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:]+(<[^(]{0,48}>)?)", name)
    name = (m.group(1) if m else name)[:70]
    gx, gy, gz = (int(r.get(k, 1) or 1) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
    wx, wy, wz = (int(r.get(k, 1) or 1) for k in ("Workgroup_Size_X", "Workgroup_Size_Y", "Workgroup_Size_Z"))
    wgs = (gx // max(wx, 1)) * (gy // max(wy, 1)) * (gz // max(wz, 1))
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += dur
    a = agg[(name, wgs)]
    a[0] += 1; a[1] += dur
print(f"total kernel time {tot / 1e3:.1f} ms")
rows = sorted(((v[1], k, v[0]) for k, v in agg.items() if k[1] < cap), reverse=True)
for d, (name, wgs), n in rows[:45]:
    print(f"{d / 1e3:8.2f} ms  n={n:5d}  avg {d / n:7.1f} us  wgs={wgs:5d}  {name}")
