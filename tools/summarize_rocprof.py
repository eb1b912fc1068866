"""Compact a rocprofv3 `*_kernel_stats.csv` (kernel names can be kilobytes long) into profiles/<name>.csv."""
import csv
import re
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:]+(<[^(]{0,40}>)?)", name)
    return (m.group(1) if m else name)[:90]


def main(src, dst, top=40):
    rows = list(csv.DictReader(open(src)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_ms", "avg_us", "pct", "min_us", "max_us"])
        for r in rows[:top]:
            w.writerow([short(r["Name"]), r["Calls"], f"{float(r['TotalDurationNs']) / 1e6:.3f}", f"{float(r['AverageNs']) / 1e3:.2f}",
                        f"{100 * float(r['TotalDurationNs']) / tot:.2f}", f"{float(r['MinNs']) / 1e3:.2f}", f"{float(r['MaxNs']) / 1e3:.2f}"])
    print(open(dst).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 40)
