"""Best (tiles, splits) per projection from a tools/ts_bench log:  python tools/ts_tune_pick.py <log> [top-n]"""
import re
import sys

rows = {}
for line in open(sys.argv[1]):
    m = re.match(r"(\S+)\s+M=\s*(\d+)\s+tiles=\s*(\d+)\s+splits=\s*(\d+)\s+wgs=\s*(\d+)\s+([\d.]+) us\s+([\d.]+) TB/s\s+max_err (\S+) (\S+)", line)
    if m:
        name, M, tiles, splits, wgs, us, tbs, err, ok = m.groups()
        rows.setdefault((name, int(M)), []).append((float(us), int(tiles), int(splits), float(tbs), ok))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 5
for key, v in rows.items():
    v.sort()
    print(key, "candidates", len(v))
    for us, tiles, splits, tbs, ok in v[:top]:
        print(f"    {tiles:5d} x {splits}   {us:8.1f} us  {tbs:5.2f} TB/s  {ok}")
