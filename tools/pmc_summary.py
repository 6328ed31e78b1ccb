"""Average PMC counter values per (kernel, grid size) from rocprofv3 results databases (`--pmc X --kernel-trace`).

    python tools/pmc_summary.py out.json name1=path1_results.db name2=path2_results.db ...
"""
import json
import sqlite3
import sys


def summarise(path):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    name_col = "kernel_name" if "kernel_name" in cols else [x for x in cols if "name" in x and "counter" not in x][0]
    grid = "grid_size" if "grid_size" in cols else ("grid_size_x" if "grid_size_x" in cols else None)
    q = f"select {name_col}, {grid or '0'}, counter_name, avg(value), count(*) from counters_collection group by 1, 2, 3"
    out = {}
    for kname, g, cname, val, n in c.execute(q):
        out.setdefault(f"{kname}|grid={g}", {})[cname] = dict(avg=val, launches=n)
    return out


if __name__ == "__main__":
    res = {}
    for arg in sys.argv[2:]:
        tag, path = arg.split("=", 1)
        res[tag] = summarise(path)
    with open(sys.argv[1], "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res, indent=1)[:3000])
