"""Summarise a rocprofv3 results database (ROCm 7.2 writes sqlite `*_results.db`) as a
per-kernel table: calls, total / average duration (the views report microseconds), share of GPU time.

    python tools/rocprof_summary.py gpurun_out/prof/r1_results.db > profiles/r01_xxx.md
"""
import sqlite3
import sys


def main(path, top=40):
    c = sqlite3.connect(path)
    rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    total = sum(r[2] for r in rows)
    print(f"source: {path}\n")
    print(f"total kernel time: {total / 1e3:.3f} ms over {sum(r[1] for r in rows)} dispatches\n")
    print("| kernel | calls | total ms | avg us | % |")
    print("|---|---:|---:|---:|---:|")
    for name, calls, tot, avg, pct in rows[:top]:
        name = name.replace("|", "/")
        if len(name) > 96:
            name = name[:93] + "..."
        print(f"| `{name}` | {calls} | {tot / 1e3:.3f} | {avg:.2f} | {pct:.2f} |")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
