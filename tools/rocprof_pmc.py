"""Per-kernel PMC averages from rocprofv3 rocpd databases (one DB per --pmc pass).
Usage: python tools/rocprof_pmc.py <db> [<db> ...]   -> markdown table per DB"""
import sqlite3
import sys


def main():
    for db in sys.argv[1:]:
        c = sqlite3.connect(db)
        raw = c.execute("select kernel_name, grid_size, counter_name, value, duration from counters_collection").fetchall()
        # the persistent Winograd kernel uses one grid for 28-item and 4-item launches: split a (kernel, grid)
        # group whose durations span more than 3x at the geometric mean of its extremes
        span = {}
        for name, grid, ctr, val, dur in raw:
            lo, hi = span.get((name, grid), (dur, dur))
            span[(name, grid)] = (min(lo, dur), max(hi, dur))
        acc = {}
        for name, grid, ctr, val, dur in raw:
            if "pfnl::" not in name:
                continue
            lo, hi = span[(name, grid)]
            cls = ""
            if hi > 3 * lo:
                cls = " [long]" if dur * dur > lo * hi else " [short]"
            key = (name.split("(")[0].replace("void ", "") + cls, grid)
            a = acc.setdefault((key, ctr), [0, 0.0, 0.0])
            a[0] += 1
            a[1] += val
            a[2] += dur
        table = {}
        for (key, ctr), (n, sv, sd) in acc.items():
            table.setdefault(key, {"n": n, "dur_us": sd / n / 1e3})[ctr] = sv / n
        ctrs = sorted({k for v in table.values() for k in v if k not in ("n", "dur_us")})
        print("\n### %s\n" % db)
        print("| kernel | grid | dispatches | avg us | " + " | ".join(ctrs) + " |")
        print("|---|---|---|---|" + "---|" * len(ctrs))
        for (name, grid), v in sorted(table.items(), key=lambda kv: -kv[1]["dur_us"] * kv[1]["n"]):
            print("| `%s` | %d | %d | %.1f | " % (name, grid, v["n"], v["dur_us"]) +
                  " | ".join("%.4g" % v.get(k, float("nan")) for k in ctrs) + " |")


if __name__ == "__main__":
    main()
