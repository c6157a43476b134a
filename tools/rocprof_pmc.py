"""Per-kernel PMC averages from rocprofv3 rocpd databases (one DB per --pmc pass).
Usage: python tools/rocprof_pmc.py <db> [<db> ...]   -> markdown table per DB"""
import sqlite3
import sys


def main():
    for db in sys.argv[1:]:
        c = sqlite3.connect(db)
        rows = c.execute("select kernel_name, grid_size, counter_name, count(*), avg(value), avg(duration) "
                         "from counters_collection group by kernel_name, grid_size, counter_name").fetchall()
        table = {}
        for name, grid, ctr, n, val, dur in rows:
            if "pfnl::" not in name:
                continue
            key = (name.split("(")[0].replace("void ", ""), grid)
            table.setdefault(key, {"n": n, "dur_us": dur / 1e3})[ctr] = val
        ctrs = sorted({k for v in table.values() for k in v if k not in ("n", "dur_us")})
        print("\n### %s\n" % db)
        print("| kernel | grid | dispatches | avg us | " + " | ".join(ctrs) + " |")
        print("|---|---|---|---|" + "---|" * len(ctrs))
        for (name, grid), v in sorted(table.items(), key=lambda kv: -kv[1]["dur_us"] * kv[1]["n"]):
            print("| `%s` | %d | %d | %.1f | " % (name, grid, v["n"], v["dur_us"]) +
                  " | ".join("%.4g" % v.get(k, float("nan")) for k in ctrs) + " |")


if __name__ == "__main__":
    main()
