"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite output) kernel trace: per kernel name and grid,
calls / total / average / min duration.  Usage: python tools/rocprof_summary.py <results.db> [out.md]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    raw = c.execute("select name, grid_x, grid_y, grid_z, workgroup_x, duration, vgpr_count, accum_vgpr_count, "
                    "lds_size, scratch_size from kernels").fetchall()
    # the persistent Winograd kernel uses one grid for 28-item and 4-item launches: split a (kernel, grid)
    # group whose durations span more than 3x at the geometric mean of its extremes
    span = {}
    for r in raw:
        k = r[:4]
        lo, hi = span.get(k, (r[5], r[5]))
        span[k] = (min(lo, r[5]), max(hi, r[5]))
    groups = {}
    for r in raw:
        lo, hi = span[r[:4]]
        cls = ""
        if hi > 3 * lo:
            cls = " [long]" if r[5] * r[5] > lo * hi else " [short]"
        g = groups.setdefault((r[0] + cls,) + tuple(r[1:5]), [0, 0, None, 0, 0, 0, 0])
        g[0] += 1
        g[1] += r[5]
        g[2] = r[5] if g[2] is None else min(g[2], r[5])
        g[3:] = [max(a, b or 0) for a, b in zip(g[3:], r[6:])]
    rows = sorted(((k[0], k[1], k[2], k[3], k[4], g[0], g[1], g[1] / g[0], g[2], g[3], g[4], g[5], g[6])
                   for k, g in groups.items()), key=lambda r: -r[6])
    total = sum(r[6] for r in rows)
    lines = ["| kernel | grid (threads) | wg | calls | total ms | avg us | min us | % | vgpr | agpr | lds B | scratch |",
             "|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for r in rows:
        name = r[0]
        if len(name) > 78:
            name = name[:67] + "..." + (name[name.index(" ["):] if " [" in name else "")
        lines.append("| `%s` | %dx%dx%d | %d | %d | %.3f | %.1f | %.1f | %.1f | %s | %s | %s | %s |" % (
            name, r[1], r[2], r[3], r[4], r[5], r[6] / 1e6, r[7] / 1e3, r[8] / 1e3, 100.0 * r[6] / total,
            r[9], r[10], r[11], r[12]))
    out = "\n".join(lines)
    pm = c.execute("select count(*) from pmc_events").fetchone()[0]
    if pm:
        out += "\n\nPMC counters (sum over dispatches, per kernel name):\n\n| kernel | counter | dispatches | sum | per dispatch |\n|---|---|---|---|---|\n"
        cur = c.execute("select * from pmc_events limit 1")
        cols = [d[0] for d in cur.description]
        out += "<!-- pmc_events columns: %s -->\n" % cols
    print(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")


if __name__ == "__main__":
    main()
