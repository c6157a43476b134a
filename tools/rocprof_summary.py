"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite output) kernel trace: per kernel name and grid,
calls / total / average / min duration.  Usage: python tools/rocprof_summary.py <results.db> [out.md]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    rows = c.execute(
        "select name, grid_x, grid_y, grid_z, workgroup_x, count(*), sum(duration), avg(duration), min(duration), "
        "max(vgpr_count), max(accum_vgpr_count), max(lds_size), max(scratch_size) from kernels "
        "group by name, grid_x, grid_y, grid_z order by sum(duration) desc").fetchall()
    total = sum(r[6] for r in rows)
    lines = ["| kernel | grid (threads) | wg | calls | total ms | avg us | min us | % | vgpr | agpr | lds B | scratch |",
             "|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for r in rows:
        name = r[0]
        if len(name) > 70:
            name = name[:67] + "..."
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
