"""profiles/r<NN>_traffic[_split16].json from the FETCH_SIZE / WRITE_SIZE rocprofv3 passes (tools/run_pmc.sh):
HBM bytes per launch of the 3x3 convolution kernels, with the gfx950 corrections of
/opt/skills/guides/MI355X_MICROARCH.md (FETCH_SIZE x1024 x2, WRITE_SIZE x1024).
Usage: python tools/make_traffic_json.py <fetch.db> <write.db> <algo> <out.json>"""
import json
import sqlite3
import sys


def per_kernel(db, ctr):
    c = sqlite3.connect(db)
    raw = c.execute("select kernel_name, grid_size, value, duration from counters_collection where counter_name=?", (ctr,)).fetchall()
    span = {}
    for name, grid, val, dur in raw:
        lo, hi = span.get((name, grid), (dur, dur))
        span[(name, grid)] = (min(lo, dur), max(hi, dur))
    out = {}
    for name, grid, val, dur in raw:
        if "conv_wino" not in name and "conv_mfma_kernel<3, 16, 2" not in name and "conv3x3_split16" not in name and "conv3x3_sf" not in name and "conv3x3_c1c10" not in name:
            continue
        lo, hi = span[(name, grid)]
        cls = ""
        if hi > 3 * lo:
            cls = " [long]" if dur * dur > lo * hi else " [short]"
        k = name.split("(")[0].replace("void ", "") + cls
        a = out.setdefault(k, [0, 0.0])
        a[0] += 1
        a[1] += val
    return out


def main():
    fdb, wdb, algo, dst = sys.argv[1:5]
    f = per_kernel(fdb, "FETCH_SIZE")
    w = per_kernel(wdb, "WRITE_SIZE")
    per = {}
    tot_b = 0.0
    tot_n = 0
    for k in f:
        if "ws_kernel<3>" in k or "split16_kernel<2" in k:        # convmerge1 (accumulating mode): not the conv3x3 class of bench.py
            continue
        n = f[k][0]
        fb = f[k][1] / n * 1024 * 2
        wb = w[k][1] / w[k][0] * 1024
        per[k] = {"dispatches": n, "fetch_bytes": round(fb, 1), "write_bytes": round(wb, 1)}
        tot_b += (fb + wb) * n
        tot_n += n
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench                                             # the sha of the kernel sources the passes were measured on
    files = bench.CONV3X3_KERNELS[algo][1]
    json.dump({"algo": algo, "kernel_src_sha": bench.kernel_source_sha(files), "kernel_src_files": files,
               "hbm_bytes_per_launch_avg": int(tot_b / tot_n),
               "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace) on "
                         "`python bench.py --steps 1 --warmup 1`; FETCH_SIZE x1024 x2 (gfx950 correction), WRITE_SIZE x1024",
               "per_kernel": per}, open(dst, "w"), indent=1)
    print(open(dst).read())


if __name__ == "__main__":
    main()
