"""profiles/r01_traffic_bf16.json from FETCH_SIZE / WRITE_SIZE passes of the bf16 path: HBM bytes per launch of the
conv3x3_bf16_kernel class (gfx950 corrections: FETCH_SIZE KiB x 2, WRITE_SIZE KiB).
usage: python tools/make_traffic_bf16.py <workload> <fetch.db> <write.db> [<workload> <fetch.db> <write.db> ...] <out.json>"""
import json, sqlite3, sys


def per_kernel(db, ctr):
    c = sqlite3.connect(db)
    out = {}
    for name, val in c.execute("select kernel_name, value from counters_collection where counter_name=?", (ctr,)).fetchall():
        if not any(k in name for k in ("conv3x3_bf16_kernel", "conv3x3_bf16_v2_kernel", "conv3x3_bf16_v3_kernel")):
            continue
        if "conv3x3_bf16_kernel<3>" in name:                       # convmerge1 (accumulating mode): not the conv3x3 class of bench.py
            continue
        k = name.split("(")[0].replace("void ", "")
        a = out.setdefault(k, [0, 0.0])
        a[0] += 1
        a[1] += val
    return out


args = sys.argv[1:]
dst = args[-1]
res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace) on `python bench.py --steps 1 --warmup 1 "
                 "--precision bf16 [--workload cfg4]`; FETCH_SIZE x1024 x2 (gfx950 correction), WRITE_SIZE x1024; average over the "
                 "three 3x3 launches of a block (conv1_i + conv10_i, shared half, per-frame half)",
       "hbm_bytes_per_launch_avg": {}, "per_kernel": {}}
for i in range(0, len(args) - 1, 3):
    wl, fdb, wdb = args[i:i + 3]
    f, w = per_kernel(fdb, "FETCH_SIZE"), per_kernel(wdb, "WRITE_SIZE")
    tot_b, tot_n, per = 0.0, 0, {}
    for k in f:
        fb, wb = f[k][1] * 1024 * 2, w[k][1] * 1024
        per[k] = {"dispatches": f[k][0], "fetch_bytes_per_launch": fb / f[k][0], "write_bytes_per_launch": wb / w[k][0]}
        tot_b += (fb / f[k][0] + wb / w[k][0]) * f[k][0]           # (the two passes need not have run the same number of launches)
        tot_n += f[k][0]
    res["hbm_bytes_per_launch_avg"][wl] = round(tot_b / tot_n)
    res["per_kernel"][wl] = per
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (the sha of the kernel sources the passes were measured on)
res["kernel_src_files"] = bench.CONV3X3_KERNELS["bf16"][1]
res["kernel_src_sha"] = bench.kernel_source_sha(res["kernel_src_files"])
json.dump(res, open(dst, "w"), indent=1)
print(json.dumps(res["hbm_bytes_per_launch_avg"]))
