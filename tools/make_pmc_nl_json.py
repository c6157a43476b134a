"""profiles/r<NN>_pmc_nl.json: matrix-pipe occupancy of the non-local attention kernel from rocprofv3 PMC passes
(SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE in ONE pass, --kernel-trace only), keyed "<H>x<W>_<fp32|bf16>" and stamped with the sha of the
kernel's sources - bench.py prints it as roofline_nl.matrix_pipe_busy only while the kernel is the one that was measured.
busy = SQ_VALU_MFMA_BUSY_CYCLES / (SIMDs x GRBM_GUI_ACTIVE / XCDs): GRBM_GUI_ACTIVE is summed over the 8 XCDs, the busy counter over all SIMDs.
Usage: python tools/make_pmc_nl_json.py out.json key1=db1 [key2=db2 ...]"""
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def busy(db):
    c = sqlite3.connect(db)
    raw = c.execute("select kernel_name, counter_name, value, duration from counters_collection where kernel_name like '%nl_attn_f16_sw_kernel%'").fetchall()
    acc = {}
    for name, ctr, val, dur in raw:
        a = acc.setdefault(ctr, [0, 0.0, 0.0])
        a[0] += 1
        a[1] += val
        a[2] += dur
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in acc or "GRBM_GUI_ACTIVE" not in acc:
        return None
    mf = acc["SQ_VALU_MFMA_BUSY_CYCLES"][1] / acc["SQ_VALU_MFMA_BUSY_CYCLES"][0]
    ga = acc["GRBM_GUI_ACTIVE"][1] / acc["GRBM_GUI_ACTIVE"][0]
    return {"busy": round(mf / (1024.0 * ga / 8.0), 4), "mfma_busy_cycles_per_launch": mf, "gui_active_per_launch": ga,
            "launches": acc["GRBM_GUI_ACTIVE"][0], "avg_us": round(acc["GRBM_GUI_ACTIVE"][2] / acc["GRBM_GUI_ACTIVE"][0] / 1e3, 2)}


def main():
    import bench
    out = {"kernel_src_sha": bench.nl_sources_sha(), "kernel_src_files": ["nonlocal_f16.hip", "nonlocal.hip"], "matrix_pipe_busy": {}, "detail": {},
           "source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace on `python bench.py --steps 1 --warmup 1 ...`; "
                     "busy = MFMA busy cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)"}
    for kv in sys.argv[2:]:
        key, db = kv.split("=", 1)
        b = busy(db)
        if b:
            out["matrix_pipe_busy"][key] = b["busy"]
            out["detail"][key] = b
    json.dump(out, open(sys.argv[1], "w"), indent=1)
    print(open(sys.argv[1]).read())


if __name__ == "__main__":
    main()
