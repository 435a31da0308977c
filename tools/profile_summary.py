#!/usr/bin/env python3
"""Writes profiles/<name>.txt from the outputs of tools/final_profile.sh <tag> <workload> <spp> <size> (merged back under
gpurun_out/) and records the HBM traffic of the PMC launch in profiles/pmc_traffic.json under the workload's name.
Usage: python tools/profile_summary.py <tag> <name> [workload spp WxH]"""
import io
import json
import os
import sqlite3
import sys
from contextlib import redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import rocpd_summary as R  # noqa: E402


def counters(db, kernel_like):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    namecol = "kernel_name" if "kernel_name" in cols else "name"
    return dict(c.execute(f"select counter_name, sum(value) from counters_collection where {namecol} like ? group by 1", (kernel_like,)).fetchall())


def kname_hint(out, tag):
    return json.loads(open(os.path.join(out, f"bench_{tag}.json")).read().strip().split("\n")[-1])["roofline"]["kernel"]


def main(tag, name, workload="cornell", spp=256, size="1024x1024"):
    w, h = (int(x) for x in size.split("x"))
    out = os.path.join(ROOT, "gpurun_out")
    prof = os.path.join(out, "prof")
    buf = io.StringIO()
    with redirect_stdout(buf):
        print(f"# {name}: rocprofv3 summaries, one MI355X (tools/final_profile.sh {tag})")
        print(f"# (a) rocprofv3 --kernel-trace --stats -- python3 bench.py --workload {workload} --no-cpu-baseline   (1 warmup + 3 timed + 3 instrumented renders)")
        R.kernel_stats(os.path.join(prof, f"{tag}_kt", "b_results.db"))
        print("\n# bench line of the profiled run:")
        print(open(os.path.join(out, f"bench_{tag}_prof.json")).read().strip().split("\n")[-1])
        print("\n# bench line of the unprofiled run (with the CPU baseline):")
        print(open(os.path.join(out, f"bench_{tag}.json")).read().strip().split("\n")[-1])
        pm = os.path.join(prof, f"{tag}_pmc")
        paths = float(w) * h * spp
        print(f"\n# (b) WORLD={workload} MIRT_SIZE={size} tools/pmc.sh {spp} {tag}_pmc: one render at {spp} spp (one k_trace launch of {int(paths):,} paths), separate --pmc passes")
        R.kernel_stats(os.path.join(pm, "kt", "c_results.db"))
        c = {}
        for p in ("sq1", "sq2", "sq3", "grbm", "fetch", "write"):
            c.update(counters(os.path.join(pm, p, "c_results.db"), "%k_trace%"))
        for k in sorted(c):
            print(f"k_trace  {k:26s} {c[k]:22.0f}")
        kt = sqlite3.connect(os.path.join(pm, "kt", "c_results.db")).execute("select sum(end-start) from kernels where name like '%k_trace%'").fetchone()[0] / 1e6
        rays = paths * json.loads(open(os.path.join(out, f"bench_{tag}.json")).read().strip().split("\n")[-1])["rays_per_sample"]
        simd_cycles = c["GRBM_GUI_ACTIVE"] / 8 * 1024
        print(f"\n# Derived ({kname_hint(out, tag)}):")
        print(f"#   launch {kt:.2f} ms for {paths / 1e6:.1f} M paths, {rays / 1e6:.1f} M rays -> {rays / kt / 1e6:.2f} Grays/s")
        print(f"#   VALU instructions per 64 rays {c['SQ_INSTS_VALU'] * 64 / rays:.0f}  (SALU {c['SQ_INSTS_SALU'] * 64 / rays:.0f}, LDS {c['SQ_INSTS_LDS'] * 64 / rays:.0f})")
        print(f"#   VALU lane utilisation  THREAD_CYCLES_VALU / (ACTIVE_INST_VALU x 64) = {100 * c['SQ_THREAD_CYCLES_VALU'] / (c['SQ_ACTIVE_INST_VALU'] * 64):.1f} %")
        print(f"#   issued lane slots {c['SQ_INSTS_VALU'] * 64 / kt / 1e9:.1f} T/s of the 78.6 T/s peak (one wave64 VALU op per 2 cycles per SIMD)")
        print(f"#   SIMD cycles: GRBM_GUI_ACTIVE/8 x 1024 SIMDs = {simd_cycles / 1e9:.1f} G; VALU issue (2 cycles each) fills {100 * 2 * c['SQ_INSTS_VALU'] / simd_cycles:.1f} % of them")
        wc = c["SQ_WAVE_CYCLES"]
        print(f"#   wave-cycles: issuing {100 * c['SQ_ACTIVE_INST_ANY'] / wc:.1f} %, issue-stalled (pipe taken by another wave) {100 * c['SQ_WAIT_INST_ANY'] / wc:.1f} %, waiting (s_waitcnt) {100 * c['SQ_WAIT_ANY'] / wc:.1f} %")
        if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE"):
            print(f"#   lds_bank_conflict_rate = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = {100 * c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE']:.1f} %"
                  f" ({100 * c['SQ_LDS_BANK_CONFLICT'] / wc:.2f} % of the wave-cycles; waiting on LDS {100 * c.get('SQ_WAIT_INST_LDS', 0) / wc:.2f} % of them; see profiles/r03_lds_conflicts.txt)")
        if "SQ_INSTS_BRANCH" in c:
            print(f"#   branches per 64 rays {c['SQ_INSTS_BRANCH'] * 64 / rays:.0f}, scalar memory instructions {c.get('SQ_INSTS_SMEM', 0) * 64 / rays:.0f}")
        hbm = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
        slot_bytes = paths * 12.0                                  # one 12-byte radiance slot per path (three floats, rt_kernels.h RadSlot)
        written = c["WRITE_SIZE"] * 1024
        print(f"#   HBM: FETCH_SIZE {c['FETCH_SIZE']:.0f} KB (x2 on gfx950), WRITE_SIZE {c['WRITE_SIZE']:.0f} KB -> {hbm / 1e9:.2f} GB per launch = {hbm / kt / 1e6:.0f} GB/s")
        print(f"#   written / algorithmic = {written / 1e9:.2f} GB / {slot_bytes / 1e9:.2f} GB of 12-byte radiance slots = {written / slot_bytes:.2f}x"
              f" (each lane ends its path at its own time and stores its own 12 bytes: partial 64-byte writes)")
        kname = json.loads(open(os.path.join(out, f"bench_{tag}.json")).read().strip().split("\n")[-1])["roofline"]["kernel"]
        path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        try:
            allw = json.load(open(path))
            if "kernel" in allw:                                   # the round-1 flat form held Cornell only
                allw = {"cornell": allw}
        except (OSError, ValueError):
            allw = {}
        allw[workload] = {"kernel": kname, "workload": f"{workload} {size}, {spp} spp in one launch (tools/pmc.sh {spp} {tag}_pmc)",
                          "paths_per_launch": int(paths), "fetch_size_kb": c["FETCH_SIZE"], "write_size_kb": c["WRITE_SIZE"], "hbm_bytes_per_launch": hbm,
                          "note": "separate --pmc passes (FETCH_SIZE, WRITE_SIZE); KB units; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of wide reads)"}
        json.dump(allw, open(path, "w"), indent=1)
    open(os.path.join(ROOT, "profiles", f"{name}.txt"), "w").write(buf.getvalue())
    print(buf.getvalue()[-1500:])


if __name__ == "__main__":
    if len(sys.argv) >= 6:
        main(sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5])
    else:
        main(sys.argv[1], sys.argv[2])
