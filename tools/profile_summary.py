#!/usr/bin/env python3
"""Writes profiles/<name>.txt from the outputs of tools/final_profile.sh <tag> (merged back under gpurun_out/).
Usage: python tools/profile_summary.py <tag> <name>"""
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


def main(tag, name):
    out = os.path.join(ROOT, "gpurun_out")
    prof = os.path.join(out, "prof")
    buf = io.StringIO()
    with redirect_stdout(buf):
        print(f"# {name}: rocprofv3 summaries, one MI355X (tools/final_profile.sh {tag})")
        print("# (a) rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline   (Cornell 1024^2 @1024 spp: 1 warmup + 3 timed + 1 instrumented render)")
        R.kernel_stats(os.path.join(prof, f"{tag}_kt", "b_results.db"))
        print("\n# bench line of the profiled run:")
        print(open(os.path.join(out, f"bench_{tag}_prof.json")).read().strip().split("\n")[-1])
        print("\n# bench line of the unprofiled run (with the CPU baseline):")
        print(open(os.path.join(out, f"bench_{tag}.json")).read().strip().split("\n")[-1])
        pm = os.path.join(prof, f"{tag}_pmc")
        print(f"\n# (b) tools/pmc.sh 256 {tag}_pmc: one render at 256 spp (one k_trace launch of 268,435,456 paths), separate --pmc passes")
        R.kernel_stats(os.path.join(pm, "kt", "c_results.db"))
        c = {}
        for p in ("sq1", "sq2", "sq3", "grbm", "fetch", "write"):
            c.update(counters(os.path.join(pm, p, "c_results.db"), "%k_trace%"))
        for k in sorted(c):
            print(f"k_trace  {k:26s} {c[k]:22.0f}")
        kt = sqlite3.connect(os.path.join(pm, "kt", "c_results.db")).execute("select sum(end-start) from kernels where name like '%k_trace%'").fetchone()[0] / 1e6
        paths = 268435456.0
        rays = paths * json.loads(open(os.path.join(out, f"bench_{tag}.json")).read().strip().split("\n")[-1])["rays_per_sample"]
        simd_cycles = c["GRBM_GUI_ACTIVE"] / 8 * 1024
        print("\n# Derived (k_trace):")
        print(f"#   launch {kt:.2f} ms for {paths / 1e6:.1f} M paths, {rays / 1e6:.1f} M rays -> {rays / kt / 1e6:.2f} Grays/s")
        print(f"#   VALU instructions per 64 rays {c['SQ_INSTS_VALU'] * 64 / rays:.0f}  (SALU {c['SQ_INSTS_SALU'] * 64 / rays:.0f}, LDS {c['SQ_INSTS_LDS'] * 64 / rays:.0f})")
        print(f"#   VALU lane utilisation  THREAD_CYCLES_VALU / (ACTIVE_INST_VALU x 64) = {100 * c['SQ_THREAD_CYCLES_VALU'] / (c['SQ_ACTIVE_INST_VALU'] * 64):.1f} %")
        print(f"#   issued lane slots {c['SQ_INSTS_VALU'] * 64 / kt / 1e9:.1f} T/s of the 78.6 T/s peak (one wave64 VALU op per 2 cycles per SIMD)")
        print(f"#   SIMD cycles: GRBM_GUI_ACTIVE/8 x 1024 SIMDs = {simd_cycles / 1e9:.1f} G; VALU issue (2 cycles each) fills {100 * 2 * c['SQ_INSTS_VALU'] / simd_cycles:.1f} % of them")
        wc = c["SQ_WAVE_CYCLES"]
        print(f"#   wave-cycles: issuing {100 * c['SQ_ACTIVE_INST_ANY'] / wc:.1f} %, issue-stalled (pipe taken by another wave) {100 * c['SQ_WAIT_INST_ANY'] / wc:.1f} %, waiting (s_waitcnt) {100 * c['SQ_WAIT_ANY'] / wc:.1f} %")
        hbm = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
        print(f"#   HBM: FETCH_SIZE {c['FETCH_SIZE']:.0f} KB (x2 on gfx950), WRITE_SIZE {c['WRITE_SIZE']:.0f} KB -> {hbm / 1e9:.2f} GB per launch for {paths * 16 / 1e9:.2f} GB of radiance slots = {hbm / kt / 1e6:.0f} GB/s")
        json.dump({"kernel": "k_trace<true,false>", "workload": f"Cornell box 3-ball 1024x1024, 256 spp in one launch (tools/pmc.sh 256 {tag}_pmc)",
                   "paths_per_launch": int(paths), "fetch_size_kb": c["FETCH_SIZE"], "write_size_kb": c["WRITE_SIZE"], "hbm_bytes_per_launch": hbm,
                   "note": "separate --pmc passes (FETCH_SIZE, WRITE_SIZE); KB units; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of wide reads)"},
                  open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
    open(os.path.join(ROOT, "profiles", f"{name}.txt"), "w").write(buf.getvalue())
    print(buf.getvalue()[-1500:])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
