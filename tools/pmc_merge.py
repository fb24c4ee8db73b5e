#!/usr/bin/env python3
"""Merge the per-pass summaries of tools/pmc_summary.py (separate rocprofv3 --pmc passes: FETCH_SIZE, WRITE_SIZE,
SQ/GRBM counters) into the per-kernel table committed as profiles/rNN_pmc_summary.json.

HBM bytes per dispatch = 2 x FETCH_SIZE + WRITE_SIZE: on gfx950 rocprofv3's FETCH_SIZE tallies the 128-byte
requests of wide coalesced reads at 64 bytes (/opt/skills/guides/MI355X_MICROARCH.md, HBM section), WRITE_SIZE is
taken as reported (uncalibrated there).  Both counters are in KiB.
    python tools/pmc_merge.py fetch.json write.json sq.json out.json"""
import json
import sys


def main(fetch, write, sq, out):
    F, W, S = (json.load(open(p)) for p in (fetch, write, sq))
    res = {}
    for k in sorted(set(F) | set(W) | set(S)):
        g = lambda d, c: d.get(k, {}).get(c, {}).get("avg_per_dispatch")
        e = {"dispatches": next((v["dispatches"] for d in (F, W, S) for v in d.get(k, {}).values()), 0)}
        f, w = g(F, "FETCH_SIZE"), g(W, "WRITE_SIZE")
        if f is not None:
            e["FETCH_SIZE_KB"] = round(f, 1)
        if w is not None:
            e["WRITE_SIZE_KB"] = round(w, 1)
        if f is not None and w is not None:
            e["hbm_bytes_per_dispatch"] = int(1024 * (2 * f + w))
        wc = g(S, "SQ_WAVE_CYCLES")
        if wc:
            for name, key in (("SQ_WAIT_ANY", "wait_any_frac"), ("SQ_WAIT_INST_ANY", "wait_inst_frac"),
                              ("SQ_ACTIVE_INST_ANY", "active_frac")):
                v = g(S, name)
                if v is not None:
                    e[key] = round(v / wc, 3)
        bc, la = g(S, "SQ_LDS_BANK_CONFLICT"), g(S, "SQ_LDS_IDX_ACTIVE")
        if bc is not None and la:
            e["lds_bank_conflict_frac"] = round(bc / la, 4)
        mb, gui = g(S, "SQ_VALU_MFMA_BUSY_CYCLES"), g(S, "GRBM_GUI_ACTIVE")
        if mb is not None and gui:
            # raw ratio of the two counters as rocprofv3 aggregates them.  Reading it as a pipe-busy fraction needs the
            # aggregation widths: with GUI_ACTIVE summed over the 8 XCDs and MFMA_BUSY over the 1024 SIMDs,
            # busy fraction = ratio * 8 / 1024 (0.48 for a ratio of 61) -- an assumption, stated in DESIGN.md.
            e["mfma_busy_cycles_over_gui_active"] = round(mb / gui, 2)
        res[k] = e
    # stamp: bench.py attaches these figures to its roofline object only while the kernel sources are the ones measured
    import os, sys as _sys
    _sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    try:
        from bench import _csrc_fingerprint
        res["_meta"] = {"csrc_sha": _csrc_fingerprint(), "hbm_bytes": "1024 * (2 * FETCH_SIZE + WRITE_SIZE), per dispatch"}
    except Exception as e:                       # noqa: BLE001
        res["_meta"] = {"csrc_sha": None, "error": str(e)[:100]}
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    print("wrote", out, len(res) - 1, "kernels")


if __name__ == "__main__":
    main(*sys.argv[1:5])
