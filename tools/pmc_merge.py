#!/usr/bin/env python3
"""Merge the per-pass summaries of tools/pmc_summary.py (separate rocprofv3 --pmc passes: FETCH_SIZE, WRITE_SIZE,
SQ/GRBM counters) into the per-kernel table committed as profiles/rNN_pmc_summary.json.

HBM bytes per dispatch = 2 x FETCH_SIZE + WRITE_SIZE: on gfx950 rocprofv3's FETCH_SIZE tallies the 128-byte
requests of wide coalesced reads at 64 bytes (/opt/skills/guides/MI355X_MICROARCH.md, HBM section), WRITE_SIZE is
taken as reported (uncalibrated there).  Both counters are in KiB.
    python tools/pmc_merge.py fetch.json write.json sq.json out.json"""
import json
import sys

N_XCD, N_CU, N_SIMD = 8, 256, 1024      # MI355X: 8 XCDs x 32 CUs x 4 SIMDs


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
            # MFMA utilisation = matrix-pipe busy cycles / (SIMDs x kernel cycles).  rocprofv3 reports SQ_VALU_MFMA_BUSY_CYCLES
            # summed over the chip's 1024 SIMDs (checked on dw_h2_kernel<128,128>: 17 104 896 = 32 cycles x 534 528
            # v_mfma_f32_32x32x16_f16, exactly the instruction count of the launch) and GRBM_GUI_ACTIVE summed over the 8 XCDs
            # (816 525 / 8 = 102 k cycles = the launch's duration under the counter run), so
            #     mfma_util = MFMA_BUSY / (1024 * GUI_ACTIVE / 8)
            # -- the fraction of SIMD-cycles in which the matrix pipe is executing, against the clock the kernel actually
            # ran at (a fraction of the DVFS-limited pipe rate, not of the quoted 2.5 PFLOP/s at the maximum clock).
            e["mfma_busy_cycles"] = int(mb)
            e["kernel_cycles"] = int(gui / N_XCD)
            e["mfma_util"] = round(mb / (N_SIMD * gui / N_XCD), 4)
            la2 = g(S, "SQ_LDS_IDX_ACTIVE")
            if la2 is not None:
                e["lds_active_frac"] = round(la2 / (N_CU * gui / N_XCD), 4)       # LDS-array cycles per CU-cycle
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
