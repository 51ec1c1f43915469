#!/usr/bin/env python3
"""Per-kernel busy cycles of the vector ALU, the LDS and the memory pipe (SQ_ACTIVE_INST_*) beside the instruction counts, for the kernels of
one joined step (tools/pmc_collect.py's bench command): which kernels keep the VALU busy for how many cycles — an instruction count prices
every instruction at 4 cycles, f64 and quarter-rate integer instructions cost more.   python tools/pmc_active.py [--pairs 64]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pmc_collect as pc  # noqa: E402


def main():
    pairs = 64
    out = {}
    for counters in (["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_ANY", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES"],):
        agg, nd = pc.run_pass(counters, pairs, "full", os.path.join(pc.ROOT, "gpurun_out", "pmc_tmp"), 6000)
        for k, v in agg.items():
            out.setdefault(k, {}).update(v)
    rows = []
    for k, v in out.items():
        unit = 2 * pairs if k.startswith(pc.ORB_KERNELS) else pairs
        per512 = (1024 if k.startswith(pc.ORB_KERNELS) else 512) / unit
        rows.append((k, {c: v.get(c, 0) * per512 for c in v}))
    rows.sort(key=lambda r: -r[1].get("SQ_ACTIVE_INST_VALU", 0))
    res = {"note": "counter totals scaled to one 512-pair step; SQ_ACTIVE_INST_* are summed over all SIMDs", "kernels": {k: v for k, v in rows}}
    json.dump(res, open(os.path.join(pc.ROOT, "gpurun_out", "r04_pmc_active.json"), "w"), indent=1)
    for k, v in rows:
        iv = v.get("SQ_INSTS_VALU", 0); av = v.get("SQ_ACTIVE_INST_VALU", 0)
        print(f"{k:30s} VALU insts {iv / 1e6:9.1f} M  active VALU {av / 1e6:9.1f} Mcyc  cycles/inst {av / max(iv, 1):5.2f}  LDS {v.get('SQ_ACTIVE_INST_LDS', 0) / 1e6:8.1f}  VMEM {v.get('SQ_ACTIVE_INST_VMEM', 0) / 1e6:8.1f}  any {v.get('SQ_ACTIVE_INST_ANY', 0) / 1e6:9.1f}  wave cycles {v.get('SQ_WAVE_CYCLES', 0) / 1e6:10.1f}")


if __name__ == "__main__":
    main()
