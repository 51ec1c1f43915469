"""bench/profiles.py — the committed measurements bench.py normalises against: the newest counter summary (profiles/r<NN>_pmc_*.json) and the newest
machine-peak file (profiles/r<NN>_peaks.json)."""
import glob
import json
import os
import re

from .config import ROOT, SYMBOL


def pmc_file():
    """The newest committed counter summary of this build family (tools/pmc_collect.py writes it): profiles/r<NN>_pmc_<tag>.json."""
    files = [f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_*.json")) if re.search(r"r(\d+)_pmc_[^/]*\.json$", f) and "traffic" not in f and "mfma" not in f]
    if not files:
        return None, None

    def key(f):
        m = re.search(r"r(\d+)_pmc_.*?(\d+)\.json$", f)
        return (int(m.group(1)), int(m.group(2))) if m else (0, 0)
    f = max(files, key=key)
    try:
        d = json.load(open(f))
    except Exception:
        return None, None
    return (d, os.path.relpath(f, ROOT)) if "kernels" in d and "calibration" in d else (None, None)


def peaks_file():
    """The newest committed machine-peak measurement (tools/peaks.py): profiles/r<NN>_peaks.json -> (summary dict, relative path)."""
    files = glob.glob(os.path.join(ROOT, "profiles", "r*_peaks.json"))
    best = (None, None, -1)
    for f in files:
        m = re.search(r"r(\d+)_peaks\.json$", f)
        if not m or int(m.group(1)) <= best[2]:
            continue
        try:
            d = json.load(open(f))
            best = (d["summary"], os.path.relpath(f, ROOT), int(m.group(1)))
        except Exception:
            pass
    return best[0], best[1]


def pmc_lookup(pmc, slot):
    """(full kernel symbol, per-image record) of the profiling slot's kernel in the counter summary"""
    if not pmc:
        return None, None
    pre = SYMBOL.get(slot)
    for name, rec in pmc["kernels"].items():
        if pre and name.startswith(pre):
            return name, rec
    return None, None
