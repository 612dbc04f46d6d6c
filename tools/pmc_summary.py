"""Per-kernel averages of rocprofv3 --pmc counter_collection CSVs, and the derived MFMA busy fraction.

    python tools/pmc_summary.py [--json out.json] [--workload "text"] [--commit sha] gpurun_out/prof_<tag>/pmc_*/**/*counter_collection.csv

Kernel names are demangled (rocprofv3 leaves some mangled: `_ZN9ammc_impl22memory_topk_f16_kernelILi2EEEv...`) and
shortened to `namespace::kernel<template args>`; only this library's kernels (namespaces ammc_*) are kept.

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KB; on gfx950 FETCH_SIZE counts 128-B requests at 64 B
(MI355X_MICROARCH.md, HBM section): tools/pmc_traffic.py doubles it.

mfma_busy_frac (per kernel, from per-launch averages of counters taken in SEPARATE passes of the same command):
    (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs)
= the share of the launch's shader cycles, at the clock the chip actually held, in which a SIMD's matrix pipe was busy.
"""
import csv
import json
import re
import subprocess
import sys
from collections import defaultdict

N_SIMD = 256 * 4
N_XCD = 8
_demangled = {}


def _itanium_lite(name: str):
    """`_ZN<len>ns<len>kernel[I<literal args>E]E...` -> `ns::kernel<args>` for the shapes this library has (integer / bool
    template literals only); binutils' c++filt does not know `DF16_` (_Float16) and returns such names unchanged."""
    m = re.match(r"_ZN", name)
    if not m:
        return None
    i, parts = 3, []
    while i < len(name) and name[i].isdigit():
        j = i
        while name[j].isdigit():
            j += 1
        n = int(name[i:j])
        parts.append(name[j:j + n])
        i = j + n
    if not parts:
        return None
    out = "::".join(parts)
    if i < len(name) and name[i] == "I":                      # template argument list of literals: L<type><value>E
        i += 1
        args = []
        while i < len(name) and name[i] == "L":
            m2 = re.match(r"L([a-z])(n?)(\d+)E", name[i:])
            if not m2:
                return out
            v = ("-" if m2.group(2) else "") + m2.group(3)
            args.append({"0": "false", "1": "true"}[v] if m2.group(1) == "b" else v)
            i += m2.end()
        out += "<" + ", ".join(args) + ">"
    return out


def demangle(name: str) -> str:
    if not name.startswith("_Z"):
        return name
    if name not in _demangled:
        out = None
        try:
            out = subprocess.run(["c++filt", name], stdout=subprocess.PIPE, timeout=10).stdout.decode().strip()
        except Exception:
            pass
        if not out or out == name:
            out = _itanium_lite(name)
        _demangled[name] = out or name
    return _demangled[name]


def short(name: str) -> str:
    """`void ammc_s16::conv_tap_s16_kernel<4, 1, 2, 4, 1, 0>(ammc_s16::TapArgs)` -> `ammc_s16::conv_tap_s16_kernel<4, 1, 2, 4, 1, 0>`"""
    name = demangle(name.strip('"'))
    name = re.sub(r"^void\s+", "", name)
    depth = 0
    for i, ch in enumerate(name):                 # cut the argument list: the first '(' outside template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return name[:i]
    return name


def collect(paths):
    acc = defaultdict(lambda: [0.0, 0])
    for path in paths:
        with open(path) as fp:
            for row in csv.DictReader(fp):
                k = short(row["Kernel_Name"])
                if not k.startswith("ammc"):
                    continue
                a = acc[(k, row["Counter_Name"])]
                a[0] += float(row["Counter_Value"])
                a[1] += 1
    return acc


def derived(acc):
    per = defaultdict(dict)
    for (k, c), (s, n) in acc.items():
        per[k][c] = s / n
        per[k].setdefault("dispatches", n)
    out = {}
    for k, c in per.items():
        row = {"dispatches": c["dispatches"]}
        for name in ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
                     "SQ_ACTIVE_INST_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_VALU_MFMA_MOPS_F32"):
            if name in c:
                row[name] = c[name]
        if c.get("GRBM_GUI_ACTIVE") and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            row["mfma_busy_frac"] = round((c["SQ_VALU_MFMA_BUSY_CYCLES"] / N_SIMD) / (c["GRBM_GUI_ACTIVE"] / N_XCD), 4)
        if c.get("SQ_WAVE_CYCLES"):
            for name, key in (("SQ_WAIT_ANY", "wait_any_frac"), ("SQ_WAIT_INST_ANY", "wait_inst_frac"),
                              ("SQ_ACTIVE_INST_ANY", "active_inst_frac")):
                if name in c:
                    row[key] = round(c[name] / c["SQ_WAVE_CYCLES"], 4)
        if c.get("SQ_LDS_IDX_ACTIVE") and "SQ_LDS_BANK_CONFLICT" in c:
            row["lds_conflict_frac"] = round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], 4)
        out[k] = row
    return out


def main(argv):
    js = workload = commit = digests = None
    paths = []
    it = iter(argv)
    for a in it:
        if a == "--json":
            js = next(it)
        elif a == "--workload":
            workload = next(it)
        elif a == "--commit":
            commit = next(it)
        elif a == "--digests":          # json of ammcnet_aaai2021_amd.build.file_digests() at profiling time
            digests = json.loads(next(it))
        else:
            paths.append(a)
    acc = collect(paths)
    print(f"{'kernel':<70} {'counter':<30} {'dispatches':>10} {'avg':>16} {'sum':>18}")
    for (k, c), (s, n) in sorted(acc.items(), key=lambda kv: (kv[0][1], -kv[1][0])):
        print(f"{k[:70]:<70} {c:<30} {n:>10} {s / n:>16.2f} {s:>18.1f}")
    der = derived(acc)
    print()
    print(f"{'kernel':<70} {'mfma_busy':>10} {'wait_any':>9} {'wait_inst':>10} {'active':>8} {'lds_confl':>10}")
    for k, r in sorted(der.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0) * kv[1]["dispatches"]):
        def g(key):
            return f"{r[key]:.3f}" if key in r else "-"
        print(f"{k[:70]:<70} {g('mfma_busy_frac'):>10} {g('wait_any_frac'):>9} {g('wait_inst_frac'):>10} {g('active_inst_frac'):>8} {g('lds_conflict_frac'):>10}")
    if js:
        with open(js, "w") as fp:
            json.dump({"source": "tools/pmc_summary.py over separate rocprofv3 --pmc passes (tools/profile_bench.sh)",
                       "formula": "mfma_busy_frac = (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs), per-launch averages",
                       "workload": workload, "commit": commit, "csrc_digests": digests, "kernels": der}, fp, indent=1)


if __name__ == "__main__":
    main(sys.argv[1:])
