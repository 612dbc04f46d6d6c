"""HBM-side bytes per launch of every conv kernel from the FETCH_SIZE / WRITE_SIZE passes of tools/profile_bench.sh
(separate rocprofv3 --pmc runs), keyed by the labels bench.py uses:

    python tools/pmc_traffic.py gpurun_out/prof_<tag> <precision> [--mode infer|train|stress] [--commit sha] [batch size n_embed] > profiles/rNN_<precision>_pmc_traffic.json

FETCH_SIZE is doubled (gfx950 tallies 128-B requests at 64 B: MI355X_MICROARCH.md, HBM section); rocprofv3 reports KB."""
import csv, glob, json, re, sys
from collections import defaultdict

root, prec = sys.argv[1], sys.argv[2]
extra = sys.argv[3:]
mode, commit, digests = "infer", None, None
if "--digests" in extra:          # json of ammcnet_aaai2021_amd.build.file_digests() at profiling time
    i = extra.index("--digests")
    digests = json.loads(extra[i + 1])
    del extra[i:i + 2]
if "--mode" in extra:
    i = extra.index("--mode")
    mode = extra[i + 1]
    del extra[i:i + 2]
if "--commit" in extra:
    i = extra.index("--commit")
    commit = extra[i + 1]
    del extra[i:i + 2]
if mode == "stress":          # bench.py --mode stress: BASELINE.json configs[4] (its --batch counts frames of 1024 rows)
    workload = {"mode": "stress", "rows": 1024 * (int(extra[0]) if extra else 256), "slots": 8192, "dim": 512, "k": 2}
elif mode == "train":         # bench.py --mode train: configs[2]
    workload = {"mode": "train", "batch": int(extra[0]) if extra else 32, "size": int(extra[1]) if len(extra) > 1 else 256,
                "n_embed": 256}
else:
    workload = {"batch": int(extra[0]) if len(extra) > 0 else 16, "size": int(extra[1]) if len(extra) > 1 else 256,
                "n_embed": int(extra[2]) if len(extra) > 2 else 2000}
    commit = commit or (extra[3] if len(extra) > 3 else None)
try:
    command = open(f"{root}/command.txt").read().strip()
except OSError:
    command = "python3 bench.py --steps 5 --warmup 2"


def label(name):
    m0 = re.search(r"conv_up_s16_kernel<(\d+)>", name)
    if m0:
        return f"conv_up_s16<{m0.group(1)}>"
    if "conv_first_s16_kernel" in name:
        return "conv_first_s16"
    if "conv_outc_s16_kernel" in name:
        return "conv_outc_s16"
    if "memory_block_s16_kernel" in name:               # round 6: the whole memory block as one launch
        return "memory_block_s16"
    if "memory_topk_s16_kernel" in name:                # (rocprofv3 leaves this one mangled)
        return "memory_topk_s16"
    if re.search(r"memory_topk_kernel", name):
        return "memory_topk"
    m = re.search(r"(conv_gemm_s16|conv_gemm_f32|conv_tap_s16)_kernel<([^>]*)>", name)
    if not m:
        m2 = re.search(r"(memory_topk_f16r|memory_topk_f16|memory_topk|wgrad_tap3_s16|wgrad_tap_s16|wgrad_s16|wgrad_f32)\w*_kernel(<[^>]*>)?", name)
        return (m2.group(1) + (m2.group(2) or "")) if m2 else None
    if m.group(1) == "conv_tap_s16":                   # labelled by its template arguments <WGM, WGN, TM, TN, AS>
        return "conv_tap_s16<" + ", ".join(v.strip() for v in m.group(2).split(",")) + ">"
    t = [int(v) if v.strip().lstrip("-").isdigit() else 0 for v in m.group(2).split(",")]
    return f"{m.group(1)}<{t[0] * t[2] * 32}x{t[1] * t[3] * 32}>"


acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for path in glob.glob(f"{root}/pmc_fetch/**/*counter_collection.csv", recursive=True) + \
        glob.glob(f"{root}/pmc_write/**/*counter_collection.csv", recursive=True):
    with open(path) as fp:
        for row in csv.DictReader(fp):
            lb = label(row["Kernel_Name"])
            if lb and row["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                a = acc[lb][row["Counter_Name"]]
                a[0] += float(row["Counter_Value"])
                a[1] += 1
out = {"source": f"tools/profile_bench.sh (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes) of "
                 f"`{command}` (precision {prec}); tools/pmc_traffic.py",
       "workload": workload, "commit": commit, "csrc_digests": digests, "kernels": {}}
for lb, c in acc.items():
    f, w = c["FETCH_SIZE"], c["WRITE_SIZE"]
    if not f[1] or not w[1]:
        continue
    fk, wk = f[0] / f[1], w[0] / w[1]
    out["kernels"][lb] = {"dispatches": f[1], "fetch_kb_raw": fk, "write_kb": wk,
                          "traffic_bytes_per_launch": (2.0 * fk + wk) * 1024.0,
                          "note": "FETCH_SIZE doubled (gfx950 counts 128-B requests at 64 B), WRITE_SIZE as is; rocprofv3 reports KB"}
print(json.dumps(out, indent=1))
