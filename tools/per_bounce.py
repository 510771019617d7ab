"""Per-bounce kernel durations from a rocprofv3 kernel-trace CSV (dispatch order = bounce order)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seq = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if "k_extend" in n or "k_trace_persist<false>" in n:
        key = "extend:" + n.split("rt355dev::")[1].split("(")[0]
    elif "k_connect" in n or "k_trace_persist<true>" in n:
        key = "connect:" + n.split("rt355dev::")[1].split("(")[0]
    else:
        continue
    seq[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in seq.items():
    if k.startswith("extend"):
        per = [[] for _ in range(7)]
        for i, d in enumerate(v):
            per[i % 7].append(d)
        print(k, "launches", len(v), "per-bounce median us:", [round(sorted(p)[len(p) // 2], 1) for p in per if p], "sum", round(sum(sorted(p)[len(p) // 2] for p in per if p), 1))
    else:
        print(k, "launches", len(v), "median us", round(sorted(v)[len(v) // 2], 1))
