"""Per-bounce kernel durations from a rocprofv3 kernel-trace CSV (dispatch order = bounce order within a frame)."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
frames, cur = [], None
for r in rows:
    n = r["Kernel_Name"]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if "k_generate" in n:
        cur = collections.defaultdict(list)
        frames.append(cur)
    if cur is None:
        continue
    for key, pat in (("extend", ("k_extend", "k_trace_persist<false", "k_trace_persist4<false", "k_trace_persist_tlas<false", "k_trace_mixed")), ("shade", ("k_shade",)),
                     ("connect", ("k_connect", "k_trace_persist<true", "k_trace_persist4<true", "k_trace_persist_tlas<true")), ("accumulate", ("k_accumulate",)),
                     ("generate", ("k_generate",))):
        if any(p in n for p in pat):
            cur[key].append(d)
frames = [f for f in frames if len(f["extend"]) == len(frames[len(frames) // 2]["extend"])]
for key in ("generate", "extend", "shade", "connect", "accumulate"):
    m = max(len(f[key]) for f in frames)
    med = []
    for i in range(m):
        v = sorted(f[key][i] for f in frames if len(f[key]) > i)
        med.append(round(v[len(v) // 2], 1))
    print(f"{key:10s} frames {len(frames)} per-launch median us {med}  sum {round(sum(med), 1)}")
