"""profiles/extend_traffic.json from a tools/pmc.sh summary: DRAM-side bytes per extend launch of a frame (1 x k_trace_persist<false, true>
for bounce 0 + 6 x k_trace_persist<false, false>), FETCH_SIZE doubled as MI355X_MICROARCH.md (HBM section) prescribes for gfx950, + WRITE_SIZE.
usage: python tools/make_traffic.py gpurun_out/<dir>/pmc_summary.csv "<how it was collected>" """
import csv, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = list(csv.DictReader(open(sys.argv[1])))
get = lambda k, c: next(float(r["per_dispatch"]) for r in rows if k in r["kernel"] and r["counter"] == c)
b0, b16 = "k_trace_persist<false, true", "k_trace_persist<false, false"
fetch = (get(b0, "FETCH_SIZE") + 6 * get(b16, "FETCH_SIZE")) / 7
write = (get(b0, "WRITE_SIZE") + 6 * get(b16, "WRITE_SIZE")) / 7
out = {"kernel": "extend = the 7 extend launches of a frame: k_trace_persist<false, true> (bounce 0) + 6 x k_trace_persist<false, false>",
       "config": {"accel": "bvh2", "detail": 1.0, "width": 1920, "height": 1080},
       "source": sys.argv[2], "FETCH_SIZE_KB_per_launch": round(fetch, 1), "WRITE_SIZE_KB_per_launch": round(write, 1),
       "correction": "gfx950: FETCH_SIZE reads half of a wide coalesced stream (MI355X_MICROARCH.md, HBM section) -> doubled; uncalibrated for this "
                     "16-B gather pattern, so the figure is an upper estimate; Infinity-Cache hits are counted by FETCH_SIZE",
       "hbm_bytes_per_launch": int(round((2 * fetch + write) * 1024))}
json.dump(out, open(os.path.join(ROOT, "profiles", "extend_traffic.json"), "w"), indent=1)
print(out["hbm_bytes_per_launch"], "bytes per extend launch (FETCH x 2 + WRITE)")
