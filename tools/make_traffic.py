"""profiles/r03_traffic.json from the round's PMC passes: bytes per launch / per frame at every memory level, for bench.py's roofline block.

usage: python tools/make_traffic.py            (reads the CSVs listed in RUNS below from profiles/, writes profiles/r03_traffic.json)

Inputs (all committed under profiles/):
  r03_pmc_calibration.csv    tools/r3_pmc.sh <dir> calib: the counters on three kernels of KNOWN byte counts (tools/pmc_calib.hip)
  r03_pmc_<label>.csv        tools/r3_pmc.sh <dir> <bench args>: per kernel and counter, summed over the dispatches of a short bench run
                             (7 separate rocprofv3 --pmc passes, --kernel-trace only)

What a count is worth, measured by the calibration (1 GiB streamed with 16 B per lane; 1.88 GB of dependent random 64-byte record fetches,
4 x global_load_dwordx4 per lane and record, from a 4 MB and a 64 MB table - the traversal's own access shape):
  FETCH_SIZE       streaming read: reports exactly 1/2 of the bytes (MI355X_MICROARCH.md, HBM section: doubled, as the guide prescribes);
                   the 64-byte requests of the record gather are counted in full (2,800 B per KB-count on the 64 MB table at 42 % L2 hits),
                   so for a kernel that mixes both, 2 x FETCH_SIZE is an UPPER estimate of the fabric-side bytes and FETCH_SIZE a lower one
  WRITE_SIZE       exact
  TCC_REQ_sum      one per L2 request: 128 B each for the streaming read, 64 B each for the streaming store and for the record gather
                   (64.2 B per request measured) -> l2 bytes = requests x 64 B (the traversal's requests are record fetches; its ray-queue
                   streams, ~10 % of the requests, move 128 B each, so this is a LOWER estimate by at most that much)
  TCP_TOTAL_CACHE_ACCESSES_sum   64 B per access for the streaming read and store; 1.25 accesses per 16-byte lane-load in the record gather
                   = 12.8 B delivered per access -> vl1d bytes = accesses x 12.8 B (bytes the vector L1 delivered to the lanes)
  SQ_INSTS_VALU / SQ_ACTIVE_INST_VALU   wave-level vector instructions: calib_valu issues a known number of v_fma_f32 (8 waves per SIMD, four
                   independent chains per lane, nothing else), calib_valu_mix the slab test's own mix (sub, mul, min, max, cmp, cndmask, add_u32,
                   and_b32) - both counters equal the known count to 0.006 %, and the run times (profiles/r03_pmc_calibration_kernel_stats.csv:
                   2.40 / 2.31 ms for 2,147 M instructions) give the chip's MEASURED vector issue rate: 893 / 928 G wave-instructions/s
                   = 0.43 per clock and SIMD at the 2.04 GHz the chip holds under that load (a wave64 FP32 instruction takes 2 clocks on
                   the 32-lane FP32 datapath; 1,024 SIMDs x 2.4 GHz / 2 = 1,229 G/s on paper)
Levels and peaks (MI355X_MICROARCH.md): hbm 8 TB/s; l2 34.5 TB/s aggregate; vl1d 64 B per clock and CU (x 256 CUs x 2.4 GHz);
vector issue: the measured 928 G wave-instructions/s.
"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")

# label -> (bench arguments of the passes, the config bench.py matches against)
RUNS = {
    "config3_lanes1": ("--lanes 1", dict(config=3, accel="bvh2", detail=1.0, width=1920, height=1080, model=False, lanes=1)),
    "config3_lanes4": ("(default: 4 lanes)", dict(config=3, accel="bvh2", detail=1.0, width=1920, height=1080, model=False, lanes=4)),
    "config4_lanes1": ("--config 4 --lanes 1", dict(config=4, accel="bvh4", detail=1.0, width=1920, height=1080, model=False, lanes=1)),
    "config4_lanes4": ("--config 4", dict(config=4, accel="bvh4", detail=1.0, width=1920, height=1080, model=False, lanes=4)),
    "config5_lanes1": ("--config 5 --lanes 1", dict(config=5, accel="bvh2", detail=1.0, width=3840, height=2160, model=False, lanes=1)),
    "config5_lanes4": ("--config 5 --lanes 4", dict(config=5, accel="bvh2", detail=1.0, width=3840, height=2160, model=False, lanes=4)),
    "config5_lanes6": ("--config 5 --lanes 6", dict(config=5, accel="bvh2", detail=1.0, width=3840, height=2160, model=False, lanes=6)),
    "config5_lanes8": ("--config 5", dict(config=5, accel="bvh2", detail=1.0, width=3840, height=2160, model=False, lanes=8)),
    "config2_lanes8": ("--config 2", dict(config=2, accel="bvh2", detail=1.0, width=1280, height=720, model=False, lanes=8)),
    "config2_lanes1": ("--config 2 --lanes 1", dict(config=2, accel="bvh2", detail=1.0, width=1280, height=720, model=False, lanes=1)),
}
EXTEND = ("k_trace_persist<false", "k_trace_persist4<false", "k_trace_persist_tlas<false", "k_extend<")
CONNECT = ("k_trace_persist<true", "k_trace_persist4<true", "k_trace_persist_tlas<true", "k_connect<")
L2_REQ_BYTES, VL1D_ACCESS_BYTES, FETCH_FACTOR = 64.0, 12.8, 2.0


def load(path):
    tab = {}
    for r in csv.DictReader(open(path)):
        tab.setdefault(r["kernel"], {})[r["counter"]] = (int(r["dispatches"]), float(r["sum"]))
    return tab


def calibration():
    path = os.path.join(PROF, "r03_pmc_calibration.csv")
    if not os.path.exists(path):
        return None
    t = load(path)
    known = {"calib_stream": 1 << 30, "calib_store": 1 << 30, "calib_gather<4>": 458752 * 64 * 64, "calib_gather<64>": 458752 * 64 * 64,
             "calib_valu_mix": 2048 * 4 * 4096 * 64, "calib_valu": 2048 * 4 * 4096 * 64}   # the last two: wave-level vector instructions, not bytes
    out = {}
    for k, c in t.items():
        name = next((n for n in known if n in k), None)
        if not name:
            continue
        out[name] = {cn: round(known[name] / (s / d), 2) for cn, (d, s) in c.items() if s > 0 and not name.startswith("calib_valu") and cn in (
            "FETCH_SIZE", "WRITE_SIZE", "TCC_REQ_sum", "TCC_READ_sum", "TCC_WRITE_sum", "TCP_TCC_READ_REQ_sum", "TCP_TCC_WRITE_REQ_sum",
            "TCP_TOTAL_CACHE_ACCESSES_sum", "TCC_EA0_RDREQ_DRAM_sum", "TCC_MISS_sum") or (name.startswith("calib_valu") and cn in (
            "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_ANY", "GRBM_GUI_ACTIVE", "SQ_THREAD_CYCLES_VALU"))}
    return {"known_bytes_per_dispatch": known, "bytes_per_count": out,
            "note_valu": "calib_valu / calib_valu_mix: known wave-level instruction count / counter value: 1.0 for SQ_INSTS_VALU and SQ_ACTIVE_INST_VALU (they count instructions)",
            "note": "FETCH_SIZE / WRITE_SIZE count KB: 2048 for calib_stream = half the bytes reported; 64 B per TCC request and 12.8 B per vector-L1 access in the record gather"}


def levels(counters, per):
    """{hbm, l2, vl1d} bytes from summed counters / `per` (launches or frames)."""
    g = lambda n: counters.get(n, 0.0)
    out = {}
    if "FETCH_SIZE" in counters or "WRITE_SIZE" in counters:
        out["hbm"] = (FETCH_FACTOR * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024 / per
        out["hbm_lower"] = (g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024 / per
    if "TCC_REQ_sum" in counters:
        out["l2"] = g("TCC_REQ_sum") * L2_REQ_BYTES / per
    elif "TCC_HIT_sum" in counters:
        out["l2"] = (g("TCC_HIT_sum") + g("TCC_MISS_sum")) * L2_REQ_BYTES / per
    if "TCP_TOTAL_CACHE_ACCESSES_sum" in counters:
        out["vl1d"] = g("TCP_TOTAL_CACHE_ACCESSES_sum") * VL1D_ACCESS_BYTES / per
    return {k: int(round(v)) for k, v in out.items()}


def entry(label, args, cfg):
    path = os.path.join(PROF, f"r03_pmc_{label}.csv")
    if not os.path.exists(path):
        return None
    t = load(path)
    ours = {k: c for k, c in t.items() if "rt355dev::" in k}
    gen = next((c for k, c in ours.items() if "k_generate" in k), None)
    if not gen:
        return None

    def summed(pred):
        """per counter: (sum over the kernels `pred` selects) normalised per frame of ITS pass (k_generate dispatches of that counter)."""
        tot, disp = {}, {}
        for k, c in ours.items():
            if not pred(k):
                continue
            for cn, (d, s) in c.items():
                frames = gen[cn][0] if cn in gen else None
                if not frames:
                    continue
                tot[cn] = tot.get(cn, 0.0) + s / frames
                disp[cn] = disp.get(cn, 0.0) + d / frames
        return tot, disp

    ext, ext_d = summed(lambda k: any(p in k for p in EXTEND))
    con, con_d = summed(lambda k: any(p in k for p in CONNECT))
    allk, _ = summed(lambda k: True)
    ext_launches = ext_d.get("TCP_TOTAL_CACHE_ACCESSES_sum") or ext_d.get("FETCH_SIZE") or 7.0
    con_launches = con_d.get("TCP_TOTAL_CACHE_ACCESSES_sum") or con_d.get("FETCH_SIZE") or 0.0
    e = {"config": cfg, "label": label,
         "source": f"profiles/r03_pmc_{label}.csv: rocprofv3 --pmc <one counter set per pass> --kernel-trace -- python3 bench.py --steps 8 --warmup 1 --no-repeat "
                   f"--no-single --no-cpu-baseline --no-profile {args} (tools/r3_pmc.sh), converted by tools/make_traffic.py",
         "extend_launches_per_frame": round(ext_launches, 3), "connect_launches_per_frame": round(con_launches, 3),
         "extend_bytes_per_launch": levels(ext, ext_launches), "frame_bytes": levels(allk, 1.0),
         "extend_counts_per_launch": {k: round(v / ext_launches, 1) for k, v in sorted(ext.items())}}
    if con_launches:
        e["connect_bytes_per_launch"] = levels(con, con_launches)
    if "SQ_INSTS_VALU" in ext:     # wave-level vector instructions (x 4 clocks of one SIMD each): the issue roofline
        e["extend_valu_insts_per_launch"] = int(ext["SQ_INSTS_VALU"] / ext_d["SQ_INSTS_VALU"])
        e["frame_valu_insts"] = int(allk.get("SQ_INSTS_VALU", 0))
        if con_launches and "SQ_INSTS_VALU" in con:
            e["connect_valu_insts_per_launch"] = int(con["SQ_INSTS_VALU"] / con_d["SQ_INSTS_VALU"])
    # what the SIMDs and the vector L1 did meanwhile (fractions of the extend launches' own cycles)
    if "SQ_WAVE_CYCLES" in ext and ext["SQ_WAVE_CYCLES"] > 0:
        e["extend_wave_time"] = {"wait_any": round(ext.get("SQ_WAIT_ANY", 0) / ext["SQ_WAVE_CYCLES"], 3),
                                 "valu_lane_utilisation": round(ext.get("SQ_THREAD_CYCLES_VALU", 0) / max(64 * ext.get("SQ_ACTIVE_INST_VALU", 1), 1), 3)
                                 if "SQ_ACTIVE_INST_VALU" in ext else None}
    if "TCP_GATE_EN2_sum" in ext and "GRBM_GUI_ACTIVE" in ext and ext["GRBM_GUI_ACTIVE"] > 0:
        e["extend_vl1d_busy"] = round(ext["TCP_GATE_EN2_sum"] / 256 / (ext["GRBM_GUI_ACTIVE"] / 8), 3)   # busy cycles per vector L1 / kernel cycles (GRBM counts the 8 XCDs)
    return e


def main():
    out = {"made_by": "tools/make_traffic.py", "units": {"l2_request_bytes": L2_REQ_BYTES, "vl1d_access_bytes": VL1D_ACCESS_BYTES, "fetch_size_factor": FETCH_FACTOR,
                                                         "doc": __doc__.split("What a count is worth")[1].strip()},
           "calibration": calibration(), "entries": []}
    for label, (args, cfg) in RUNS.items():
        e = entry(label, args, cfg)
        if e:
            out["entries"].append(e)
            print(label, "extend/launch", e["extend_bytes_per_launch"], "frame", e["frame_bytes"], e.get("extend_vl1d_busy"), e.get("extend_wave_time"))
    json.dump(out, open(os.path.join(PROF, "r03_traffic.json"), "w"), indent=1)
    print("wrote profiles/r03_traffic.json with", len(out["entries"]), "entries")


if __name__ == "__main__":
    main()
