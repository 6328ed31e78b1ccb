"""Round-5 PMC summaries from the raw rocprofv3 passes of tools/gpu_r05.sh (pmc_summary.py output).

    python tools/pmc_r05_summary.py northstar raw.json out.json   samplers + verifier (the kernels BASELINE.json's north_star names)
    python tools/pmc_r05_summary.py l2        raw.json out.json   L2 / TA / SQ view of the 7B projections at the shipped 128-row plans

HBM bytes per launch follow MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE come from SEPARATE passes, are KiB, and
on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads: hbm = (2 FETCH_SIZE + WRITE_SIZE) x 1024.  The inputs of
tools/kbench.py (8 MB of logits, 8 MB of noise) fit the 256 MiB Infinity Cache, whose hits the fabric-side counters include:
the figure is "bytes requested from the memory side of the L2", which is what the algorithmic bytes are compared with.
SQ_* counters are summed over the chip; SQ_WAVE_CYCLES / SQ_WAIT_* count quad-cycles."""
import hashlib
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
mode, raw_path, out_path = sys.argv[1:4]
raw = json.load(open(raw_path))


def sha(*names):
    h = hashlib.sha256()
    for n in names:
        with open(os.path.join(REPO, "sequoia_amd", "csrc", n), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def kernels_of(tag):
    return raw.get(tag, {})


def short(kname):
    k = kname.split("|grid=")
    base = k[0].split("(")[0]
    for cut in ("void ", "sequoia::"):
        base = base.replace(cut, "")
    return base.strip(), int(k[1]) if len(k) > 1 and k[1].isdigit() else 0


out = {"kernels": {}}
if mode == "northstar":
    out["note"] = ("rocprofv3 --kernel-trace --pmc over tools/kbench.py samp | verify (config-B growmap, 128 nodes, V = 32000, synthetic logits): "
                   "FETCH_SIZE, WRITE_SIZE and the SQ group in separate passes.  hbm_bytes = (2 FETCH_SIZE + WRITE_SIZE) KiB per launch, "
                   "averaged over the launches of that (kernel, grid).  lds_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; "
                   "wait_inst_frac / wait_any_frac / active_frac = share of SQ_WAVE_CYCLES.")
    want = ("sample_parts", "sample_merge", "logits_stats", "verify_nodes", "verify_walk", "topk", "argmax_rows")
    for what in ("samp", "verify"):
        f, w, s = (kernels_of(f"ns_{what}_{p}") for p in ("FETCH", "WRITE", "SQ"))
        for kname, rec in s.items():
            base, grid = short(kname)
            if not any(x in base for x in want):
                continue
            fr, wr = f.get(kname, {}), w.get(kname, {})
            fetch = fr.get("FETCH_SIZE", {}).get("avg")
            write = wr.get("WRITE_SIZE", {}).get("avg")
            wave = rec.get("SQ_WAVE_CYCLES", {}).get("avg") or 0.0
            lds_act = rec.get("SQ_LDS_IDX_ACTIVE", {}).get("avg") or 0.0
            e = dict(kernel=base, grid=grid, launches=rec.get("SQ_WAVE_CYCLES", {}).get("launches"),
                     FETCH_SIZE_KiB=None if fetch is None else round(fetch, 1), WRITE_SIZE_KiB=None if write is None else round(write, 1),
                     hbm_bytes_per_launch=None if fetch is None or write is None else int((2 * fetch + write) * 1024))
            for c in ("SQ_INSTS_VALU", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY",
                      "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"):
                if c in rec:
                    e[c] = round(rec[c]["avg"], 1)
            if wave:
                e["wait_inst_frac"] = round((rec.get("SQ_WAIT_INST_ANY", {}).get("avg") or 0.0) / wave, 3)
                e["wait_any_frac"] = round((rec.get("SQ_WAIT_ANY", {}).get("avg") or 0.0) / wave, 3)
                e["active_frac"] = round((rec.get("SQ_ACTIVE_INST_ANY", {}).get("avg") or 0.0) / wave, 3)
            if lds_act:
                e["lds_conflict_frac"] = round((rec.get("SQ_LDS_BANK_CONFLICT", {}).get("avg") or 0.0) / lds_act, 3)
            out["kernels"][f"{base}|grid={grid}"] = e
    out["source_sha"] = {"sampler.hip": sha("sampler.hip", "common.h"), "verify.hip": sha("verify.hip", "common.h")}
else:
    out["note"] = ("rocprofv3 --kernel-trace --pmc over tools/ts_bench 128 at the shipped 7B launch plans (weights rotate over > 640 MB): TCC, "
                   "TCP / TA and SQ groups in separate passes.  l2_hit_rate = TCC_HIT / (TCC_HIT + TCC_MISS); l2_request_bytes = TCC_REQ x 128 B; "
                   "l2_busy_frac = TCC_BUSY_avr / (GRBM_GUI_ACTIVE / 8); mean L1->L2 read latency = TCP_TCC_READ_REQ_LATENCY / TCP_TCC_READ_REQ.")
    for tag in ("qkv", "o", "gate_up", "down"):
        tcc, tcp, sq = (kernels_of(f"l2_{tag}_{p}") for p in ("TCC", "TCP", "SQ"))
        for kname, rec in tcc.items():
            base, grid = short(kname)
            if "ts_linear_kernel" not in base:
                continue
            g = lambda d, c: (d.get(kname, {}).get(c, {}) or {}).get("avg")
            hit, miss, req, busy, gui = (g(tcc, c) for c in ("TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum", "TCC_BUSY_avr", "GRBM_GUI_ACTIVE"))
            e = dict(kernel=kname.split("|")[0][:120], grid=grid, TCC_HIT_sum=hit, TCC_MISS_sum=miss, TCC_REQ_sum=req, TCC_BUSY_avr=busy,
                     GRBM_GUI_ACTIVE=gui)
            if hit is not None and miss is not None and hit + miss > 0:
                e["l2_hit_rate"] = round(hit / (hit + miss), 3)
            if req is not None:
                e["l2_request_bytes"] = int(req * 128)
            if busy is not None and gui:
                e["l2_busy_frac"] = round(busy / (gui / 8), 3)
            rd, lat, ta = g(tcp, "TCP_TCC_READ_REQ_sum"), g(tcp, "TCP_TCC_READ_REQ_LATENCY_sum"), g(tcp, "TA_BUSY_avr")
            e.update(TCP_TCC_READ_REQ_sum=rd, TA_BUSY_avr=ta, TCP_PENDING_STALL_CYCLES_sum=g(tcp, "TCP_PENDING_STALL_CYCLES_sum"))
            if rd and lat:
                e["mean_l1_to_l2_read_latency_cycles"] = round(lat / rd)
            wave = g(sq, "SQ_WAVE_CYCLES")
            for c in ("SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU"):
                e[c] = g(sq, c)
            if wave:
                e["wait_any_frac"] = round((g(sq, "SQ_WAIT_ANY") or 0.0) / wave, 3)
                e["wait_inst_frac"] = round((g(sq, "SQ_WAIT_INST_ANY") or 0.0) / wave, 3)
            out["kernels"][f"{tag}|grid={grid}"] = e
    out["source_sha"] = {"ts_linear.hip": sha("ts_linear.hip", "common.h")}
json.dump(out, open(out_path, "w"), indent=1)
for k, v in out["kernels"].items():
    print(k, {a: b for a, b in v.items() if a in ("hbm_bytes_per_launch", "lds_conflict_frac", "wait_inst_frac", "wait_any_frac", "active_frac",
                                                  "l2_hit_rate", "l2_busy_frac", "l2_request_bytes", "mean_l1_to_l2_read_latency_cycles", "launches")})
