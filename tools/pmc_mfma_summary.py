"""Per kernel VARIANT: MFMA utilisation, LDS and wait shares from two rocprofv3 --pmc passes (tools/gpu_pmc_mfma.sh).
usage: pmc_mfma_summary.py <pass A dir> <pass B dir> <out.json> ["<command>"]

Derived columns (rocprofv3 sums a counter over its instances: 8 XCDs x SEs x CUs x SIMDs as the counter has them):
  cycles        = GRBM_GUI_ACTIVE / 8                      (GPU-active cycles of the dispatch, one XCD's worth)
  mfma_util     = SQ_VALU_MFMA_BUSY_CYCLES / (cycles * 4 SIMDs * n_cus)      (MI355X_MICROARCH.md: the counter counts
                  cycles, 32 per v_mfma_f32_32x32x16_bf16; 1.0 = every SIMD's matrix pipe busy in every cycle)
  wave_cycles   = 4 * SQ_WAVE_CYCLES (the SQ wave counters count quad-cycles); shares below are of SQ_WAVE_CYCLES:
  wait_any      = SQ_WAIT_ANY / SQ_WAVE_CYCLES             (parked at s_waitcnt / s_barrier)
  wait_inst     = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES        (issue stalls), of which lds_issue = SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES
  active        = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES
  bank_conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (pass B's LDS-active cycles; None if that counter is missing)
  clock_GHz     = cycles / duration
"""
import collections, csv, glob, hashlib, json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_CUS, SIMDS = 256, 4


def variant(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"(?:[\w:]*::)?(conv_igemm_(?:fwd|dgrad)_kernel|wgrad_kernel|wgrad_multi_kernel)<([^>]*)>", n)
    if m:
        return f"{m.group(1)}<{m.group(2).replace(' ', '')}>"
    n = re.sub(r"<.*", "", n.split("(")[0])
    return n.split("::")[-1]


def load(d):
    dur = {}
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    seen = collections.defaultdict(set)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = variant(r["Kernel_Name"])
            out[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Dispatch_Id"] not in seen[k] and r["Dispatch_Id"] in dur:
                seen[k].add(r["Dispatch_Id"])
                out[k]["_duration_ns"].append(dur[r["Dispatch_Id"]])
    return out


A, B = load(sys.argv[1]), load(sys.argv[2])
avg = lambda v: sum(v) / len(v) if v else None
summary = {}
for k in sorted(A, key=lambda k: -sum(A[k].get("_duration_ns", [0]))):
    a, b = A[k], B.get(k, {})
    g = lambda d, c: avg(d.get(c, []))
    cyc = g(a, "GRBM_GUI_ACTIVE")
    cyc = cyc / 8 if cyc else None
    wave = g(a, "SQ_WAVE_CYCLES")
    e = {"launches": len(a.get("_duration_ns", [])), "avg_us_under_pmc": round(g(a, "_duration_ns") / 1e3, 2) if a.get("_duration_ns") else None,
         "cycles": cyc}
    mf = g(a, "SQ_VALU_MFMA_BUSY_CYCLES")
    if cyc and mf is not None:
        e["mfma_util"] = round(mf / (cyc * SIMDS * N_CUS), 4)
        e["clock_GHz"] = round(cyc / g(a, "_duration_ns"), 3) if a.get("_duration_ns") else None
    if wave:
        for name, c in (("wait_any", "SQ_WAIT_ANY"), ("wait_inst", "SQ_WAIT_INST_ANY"), ("lds_issue", "SQ_WAIT_INST_LDS"),
                        ("active", "SQ_ACTIVE_INST_ANY")):
            v = g(a, c)
            e[name] = round(v / wave, 4) if v is not None else None
    bc, la = g(a, "SQ_LDS_BANK_CONFLICT"), g(b, "SQ_LDS_IDX_ACTIVE")
    e["bank_conflict"] = round(bc / la, 4) if (bc is not None and la) else None
    e["raw"] = {c: round(avg(v), 1) for d in (a, b) for c, v in d.items() if not c.startswith("_")}
    summary[k] = e
h = hashlib.sha256()
for f in sorted(glob.glob(os.path.join(ROOT, "cv_a-fan_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "cv_a-fan_amd", "csrc", "*.h"))):
    h.update(open(f, "rb").read())
sys.path.insert(0, ROOT)
import bench  # noqa: E402 (stdlib-only at import time): per-file hashes, so that staleness is judged per kernel
summary["_meta"] = {"kernel_sources_sha": h.hexdigest()[:16], "kernel_sources_sha_files": bench.source_shas(), "command": sys.argv[4] if len(sys.argv) > 4 else None,
                    "n_cus": N_CUS, "formulas": __doc__.split("Derived columns")[1].strip()}
json.dump(summary, open(sys.argv[3], "w"), indent=1)
for k, e in summary.items():
    if not k.startswith("_") and e.get("launches"):
        print(f"{k[:64]:64s} n={e['launches']:4d} {e['avg_us_under_pmc']:8.1f} us  mfma {e.get('mfma_util')}  wait {e.get('wait_any')} "
              f"inst {e.get('wait_inst')} lds {e.get('lds_issue')} act {e.get('active')} bc {e.get('bank_conflict')} clk {e.get('clock_GHz')}")
