"""Writes profiles/traffic.json from a tools/profile.sh summary: HBM-side bytes per k_search launch from
the PMC counters FETCH_SIZE and WRITE_SIZE (separate rocprofv3 passes, kernel-trace only).

Calibration (tools/micro/calib.sh, gpurun_out/calib): for this kernel's access pattern -- random 16-byte
gathers that miss L2 -- TCC_EA0_RDREQ counts exactly one request per gather and FETCH_SIZE[KB]*1024 equals
TCC_EA0_RDREQ*64 B (838.9 M gathers -> 838.0 M requests, 53.6 GB), so FETCH_SIZE is taken as is; the
guide's x2 correction applies to wide coalesced streams (128-byte requests tallied at 64 B), which here
are only the k_encode pass and the packed-base reloads (< 3 % of the requests)."""
import json, re, sys
summary = sys.argv[1]
tag = sys.argv[2] if len(sys.argv) > 2 else summary
vals = {}
for line in open(summary):
    m = re.match(r"void k_search\S*.*\s(FETCH_SIZE|WRITE_SIZE|TCC_EA0_RDREQ_sum|TCC_EA0_WRREQ_sum|TCC_HIT_sum|TCC_MISS_sum)\s+n=\s*\d+ avg=([0-9.e+]+)", line)
    if m:
        vals[m.group(1)] = float(m.group(2))
out = {
    "hbm_bytes_per_launch": (vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024,
    "fetch_bytes": vals["FETCH_SIZE"] * 1024, "write_bytes": vals["WRITE_SIZE"] * 1024,
    "read_requests": vals.get("TCC_EA0_RDREQ_sum"), "write_requests": vals.get("TCC_EA0_WRREQ_sum"),
    "l2_hit_rate": vals["TCC_HIT_sum"] / (vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"]) if "TCC_HIT_sum" in vals else None,
    "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on `python bench.py --steps 3 --warmup 1`, "
              "summary " + tag + "; per k_search launch over 10 M reads; fabric-side (L2 miss) bytes, Infinity "
              "Cache hits included; FETCH_SIZE calibrated 1:1 for 16-byte gathers (tools/micro/calib.sh)",
}
json.dump(out, open("profiles/traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
