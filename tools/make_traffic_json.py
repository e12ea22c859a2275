"""Adds one workload's entry to profiles/traffic.json from a tools/profile.sh summary: HBM-side bytes per k_search
launch (k_search_fused where the fused route ran) from the PMC counters FETCH_SIZE and WRITE_SIZE (separate rocprofv3 passes, kernel-trace only).

Correction (MI355X_MICROARCH.md, HBM section; re-measured here): FETCH_SIZE = TCC_EA0_RDREQ x 64 B, but on gfx950 every
L2 read miss is a 128-byte fabric request -- TCC_EA0_RDREQ_128B equals TCC_EA0_RDREQ for this kernel (gpurun pmc_reqsize)
and for every access shape of tools/micro/ceilings.hip, 16-byte random gathers included -- so the bytes read are
2 x FETCH_SIZE.  WRITE_SIZE counts the 64-byte write requests exactly.
The entry is keyed by (config, image level, search variant) and records the dominant kernel's name: bench.py attaches it
only to a line of that same kernel (bench.load_traffic).
usage: python tools/make_traffic_json.py <summary.txt> <config number> <reads per gpu> [tag] [image level] [search variant]"""
import hashlib, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
summary, config, reads = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
tag = sys.argv[4] if len(sys.argv) > 4 else os.path.basename(summary)
level = int(sys.argv[5]) if len(sys.argv) > 5 else 0
variant = int(sys.argv[6]) if len(sys.argv) > 6 else 5
# the dominant kernel of the run: the fused kernel when it appears in the summary, else the general search kernel
text = open(summary).read()
fused = variant == 5 and level == 0 and "k_search_fused" in text
KERNEL = r"(?:void )?k_search_fused" if fused else r"void k_search(?!_fused)\S*"
kernel_name = bench.dominant_kernel(variant, level)
vals = {}
for line in open(summary):
    m = re.match(KERNEL + r".*\s(FETCH_SIZE|WRITE_SIZE|TCC_EA0_RDREQ_sum|TCC_EA0_WRREQ_sum|TCC_HIT_sum|TCC_MISS_sum|SQ_INSTS_VALU|SQ_INSTS_SALU)\s+n=\s*\d+ avg=([0-9.e+]+)", line)
    if m:       # (the fused route launches two instantiations, of which one returns at once: a launch = both)
        vals[m.group(1)] = vals.get(m.group(1), 0.0) + float(m.group(2))
path = os.path.join(ROOT, "profiles", "traffic.json")
try:
    allv = json.load(open(path))
    if "hbm_bytes_per_launch" in allv:      # round-1 format: one flat entry
        allv = {}
except Exception:
    allv = {}
fetch = 2 * vals["FETCH_SIZE"] * 1024
write = vals["WRITE_SIZE"] * 1024
key = bench.traffic_key(config, level, variant)
allv[key] = {
    "kernel": kernel_name, "image_level": level, "search_variant": variant,
    "hbm_bytes_per_launch": fetch + write, "fetch_bytes": fetch, "write_bytes": write,
    "fetch_size_counter_bytes": vals["FETCH_SIZE"] * 1024,
    "read_requests_128B": vals.get("TCC_EA0_RDREQ_sum"), "write_requests_64B": vals.get("TCC_EA0_WRREQ_sum"),
    "l2_hit_rate": vals["TCC_HIT_sum"] / (vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"]) if "TCC_HIT_sum" in vals else None,
    "valu_instructions": vals.get("SQ_INSTS_VALU"), "salu_instructions": vals.get("SQ_INSTS_SALU"),
    "reads_per_gpu": reads, "kernel_source_sha16": bench.kernel_source_sha16(),
    "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on bench.py, summary profiles/" + tag +
              "; per k_search launch; bytes read = 2 x FETCH_SIZE (every L2 read miss is a 128-byte fabric request on "
              "gfx950: TCC_EA0_RDREQ_128B == TCC_EA0_RDREQ), bytes written = WRITE_SIZE; Infinity Cache hits included",
}
json.dump(allv, open(path, "w"), indent=1)
print(json.dumps(allv[key], indent=1))
