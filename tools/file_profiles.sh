#!/bin/bash
# Files a final session's outputs (gpurun_out/final_<tag>/) under profiles/ as the round's set: tools/file_profiles.sh <tag> <round, e.g. r04>
TAG=$1; R=${2:-r04}; S=gpurun_out/final_$TAG
for f in c2_bench.json c2_bench_under_rocprof.json c2_level1.json c2_level1_rocprof_summary.txt c2_level2.json c2_level2_rocprof_summary.txt \
         c2_line.json c2_rocprof_summary.txt c3_bench.json c3_bench_under_rocprof.json c3_line.json c3_rocprof_summary.txt \
         c5_bench.json c5_bench_under_rocprof.json c5_line.json c5_rocprof_summary.txt robustness.jsonl batch_size.txt knob_sweep.log \
         lane_stats.txt timeline.txt two_in_flight.txt c6_hbm_bench.json c6_big_index_bench.json c6_big_index_k32.json long_reads.txt ab_prev_round.txt; do
  cp $S/$f profiles/${R}_$f
done
( grep -E "passed|failed" $S/gputest.log | tail -2; tail -4 $S/gputest.log ) > profiles/${R}_gputest_tail.txt
sed "s/${TAG}_/${R}_/g" $S/traffic.json > profiles/traffic.json
python - <<PY
import json, bench
h = bench.kernel_source_sha16()
t = json.load(open("profiles/traffic.json"))
ents = t["entries"] if "entries" in t else t
bad = [k for k, v in ents.items() if isinstance(v, dict) and v.get("kernel_source_sha16") != h]
print("kernel sources", h, "entries", len(ents), "of another hash:", bad)
PY
