#!/bin/bash
# The measurements a round's profiles/ are made of (GPU box, from the repo root): the driver's suite, fuzz, bench lines and
# rocprofv3 summaries (kernel trace + PMC traffic, separate passes) of BASELINE configs 2, 5 and 3, the image levels with
# their own counters, robustness and batch-size rows.  usage: tools/final_session.sh <tag, e.g. r05> [1|2|all]
#   part 1: the -m gpu suite, fuzz, bench lines, rocprofv3 kernel traces + PMC passes, the committed lines
#   part 2: robustness and batch-size rows, knob sweep, instrumented builds, two batches in flight, the 10^9- and 2.25 x 10^9-column
#           indexes, this round's library against last round's (sbwt_amd/lib/lib_prev.so, tools/build_rev_lib.sh)
TAG=${1:-r06}
PART=${2:-all}
O=gpurun_out/final_$TAG
mkdir -p $O
if [ "$PART" = 1 ] || [ "$PART" = all ]; then
( time timeout 1500 python -m pytest tests/ -q -m gpu ) > $O/gputest.log 2>&1; tail -4 $O/gputest.log
( for seed in 1 2 3 4; do SEED=$seed timeout 150 python tools/fuzz_gpu.py 100 2>&1 | tail -1 | sed "s/^/seed $seed: /"; done ) > $O/fuzz.log 2>&1; cat $O/fuzz.log
timeout 900 python bench.py --steps 25 --warmup 3 > $O/c2_bench.json 2> $O/c2_bench.err
timeout 400 python bench.py --config 5 --steps 10 --no-cli-full > $O/c5_bench.json 2> $O/c5_bench.err
timeout 900 python bench.py --config 3 --steps 5 > $O/c3_bench.json 2> $O/c3_bench.err
PMC="FETCH_SIZE|WRITE_SIZE|TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_128B_sum|TCC_HIT_sum TCC_MISS_sum|SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES"
prof() {   # prof <name> <config> <level> <variant> <reads> <bench args...>
  local name=$1 cfg=$2 lvl=$3 var=$4 reads=$5; shift 5
  PMC_GROUPS="$PMC" bash tools/profile.sh ${TAG}_$name "$@" --warmup 2 --no-cpu-baseline --no-end-to-end --no-two-in-flight --no-int32-leg > /dev/null 2>&1
  cp gpurun_out/prof_${TAG}_$name/summary.txt $O/${name}_rocprof_summary.txt
  cp gpurun_out/prof_${TAG}_$name/trace_bench.json $O/${name}_bench_under_rocprof.json
  python tools/make_traffic_json.py $O/${name}_rocprof_summary.txt $cfg $reads ${TAG}_${name}_rocprof_summary.txt $lvl $var > /dev/null 2>> $O/traffic.err
}
prof c2 2 0 5 10000000 --steps 5
prof c5 5 0 5 10000000 --config 5 --steps 5
prof c3 3 0 5 100000000 --config 3 --steps 3
prof c2_level1 2 1 1 10000000 --steps 3 --image-level 1
prof c2_level2 2 2 1 10000000 --steps 3 --image-level 2
cp profiles/traffic.json $O/traffic.json
# the committed lines: the same commands once more, now that traffic.json belongs to this code (roofline.traffic attached)
timeout 900 python bench.py --steps 25 --warmup 3 --no-cpu-baseline --no-end-to-end > $O/c2_line.json 2>/dev/null
timeout 400 python bench.py --config 5 --steps 10 --no-cpu-baseline --no-end-to-end > $O/c5_line.json 2>/dev/null
timeout 900 python bench.py --config 3 --steps 5 --no-cpu-baseline --no-end-to-end > $O/c3_line.json 2>/dev/null
timeout 600 python bench.py --steps 10 --image-level 1 --no-cpu-baseline --no-end-to-end > $O/c2_level1.json 2>/dev/null
timeout 600 python bench.py --steps 10 --image-level 2 --no-cpu-baseline --no-end-to-end > $O/c2_level2.json 2>/dev/null
fi   # part 1
if [ "$PART" = 2 ] || [ "$PART" = all ]; then
NREADS=10000000 timeout 900 python tools/robustness_bench.py > $O/robustness.jsonl 2> $O/robustness.err
for n in 1000000 4000000; do NREADS=$n ROUNDS=9 CONFIGS='[[5,0]]' python tools/ab_step.py 2>&1 | grep "^variant" | sed "s/^/config 2, $n reads: /"; done > $O/batch_size.txt 2>&1
python - <<PY
import json
for c in ("c2", "c5", "c3", "c2_line", "c5_line", "c3_line", "c2_level1", "c2_level2"):
    try:
        d = json.load(open("$O/%s_bench.json" % c if "_" not in c else "$O/%s.json" % c))
        r = d["roofline"]
        print(c, round(d["value"] / 1e9, 1), "G k-mers/s  step", round(d["ms_per_step"], 3), "kernel", r["kernel"], round(r["kernel_ms"], 3),
              "frac", round(r["frac"], 3), "traffic/alg", r.get("traffic_over_algorithmic"), "lines/read", r.get("read_lines_per_read"))
    except Exception as e:
        print(c, "ERR", e)
PY
# results must not depend on the builder / route knobs: the parity tests under non-default environments
for e in SBWTGPU_PATH_LOOKAHEAD=0 SBWTGPU_PATH_RANK=0 SBWTGPU_SCRATCH_ARENA=0 SBWTGPU_IMAGE_LEVEL=1 SBWTGPU_IMAGE_LEVEL=2 SBWTGPU_PATH_SAFE=0 SBWTGPU_PATH_STITCH=0 SBWTGPU_FUSED_RAGGED=0 SBWTGPU_SPLIT_LONG=0 SBWTGPU_FUSED_PIECES=3 SBWTGPU_FUSED_PIECES=1; do
  env $e timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_workloads.py -x -q -m gpu 2>&1 | tail -1 | sed "s/^/$e: /"
done | tee $O/knob_sweep.log
# instrumented builds (tools/build_stats_lib.sh [timeline]; not the shipped library): lane-iterations by kind, the tail by
# iteration, and when the waves of one launch start / see the tickets run out / leave; two batches in flight
if [ -f sbwt_amd/lib/lib_stats.so ]; then
  ( export SBWTGPU_LIB=$PWD/sbwt_amd/lib/lib_stats.so
    echo "== config 2 (coli3-like, k=30, 10 M reads) =="; python tools/lane_stats_fused.py 2>&1 | grep -v "^config\|^index\|^variant\|amdgpu.ids"
    echo "== config 5 (k=63, no streaming support, 10 M reads) =="; K=63 STREAMING=0 python tools/lane_stats_fused.py 2>&1 | grep -v "^config\|^index\|^variant\|amdgpu.ids"
    echo "== config 3 index type (pan64, k=31), 20 M reads =="; GENOMES=pan64 K=31 NREADS=20000000 python tools/lane_stats_fused.py 2>&1 | grep -v "^config\|^index\|^variant\|amdgpu.ids" ) > $O/lane_stats.txt 2>&1
fi
if [ -f sbwt_amd/lib/lib_timeline.so ]; then
  ( export SBWTGPU_LIB=$PWD/sbwt_amd/lib/lib_timeline.so
    echo "(the unsorted instantiation, SBWTGPU_FUSED_SORT=0: follower waves of the sorted one never see the ticket counter themselves)"
    for n in 10000000 1000000; do SBWTGPU_FUSED_SORT=0 NREADS=$n python tools/timeline_fused.py 2>&1 | grep -v "^config\|^index\|amdgpu.ids"; done ) > $O/timeline.txt 2>&1
fi
( NREADS=10000000 python tools/overlap_steps.py 2>&1 | grep "^reads"
  NREADS=1000000 STEPS=50 python tools/overlap_steps.py 2>&1 | grep "^reads"
  K=63 STREAMING=0 NREADS=10000000 python tools/overlap_steps.py 2>&1 | grep "^reads" ) > $O/two_in_flight.txt 2>&1
# the indexes beyond the Infinity Cache and beyond 2^31 columns (SURVEY 8d "G-hbm")
timeout 900 python bench.py --config 6 --steps 5 --warmup 2 --no-end-to-end > $O/c6_hbm_bench.json 2> $O/c6_hbm_bench.err
timeout 1200 python bench.py --config 6 --hbm-genome-len 2250000000 --steps 5 --warmup 2 --no-end-to-end > $O/c6_big_index_bench.json 2> $O/c6_big_index_bench.err
# ... and 31 < k <= 63 there (round 6: the full image; k = 32 is what the GPU builder's 64-bit keys hold at that size)
timeout 1200 python bench.py --config 6 --hbm-genome-len 2250000000 --hbm-k 32 --steps 5 --warmup 2 --no-end-to-end --no-cpu-baseline > $O/c6_big_index_k32.json 2> $O/c6_big_index_k32.err
# long reads (round 6: the fused kernel's ticket table): whole genomes, 1 kbp and 10 kbp reads, >= 10^9 bases per batch, k = 30 and 63,
# 1 % and 5 % substitutions, with the table and without (the general kernel)
( for K in 30 63; do for SUBS in 0.01 0.05; do K=$K SUBS=$SUBS timeout 600 python tools/long_read_bench.py 2>&1 | grep "^k="; done; done ) > $O/long_reads.txt 2>&1
# this round's library against last round's, interleaved on this box (configs 2, 3's index type, 5)
if [ -f sbwt_amd/lib/lib_prev.so ]; then
  ( ROUNDS=5 bash tools/ab_libs5.sh "sbwt_amd/lib/lib_prev.so sbwt_amd/lib/libsbwtgpu.so" 2 | sed "s/^/c2 /"
    GENOMES=pan64 K=31 ROUNDS=3 bash tools/ab_libs5.sh "sbwt_amd/lib/lib_prev.so sbwt_amd/lib/libsbwtgpu.so" 2 | sed "s/^/c3 /"
    K=63 STREAMING=0 ROUNDS=3 bash tools/ab_libs5.sh "sbwt_amd/lib/lib_prev.so sbwt_amd/lib/libsbwtgpu.so" 2 | sed "s/^/c5 /" ) > $O/ab_prev_round.txt 2>&1
fi
fi   # part 2
