#!/bin/bash
# The measurements a round's profiles/ are made of (GPU box, from the repo root): tests, fuzz, bench lines and rocprofv3
# summaries of BASELINE configs 2, 5 and 3.  usage: tools/final_session.sh <tag, e.g. r03>
TAG=${1:-r03}
O=gpurun_out/final_$TAG
mkdir -p $O
timeout 900 python -m pytest tests/ -q -m gpu > $O/gputest.log 2>&1; tail -3 $O/gputest.log
( for seed in 1 2 3 4; do SEED=$seed timeout 150 python tools/fuzz_gpu.py 100 2>&1 | tail -1 | sed "s/^/seed $seed: /"; done ) > $O/fuzz.log 2>&1; cat $O/fuzz.log
timeout 600 python bench.py --steps 25 --warmup 3 > $O/c2_bench.json 2> $O/c2_bench.err
timeout 400 python bench.py --config 5 --steps 10 > $O/c5_bench.json 2> $O/c5_bench.err
timeout 900 python bench.py --config 3 --steps 5 > $O/c3_bench.json 2> $O/c3_bench.err
bash tools/profile.sh ${TAG}_c2 --steps 5 --warmup 1 --no-cpu-baseline --no-end-to-end > /dev/null 2>&1
bash tools/profile.sh ${TAG}_c5 --config 5 --steps 5 --warmup 1 --no-cpu-baseline --no-end-to-end > /dev/null 2>&1
PMC_GROUPS="FETCH_SIZE|WRITE_SIZE|TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_128B_sum|TCC_HIT_sum TCC_MISS_sum|SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" \
  bash tools/profile.sh ${TAG}_c3 --config 3 --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end > /dev/null 2>&1
for c in 2 5 3; do cp gpurun_out/prof_${TAG}_c$c/summary.txt $O/c${c}_rocprof_summary.txt; done
python - <<PY
import json
for c in (2, 5, 3):
    try:
        d = json.load(open("$O/c%d_bench.json" % c))
        print("config", c, round(d["value"] / 1e9, 1), "G k-mers/s  step", round(d["ms_per_step"], 3), "kernel", round(d["roofline"]["kernel_ms"], 3),
              "frac", round(d["roofline"]["frac"], 3), "B/col", round(d["index_build"]["image_bytes_per_column"], 1), "parity", d["cpu_baseline"]["gpu_output_bit_identical_on_sample"])
    except Exception as e:
        print("config", c, "ERR", e)
PY
# results must not depend on the builder / route knobs: the parity tests under non-default environments
for e in SBWTGPU_PATH_LOOKAHEAD=0 SBWTGPU_IMAGE_LEVEL=1 SBWTGPU_IMAGE_LEVEL=2 SBWTGPU_PATH_SAFE=0 SBWTGPU_PATH_STITCH=0 SBWTGPU_FUSED_RAGGED=0 SBWTGPU_SPLIT_LONG=0; do
  env $e timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_workloads.py -x -q -m gpu 2>&1 | tail -1 | sed "s/^/$e: /"
done | tee $O/knob_sweep.log
