#!/bin/bash
# rocprofv3 recipe for the bench (run on the GPU box from the repo root):
#   pass 1: kernel trace + stats; passes 2..: PMC counters, one small group per run (FETCH_SIZE and
#   WRITE_SIZE do not fit one pass; counters are never combined with the API trace domains).
# usage: tools/profile.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
ARGS=${@:---steps 3 --warmup 2 --no-cpu-baseline --no-end-to-end --no-two-in-flight --no-int32-leg}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/trace_bench.json 2> $OUT/trace.err
# PMC_GROUPS (env): '|'-separated counter groups, e.g. "SQ_INSTS_VALU SQ_INSTS_SALU|TCC_EA0_RDREQ_sum"; default = the full set
DEFAULT_GROUPS="FETCH_SIZE|WRITE_SIZE|TCC_HIT_sum TCC_MISS_sum|TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_128B_sum|SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY|SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES|GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum"
IFS='|' read -ra GROUPS_ARR <<< "${PMC_GROUPS:-$DEFAULT_GROUPS}"
for grp in "${GROUPS_ARR[@]}"; do
  name=$(echo $grp | tr ' ' '_')
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc_$name -- python3 bench.py $ARGS > /dev/null 2> $OUT/pmc_$name.err || echo "pmc $grp failed" >> $OUT/errors.txt
done
python3 tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
# the raw rocprofv3 output is large (gpurun returns at most 64 MiB): keep the summary only
rm -rf $OUT/trace $OUT/pmc_*/
cat $OUT/summary.txt
