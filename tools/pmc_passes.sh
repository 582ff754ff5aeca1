#!/bin/bash
# Counter evidence for the bench run (GPU box): one rocprofv3 --pmc pass per counter group over
#     python3 bench.py --no-cpu-baseline --steps 6 --warmup 2
# (separate passes: FETCH_SIZE and WRITE_SIZE do not fit one TCC pass, SQ has 8 slots; never combined with
# --sys-trace).  The first (unprofiled) run tunes the launch shapes and stores them in RTPE_AUTOTUNE_CACHE, so the
# profiled passes import them and time the same kernels.  Output: gpurun_out/<tag>/pmc_<group>/..._counter_collection.csv
# and the per-kernel summary profiles/<tag>_pmc_summary.json (tools/pmc_summary.py).
#     bash tools/pmc_passes.sh r02
set -e
TAG=${1:-r02}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export RTPE_AUTOTUNE_CACHE=$PWD/$OUT/autotune.json
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - >/dev/null
python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 > $OUT/bench_unprofiled.json 2> $OUT/bench_unprofiled.err
echo "unprofiled: $(cut -c1-120 $OUT/bench_unprofiled.json)"
# the profiled passes run one forward at a time with the branches of a module one after another (option "lanes" off,
# RTPE_FORWARDS_IN_FLIGHT=1): with the lanes on or two forwards in flight, kernels of
# different branches overlap in time and a kernel's duration in the trace is not its own (bench.py times the roofline
# kernel on single-stream recorded steps for the same reason)
export RTPE_LANES=0
export RTPE_FORWARDS_IN_FLIGHT=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 > $OUT/bench_traced.json 2> $OUT/bench_traced.err
echo "traced: $(cut -c1-120 $OUT/bench_traced.json)"
run_pass () {
  name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/pmc_$name -o pmc -- python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 > $OUT/bench_pmc_$name.json 2> $OUT/bench_pmc_$name.err
  echo "pass $name done: $(ls $OUT/pmc_$name/*counter_collection.csv 2>/dev/null | head -1)"
}
run_pass fetch FETCH_SIZE
run_pass write WRITE_SIZE
run_pass mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
run_pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
python3 tools/pmc_summary.py $OUT profiles/${TAG}_pmc_summary.json > $OUT/pmc_summary.log 2>&1 || tail -5 $OUT/pmc_summary.log
tail -30 $OUT/pmc_summary.log
