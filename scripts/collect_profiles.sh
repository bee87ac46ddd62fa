#!/bin/bash
# One GPU call that regenerates every profile the bench line and DESIGN.md cite (run from the repo root on the box):
#   kernel trace + stats of bench.py, three PMC passes (FETCH_SIZE, WRITE_SIZE, SQ_*) and a GRBM pass over prof_step.py
# usage: scripts/collect_profiles.sh TAG      -> gpurun_out/TAG_*
set -e
tag=${1:-prof}
R=$PWD
out=$R/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-formats > $out/${tag}_stats_bench.json 2> $out/${tag}_stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/${tag}_FETCH_SIZE -- python3 $R/scripts/prof_step.py 2048 2 > /dev/null 2> $out/${tag}_pmc1.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/${tag}_WRITE_SIZE -- python3 $R/scripts/prof_step.py 2048 2 > /dev/null 2> $out/${tag}_pmc2.err
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $out/${tag}_sq -- python3 $R/scripts/prof_step.py 2048 2 > /dev/null 2> $out/${tag}_pmc3.err
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $out/${tag}_grbm -- python3 $R/scripts/prof_step.py 2048 3 > /dev/null 2> $out/${tag}_pmc4.err
cd $R
# keep only the small summaries (the merge back is capped at 64 MiB)
find $out/${tag}_stats -name "*kernel_trace.csv" -delete
python3 bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
cat $out/${tag}_bench.json
