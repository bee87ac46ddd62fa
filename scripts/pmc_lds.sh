#!/bin/bash
# LDS-side SQ counters of scripts/prof_step.py (one 4 Mi-pixel chunk per launch): usage scripts/pmc_lds.sh TAG
tag=${1:-lds}
R=$PWD
out=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_UNALIGNED_STALL SQ_WAVE_CYCLES --output-format csv -d $out/${tag}_c -- python3 $R/scripts/prof_step.py 2048 2 > /dev/null 2> $out/${tag}_c.err
cd $R
python3 scripts/pmc_summary.py $out/${tag}_c > $out/${tag}_summary.txt 2>&1
find $out/${tag}_c -name "*kernel_trace.csv" -delete
grep -E "k_bwd8h|k_fwd_pipe|k_bwd8<" $out/${tag}_summary.txt
