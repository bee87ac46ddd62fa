#!/bin/bash
# SQ counter passes over scripts/prof_step.py (one 4 Mi-pixel chunk per launch at 2048^2): usage scripts/pmc_sq.sh TAG
tag=${1:-sq}
R=$PWD
out=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $out/${tag}_a -- python3 $R/scripts/prof_step.py 2048 2 > /dev/null 2> $out/${tag}_a.err
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_MFMA --output-format csv -d $out/${tag}_b -- python3 $R/scripts/prof_step.py 2048 2 > /dev/null 2> $out/${tag}_b.err
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $out/${tag}_g -- python3 $R/scripts/prof_step.py 2048 3 > /dev/null 2> $out/${tag}_g.err
cd $R
python3 scripts/pmc_summary.py $out/${tag}_a $out/${tag}_b $out/${tag}_g > $out/${tag}_summary.txt 2>&1
find $out/${tag}_a $out/${tag}_b $out/${tag}_g -name "*kernel_trace.csv" -delete
grep -E "k_bwd8h|k_fwd_pipe" $out/${tag}_summary.txt
