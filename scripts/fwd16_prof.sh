#!/bin/bash
# kernel stats + effective clock of both forward tilings on one box: usage scripts/fwd16_prof.sh TAG
set -e
tag=${1:-fwd16}
R=$PWD; out=$R/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for knob in 0 1; do
  export SIREN_FIT_FWD16=$knob
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats_$knob -- python3 $R/scripts/prof_step.py 4096 6 > /dev/null 2> $out/${tag}_stats_$knob.err
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $out/${tag}_clk_$knob -- python3 $R/scripts/prof_step.py 4096 4 > /dev/null 2> $out/${tag}_clk_$knob.err
  find $out/${tag}_stats_$knob -name "*kernel_trace.csv" -delete
done
cd $R
bash scripts/fwd16_ab.sh > $out/${tag}_ab.txt 2>&1
cat $out/${tag}_ab.txt
