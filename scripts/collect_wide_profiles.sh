#!/bin/bash
# Wide-path (hidden 512 / 1024) evidence in one GPU call: per-kernel HIP-event table, rocprofv3 kernel stats, HBM traffic
# (FETCH_SIZE / WRITE_SIZE, separate passes) and the effective clock (GRBM_GUI_ACTIVE) for
#   SIREN 512x8 on 2048^2  and  SIREN 1024x12 on 1024^2 (one 1 Mi-pixel chunk)
# usage: scripts/collect_wide_profiles.sh TAG   -> gpurun_out/TAG_*
tag=${1:-wide}
R=$PWD
out=$R/gpurun_out
mkdir -p $out
python3 scripts/wide_bench.py > $out/${tag}_bench.jsonl 2> $out/${tag}_bench.err
cd /tmp && export TMPDIR=/tmp
for cfg in "512 8 2048" "1024 12 1024"; do
  set -- $cfg; n=${1}x${2}
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats_$n -- python3 $R/scripts/wide_one.py $1 $2 $3 > /dev/null 2> $out/${tag}_stats_$n.err
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/${tag}_FETCH_$n -- python3 $R/scripts/wide_one.py $1 $2 $3 > /dev/null 2> $out/${tag}_f_$n.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/${tag}_WRITE_$n -- python3 $R/scripts/wide_one.py $1 $2 $3 > /dev/null 2> $out/${tag}_w_$n.err
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/${tag}_clk_$n -- python3 $R/scripts/wide_one.py $1 $2 $3 > /dev/null 2> $out/${tag}_c_$n.err
done
cd $R
find $out -path "*${tag}_*" -name "*kernel_trace.csv" -delete
for n in 512x8 1024x12; do
  echo "== $n" >> $out/${tag}_summary.txt
  python3 scripts/pmc_summary.py $out/${tag}_FETCH_$n $out/${tag}_WRITE_$n $out/${tag}_clk_$n >> $out/${tag}_summary.txt 2>&1
done
cat $out/${tag}_bench.jsonl
