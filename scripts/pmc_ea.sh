#!/bin/bash
# L2 <-> HBM interface counters (separate passes) for the wide path (512x8, one step) and the width-256 path: outstanding-request levels
# (average latency = LEVEL / REQ), DRAM credit stalls.  usage: scripts/pmc_ea.sh TAG -> gpurun_out/TAG_*
tag=${1:-ea}; R=$PWD; out=$R/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_sum --output-format csv -d $out/${tag}_rd_wide -- python3 $R/scripts/wide_one.py 512 8 2048 > /dev/null 2> $out/${tag}_1.err
rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $out/${tag}_wr_wide -- python3 $R/scripts/wide_one.py 512 8 2048 > /dev/null 2> $out/${tag}_2.err
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_sum --output-format csv -d $out/${tag}_rd_256 -- python3 $R/scripts/prof_step.py 2048 2 > /dev/null 2> $out/${tag}_3.err
rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $out/${tag}_wr_256 -- python3 $R/scripts/prof_step.py 2048 2 > /dev/null 2> $out/${tag}_4.err
cd $R
find $out -path "*${tag}_*" -name "*kernel_trace.csv" -delete
python3 scripts/pmc_summary.py $out/${tag}_rd_wide $out/${tag}_wr_wide > $out/${tag}_wide.txt 2>&1
python3 scripts/pmc_summary.py $out/${tag}_rd_256 $out/${tag}_wr_256 > $out/${tag}_256.txt 2>&1
