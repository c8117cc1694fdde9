#!/bin/bash
# round-3 probe (second session): instruction-cache behaviour of the fully unrolled k5 kernels (code objects of 35-64 KB against a 64 KB I-cache per CU pair)
mkdir -p gpurun_out/r3
export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_INST_LEVEL_[A-Z_]*\|SQC_TC_INST[A-Z_]*\|SQ_WAIT_INST_ANY\|SQ_WAIT_IFETCH[A-Z_]*\|SQ_IFETCH_LEVEL" | sort -u | tr '\n' ' '; echo
tools/pmc.sh ic1 "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" --quick --rounds 1 > /dev/null 2>&1; cut -c1-330 gpurun_out/pmc_ic1.csv | head -14
tools/pmc.sh ic2 "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" --quick --rounds 1 > /dev/null 2>&1; cut -c1-330 gpurun_out/pmc_ic2.csv | head -14
