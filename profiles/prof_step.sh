#!/bin/bash
# rocprofv3 passes over one bench.py command on the GPU box (run from the repo root): kernel stats, then one PMC pass per counter
# group (each its own run with --kernel-trace only; FETCH_SIZE and WRITE_SIZE do not fit one pass, MI355X_MICROARCH.md "PMC slots").
#   profiles/prof_step.sh <tag> [bench.py flags...]      -> gpurun_out/prof_<tag>/{stats,fetch,write,sq,tcc,inst}/ + summary.json
TAG=$1; shift
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
ARGS="bench.py --steps 2 --warmup 1 --no-cpu --no-e2e --no-index-leg --parity-sample 0 $*"
export PROF_STEPS=2
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ARGS > $OUT/bench.json 2> $OUT/stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $ARGS > /dev/null 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $ARGS > /dev/null 2> $OUT/write.err
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --output-format csv -d $OUT/sq -- python3 $ARGS > /dev/null 2> $OUT/sq.err
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/tcc -- python3 $ARGS > /dev/null 2> $OUT/tcc.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --output-format csv -d $OUT/inst -- python3 $ARGS > /dev/null 2> $OUT/inst.err
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum TCC_EA0_RDREQ_DRAM_sum --output-format csv -d $OUT/ea -- python3 $ARGS > /dev/null 2> $OUT/ea.err
python3 profiles/summarize_pmc.py $OUT > $OUT/summary.json
# the raw per-dispatch tables are large: keep the per-kernel statistics and the summary
find $OUT -name "*counter_collection.csv" -delete
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*agent_info.csv" -delete
tail -c 300 $OUT/stats.err
