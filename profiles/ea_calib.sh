#!/bin/bash
# the calibration kernels of profiles/micro/ea_calib.hip under rocprofv3, one pass per counter group (PMC slots: MI355X_MICROARCH.md)
#   profiles/ea_calib.sh   -> gpurun_out/ea_calib/{rd,dram,fetch}/ + gpurun_out/ea_calib/summary.json
export TMPDIR=/tmp
OUT=gpurun_out/ea_calib
mkdir -p $OUT
[ -x profiles/micro/ea_calib ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o profiles/micro/ea_calib profiles/micro/ea_calib.hip
./profiles/micro/ea_calib > $OUT/expected.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- ./profiles/micro/ea_calib > /dev/null 2> $OUT/stats.err
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum --output-format csv -d $OUT/rd -- ./profiles/micro/ea_calib > /dev/null 2> $OUT/rd.err
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT/dram -- ./profiles/micro/ea_calib > /dev/null 2> $OUT/dram.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- ./profiles/micro/ea_calib > /dev/null 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/tcc -- ./profiles/micro/ea_calib > /dev/null 2> $OUT/tcc.err
python3 profiles/ea_calib_summary.py $OUT > $OUT/summary.json
tail -c 400 $OUT/rd.err
