cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $R/gpurun_out/sq_now --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/trace_now --output-format csv -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline > /dev/null 2>&1
echo ok
