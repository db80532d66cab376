set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r03y
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o b -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/r03y/bench_driver_cmd_under_rocprof.json 2>/tmp/prof.err
cp $(find /tmp/prof -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r03y/bench_default_cmd_kernel_stats.csv
python3 $R/tools/trace_summary.py $(find /tmp/prof -name "*kernel_trace.csv" | head -1) > $R/gpurun_out/r03y/bench_default_cmd_kernel_summary.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o p -- python3 $R/tools/bench_vecstep.py --reps 1 > /tmp/pmc_$c.out 2>/tmp/pmc_$c.err
  f=$(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1)
  head -1 $f > $R/gpurun_out/r03y/k13_pmc_$c.csv
  grep a3_rollout_kernel $f >> $R/gpurun_out/r03y/k13_pmc_$c.csv
done
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d /tmp/pmc_sq -o p -- python3 $R/tools/bench_vecstep.py --reps 1 > /tmp/pmc_sq.out 2>/tmp/pmc_sq.err
f=$(find /tmp/pmc_sq -name "*counter_collection.csv" | head -1)
head -1 $f > $R/gpurun_out/r03y/k13_pmc_sq.csv
grep a3_rollout_kernel $f >> $R/gpurun_out/r03y/k13_pmc_sq.csv
grep a3_rollout $R/gpurun_out/r03y/bench_default_cmd_kernel_summary.csv
wc -l $R/gpurun_out/r03y/*.csv
