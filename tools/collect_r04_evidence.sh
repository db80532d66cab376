# Round-4 evidence, one GPU call:  gpurun -- 'bash tools/collect_r04_evidence.sh'  -> gpurun_out/r04/ (copied into profiles/r04)
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# 1. the driver's command, un-profiled and under rocprofv3 (kernel trace + stats)
python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2>/tmp/bench.err || tail -5 /tmp/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o b -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_cmd_under_rocprof.json 2>/tmp/prof.err || tail -5 /tmp/prof.err
cp $(find /tmp/prof -name "*kernel_stats.csv" | head -1) $O/bench_default_cmd_kernel_stats.csv
python3 $R/tools/trace_summary.py $(find /tmp/prof -name "*kernel_trace.csv" | head -1) > $O/bench_default_cmd_kernel_summary.csv
# 2. whole config-3 iterations with the kernel update, under rocprofv3
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof2 -o it -- python3 $R/tools/bench_ppo_iter.py --rollout device --mode kernel --itr 5 > $O/ppo_iteration_kernel_update_under_rocprof.json 2>/tmp/prof2.err || tail -5 /tmp/prof2.err
cp $(find /tmp/prof2 -name "*kernel_stats.csv" | head -1) $O/ppo_iteration_kernel_update_kernel_stats.csv
python3 $R/tools/bench_ppo_iter.py --rollout device --mode kernel --itr 6 > $O/ppo_iteration_kernel_update.json 2>/tmp/it.err || tail -5 /tmp/it.err
python3 $R/tools/bench_ppo_iter.py --rollout device --mode fused_graph --itr 4 > $O/ppo_iteration_torch_graph_update.json 2>/tmp/it2.err || tail -5 /tmp/it2.err
# 3. the update step alone: every path, wall clock; the kernel path's phases
python3 $R/tools/bench_ppo_update.py > $O/ppo_update_step.json 2>/tmp/u.err || tail -5 /tmp/u.err
python3 $R/tools/profile_update_phase.py 1638400 65536 > $O/update_phase_wall_minibatch_65536.json 2>/tmp/u2.err || tail -5 /tmp/u2.err
python3 $R/tools/profile_update_phase.py 1638400 64 > $O/update_phase_wall_minibatch_64.json 2>/tmp/u3.err || tail -5 /tmp/u3.err
python3 $R/tools/bench_ppo_update_kernel.py --out $O/k14_kernel_by_minibatch.json > /dev/null 2>/tmp/u4.err || tail -5 /tmp/u4.err
python3 $R/tools/bench_ppo_update_kernel.py --mirror --out $O/k14_kernel_by_minibatch_mirror.json > /dev/null 2>/tmp/u5.err || tail -5 /tmp/u5.err
python3 $R/tools/time_k14.py 65536 > /tmp/k14a.txt 2>/tmp/u6.err; python3 -c "t=open('/tmp/k14a.txt').read(); open('$O/k14_phase_cycles.json','w').write(t[t.find('{'):])"
python3 $R/tools/time_k14.py 65536 --mirror > /tmp/k14b.txt 2>/tmp/u7.err; python3 -c "t=open('/tmp/k14b.txt').read(); open('$O/k14_phase_cycles_mirror.json','w').write(t[t.find('{'):])"
# 4. K14 counters (separate passes, kernel trace only): matrix-pipe busy, instruction mix; HBM traffic
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d /tmp/pmc_sq -o p -- python3 $R/tools/bench_ppo_update_kernel.py --batches 65536 --reps 5 > /tmp/pmc_sq.out 2>/tmp/pmc_sq.err
f=$(find /tmp/pmc_sq -name "*counter_collection.csv" | head -1)
python3 $R/tools/pmc_summary.py $f ppo_update > $O/k14_sq_counters.json
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o p -- python3 $R/tools/bench_ppo_update_kernel.py --batches 65536 --reps 5 > /tmp/pmc_$c.out 2>/tmp/pmc_$c.err
  f=$(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1)
  python3 $R/tools/pmc_summary.py $f ppo_update > $O/k14_pmc_$c.json
done
# 5. how a vec step's bytes can cross PCIe (host batcher design data)
$R/tools/hip/pcie_paths.bin > $O/pcie_paths.json 2>/tmp/pcie.err || tail -3 /tmp/pcie.err
ls -la $O
