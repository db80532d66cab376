R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r03z
cd /tmp && export TMPDIR=/tmp
for rows in 16 32; do for n in 1024 4096 8192 16384; do
  OLY_K11_ROWS=$rows rocprofv3 --kernel-trace --output-format csv -d /tmp/k11_${rows}_$n -o t -- python3 $R/tools/time_k11.py $n > /dev/null 2>&1
  f=$(find /tmp/k11_${rows}_$n -name "*kernel_trace.csv" | head -1)
  echo "rows=$rows N=$n $(python3 $R/tools/trace_summary.py $f mlp_forward | tail -n +2 | cut -d, -f1,2,8,9,10)"
done; done
