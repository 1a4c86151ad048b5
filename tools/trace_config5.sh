mkdir -p gpurun_out/r04i; cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04i/stats_cfg5 -o s -- python $R/bench.py --config 5 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $R/gpurun_out/r04i/stats_cfg5.log 2>&1
cp $R/gpurun_out/r04i/stats_cfg5/*/s_kernel_stats.csv $R/gpurun_out/r04i/r04i_bench_config5_kernel_stats.csv 2>/dev/null || cp $R/gpurun_out/r04i/stats_cfg5/s_kernel_stats.csv $R/gpurun_out/r04i/r04i_bench_config5_kernel_stats.csv
tail -2 $R/gpurun_out/r04i/stats_cfg5.log | cut -c1-300
head -30 $R/gpurun_out/r04i/r04i_bench_config5_kernel_stats.csv | cut -c1-200
