mkdir -p gpurun_out/r04s; cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04s/stats_vae -o s -- python $R/tools/vae_bench.py > $R/gpurun_out/r04s/stats_vae.log 2>&1
cp $R/gpurun_out/r04s/stats_vae/*/s_kernel_stats.csv $R/gpurun_out/r04s/r04s_vae_kernel_stats.csv 2>/dev/null || cp $R/gpurun_out/r04s/stats_vae/s_kernel_stats.csv $R/gpurun_out/r04s/r04s_vae_kernel_stats.csv
head -16 $R/gpurun_out/r04s/r04s_vae_kernel_stats.csv | cut -c1-260
