# Copies what profiles/collect.sh left in gpurun_out/ to the tracked names of this round (run from the repo root).
R=${1:-r06}
cp gpurun_out/final_pytest.txt profiles/${R}_pytest_gpu.txt
for n in default driver_args zerocopy seq lfu staged unchanged_trainer sharded_w1 torchrun1 interleaved overlap share2 share3; do
  cp gpurun_out/bench_$n.json profiles/${R}_bench_$n.json
done
cp gpurun_out/stats_seq.txt profiles/${R}_kernel_stats_criteo1tb_seq.txt
cp gpurun_out/stats_ov.txt profiles/${R}_kernel_stats_criteo1tb_default_profiled.txt
cp gpurun_out/stats_il.txt profiles/${R}_kernel_stats_criteo1tb_interleaved_profiled.txt
cp gpurun_out/stats_k1.txt profiles/${R}_kernel_stats_kaggle_p1_interleaved_profiled.txt
cp gpurun_out/timeline_il.txt profiles/${R}_timeline_one_window_interleaved_profiled.txt
cp gpurun_out/timeline_k1.txt profiles/${R}_timeline_kaggle_p1_interleaved_profiled.txt
cp gpurun_out/sharded_terms.md profiles/${R}_sharded_terms.md
cp gpurun_out/timeline_seq.txt profiles/${R}_timeline_one_window_seq.txt
cp gpurun_out/timeline_ov.txt profiles/${R}_timeline_one_window_default_profiled.txt
cp gpurun_out/pmc_hbm_traffic.txt profiles/${R}_pmc_hbm_traffic.txt
cp gpurun_out/traffic.json profiles/traffic.json
cp gpurun_out/config_matrix.md profiles/${R}_config_matrix.md
for n in overlap interleaved auto unchanged tunable graph graph_tunable; do
  [ -f gpurun_out/dlrm_main_$n.json ] && cp gpurun_out/dlrm_main_$n.json profiles/${R}_dlrm_main_criteo1tb_$n.json
done
