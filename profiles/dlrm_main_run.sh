# Whole-model it/s of examples/dlrm_main.py at BASELINE.json configs[2]: the trainer of
# recsys/dlrm_main.py:206-297 around this operator, by surface.  Writes gpurun_out/dlrm_main_*.json; merge with
# `python profiles/dlrm_main_run.sh.py` is not needed: publish copies them as profiles/r06_dlrm_main_criteo1tb_<surface>.json
set -x
mkdir -p gpurun_out
COMMON="--dataset criteo_1tb --use_cache --cache_ratio 0.01 --use_freq --batch_size 16384 --prefetch_num 8 --use_overlap --limit_train_batches 616 --warmup_batches 16"
python examples/dlrm_main.py $COMMON --overlap_cache_op --fused_sgd --fold_hook --window_keys --json_out gpurun_out/dlrm_main_overlap.json 2>&1 | tail -4
python examples/dlrm_main.py $COMMON --overlap_cache_op --fused_sgd --fold_hook --window_keys --arrangement interleaved --json_out gpurun_out/dlrm_main_interleaved.json 2>&1 | tail -3
python examples/dlrm_main.py $COMMON --overlap_cache_op --fused_sgd --fold_hook --window_keys --arrangement auto --json_out gpurun_out/dlrm_main_auto.json 2>&1 | tail -3
python examples/dlrm_main.py $COMMON --use_sparse_embed_grad --json_out gpurun_out/dlrm_main_unchanged.json 2>&1 | tail -3
# the dense part's GEMMs picked by torch's TunableOp (warm-up long enough for the tuning to finish)
python examples/dlrm_main.py $COMMON --overlap_cache_op --fused_sgd --fold_hook --window_keys --arrangement overlap --tunable_gemm --warmup_batches 96 --json_out gpurun_out/dlrm_main_tunable.json 2>&1 | tail -4
# ... the whole iteration replayed from one hipGraph (the launch thread no longer sets the pace), and both together
python examples/dlrm_main.py $COMMON --overlap_cache_op --fused_sgd --fold_hook --window_keys --arrangement overlap --graph_step --json_out gpurun_out/dlrm_main_graph.json 2>&1 | tail -4
python examples/dlrm_main.py $COMMON --overlap_cache_op --fused_sgd --fold_hook --window_keys --arrangement overlap --graph_step --graph_after 80 --tunable_gemm --warmup_batches 96 --json_out gpurun_out/dlrm_main_graph_tunable.json 2>&1 | tail -4
