# Last collection of round 6 (one gpurun call, run from the repo root): the evaluation tests first, the whole GPU suite,
# the default / driver-argument bench lines of the final tree, the whole model with --eval_acc at the Kaggle table, and
# the reuse sweep's bench lines (NO_PMC=1).  Everything lands in gpurun_out/final/.
set -x
R=$PWD
O=$R/gpurun_out/final
mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_modules.py -m gpu -x -q -k evaluate 2>&1 | tail -15 > $O/eval_tests.txt; cat $O/eval_tests.txt
timeout 1300 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $O/final_pytest.txt; cat $O/final_pytest.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -2 $O/bench_default.err
python bench.py --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err
timeout 400 python examples/dlrm_main.py --dataset criteo_kaggle --use_cache --cache_ratio 0.05 --use_freq --batch_size 16384 \
    --prefetch_num 8 --use_overlap --use_sparse_embed_grad --overlap_cache_op --fused_sgd --fold_hook --window_keys \
    --limit_train_batches 400 --eval_acc --limit_val_batches 16 --limit_test_batches 16 --learning_rate 0.1 \
    --json_out $O/dlrm_main_kaggle_eval.json > $O/dlrm_main_kaggle_eval.log 2>&1; tail -8 $O/dlrm_main_kaggle_eval.log
NO_PMC=1 bash profiles/reuse_sweep.sh > $O/reuse_sweep.log 2>&1
python profiles/reuse_sweep.py gpurun_out/sweep > $O/reuse_sweep.md 2>$O/reuse_sweep.err; head -20 $O/reuse_sweep.md
rm -rf gpurun_out/sweep/pmc
du -sh gpurun_out
