run() { timeout 600 python bench.py --no_cpu_baseline "$@" 2>gpurun_out/err.txt | tail -1 | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); c=d['cache']; v=d.get('verified') or {}; print('| \`%s\` | %.0f M | %.3f | %.0f | %.3f | %d / %d | %s | %s |' % ('$*', d['value']/1e6, d['ms_per_step'], d['it_per_s'], c['unique_hit_rate'], c['rows_in'], c['rows_out'], ('%d rows, %d violations, max err/bound %.2f' % (v['rows'], v['bound_violations'] + v.get('untouched_mismatch', 0), v['max_err_over_bound'])) if v else 'not run', (d['config'].get('arrangement') or {}).get('mode')))
except Exception as e:
    print('  FAILED $*', e)" ; grep -E "Error|error" gpurun_out/err.txt | tail -2; }
echo "| bench.py flags | lookups/s | ms/step | it/s | unique-row hit rate | rows in / out (timed+warmup) | end-of-run check (closed form of SGD) | arrangement (the library's choice unless pinned) |"
echo "|---|---|---|---|---|---|---|---|"
run --workload criteo_kaggle --cache_ratio 1.0 --prefetch_num 1
run --workload criteo_kaggle --cache_ratio 0.05 --prefetch_num 1
run --workload criteo_kaggle --cache_ratio 0.05 --prefetch_num 1 --use_lfu
run --workload criteo_kaggle --cache_ratio 0.05 --prefetch_num 8
run
run --use_lfu
run --no_overlap
run --workload avazu --cache_ratio 0.01
run --workload avazu --cache_ratio 0.01 --use_lfu
run --workload avazu --cache_ratio 0.01 --use_lfu --batch_size 2048 --embedding_dim 32 --prefetch_num 1
run --workload custom --cache_ratio 0.01 --pooling 2
run --workload custom --cache_ratio 0.01 --pooling 8 --batch_size 4096
run --dist uniform --batch_size 4096 --prefetch_num 4
# BASELINE.json configs[0]: the Kaggle table whole in the cache (cache_ratio 1.0, no swaps), and beside it the repo's
# pure-PyTorch CPU EmbeddingBag path over the SAME Kaggle table on the host's cores (cpu_baseline, not suppressed here)
echo
echo "configs[0] (Criteo-Kaggle, cache_ratio 1.0): GPU line and the torch-CPU EmbeddingBag path on the Kaggle table"
timeout 900 python bench.py --workload criteo_kaggle --cache_ratio 1.0 --prefetch_num 1 2>gpurun_out/err.txt | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['cpu_baseline']
print('| GPU lookups/s | ms/step | CPU lookups/s (torch F.embedding_bag fwd + sparse bwd + SGD.step) | CPU it/s | threads | sample |')
print('|---|---|---|---|---|---|')
print('| %.0f M | %.3f | %.2f M | %.1f | %d | %s |' % (d['value'] / 1e6, d['ms_per_step'], c['value'] / 1e6, c['it_per_s'], c['cores'], c['sample']))
"
