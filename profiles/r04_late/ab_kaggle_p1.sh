#!/bin/bash
cd /root/repo
one() {
env $2 python bench.py --no_cpu_baseline --no_verify --workload criteo_kaggle --cache_ratio 0.05 --prefetch_num 1 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['roofline_other'][1]
print('$1', round(d['value']/1e9,3), 'out_busy', round(s.get('worker_out_busy_ms',0),3), 'in_busy', round(s.get('worker_in_busy_ms',0),3), 'in_wait', round(s.get('worker_in_wait_ms',0),3), {k:round(v,3) for k,v in d['cache']['cache_op_ms_by_phase'].items()})"
}
for rep in 1 2; do
one new "CE_X=1"
one old "CE_WB_RELAX=0 CE_EARLY_MAPS=0 CE_FWDK_BLOCKS_PER_CU=8 CE_BWD_BLOCKS_PER_CU=2 CE_HOST_THP=0 CE_NUMA_BIND=0"
one norelax "CE_WB_RELAX=0"
one noearly "CE_EARLY_MAPS=0"
one oldgrid "CE_FWDK_BLOCKS_PER_CU=8 CE_BWD_BLOCKS_PER_CU=2"
one nobind "CE_NUMA_BIND=0"
done
