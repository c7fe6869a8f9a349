#!/bin/bash
cd /root/repo
run() {
  for rep in 1 2 3 4; do
  env $2 python bench.py --no_cpu_baseline --no_verify 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; o=d['roofline_other'][0]
print('$1', round(d['value']/1e9,3), 'bwd', round(r['avg_ms']*1e3,1), round(r['avg_ms_in_pipeline']*1e3,1), 'fwd', round(o['avg_ms']*1e3,1), round(o['avg_ms_in_pipeline']*1e3,1), 'chain', round(sum(v for k,v in d['cache']['cache_op_ms_by_phase'].items() if k!='admit_swap'),3))
"
  done
}
run base "CE_X=0"
run f16b4 "CE_FWDK_BLOCKS_PER_CU=16 CE_BWD_BLOCKS_PER_CU=4"
run f16b6 "CE_FWDK_BLOCKS_PER_CU=16 CE_BWD_BLOCKS_PER_CU=6"
run f16b8 "CE_FWDK_BLOCKS_PER_CU=16 CE_BWD_BLOCKS_PER_CU=8"
run f32b6 "CE_FWDK_BLOCKS_PER_CU=32 CE_BWD_BLOCKS_PER_CU=6"
run f16b12 "CE_FWDK_BLOCKS_PER_CU=16 CE_BWD_BLOCKS_PER_CU=12"
run base "CE_X=0"
