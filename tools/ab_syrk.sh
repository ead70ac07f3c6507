#!/bin/bash
# A/B of SYRK launch parameters on the headline config (prints syrk ms/launch)
for kb in 8 16 32; do for wg in 1 2; do
  OAK_SYRK_KB=$kb OAK_SYRK_WG_PER_CU=$wg python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('kb=$kb wg=$wg', 'syrk ms', round(d['phase_ms_per_step']['syrk'],2), 'TF/s', round(d['roofline']['achieved'],1), 'step ms', round(d['ms_per_step'],1))"
done; done
