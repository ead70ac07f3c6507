#!/bin/bash
# A/B of SYRK launch parameters on the headline config (prints syrk ms/launch)
for xcd in 0 1; do
  OAK_SYRK_XCD=$xcd python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('xcd=$xcd', 'syrk ms', round(d['phase_ms_per_step']['syrk'],2), 'TF/s', round(d['roofline']['achieved'],1), 'reduce', round(d['phase_ms_per_step']['reduce'],2), 'step ms', round(d['ms_per_step'],1), 'loss', d['loss'])"
done
