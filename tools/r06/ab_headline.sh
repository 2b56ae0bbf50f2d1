#!/usr/bin/env bash
# same-box A/B of library builds on the headline (and the accurate search): tools/r06/ab_headline.sh LIB...
for rep in 1 2 3; do for L in "$@"; do
  LIMG_HIP_LIB=$L python bench.py --steps 50 --no-cpu-baseline --no-host-rate 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$L', d['ms_per_step'], d['roofline']['kernels_ms'])"
done; done
for L in "$@"; do
  LIMG_HIP_LIB=$L python bench.py --accurate --steps 10 --no-cpu-baseline --no-host-rate 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$L accurate', d['ms_per_step'], d['roofline']['kernels_ms'])"
done
