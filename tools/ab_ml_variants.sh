#!/bin/bash
# model-level kernel: parity tests, then bench C3z (steps in one launch / one launch per step) for the default
# library and any experimental builds given as tags (mptrac_amd/lib/libmptrac_hip_<tag>.so)
for v in "" "$@"; do
  L=$PWD/mptrac_amd/lib/libmptrac_hip${v:+_$v}.so
  echo "== ${v:-default}"
  MPHIP_LIB=$L timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "zeta or mlp or model or non_monotonic or run_timesteps" 2>&1 | tail -1
  for m in on off; do MPHIP_LIB=$L timeout 120 python bench.py --workload C3z --multi-step $m --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('multi-step $m', '%.4g p-steps/s' % d['value'], '%.3f ms/step' % d['ms_per_step'], 'kernel %.3f ms' % d['roofline'].get('kernel_ms'))"; done
done
