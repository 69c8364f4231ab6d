#!/bin/bash
# One-box A/B of the AnticipationRNN step between two checkouts: build/wt_<name> worktrees built
# in the container (python -c "from inpaintnet_amd import _lib; _lib.build()" inside each), alternated with HEAD.
one() { (cd "$1" && HIP_FORCE_DEV_KERNARG=1 python -c "
import sys, torch, bench
sys.stdout = sys.stderr
r = bench.arnn_extra(steps=int('$2'), warmup=3)['anticipation_rnn_train']
print('$1', '$2 steps', r['ms_per_step'], 'ms/step')
" 2>&1 | grep 'ms/step'); }
for rep in 1 2 3; do
  for wt in "$@"; do one "$wt" 8; one "$wt" 40; done
done
