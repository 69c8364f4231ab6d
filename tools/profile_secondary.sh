#!/bin/bash
# end-of-round refresh of the secondary workloads' traces (the headline kernels did not change after profiles/r04_y_*):
#   kernel stats + one-step timeline of the AnticipationRNN and LatentRNN steps.   usage: tools/profile_secondary.sh <tag>
set -u
TAG=${1:-r04z}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
bash $ROOT/tools/trace_cmd.sh $TAG/arnn 10 tools/arnn_bench.py 20
bash $ROOT/tools/trace_cmd.sh $TAG/latent 8 tools/latent_time.py
head -6 $ROOT/gpurun_out/$TAG/arnn/kernel_stats.txt $ROOT/gpurun_out/$TAG/latent/kernel_stats.txt
head -8 $ROOT/gpurun_out/$TAG/arnn/timeline.txt
