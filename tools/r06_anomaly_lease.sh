#!/bin/bash
# One fresh lease of the anomaly hunt: (c) five fresh-process runs of the driver's command, short form, nothing in front
# of --warmup 5 -- value, slowest timed step, recorder; (d) one repetition of the AnticipationRNN table under per-launch events.
mkdir -p gpurun_out
tag=$(date +%s)
for i in 1 2 3 4 5; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-roofline --no-parity 2>/dev/null
done | python -c "
import json, sys
rows = [json.loads(l) for l in sys.stdin if l.startswith('{')]
print(json.dumps({'lease': $tag, 'what': 'c', 'value': [r['value'] for r in rows], 'slowest_step_ms': [max(r['first_steps_ms']) for r in rows],
                  'first_step_ms': [r['first_steps_ms'][0] for r in rows], 'slow_waits': [r.get('slow_waits') for r in rows], 'waits_noted': [r.get('waits_noted') for r in rows], 'chain_timeouts': [r['chain_timeouts'] for r in rows]}))
" > gpurun_out/r06_anomaly_lease_$tag.jsonl
python tools/arnn_anomaly_rep.py 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read()); print(json.dumps({'lease': $tag, 'what': 'd', **d}))" >> gpurun_out/r06_anomaly_lease_$tag.jsonl
cat gpurun_out/r06_anomaly_lease_$tag.jsonl
