#!/usr/bin/env python3
"""AnticipationRNN training step alone (bench.py's `anticipation_rnn_train` extra): python tools/arnn_bench.py [steps]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

print(json.dumps(bench.arnn_extra(steps=int(sys.argv[1]) if len(sys.argv) > 1 else 20, warmup=3, tables=False)))
