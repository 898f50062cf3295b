#!/bin/bash
# quick correctness + perf check of a traversal change: GPU parity suite, then env-variant A/B on killeroo-like 16 spp
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
bash tools/gpu_ab.sh "$@"
