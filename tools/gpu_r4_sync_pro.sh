#!/bin/bash
# k_sync with its prologue in one burst + the bookkeeping on a snapshot: phases, the parity tests that see it, the FM pass
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
python tools/gpu_sync_phases_batch.py 2>&1 | tail -10
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "golden or oracle_end_to_end or batch_equals_streaming or zero_copy or deferred or noise_only or replay or interleaved" 2>&1 | tail -3
bash tools/gpu_r4_ab.sh "$@"
