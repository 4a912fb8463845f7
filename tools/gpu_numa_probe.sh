#!/bin/bash
# where does the drop-in's host thread want to run?  GPU NUMA node, CPU lists, and the drop-in leg pinned to each node's CPUs
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
for d in /sys/class/drm/card*/device; do [ -f $d/numa_node ] && echo "$d numa_node=$(cat $d/numa_node) $(cat $d/vendor 2>/dev/null)"; done | head -8
ls /sys/devices/system/node/ | grep node | head; for n in /sys/devices/system/node/node*; do echo "$n cpus=$(cat $n/cpulist)"; done | head -8
nproc; python - <<'PY'
import os; print("affinity", len(os.sched_getaffinity(0)), sorted(os.sched_getaffinity(0))[:8], "...")
PY
which numactl taskset
for n in /sys/devices/system/node/node*; do
  cpus=$(cat $n/cpulist)
  echo "== taskset -c $cpus"
  taskset -c $cpus python tools/gpu_dropin.py 2 2>&1 | grep "^{" | cut -c1-110
done
echo "== unpinned"; python tools/gpu_dropin.py 2 2>&1 | grep "^{" | cut -c1-110
