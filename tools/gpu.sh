#!/bin/bash
# usage: tools/gpu.sh TIMEOUT 'command' [tail-lines]  -- rebuilds the library (content-hash incremental), sends the tree to a GPU box;
# retries while the pod's GPU slots are busy (status=transient: nothing charged)
cd /root/repo
python -m multiplanarunet_amd.build 2>&1 | grep -v "^/opt/rocm" | tail -2
for attempt in 1 2 3 4 5 6 7 8; do
  out=$(/usr/local/graft/bin/gpurun --timeout "$1" -- "$2" 2>&1)
  if echo "$out" | grep -q "status=transient"; then echo "[gpu.sh] slots busy (attempt $attempt), retrying in 90 s"; sleep 90; continue; fi
  echo "$out" | tail -${3:-80}; exit 0
done
echo "$out" | tail -20
