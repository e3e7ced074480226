#!/bin/bash
# two fresh processes on one GPU, one grid-synchronised computation through IPC-shared device memory (tools/ipc_probe.hip)
for mode in 0 1 2; do
  f=/tmp/ipc_handle_$mode.bin; rm -f $f
  echo "== mode $mode (0 hipMalloc, 1 uncached, 2 fine-grained)"
  timeout 120 tools/bin/ipc_probe owner $mode $f 20008 &
  opid=$!
  timeout 120 tools/bin/ipc_probe peer $f 20008
  wait $opid
  echo "owner rc $?"
done
