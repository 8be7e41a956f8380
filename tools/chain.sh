#!/bin/bash
# Runs the given shell commands one after the other on the GPU box (each under its own `timeout`), and starts no further step once
# one was killed by its time limit (exit 124 / 137): a hung GPU step says something — read it before running more.
#   bash tools/chain.sh 'cmd 1' 'cmd 2' ...      (a step that merely FAILS does not stop the chain)
for c in "$@"; do
  echo "== $c"
  timeout -k 10 ${STEP_TIMEOUT:-420} bash -c "$c"
  rc=$?
  echo "== rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "== step killed at its limit: chain stopped"; exit $rc; fi
done
exit 0
